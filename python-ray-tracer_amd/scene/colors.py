"""The six colour constants of the reference's scene/colors.py:1-6 (0-255 RGB triples)."""
RED, GREEN, BLUE = [255, 70, 70], [70, 255, 70], [70, 70, 255]
YELLOW, GREY, MAGENTA = [255, 255, 70], [125, 125, 125], [139, 0, 139]
