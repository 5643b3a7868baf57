from .image import convert_array_to_image, frame_to_hwc  # noqa: F401
