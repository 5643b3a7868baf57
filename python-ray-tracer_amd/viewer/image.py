"""Frame sink, reference viewer/image.py:7-19: the (3,w,h) uint8 frame -> an RGB image.

On square frames the reference's `rotate(270)` + `mirror` is a transpose: image[row=y, col=x] =
frame[:, x, y].  On non-square frames its fixed-size canvas crops; here the
transpose is applied for every shape.  The frame's channel planes are used as stored (the
reference stores R,B,G — common.py:63 — and displays them as if R,G,B; `undo_swap=True` shows
true colours instead)."""
import numpy as np


def frame_to_hwc(x, undo_swap=False):
    x = np.asarray(x)
    if x.ndim != 3 or x.shape[0] != 3:
        raise ValueError(f"expected a (3, w, h) frame, got {x.shape}")
    planes = x[[0, 2, 1]] if undo_swap else x
    return np.ascontiguousarray(planes.transpose(2, 1, 0).astype(np.uint8))


def convert_array_to_image(x, undo_swap=False):
    """(3,w,h) frame -> PIL image; an (h,w,3) array (rendered with RT_FLAG_U8_HWC on the device) is used as is."""
    from PIL import Image
    x = np.asarray(x)
    if x.ndim == 3 and x.shape[2] == 3 and x.shape[0] != 3:
        return Image.fromarray(np.ascontiguousarray(x.astype(np.uint8)))
    return Image.fromarray(frame_to_hwc(x, undo_swap))
