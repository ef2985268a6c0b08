"""Functional stand-in for the 3 cv2 calls + constants reached by loader/postproc.py P2-P4 (OpenCV is not installed in
this container).  Restated from OpenCV's documented semantics (see oracle/postproc_ref.c header).  Used ONLY by
oracle/gen_golden_postproc.py to let the reference's own post_process run against the real skimage / scipy."""
import numpy as np

MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE = 0, 1, 2
INTER_NEAREST, INTER_LINEAR = 0, 1
RETR_TREE, CHAIN_APPROX_SIMPLE = 3, 2
COLOR_BGR2RGB = 4


def getStructuringElement(shape, ksize):
    assert shape == MORPH_ELLIPSE
    w, h = ksize
    r, c = h // 2, w // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    se = np.zeros((h, w), np.uint8)
    for i in range(h):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            se[i, max(c - dx, 0):min(c + dx + 1, w)] = 1
    return se


def _morph(src, kernel, is_dilate):
    src = np.asarray(src)
    H, W = src.shape
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    out = np.full((H, W), 0 if is_dilate else 255, dtype=src.dtype)
    for i in range(kh):
        for j in range(kw):
            if not kernel[i, j]:
                continue
            dy, dx = i - ay, j - ax  # dst(y,x) op= src(y+dy, x+dx)
            y0, y1 = max(0, -dy), min(H, H - dy)
            x0, x1 = max(0, -dx), min(W, W - dx)
            if y0 >= y1 or x0 >= x1:
                continue
            s = src[y0 + dy:y1 + dy, x0 + dx:x1 + dx]
            d = out[y0:y1, x0:x1]
            out[y0:y1, x0:x1] = np.maximum(d, s) if is_dilate else np.minimum(d, s)
    if not is_dilate:
        # pixels whose every tap fell outside cannot occur (the anchor tap is always inside for the 3x3 cross)
        pass
    return out


def dilate(src, kernel, iterations=1):
    assert iterations == 1
    return _morph(src, kernel, True)


def erode(src, kernel, iterations=1):
    assert iterations == 1
    return _morph(src, kernel, False)
