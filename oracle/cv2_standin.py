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


# ---- findContours (outer border of the first returned contour only) -----------------------------------------------------
# TEST INFRASTRUCTURE.  Restatement of what loader/postproc.py:29-41 consumes: cv2.findContours(mask, RETR_TREE,
# CHAIN_APPROX_SIMPLE)[0][0].  OpenCV (pinned opencv-python 4.6.0.66, environment.yml:31) is absent from this image, so this
# follows Suzuki & Abe, "Topological structural analysis of digitized binary images by border following" (CVGIP 1985),
# with OpenCV's documented conventions: 8-connected foreground; Freeman codes 0..7 = E, NE, N, NW, W, SW, S, SE (y down);
# raster scan finds outer-border start pixels; top-level contours are returned most-recently-found first, so [0][0] is the
# outer border of the component whose start pixel comes LAST in raster order; CHAIN_APPROX_SIMPLE keeps the points at which
# the chain code changes.  Not modelled: a piece nested in a hole of another piece (not top-level under RETR_TREE).
# PARITY UNPINNED against the real library.
RETR_TREE, CHAIN_APPROX_SIMPLE, CHAIN_APPROX_NONE = 3, 2, 1
_CODE = [(1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1)]  # (dx, dy)


def _trace_outer(mask, x0, y0, method):
    h, w = mask.shape

    def fg(x, y):
        return 0 <= x < w and 0 <= y < h and mask[y, x] != 0

    pts = []
    s = 4
    while True:  # clockwise from NW for the first neighbour
        s = (s - 1) & 7
        x1, y1 = x0 + _CODE[s][0], y0 + _CODE[s][1]
        if fg(x1, y1) or s == 4:
            break
    if s == 4 and not fg(x1, y1):
        return [(x0, y0)]
    x3, y3, px, py, prev = x0, y0, x0, y0, s ^ 4
    while True:
        while True:  # counter-clockwise, starting right after the direction we arrived from
            s += 1
            x4, y4 = x3 + _CODE[s & 7][0], y3 + _CODE[s & 7][1]
            if fg(x4, y4):
                break
        s &= 7
        if s != prev or method == CHAIN_APPROX_NONE:
            pts.append((px, py))
            prev = s
        px, py = px + _CODE[s][0], py + _CODE[s][1]
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            return pts
        x3, y3 = x4, y4
        s = (s + 4) & 7


def findContours(mask, mode=RETR_TREE, method=CHAIN_APPROX_SIMPLE):
    """Returns ([first_contour], None) with first_contour int32 (K, 1, 2) of (x, y) -- only element [0][0] is reproduced."""
    from scipy import ndimage

    mask = np.asarray(mask)
    lab, n = ndimage.label(mask != 0, structure=np.ones((3, 3), int))
    if n == 0:
        return [], None
    starts = ndimage.minimum(np.arange(mask.size).reshape(mask.shape), lab, index=np.arange(1, n + 1))
    last = int(np.max(starts))
    pts = _trace_outer(mask, last % mask.shape[1], last // mask.shape[1], method)
    return [np.array(pts, dtype=np.int32).reshape(-1, 1, 2)], None


def _round_half_even(v):
    return int(np.rint(v))  # cvRound


def resize(src, dsize, fx=0.0, fy=0.0, interpolation=INTER_LINEAR):
    """cv2.resize restated for the two modes infer/wsi.py reaches (:691-697, :706-710, :761-765, :786-788).  Unpinned: OpenCV
    is absent.  Size: dsize, or cvRound(src * f) when dsize is (0, 0); inverse scale = f in the second case and dsize / ssize in
    the first, source scale = 1 / inverse scale (double).  INTER_NEAREST: s = min(floor(d * scale), ssize - 1).  INTER_LINEAR:
    f = (d + 0.5) * scale - 0.5, s = floor(f), w = f - s; s < 0 -> (0, w = 0); s >= ssize - 1 -> (ssize - 1, w = 0); float32
    weights, horizontal pass then vertical pass, each d = a * (1 - w) + b * w."""
    src = np.asarray(src)
    H, W = src.shape[:2]
    if dsize is None or tuple(dsize) == (0, 0):
        inv_x, inv_y = float(fx), float(fy)
        dw, dh = _round_half_even(W * inv_x), _round_half_even(H * inv_y)
    else:
        dw, dh = int(dsize[0]), int(dsize[1])
        inv_x, inv_y = dw / W, dh / H
    sx_scale, sy_scale = 1.0 / inv_x, 1.0 / inv_y

    def nearest(n_dst, n_src, scale):
        return np.minimum(np.floor(np.arange(n_dst) * scale).astype(np.int64), n_src - 1)

    if interpolation == INTER_NEAREST:
        return src[nearest(dh, H, sy_scale)][:, nearest(dw, W, sx_scale)].copy()
    assert interpolation == INTER_LINEAR

    def taps(n_dst, n_src, scale):
        f = (np.arange(n_dst) + 0.5) * scale - 0.5
        s = np.floor(f).astype(np.int64)
        w = (f - s).astype(np.float32)
        lo = s < 0
        s[lo], w[lo] = 0, 0.0
        hi = s >= n_src - 1
        s[hi], w[hi] = n_src - 1, 0.0
        return s, np.minimum(s + 1, n_src - 1), w

    a = src.astype(np.float32)
    x0, x1, wx = taps(dw, W, sx_scale)
    y0, y1, wy = taps(dh, H, sy_scale)
    shape_x = (1, dw) + (1,) * (a.ndim - 2)
    hor = a[:, x0] * (np.float32(1) - wx).reshape(shape_x) + a[:, x1] * wx.reshape(shape_x)
    shape_y = (dh, 1) + (1,) * (a.ndim - 2)
    return (hor[y0] * (np.float32(1) - wy).reshape(shape_y) + hor[y1] * wy.reshape(shape_y)).astype(np.float32)
