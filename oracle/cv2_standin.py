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
# the chain code changes.  find_contours_tree below is the whole algorithm (holes, hierarchy); _trace_outer is the outer-border
# walk alone.  PARITY UNPINNED against the real library; external pins: tests/golden/cv2_documented.json.
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


def find_contours_tree(mask, method=CHAIN_APPROX_SIMPLE):
    """Suzuki & Abe's Algorithm 1 in full (outer AND hole borders, parent bookkeeping through LNBD, border marking), flattened
    the way cv::findContours(RETR_TREE) flattens its tree: OpenCV links every finished border at the FRONT of its parent's child
    list (cvInsertNodeIntoTree in the 4.x legacy implementation behind opencv-python 4.6.0.66) and walks the tree in pre-order
    (cvTreeToNodeSeq), so siblings come newest-first and element 0 is the top-level outer border whose start pixel comes LAST in
    the raster scan -- a piece lying inside a hole of another piece is a grand-child and never element 0.
    Returns (contours, hierarchy): contours[k] int32 (K, 1, 2) of (x, y), hierarchy int32 (1, n, 4) = [next, prev, first_child, parent].
    Foreground is 8-connected (outer borders), holes 4-connected, as in OpenCV."""
    m = np.asarray(mask)
    h, w = m.shape
    f = np.zeros((h + 2, w + 2), np.int32)
    f[1:-1, 1:-1] = (m != 0)
    # neighbour k of (i, j) in CLOCKWISE order starting east (row i down): E, SE, S, SW, W, NW, N, NE
    cw = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]
    idx_of = {d: k for k, d in enumerate(cw)}
    borders = {1: dict(hole=True, parent=0, pts=None)}  # the frame
    order = []
    nbd = 1
    for i in range(1, h + 1):
        lnbd = 1
        for j in range(1, w + 1):
            v = f[i, j]
            if v == 0:
                continue
            is_outer = v == 1 and f[i, j - 1] == 0
            is_hole = (not is_outer) and v >= 1 and f[i, j + 1] == 0
            if is_outer or is_hole:
                nbd += 1
                i2, j2 = (i, j - 1) if is_outer else (i, j + 1)
                if is_hole and v > 1:
                    lnbd = v
                bp = borders[lnbd]
                parent = (bp["parent"] if not bp["hole"] else lnbd) if is_outer else (lnbd if not bp["hole"] else bp["parent"])
                if is_hole and not bp["hole"]:
                    parent = lnbd
                elif is_hole and bp["hole"]:
                    parent = bp["parent"]
                # (3.1) clockwise around (i, j) from (i2, j2): first non-zero pixel
                k0 = idx_of[(i2 - i, j2 - j)]
                i1 = j1 = None
                for t in range(8):
                    di, dj = cw[(k0 + t) % 8]
                    if f[i + di, j + dj] != 0:
                        i1, j1 = i + di, j + dj
                        break
                pts = []
                if i1 is None:
                    f[i, j] = -nbd
                    pts = [(j - 1, i - 1)]
                else:
                    i2, j2, i3, j3 = i1, j1, i, j
                    prev_dir = None
                    while True:
                        # (3.3) counter-clockwise around (i3, j3) starting after (i2, j2)
                        k0 = idx_of[(i2 - i3, j2 - j3)]
                        east_zero = False
                        for t in range(1, 9):
                            k = (k0 - t) % 8
                            di, dj = cw[k]
                            if f[i3 + di, j3 + dj] != 0:
                                i4, j4 = i3 + di, j3 + dj
                                break
                            if k == 0:
                                east_zero = True  # (i3, j3 + 1) is a 0-pixel examined in this step
                        if east_zero:
                            f[i3, j3] = -nbd
                        elif f[i3, j3] == 1:
                            f[i3, j3] = nbd
                        step = (i4 - i3, j4 - j3)
                        if method == CHAIN_APPROX_NONE or step != prev_dir:
                            pts.append((j3 - 1, i3 - 1))
                            prev_dir = step
                        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                            break
                        i2, j2, i3, j3 = i3, j3, i4, j4
                    borders[nbd] = dict(hole=is_hole, parent=parent, pts=pts)
                    order.append(nbd)
                    if f[i, j] != 1:
                        lnbd = abs(f[i, j])
                    continue
                borders[nbd] = dict(hole=is_hole, parent=parent, pts=pts)
                order.append(nbd)
            if f[i, j] != 1:
                lnbd = abs(f[i, j])
    # flatten: children newest-first, pre-order
    kids = {}
    for b in order:
        kids.setdefault(borders[b]["parent"], []).insert(0, b)
    flat = []

    def walk(p):
        for c in kids.get(p, []):
            flat.append(c)
            walk(c)

    walk(1)
    pos = {b: k for k, b in enumerate(flat)}
    hier = np.full((1, len(flat), 4), -1, np.int32)
    for p, cs in kids.items():
        for a_, b_ in zip(cs[:-1], cs[1:]):
            hier[0, pos[a_], 0] = pos[b_]
            hier[0, pos[b_], 1] = pos[a_]
        if p != 1 and cs:
            hier[0, pos[p], 2] = pos[cs[0]]
        for c in cs:
            hier[0, pos[c], 3] = pos[p] if p != 1 else -1
    return [np.array(borders[b]["pts"], dtype=np.int32).reshape(-1, 1, 2) for b in flat], hier


def findContours(mask, mode=RETR_TREE, method=CHAIN_APPROX_SIMPLE):
    """(contours, hierarchy) of cv2.findContours(mask, RETR_TREE, method) -- see find_contours_tree."""
    assert mode == RETR_TREE
    contours, hier = find_contours_tree(mask, method)
    return contours, (hier if len(contours) else None)


def moments(array, binaryImage=False):
    """cv2.moments of a single-channel image as loader/postproc.py:27 consumes it (m00, m10, m01 only): raw spatial moments
    m_pq = sum_{x,y} x^p y^q I(x, y) in double precision (OpenCV's documented definition; the mask is 0 / 1, so no rounding is
    involved below 2^53).  Restated, PARITY UNPINNED against the library."""
    a = np.asarray(array)
    assert a.ndim == 2
    v = (a != 0).astype(np.float64) if binaryImage else a.astype(np.float64)
    ys, xs = np.mgrid[0:a.shape[0], 0:a.shape[1]]
    return {"m00": float(v.sum()), "m10": float((v * xs).sum()), "m01": float((v * ys).sum())}


def findContours_first_piece(mask, method=CHAIN_APPROX_SIMPLE):
    """The shortcut the device kernels take (cerb_inst_contour_start): outer border of the 8-connected piece whose first pixel
    comes last in the raster scan.  Equals findContours(...)[0][0] whenever no piece lies inside a hole of another piece --
    always true for the label maps post_process emits (tests/test_oracle_postproc.py)."""
    from scipy import ndimage

    mask = np.asarray(mask)
    lab, n = ndimage.label(mask != 0, structure=np.ones((3, 3), int))
    if n == 0:
        return None
    starts = ndimage.minimum(np.arange(mask.size).reshape(mask.shape), lab, index=np.arange(1, n + 1))
    last = int(np.max(starts))
    return np.array(_trace_outer(mask, last % mask.shape[1], last // mask.shape[1], method), dtype=np.int32).reshape(-1, 1, 2)


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
