"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the reference's TILED nuclei post-processing of a
slide -- what `cerberus_amd/shard_postproc.py` replaces with exact band ownership.

  tile sets (grid, vertical strips, horizontal strips, cross sections) + removal flags
        infer/wsi.py:290-317, 643 -> tiatoolbox 1.3.1 `NucleusInstanceSegmentor._get_tile_info` (un-vendored and absent from this image:
        restated from its published source as remembered -- PARITY UNPINNED)
  per tile: post_process -> instance boxes -> margin logic -> offsets         infer/wsi.py:81-268 (`_process_tile_predictions`; in the repo)
  accumulation: update + remove                                               infer/wsi.py:471-480 (`_merge_inst_results` callback)

shapely is absent: `box.contains(box)` and `STRtree.query` (envelope intersection, touching included) are closed-interval box tests.
post_process is oracle/postproc_ref.py (pinned against the real reference).  Instances are identified by their bounding box
[x0, y0, x1, y1) in slide coordinates (HoVerNet.get_instance_info keeps box / centroid / contour / type; the box is what the margin logic uses).
"""
import numpy as np

from . import postproc_ref as pr


def get_coordinates(image_wh, tile_wh):
    """PatchExtractor.get_coordinates(image_shape, patch_input_shape = patch_output_shape = stride_shape = tile): output boxes
    [x0, y0, x1, y1] on a regular grid from the origin; the last row / column may reach past the image."""
    w, h = image_wh
    tw, th = tile_wh
    xs = np.arange(0, max(int(np.ceil(w / tw)), 1)) * tw
    ys = np.arange(0, max(int(np.ceil(h / th)), 1)) * th
    out = [[x, y, x + tw, y + th] for y in ys for x in xs]
    return np.array(out, np.int64)


def _intersects(a, b):  # closed boxes (shapely envelope intersection: touching counts)
    return a[0] <= b[2] and b[0] <= a[2] and a[1] <= b[3] and b[1] <= a[3]


def _contains(outer, inner):  # shapely `outer.contains(inner)` for boxes with non-empty interiors
    return outer[0] <= inner[0] and outer[1] <= inner[1] and inner[2] <= outer[2] and inner[3] <= outer[3]


def get_tile_info(image_wh, tile_shape, margin, patch_output_shape):
    """-> list of (boxes [n, 4], flags [n, 4] = remove-side [top, bottom, left, right]) for modes 0 (grid), 1 (vertical strips),
    2 (horizontal strips), 3 (cross sections)."""
    w, h = int(image_wh[0]), int(image_wh[1])
    tile = (np.floor(np.array(tile_shape) / np.array(patch_output_shape)) * np.array(patch_output_shape)).astype(np.int64)
    boxes = get_coordinates((w, h), tile)
    if w <= tile[0] and h <= tile[1]:
        return [(boxes, np.zeros((len(boxes), 4), np.int64))]
    edges = [(0, 0, w, 0), (0, h, w, h), (0, 0, 0, h), (w, 0, w, h)]  # top, bottom, left, right image edges

    def unset_removal_flag(bx, flag):
        for idx, e in enumerate(edges):
            for i, b in enumerate(bx):
                if _intersects(b, e):
                    flag[i, idx] = 0
        return flag

    boxes_br = boxes[:, 2:]
    boxes_tr = np.stack([boxes[:, 2], boxes[:, 1]], axis=1)
    boxes_bl = np.stack([boxes[:, 0], boxes[:, 3]], axis=1)
    flag = unset_removal_flag(boxes, np.ones((len(boxes), 4), np.int64))
    info = [(boxes, flag)]
    sel = np.nonzero(flag[:, 3])[0]  # tiles whose right edge is inside the slide: a vertical strip astride that edge
    vb = np.concatenate([boxes_tr[sel] - np.array([margin, 0]), boxes_br[sel] + np.array([margin, 0])], axis=1)
    vf = np.zeros((len(vb), 4), np.int64)
    vf[:, [0, 1]] = 1
    info.append((vb, unset_removal_flag(vb, vf)))
    sel = np.nonzero(flag[:, 1])[0]  # tiles whose bottom edge is inside the slide: a horizontal strip astride it
    hb = np.concatenate([boxes_bl[sel] - np.array([0, margin]), boxes_br[sel] + np.array([0, margin])], axis=1)
    hf = np.zeros((len(hb), 4), np.int64)
    hf[:, [2, 3]] = 1
    info.append((hb, unset_removal_flag(hb, hf)))
    sel = np.nonzero(flag[:, 1] * flag[:, 3])[0]  # bottom-right corners inside the slide: a cross section of 4 margins square
    cb = np.concatenate([boxes_br[sel] - 2 * margin, boxes_br[sel] + 2 * margin], axis=1)
    info.append((cb, np.ones((len(cb), 4), np.int64)))
    return info


def _inst_boxes(lab):
    """{id: [x0, y0, x1, y1)} of a label map (misc/utils.py:82-91 get_bounding_box, max exclusive)."""
    out = {}
    ids = np.unique(lab)
    for i in ids[ids > 0]:
        ys, xs = np.nonzero(lab == i)
        out[int(i)] = np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1], np.int64)
    return out


def process_tile(inst_canvas, tile_bounds, tile_flag, tile_mode, ref_boxes, margin, labeller=None):
    """infer/wsi.py:81-268.  ref_boxes: {uid: box} accumulated so far.  -> (new {local key: slide box}, [uids to remove from ref_boxes]).
    labeller(crop [h, w, 2]) -> label map: the C oracle by default; the GPU tests pass the HIP kernel here."""
    H, W = inst_canvas.shape[:2]
    tl, br = np.array(tile_bounds[:2]), np.array(tile_bounds[2:])
    w, h = (br - tl).tolist()
    crop = inst_canvas[max(tl[1], 0):br[1], max(tl[0], 0):br[0]]
    if crop.size == 0:
        return {}, []
    lab = (labeller(np.ascontiguousarray(crop)) if labeller is not None else pr.proc(np.ascontiguousarray(crop), "Nuclei")).astype(np.int64)
    boxes = _inst_boxes(lab)
    if not boxes:
        return {}, []
    m = margin
    boundary_lines = [(0, 0, w, 1), (0, h - 1, w, h), (0, 0, 1, h), (w - 1, 0, w, h)]
    margin_boxes = [(0, 0, w, m), (0, h - m, w, h), (0, 0, m, h), (w - m, 0, w, h)]
    margin_lines = [(m, m, w - m, m), (m, h - m, w - m, h - m), (m, m, m, h - m), (w - m, m, w - m, h - m)]
    margin_lines = [(a + tl[0], b + tl[1], c + tl[0], d + tl[1]) for a, b, c, d in margin_lines]
    remove = set()
    if tile_mode in (0, 3):
        sel = [bx for i, bx in enumerate(margin_boxes) if tile_flag[i] or tile_mode == 3]
        for bounds in sel:
            for k, b in boxes.items():
                if _intersects(b, bounds) and _contains(bounds, b):
                    remove.add(k)
    elif tile_mode in (1, 2):
        sel = [margin_boxes[i] if f else boundary_lines[i] for i, f in enumerate(tile_flag)]
        for bounds in sel:
            for k, b in boxes.items():
                if _intersects(b, bounds):
                    remove.add(k)
    else:
        raise ValueError(tile_mode)
    remove_in_orig = []
    if tile_mode == 3:
        for uid, b in ref_boxes.items():
            if any(_intersects(b, ml) for ml in margin_lines):
                remove_in_orig.append(uid)
    off = np.concatenate([tl, tl])
    new = {k: b + off for k, b in boxes.items() if k not in remove}
    return new, remove_in_orig


def reference_tiled_nuclei(inst_canvas, tile_shape=4096, margin=64, patch_output_shape=144, labeller=None):
    """The whole nuclei loop of infer/wsi.py:642-682 on one (H, W, 2) probability canvas -> sorted list of kept instance boxes."""
    H, W = inst_canvas.shape[:2]
    acc, uid = {}, 0
    for mode, (bounds, flags) in enumerate(get_tile_info((W, H), [tile_shape, tile_shape], margin, [patch_output_shape, patch_output_shape])):
        results = []
        for tb, tf in zip(bounds, flags):  # all tiles of a set see the dictionary as it was BEFORE the set (futures are merged after)
            results.append(process_tile(inst_canvas, tb, tf, mode, acc, margin, labeller))
        for new, rem in results:
            for b in new.values():
                acc[uid] = b
                uid += 1
            for r in rem:
                acc.pop(r, None)
    return sorted(tuple(int(v) for v in b) for b in acc.values())


def whole_map_nuclei(inst_canvas):
    lab = pr.proc(np.ascontiguousarray(inst_canvas), "Nuclei").astype(np.int64)
    return sorted(tuple(int(v) for v in b) for b in _inst_boxes(lab).values())
