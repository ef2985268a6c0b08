"""The REFERENCE's tiled nuclei post-processing of a slide (`--reference_tiling`), on the GPU.

By default this package labels a slide band in one pass with exact ownership of every instance (cerberus_amd/shard_postproc.py); the
reference instead post-processes the nuclei canvas in 4096 x 4096 tiles and repairs the seams with three more tile sets
(infer/wsi.py:642-684): vertical strips and horizontal strips astride the inner tile edges and `cross sections` at the inner corners,
each with its own rule for which instances of the tile are dropped and which of the already accumulated ones are removed
(`_process_tile_predictions`, infer/wsi.py:81-268).  That scheme loses the few instances that lie wholly inside a margin zone and touch
the edge line of the strip that should have re-found them; the band scheme keeps them -- so the two results are NOT identical, and a user
who needs the reference's own instance set bit for bit asks for it here.

What runs where: every tile is labelled by cerb_postproc_nuclei with `exact_ties` on (the tile is exactly what the reference hands to
skimage, so its heap order is reproducible) and turned into the instance dictionary by cerb_inst_table / cerb_inst_contour_*; the margin
logic is a few vectorised closed-interval box tests on the host (shapely's `STRtree.query` = envelope intersection, touching included;
`box.contains(box)`).  Tile sets follow tiatoolbox 1.3.1 `NucleusInstanceSegmentor._get_tile_info` (un-vendored in the reference's tree,
absent from this image: restated, unpinned -- the test-suite compares this module with an independent CPU restatement of the same scheme).
"""
from collections import OrderedDict

import numpy as np
import torch

from .postproc import get_inst_info_dict, postproc_device


def _grid(image_wh, tile_wh):
    """Output boxes [x0, y0, x1, y1] of a regular tile grid from the origin; the last row / column may reach past the image."""
    w, h = image_wh
    tw, th = int(tile_wh[0]), int(tile_wh[1])
    xs = np.arange(max(-(-w // tw), 1), dtype=np.int64) * tw
    ys = np.arange(max(-(-h // th), 1), dtype=np.int64) * th
    gx, gy = np.meshgrid(xs, ys)
    gx, gy = gx.ravel(), gy.ravel()
    return np.stack([gx, gy, gx + tw, gy + th], axis=1)


def _hits(boxes, bounds):
    """closed-interval intersection of every box [n, 4] with one box (touching counts)"""
    return (boxes[:, 0] <= bounds[2]) & (bounds[0] <= boxes[:, 2]) & (boxes[:, 1] <= bounds[3]) & (bounds[1] <= boxes[:, 3])


def _inside(boxes, bounds):
    return (bounds[0] <= boxes[:, 0]) & (bounds[1] <= boxes[:, 1]) & (boxes[:, 2] <= bounds[2]) & (boxes[:, 3] <= bounds[3])


def get_tile_info(image_wh, tile_shape, margin, patch_output_shape):
    """[(boxes [n, 4], removal flags [n, 4] = (top, bottom, left, right))] for tile modes 0 (grid), 1 (vertical strips), 2 (horizontal
    strips), 3 (cross sections); a slide that fits one tile has the grid set alone with no flag raised."""
    w, h = int(image_wh[0]), int(image_wh[1])
    pos = np.array(patch_output_shape, dtype=np.int64).reshape(-1)
    pos = np.array([pos[0], pos[-1]])
    tile = (np.array(tile_shape, dtype=np.int64).reshape(-1)[:2] // pos) * pos  # whole output patches per tile
    grid = _grid((w, h), tile)
    if w <= tile[0] and h <= tile[1]:
        return [(grid, np.zeros((len(grid), 4), np.int64))]
    image_edges = [(0, 0, w, 0), (0, h, w, h), (0, 0, 0, h), (w, 0, w, h)]  # top, bottom, left, right

    def clear_at_image_edges(boxes, flags):
        for side, edge in enumerate(image_edges):
            flags[_hits(boxes, edge), side] = 0
        return flags

    flags = clear_at_image_edges(grid, np.ones((len(grid), 4), np.int64))
    info = [(grid, flags)]
    right_inner = np.nonzero(flags[:, 3])[0]  # a vertical strip astride every right edge that lies inside the slide
    vb = np.stack([grid[right_inner, 2] - margin, grid[right_inner, 1], grid[right_inner, 2] + margin, grid[right_inner, 3]], axis=1)
    vf = np.zeros((len(vb), 4), np.int64)
    vf[:, :2] = 1
    info.append((vb, clear_at_image_edges(vb, vf)))
    bottom_inner = np.nonzero(flags[:, 1])[0]  # a horizontal strip astride every bottom edge inside the slide
    hb = np.stack([grid[bottom_inner, 0], grid[bottom_inner, 3] - margin, grid[bottom_inner, 2], grid[bottom_inner, 3] + margin], axis=1)
    hf = np.zeros((len(hb), 4), np.int64)
    hf[:, 2:] = 1
    info.append((hb, clear_at_image_edges(hb, hf)))
    corner = np.nonzero(flags[:, 1] * flags[:, 3])[0]  # a square of four margins around every inner bottom-right corner
    cb = np.stack([grid[corner, 2] - 2 * margin, grid[corner, 3] - 2 * margin, grid[corner, 2] + 2 * margin, grid[corner, 3] + 2 * margin], axis=1)
    info.append((cb, np.ones((len(cb), 4), np.int64)))
    return info


def _tile_instances(inst_canvas, type_canvas, bounds, exact_ties, y_off=0, slide_hw=None):
    """Label one tile on the GPU -> (list of per-instance dictionaries in TILE coordinates, boxes [n, 4] as x0, y0, x1, y1).
    inst_canvas holds slide rows y_off .. y_off + rows (a rank's band plus the rows it fetched from its neighbours) and may be wider than the
    slide (canvases are whole output patches wide); slide_hw = the slide's own height / width, the clip of every tile."""
    H, W = (int(inst_canvas.shape[0]) + int(y_off), int(inst_canvas.shape[1])) if slide_hw is None else (int(slide_hw[0]), int(slide_hw[1]))
    x0, y0, x1, y1 = max(int(bounds[0]), 0), max(int(bounds[1]), 0), min(int(bounds[2]), W), min(int(bounds[3]), H)
    if x1 <= x0 or y1 <= y0:
        return [], np.zeros((0, 4), np.int64)
    assert y0 >= y_off and y1 - y_off <= int(inst_canvas.shape[0]), "tile rows outside the rows this rank holds"
    lab, _ = postproc_device(inst_canvas[y0 - y_off:y1 - y_off, x0:x1], "Nuclei", exact_ties=exact_ties)
    tmap = None if type_canvas is None else type_canvas[y0 - y_off:y1 - y_off, x0:x1].contiguous()
    info = get_inst_info_dict(lab, tmap, flat_box=True)
    items = list(info.values())
    boxes = np.array([np.asarray(v["box"]) for v in items], dtype=np.int64).reshape(-1, 4)
    return items, boxes


def _tile_arrays(inst_canvas, type_canvas, bounds, exact_ties, y_off=0, slide_hw=None):
    """_tile_instances without a Python object per instance: the valid rows (area > 0, contour of >= 3 points: loader/postproc.py:34-35) of the tile's
    instance table, their contour runs re-packed, boxes [n, 4] as x0, y0, x1, y1 -- all in TILE coordinates.  -> (tab, cnts, pts, offs, boxes, has_type)"""
    from .postproc import inst_contours_device, inst_table_device

    H, W = (int(inst_canvas.shape[0]) + int(y_off), int(inst_canvas.shape[1])) if slide_hw is None else (int(slide_hw[0]), int(slide_hw[1]))
    x0, y0, x1, y1 = max(int(bounds[0]), 0), max(int(bounds[1]), 0), min(int(bounds[2]), W), min(int(bounds[3]), H)
    empty = (np.zeros((0, 16), np.int64), np.zeros(0, np.int64), np.zeros((0, 2), np.int64), np.zeros(0, np.int64), np.zeros((0, 4), np.int64), type_canvas is not None)
    if x1 <= x0 or y1 <= y0:
        return empty
    assert y0 >= y_off and y1 - y_off <= int(inst_canvas.shape[0]), "tile rows outside the rows this rank holds"
    lab, _ = postproc_device(inst_canvas[y0 - y_off:y1 - y_off, x0:x1], "Nuclei", exact_ties=exact_ties)
    tmap = None if type_canvas is None else type_canvas[y0 - y_off:y1 - y_off, x0:x1].contiguous()
    lab = lab.contiguous()
    tab_dev = inst_table_device(lab, tmap)
    cnts, pts, offs = inst_contours_device(lab, tab_dev)
    tab = tab_dev.cpu().numpy()
    cnts, offs = np.asarray(cnts, dtype=np.int64), np.asarray(offs, dtype=np.int64)
    valid = np.nonzero((tab[:, 0] > 0) & (cnts >= 3))[0] if len(tab) else np.zeros(0, np.int64)
    if valid.size == 0:
        return empty
    tab, cnts, offs = tab[valid], cnts[valid], offs[valid]
    idx = np.repeat(offs, cnts) + (np.arange(int(cnts.sum())) - np.repeat(np.cumsum(cnts) - cnts, cnts))  # the kept runs, packed
    pts = np.asarray(pts)[idx]
    offs = np.cumsum(cnts) - cnts
    return tab, cnts, pts, offs, tab[:, [5, 3, 6, 4]].astype(np.int64), type_canvas is not None


def process_tile_arrays(inst_canvas, type_canvas, tile_bounds, tile_flag, tile_mode, margin, exact_ties=True, y_off=0, slide_hw=None):
    """process_tile_predictions on arrays: the kept instances of one tile as (tab, cnts, pts, origin [n, 2]) -- table rows and contour points in
    TILE coordinates plus the tile's (x, y) origin per instance (cerberus_amd.inst_info places them) -- and their boxes in SLIDE coordinates."""
    tl = np.array([int(tile_bounds[0]), int(tile_bounds[1])], dtype=np.int64)
    w, h = int(tile_bounds[2]) - int(tile_bounds[0]), int(tile_bounds[3]) - int(tile_bounds[1])
    tab, cnts, pts, offs, boxes, has_type = _tile_arrays(inst_canvas, type_canvas, tile_bounds, exact_ties, y_off, slide_hw)
    if len(tab) == 0:
        return {"tab": tab, "cnts": cnts, "pts": pts, "origin": np.zeros((0, 2), np.int64), "boxes": boxes, "has_type": has_type}
    m = int(margin)
    boundary_lines = [(0, 0, w, 1), (0, h - 1, w, h), (0, 0, 1, h), (w - 1, 0, w, h)]
    margin_boxes = [(0, 0, w, m), (0, h - m, w, h), (0, 0, m, h), (w - m, 0, w, h)]
    drop = np.zeros(len(tab), bool)
    if tile_mode in (0, 3):
        for side, zone in enumerate(margin_boxes):
            if tile_flag[side] or tile_mode == 3:
                drop |= _hits(boxes, zone) & _inside(boxes, zone)
    elif tile_mode in (1, 2):
        for side, flag in enumerate(tile_flag):
            drop |= _hits(boxes, margin_boxes[side] if flag else boundary_lines[side])
    else:
        raise ValueError("Unknown tile mode %r." % (tile_mode,))
    keep = np.nonzero(~drop)[0]
    kc, ko = cnts[keep], offs[keep]
    idx = np.repeat(ko, kc) + (np.arange(int(kc.sum())) - np.repeat(np.cumsum(kc) - kc, kc))
    return {"tab": tab[keep], "cnts": kc, "pts": pts[idx], "origin": np.tile(tl, (len(keep), 1)), "boxes": boxes[keep] + np.concatenate([tl, tl]),
            "has_type": has_type}


def merge_tile_arrays(parts, tile_info, margin):
    """merge_tile_results on arrays -> the ("Nuclei", tab, cnts, pts, offs, has_type, 1.0, origin) tuple cerberus_amd.inst_info builds the
    dictionary (or the .dat stream) from: sets in order 0..3, tiles of a set in order, a cross section evicting what was accumulated before ITS SET
    and touches one of its inner margin lines."""
    chunks, alive, has_type = [], [], False
    for mode, (bounds, _) in enumerate(tile_info):
        n_before = len(chunks)
        ref_boxes = np.concatenate([c["boxes"] for c in chunks], axis=0) if (mode == 3 and chunks) else np.zeros((0, 4), np.int64)
        ref_alive = np.concatenate(alive) if (mode == 3 and alive) else np.zeros(0, bool)
        evict = np.zeros(len(ref_boxes), bool)
        cand = np.zeros(0, np.int64)
        if mode == 3 and len(ref_boxes) and len(bounds):
            xs = np.unique([int(b[0]) for b in bounds])
            ys = np.unique([int(b[1]) for b in bounds])
            wx, wy = int(bounds[0][2]) - int(bounds[0][0]), int(bounds[0][3]) - int(bounds[0][1])
            ix = np.searchsorted(xs, ref_boxes[:, 2], side="right") - 1
            iy = np.searchsorted(ys, ref_boxes[:, 3], side="right") - 1
            near = (ix >= 0) & (ref_boxes[:, 0] <= xs[np.maximum(ix, 0)] + wx) & (iy >= 0) & (ref_boxes[:, 1] <= ys[np.maximum(iy, 0)] + wy)
            cand = np.nonzero(near & ref_alive)[0]
        for ti in range(len(bounds)):
            c = parts.get((mode, ti))
            if mode == 3 and len(cand):
                sel = np.zeros(len(cand), bool)
                for line in eviction_lines(bounds[ti], margin):
                    sel |= _hits(ref_boxes[cand], line)
                evict[cand[sel]] = True
            if c is not None and len(c["tab"]):
                chunks.append(c)
                alive.append(np.ones(len(c["tab"]), bool))
                has_type = has_type or bool(c["has_type"])
        if mode == 3 and evict.any():
            pos = 0
            for j in range(n_before):
                k = len(alive[j])
                alive[j] &= ~evict[pos:pos + k]
                pos += k
    if not chunks:
        return ("Nuclei", np.zeros((0, 16), np.int64), np.zeros(0, np.int64), np.zeros((0, 2), np.int64), np.zeros(0, np.int64), has_type, 1.0, np.zeros((0, 2), np.int64))
    tabs, cntss, ptss, orgs = [], [], [], []
    for c, a in zip(chunks, alive):
        keep = np.nonzero(a)[0]
        if keep.size == len(a):
            tabs.append(c["tab"]); cntss.append(c["cnts"]); ptss.append(c["pts"]); orgs.append(c["origin"])
            continue
        kc = c["cnts"][keep]
        ko = (np.cumsum(c["cnts"]) - c["cnts"])[keep]
        idx = np.repeat(ko, kc) + (np.arange(int(kc.sum())) - np.repeat(np.cumsum(kc) - kc, kc))
        tabs.append(c["tab"][keep]); cntss.append(kc); ptss.append(c["pts"][idx]); orgs.append(c["origin"][keep])
    cnts = np.concatenate(cntss)
    return ("Nuclei", np.concatenate(tabs), cnts, np.concatenate(ptss), np.cumsum(cnts) - cnts, has_type, 1.0, np.concatenate(orgs))


def nuclei_dict_from_part(part):
    """{uuid4 hex -> {'box': [x1, y1, x2, y2], 'centroid', 'contour', 'type', 'type_prob'}} of a merge_tile_arrays result (slide coordinates)."""
    from .inst_info import _uuid4_hex, info_from_table

    info = info_from_table(part[1], part[2], part[3], part[4], bool(part[5]), 1.0, flat_box=True, origin=part[7])
    return OrderedDict(zip(_uuid4_hex(len(info)), info.values()))


def eviction_lines(tile_bounds, margin):
    """The four inner margin lines of a cross section in slide coordinates (infer/wsi.py:241-255): accumulated instances touching one are removed."""
    tl = (int(tile_bounds[0]), int(tile_bounds[1]))
    w, h = int(tile_bounds[2]) - tl[0], int(tile_bounds[3]) - tl[1]
    m = int(margin)
    return [(a + tl[0], b + tl[1], c + tl[0], d + tl[1]) for a, b, c, d in ((m, m, w - m, m), (m, h - m, w - m, h - m), (m, m, m, h - m), (w - m, m, w - m, h - m))]


def process_tile_predictions(inst_canvas, type_canvas, tile_bounds, tile_flag, tile_mode, ref_boxes, margin, exact_ties=True, y_off=0, slide_hw=None):
    """infer/wsi.py:81-268 for one tile.  ref_boxes: [k, 4] boxes (slide coordinates) of what has been accumulated before this tile SET, or None
    when the caller evicts at merge time (merge_tile_results).  -> (kept instance dictionaries in slide coordinates, indices into ref_boxes to remove)."""
    tl = np.array([int(tile_bounds[0]), int(tile_bounds[1])], dtype=np.int64)
    w, h = int(tile_bounds[2]) - int(tile_bounds[0]), int(tile_bounds[3]) - int(tile_bounds[1])
    items, boxes = _tile_instances(inst_canvas, type_canvas, tile_bounds, exact_ties, y_off, slide_hw)
    remove = np.zeros(0, np.int64)
    if tile_mode == 3 and ref_boxes is not None and len(ref_boxes):  # a cross section also evicts accumulated instances touching its inner margin lines
        sel = np.zeros(len(ref_boxes), bool)
        for line in eviction_lines(tile_bounds, margin):
            sel |= _hits(ref_boxes, line)
        remove = np.nonzero(sel)[0]
    if not items:
        return [], remove
    m = int(margin)
    boundary_lines = [(0, 0, w, 1), (0, h - 1, w, h), (0, 0, 1, h), (w - 1, 0, w, h)]
    margin_boxes = [(0, 0, w, m), (0, h - m, w, h), (0, 0, m, h), (w - m, 0, w, h)]
    drop = np.zeros(len(items), bool)
    if tile_mode in (0, 3):  # grid tiles / cross sections: instances lying wholly inside a flagged margin zone
        for side, zone in enumerate(margin_boxes):
            if tile_flag[side] or tile_mode == 3:
                drop |= _hits(boxes, zone) & _inside(boxes, zone)
    elif tile_mode in (1, 2):  # strips: everything touching a flagged margin zone, or the one-pixel line of an unflagged side
        for side, flag in enumerate(tile_flag):
            drop |= _hits(boxes, margin_boxes[side] if flag else boundary_lines[side])
    else:
        raise ValueError("Unknown tile mode %r." % (tile_mode,))
    off = np.concatenate([tl, tl])
    kept = []
    for i in np.nonzero(~drop)[0].tolist():
        v = dict(items[i])
        v["box"] = np.asarray(v["box"]) + off
        v["centroid"] = np.asarray(v["centroid"]) + tl
        v["contour"] = np.asarray(v["contour"]) + tl
        kept.append(v)
    return kept, remove


def merge_tile_results(parts, tile_info, margin):
    """parts: {(mode, tile index): kept dictionaries}, from one rank or gathered from all.  The accumulation of infer/wsi.py:642-684: sets in
    order 0..3, tiles of a set in order; a cross section (set 3) evicts what had been accumulated BEFORE its set and touches one of its inner
    margin lines (every tile of a set sees the dictionary as it was when the set was submitted), then adds its own instances."""
    from .inst_info import _uuid4_hex

    acc = OrderedDict()
    for mode, (bounds, _) in enumerate(tile_info):
        keys = list(acc.keys())
        ref_boxes = np.array([np.asarray(acc[k]["box"]) for k in keys], dtype=np.int64).reshape(-1, 4) if (mode == 3 and keys) else np.zeros((0, 4), np.int64)
        cand = np.zeros(0, np.int64)
        if mode == 3 and len(ref_boxes) and len(bounds):
            # candidates once for the whole set: an instance can only touch a cross section's lines if it reaches within two margins of an
            # inner tile corner in x AND in y -- a fraction of a per cent of a slide's instances; the per-tile tests then run on those
            xs = np.unique([int(b[0]) for b in bounds])
            ys = np.unique([int(b[1]) for b in bounds])
            wx, wy = int(bounds[0][2]) - int(bounds[0][0]), int(bounds[0][3]) - int(bounds[0][1])
            ix = np.searchsorted(xs, ref_boxes[:, 2], side="right") - 1  # the last cross-section column starting at or before the box's right edge
            iy = np.searchsorted(ys, ref_boxes[:, 3], side="right") - 1
            near = (ix >= 0) & (ref_boxes[:, 0] <= xs[np.maximum(ix, 0)] + wx) & (iy >= 0) & (ref_boxes[:, 1] <= ys[np.maximum(iy, 0)] + wy)
            cand = np.nonzero(near)[0]
        for ti in range(len(bounds)):
            kept = parts.get((mode, ti), [])
            if mode == 3 and len(cand):
                sel = np.zeros(len(cand), bool)
                for line in eviction_lines(bounds[ti], margin):
                    sel |= _hits(ref_boxes[cand], line)
                for r in cand[sel].tolist():
                    acc.pop(keys[r], None)
            for k, v in zip(_uuid4_hex(len(kept)), kept):  # (one os.urandom call per tile: uuid.uuid4() per instance was 1.3 s of a slide's merge)
                acc[k] = v
    return acc


def _rank_tiles(tile_info, band_bounds, slide_h):
    """Owner of every tile = the rank whose band holds the tile's first slide row.  -> per rank [(mode, tile index)], per rank last row needed."""
    world = len(band_bounds)
    starts = np.array([b[0] for b in band_bounds], dtype=np.int64)
    mine = [[] for _ in range(world)]
    need_hi = [int(b[1]) for b in band_bounds]
    for mode, (bounds, _) in enumerate(tile_info):
        for ti, b in enumerate(bounds):
            y0 = min(max(int(b[1]), 0), slide_h - 1)
            r = int(np.searchsorted(starts, y0, side="right") - 1)
            mine[r].append((mode, ti))
            need_hi[r] = max(need_hi[r], min(int(b[3]), slide_h))
    return mine, need_hi


def reference_tiled_nuclei(inst_canvas, type_canvas=None, tile_shape=4096, margin=64, patch_output_shape=144, exact_ties=True, prof=None, as_part=False):
    """The nuclei loop of infer/wsi.py:642-684 over a device-resident INST canvas [H, W, 2] (and uint8 TYPE canvas [H, W]):
    OrderedDict {uuid4 hex -> {'box': [x1, y1, x2, y2], 'centroid', 'contour', 'type', 'type_prob'}} in slide coordinates."""
    return reference_tiled_nuclei_sharded(inst_canvas, type_canvas, 0, (int(inst_canvas.shape[0]), int(inst_canvas.shape[1])), 0, 1, None, tile_shape, margin,
                                          patch_output_shape, exact_ties, prof=prof, as_part=as_part)


def reference_tiled_nuclei_sharded(band_inst, band_type, band_y0, slide_hw, rank, world, dist, tile_shape=4096, margin=64, patch_output_shape=144,
                                   exact_ties=True, watch=None, prof=None, as_part=False):
    """The same over N ranks that each hold a band of the slide's canvases (rows band_y0 .. band_y0 + band_inst.shape[0]): tiles are
    independent, so every rank labels the tiles whose first row lies in its band -- after fetching the rows those tiles reach into below its
    band from the ranks that hold them (point-to-point, upwards only: a tile never starts above its owner's band) -- and the root merges the
    kept instances in the reference's order, applying the cross sections' evictions there (they look at instances of any rank).
    Returns the dictionary on rank 0, None elsewhere.  prof: dict receiving seconds per phase.
    Tiles and merge work on ARRAYS (instance-table rows, contour runs, boxes): no Python object per instance until -- unless as_part -- the
    dictionary is built once at the end; as_part=True returns the ("Nuclei", tab, cnts, pts, offs, has_type, 1.0, origin) tuple instead, which
    cerberus_amd.inst_info turns into the `.dat` stream directly (write_dat_fast) or into the same dictionary (nuclei_dict_from_part)."""
    import time

    from .launch import null_watch

    watch = watch or null_watch()
    assert band_inst.is_cuda and band_inst.dim() == 3
    H, W = int(slide_hw[0]), int(slide_hw[1])
    rows = min(int(band_inst.shape[0]), H - int(band_y0))
    shape2 = [tile_shape, tile_shape] if np.isscalar(tile_shape) else tile_shape
    pos2 = [patch_output_shape, patch_output_shape] if np.isscalar(patch_output_shape) else patch_output_shape
    tile_info = get_tile_info((W, H), shape2, margin, pos2)
    t0 = time.perf_counter()
    if dist is None or world == 1:
        band_bounds = [(0, H)]
        assert int(band_y0) == 0 and rows == H, "a single rank holds the whole canvas"
    else:
        me = torch.tensor([int(band_y0), int(band_y0) + rows], dtype=torch.int64, device=band_inst.device)
        allb = [torch.zeros_like(me) for _ in range(world)]
        with watch.phase("reference tiling: band-bounds all-gather"):
            dist.all_gather(allb, me)
        band_bounds = [(int(b[0].item()), int(b[1].item())) for b in allb]
    mine, need_hi = _rank_tiles(tile_info, band_bounds, H)
    y0, y1 = band_bounds[rank]
    canvas, tcanvas = band_inst[:rows], (None if band_type is None else band_type[:rows])
    if world > 1 and dist is not None:
        ext = max(need_hi[rank], y1) - y0
        if ext > rows:
            big = torch.empty((ext, W, 2), dtype=band_inst.dtype, device=band_inst.device)
            big[:rows] = canvas[:, :W]
            canvas = big
            if tcanvas is not None:
                tb = torch.empty((ext, W), dtype=tcanvas.dtype, device=tcanvas.device)
                tb[:rows] = tcanvas[:, :W]
                tcanvas = tb
        with watch.phase("reference tiling: canvas rows from the ranks below"):
            for d in range(1, world):
                ops = []
                src = rank + d
                if src < world:  # rows of `src`'s band that my tiles reach into
                    a, b = max(band_bounds[src][0], y1), min(band_bounds[src][1], need_hi[rank])
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, canvas[a - y0:b - y0], src))
                        if tcanvas is not None:
                            ops.append(dist.P2POp(dist.irecv, tcanvas[a - y0:b - y0], src))
                dst = rank - d
                if dst >= 0:
                    a, b = max(y0, band_bounds[dst][1]), min(y1, need_hi[dst])
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, band_inst[a - y0:b - y0, :W].contiguous(), dst))
                        if band_type is not None:
                            ops.append(dist.P2POp(dist.isend, band_type[a - y0:b - y0, :W].contiguous(), dst))
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
    t1 = time.perf_counter()
    parts = {}
    for mode, ti in mine[rank]:
        bounds, flags = tile_info[mode]
        parts[(mode, ti)] = process_tile_arrays(canvas, tcanvas, bounds[ti], flags[ti], mode, margin, exact_ties, y_off=y0, slide_hw=(H, W))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if world > 1 and dist is not None:
        lst = [None] * world if rank == 0 else None
        with watch.phase("reference tiling: per-tile instance arrays to rank 0"):
            dist.gather_object(parts, lst, dst=0)
        if rank != 0:
            if prof is not None:
                prof.update(exchange_s=t1 - t0, tiles_s=t2 - t1, tiles=len(mine[rank]))
            return None
        parts = {}
        for p in lst:
            parts.update(p)
    part = merge_tile_arrays(parts, tile_info, margin)
    t3 = time.perf_counter()
    acc = part if as_part else nuclei_dict_from_part(part)
    if prof is not None:
        prof.update(exchange_s=t1 - t0, tiles_s=t2 - t1, merge_s=t3 - t2, dictionary_s=time.perf_counter() - t3, instances=int(len(part[1])),
                    tiles=len(mine[rank]), tiles_total=sum(len(b) for b, _ in tile_info))
    return acc
