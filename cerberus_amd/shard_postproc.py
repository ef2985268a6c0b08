"""Band-local post-processing for the multi-GPU slide path.

The reference post-processes a slide in 4096^2 tiles and repairs the seams with 64-px margins, boundary strips and shapely
de-duplication (infer/wsi.py:81-268), and names instances with uuid4 (infer/wsi.py:265,831).  Here every rank owns one
contiguous band of canvas rows and labels it in ONE pass:

  1. halo exchange: each rank receives `margin` rows of the probability canvas from the rank above and below (two
     point-to-point messages per rank over the direct xGMI link; the only data-path traffic of this stage);
  2. local labelling of window = halo + band + halo with the ordinary single-GPU kernels (cerb_postproc_*), instance table
     (cerb_inst_table: bounding rows + first pixel of every instance);
  3. ownership: an instance belongs to the rank whose band contains its FIRST pixel in raster order (its top row), so
     every instance has exactly one owner and only ever extends downwards out of its owner's band;
  4. one all-gather of the per-rank owned counts -> exclusive prefix sum -> slide-global ids that are unique and ordered by
     (band, first pixel); one all-gather of the few (first pixel -> global id) pairs of instances that cross a band's lower
     edge, so the rank below names its part of them identically;
  5. cerb_relabel writes the band's rows with global ids.

An instance whose bounding rows come within `guard` rows of an artificial window edge may be cut (or, for glands, padded
differently before dilation); those are counted in info['n_truncated'] -- 0 means the band result is provably the one a
single GPU computes on the whole slide (up to the id bijection).  Choose margin >= tallest instance + guard.

The phase functions are pure per-rank steps; `run_local` chains them for ranks simulated in one process (GPU test),
`run_distributed` uses torch.distributed (RCCL on the GPU box, gloo in the CPU tests).  `label_fn` / `table_fn` default to the
HIP kernels and fail loudly without a GPU; the gloo test injects numpy stand-ins to exercise the protocol itself."""
from collections import OrderedDict

import numpy as np
import torch


def _device_label_fn(window, tissue, ds_factor):
    from .postproc import postproc_device

    lab, info = postproc_device(window, tissue, ds_factor, exact_ties=False)  # WSI bands: see postproc_device
    n = int(info["n_inst"].item())
    return lab, max(n, 0)


def _device_table_fn(lab, n):
    from .postproc import inst_table_device

    return inst_table_device(lab, None, n)


def _device_relabel_fn(lab_rows, mapping):
    import ctypes as C

    from . import _lib

    out = torch.empty_like(lab_rows)
    stream = torch.cuda.current_stream(lab_rows.device).cuda_stream
    with torch.cuda.device(lab_rows.device):
        _lib.check(_lib.lib().cerb_relabel(lab_rows.data_ptr(), lab_rows.stride(0), mapping.data_ptr(), int(mapping.numel()),
                                           int(lab_rows.shape[0]), int(lab_rows.shape[1]), out.data_ptr(), out.stride(0), C.c_void_p(stream)))
    return out


def _device_arrays_fn(lab, n, type_window, owned):
    """cerb_inst_table (with the class votes) + cerb_inst_contour_* over a rank's WINDOW label map, for the instances it owns (`owned`: 0-based
    local ids; the other rows are zeroed, which is how the contour kernels skip an id).  -> host arrays (tab [n, 16], cnts [n], pts [P, 2], offs [n])."""
    from .postproc import inst_contours_device, inst_table_device

    tab = inst_table_device(lab, type_window, n)
    keep = torch.zeros(max(n, 0), dtype=torch.bool, device=lab.device)
    if len(owned):
        keep[torch.from_numpy(np.ascontiguousarray(owned, dtype=np.int64)).to(lab.device)] = True
    tab = tab * keep[:, None].to(tab.dtype)
    cnts, pts, offs = inst_contours_device(lab, tab)
    return tab.cpu().numpy(), cnts, pts, offs


def _device_mask_fn(lumen_window, gland_rows):
    from .postproc import mask_lumen_by_gland

    mask_lumen_by_gland(lumen_window, gland_rows)


class BandState(object):
    """Per-rank, per-tissue state carried between the phases."""

    def __init__(self, rank, world, band, y0_global, margin, guard, tissue, ds_factor=1.0, type_band=None):
        assert band.dim() == 3 and band.shape[2] == 2, "band: (rows, W, 2) probability canvas of this rank"
        self.rank, self.world = rank, world
        self.band = band
        # the tissue's class map over the same rows (uint8, at the band's resolution), or None: its halo rows travel with the probability halos so
        # that the owner of an instance can take the instance's class votes over ALL of its pixels (owned_parts)
        self.type_band = type_band
        self.type_window = None
        assert type_band is None or tuple(type_band.shape) == tuple(band.shape[:2]), "type_band: (rows, W) class ids over the band's rows"
        self.y0 = int(y0_global)  # global canvas row of band row 0 (at the resolution of `band`)
        self.margin, self.guard = int(margin), int(guard)
        self.tissue, self.ds = tissue, float(ds_factor)
        if world > 1:
            assert band.shape[0] >= margin, "band shorter than the halo margin: use fewer ranks or a smaller margin"

    # ---- phase 1: what the neighbours need -------------------------------------------------------------------------
    def strips(self):
        """(rows for the rank above = my first `margin` rows, rows for the rank below = my last `margin` rows)"""
        up = self.band[: self.margin].contiguous() if self.rank > 0 else None
        down = self.band[self.band.shape[0] - self.margin:].contiguous() if self.rank < self.world - 1 else None
        return up, down

    def type_strips(self):
        """The same rows of the class map (None without one)."""
        if self.type_band is None:
            return None, None
        up = self.type_band[: self.margin].contiguous() if self.rank > 0 else None
        down = self.type_band[self.type_band.shape[0] - self.margin:].contiguous() if self.rank < self.world - 1 else None
        return up, down

    def set_type_window(self, type_above, type_below):
        if self.type_band is None:
            return
        parts = [p for p in (type_above, self.type_band, type_below) if p is not None]
        tw = torch.cat(parts, dim=0) if torch.is_tensor(self.type_band) else np.concatenate([np.asarray(p) for p in parts], axis=0)
        self.type_window = tw.contiguous() if torch.is_tensor(tw) else np.ascontiguousarray(tw)

    # ---- phase 2: label the window ------------------------------------------------------------------------------------
    def label(self, from_above, from_below, label_fn=_device_label_fn, table_fn=_device_table_fn):
        parts = [p for p in (from_above, self.band, from_below) if p is not None]
        window = torch.cat(parts, dim=0) if len(parts) > 1 else self.band
        self.top = 0 if from_above is None else int(from_above.shape[0])
        self.h_band = int(self.band.shape[0])
        self.h_win, self.w = int(window.shape[0]), int(window.shape[1])
        self.lab, self.n = label_fn(window, self.tissue, self.ds)
        if self.n > 0:
            tab = table_fn(self.lab, self.n)
            tab = tab.cpu().numpy() if torch.is_tensor(tab) else np.asarray(tab)
        else:
            tab = np.zeros((0, 16), np.int64)
        area, y1, y2, first = tab[:, 0], tab[:, 3], tab[:, 4], tab[:, 7]
        self.y1, self.y2, self.x1, self.x2 = y1, y2, tab[:, 5], tab[:, 6]
        alive = area > 0
        fy = first // self.w
        # global key of an instance: its first pixel in slide raster order
        self.key = (fy - self.top + self.y0) * self.w + (first % self.w)
        self.owned = alive & (fy >= self.top) & (fy < self.top + self.h_band)
        self.in_band = alive & (y2 > self.top) & (y1 < self.top + self.h_band)
        art_top, art_bot = from_above is not None, from_below is not None
        cut = np.zeros(len(tab), bool)
        if art_top:
            cut |= y1 < self.guard
        if art_bot:
            cut |= y2 > self.h_win - self.guard
        self.cut = cut & alive
        self.n_truncated = int((cut & self.in_band).sum())
        # rank-local order of the owned instances = order of their first pixels
        own_idx = np.nonzero(self.owned)[0]
        self.own_sorted = own_idx[np.argsort(self.key[own_idx], kind="stable")]
        self.n_owned = int(len(own_idx))
        return self.n_owned

    # ---- phase 3: global ids -------------------------------------------------------------------------------------------
    def publish(self, offset):
        """Global ids of my owned instances (offset = owned counts of the ranks above); returns the int64 [k, 2] table
        (key, global id) of the owned instances that extend below my band -- the only ones another rank has to name."""
        self.gid = np.zeros(len(self.key), np.int64)
        self.gid[self.own_sorted] = offset + 1 + np.arange(self.n_owned)
        crossing = self.owned & (self.y2 > self.top + self.h_band)
        return np.stack([self.key[crossing], self.gid[crossing]], axis=1).astype(np.int64).reshape(-1, 2)

    def resolve(self, published_from_above, relabel_fn=_device_relabel_fn, defer=None):
        """Name the instances I see but do not own (their owner is a rank above), relabel my band rows.
        defer: a list -- instances whose owner has not published yet are not counted as unresolved: their keys are appended to it and their pixels
        get the NEGATIVE placeholder -(position in the list + 1), to be renamed by the caller (cerberus_amd.stream_bands: the first sub-band of
        a rank whose neighbour above is still walking its band)."""
        lut = {}
        for tab in published_from_above:
            for k, g in np.asarray(tab).reshape(-1, 2):
                lut[int(k)] = int(g)
        foreign = np.nonzero(self.in_band & ~self.owned)[0]
        unresolved = 0
        for i in foreign:
            g = lut.get(int(self.key[i]))
            if g is None and defer is not None:
                defer.append(int(self.key[i]))
                self.gid[i] = -len(defer)
            elif g is None:
                unresolved += 1
            else:
                self.gid[i] = g
        self.n_unresolved = unresolved
        mapping = np.zeros(len(self.key) + 1, np.int32)
        mapping[1:] = self.gid
        rows = self.lab[self.top: self.top + self.h_band]
        if torch.is_tensor(rows):
            out = relabel_fn(rows, torch.from_numpy(mapping).to(rows.device))
        else:
            out = relabel_fn(rows, mapping)
        info = {"n_owned": self.n_owned, "n_truncated": self.n_truncated, "n_unresolved": self.n_unresolved}
        return out, info

    # ---- lumen *= (gland > 0) on the WINDOWS (infer/wsi.py:799-804), before the band rows are cut out of them ------------------------
    def mask_by(self, gland, mask_fn=_device_mask_fn):
        """self: the lumen state, gland: the gland state of the same band (both labelled).  The lumen window's rows are taken out of the gland
        window (its halo is at least as tall), so that an owned lumen is masked over ALL of its pixels -- the halo part included, which is where
        its table and contour come from (owned_parts).  A gland the window cut could mask differently from the whole-slide labelling inside its
        bounding box (a hole the cut opened is not filled): every OWNED lumen whose box meets the box of such a gland is added to the lumen's
        n_truncated (in-band rows are covered by the gland's own count)."""
        d = gland.top - self.top
        if d < 0 or d + self.h_win > gland.h_win or gland.w != self.w or gland.y0 != self.y0:
            raise ValueError("lumen masking on windows needs the gland halo to cover the lumen halo (margins: gland >= lumen)")
        rows = gland.lab[d: d + self.h_win]
        mask_fn(self.lab, rows)
        own = np.nonzero(self.owned)[0]
        hit = np.zeros(len(own), bool)
        for g in np.nonzero(gland.cut)[0]:
            hit |= (self.y1[own] + d < gland.y2[g]) & (self.y2[own] + d > gland.y1[g]) & (self.x1[own] < gland.x2[g]) & (self.x2[own] > gland.x1[g])
        self.n_truncated += int(hit.sum())

    # ---- per-rank instance arrays: what the root needs for dat/<slide>.dat, without the label maps -----------------
    def owned_parts(self, arrays_fn=_device_arrays_fn):
        """(tab int64 [k, 16], cnts int32 [k], pts int32 [P, 2]) of the k instances this rank OWNS, in the order of their slide-global ids and in
        SLIDE coordinates (of this tissue's resolution): cerb_inst_table rows (class votes over the whole instance: the window holds all of it
        when n_truncated == 0) and the contour runs, shifted from window rows to slide rows in the integer domain -- sum_y += area * dy,
        y1 / y2 += dy, first += dy * w, contour y += dy -- so that the ranks' arrays, concatenated in rank order, ARE the arrays a one-GPU run
        computes on the whole label map (ids are ordered by (band, first pixel) = the raster order of the first pixels)."""
        order = np.asarray(self.own_sorted, dtype=np.int64)
        if self.n <= 0 or order.size == 0:
            return np.zeros((0, 16), np.int64), np.zeros(0, np.int32), np.zeros((0, 2), np.int32)
        tab, cnts, pts, offs = arrays_fn(self.lab, self.n, self.type_window, order)
        tab, cnts, pts, offs = np.asarray(tab), np.asarray(cnts), np.asarray(pts).reshape(-1, 2), np.asarray(offs)
        dy = int(self.y0 - self.top)
        sel = tab[order].astype(np.int64, copy=True)
        alive = sel[:, 0] > 0  # (a lumen its gland masked away entirely stays as an all-zero row: its id exists, its entry does not)
        sel[~alive] = 0
        sel[alive, 2] += sel[alive, 0] * dy
        sel[alive, 3] += dy
        sel[alive, 4] += dy
        sel[alive, 7] += dy * self.w
        c = cnts[order].astype(np.int64)
        c[~alive] = 0
        total = int(c.sum())
        new_offs = np.cumsum(c) - c
        idx = np.repeat(offs[order].astype(np.int64) - new_offs, c) + np.arange(total, dtype=np.int64)
        p = pts[idx].astype(np.int32, copy=True) if total else np.zeros((0, 2), np.int32)
        p[:, 1] += dy
        return sel, c.astype(np.int32), p


def run_local(bands, tissue, margin, guard, ds_factor=1.0, label_fn=_device_label_fn, table_fn=_device_table_fn,
              relabel_fn=_device_relabel_fn):
    """All ranks simulated in one process: `bands` = list of (rows_r, W, 2) tensors in slide order.
    Returns (list of int32 band label maps with global ids, total instance count, list of per-rank info)."""
    world = len(bands)
    y0 = np.concatenate([[0], np.cumsum([b.shape[0] for b in bands])]).astype(np.int64)
    states = [BandState(r, world, bands[r], y0[r], margin, guard, tissue, ds_factor) for r in range(world)]
    strips = [s.strips() for s in states]
    counts = []
    for r, s in enumerate(states):
        above = strips[r - 1][1] if r > 0 else None          # the rank above sends its last rows down
        below = strips[r + 1][0] if r < world - 1 else None  # the rank below sends its first rows up
        counts.append(s.label(above, below, label_fn, table_fn))
    offs = np.concatenate([[0], np.cumsum(counts)])
    pubs = [s.publish(int(offs[r])) for r, s in enumerate(states)]
    outs, infos = [], []
    for r, s in enumerate(states):
        o, i = s.resolve(pubs[:r], relabel_fn)
        outs.append(o)
        infos.append(i)
    return outs, int(offs[-1]), infos


class IncrementalLocalLabeller(object):
    """run_local for ONE tissue on ONE GPU, fed while the rows further down are still being inferred: a local band is labelled (on a side stream)
    as soon as its own rows and the halo rows below it are final, so that of the slide's tail only the last band, the id protocol and the
    relabelling remain when the inference ends.  Same bands, same protocol, same results as run_local -- only earlier.
        lab = IncrementalLocalLabeller(band, "Nuclei", margin, guard, max_band_px)
        lab.feed(rows_final, events)     after some inference was queued: `events` complete => canvas rows [0, rows_final) are final
        outs, n, infos = lab.finish()    on the caller's stream (waits for the side stream)"""

    def __init__(self, band, tissue, margin, guard, max_band_px, ds_factor=1.0, label_fn=_device_label_fn, table_fn=_device_table_fn,
                 relabel_fn=_device_relabel_fn):
        rows, cols = int(band.shape[0]), int(band.shape[1])
        self.nb = local_band_count(rows, cols, max_band_px, margin)
        self.cuts = [int(round(i * rows / self.nb)) for i in range(self.nb + 1)]
        self.margin = int(margin)
        self.states = [BandState(r, self.nb, band[self.cuts[r]:self.cuts[r + 1]], self.cuts[r], margin, guard, tissue, ds_factor) for r in range(self.nb)]
        self.counts = [None] * self.nb
        self.done = 0
        self.fns = (label_fn, table_fn, relabel_fn)
        self.side = torch.cuda.Stream(band.device) if band.is_cuda else None
        self.early = 0  # bands labelled before finish()

    def rows_needed(self, r):
        return self.cuts[r + 1] + (self.margin if r < self.nb - 1 else 0)

    def _label(self, r):
        above = self.states[r - 1].strips()[1] if r > 0 else None
        below = self.states[r + 1].strips()[0] if r < self.nb - 1 else None
        self.counts[r] = self.states[r].label(above, below, self.fns[0], self.fns[1])

    def feed(self, rows_final, events=()):
        while self.done < self.nb and self.rows_needed(self.done) <= rows_final:
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    for e in events:
                        self.side.wait_event(e)
                    self._label(self.done)
            else:
                self._label(self.done)
            self.done += 1
            self.early += 1

    def finish(self):
        if self.side is not None:
            cur = torch.cuda.current_stream(self.side.device)
            cur.wait_stream(self.side)
            for s in self.states[: self.done]:
                # labelled on the side stream, relabelled and handed on on the caller's: without this the caching allocator may give a freed window back to
                # the side stream while the caller's stream still reads it
                if torch.is_tensor(getattr(s, "lab", None)) and s.lab.is_cuda:
                    s.lab.record_stream(cur)
        for r in range(self.done, self.nb):
            self._label(r)
        self.done = self.nb
        offs = np.concatenate([[0], np.cumsum(self.counts)])
        pubs = [s.publish(int(offs[r])) for r, s in enumerate(self.states)]
        outs, infos = [], []
        for r, s in enumerate(self.states):
            o, i = s.resolve(pubs[:r], self.fns[2])
            outs.append(o)
            infos.append(i)
        return outs, int(offs[-1]), infos


def make_incremental(band_canv, dist, wsi_mode=True, margin=512, guard=48, max_band_px=None, tissues=("Nuclei",)):
    """{tissue: IncrementalLocalLabeller} for the full-resolution tissues of a one-GPU job whose canvas is labelled in several local bands (nothing to
    overlap otherwise, and the multi-rank protocol labels one band per rank after its halo exchange).  band_canv: the canvases sharded_postprocess
    will get (same rows, same columns).  The half-resolution tissues (gland, lumen: 0.04 s of a 40000^2 slide) stay in the tail."""
    pre = OrderedDict()
    if dist is not None or not max_band_px:
        return pre
    for t in tissues:
        key = t + "-INST"
        if key not in band_canv or (wsi_mode and t != "Nuclei"):
            continue
        band = band_canv[key]
        mt = margin.get(t, margin.get("default", 512)) if isinstance(margin, dict) else margin
        if local_band_count(int(band.shape[0]), int(band.shape[1]), max_band_px, mt) > 1:
            pre[t] = IncrementalLocalLabeller(band, t, mt, guard, max_band_px)
    return pre


def _tick(prof, key, nbytes, t0):
    """prof: None, or a dict collecting (bytes moved INTO this rank or out of it, seconds) per phase -- bench.py's xGMI figures."""
    if prof is None:
        return
    import time

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    e = prof.setdefault(key, {"bytes": 0, "s": 0.0})
    e["bytes"] += int(nbytes)
    e["s"] += time.perf_counter() - t0


def _tock(prof):
    if prof is None:
        return 0.0
    import time

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.perf_counter()


def dist_label(band, y0_global, tissue, margin, guard, dist, ds_factor=1.0, label_fn=_device_label_fn, table_fn=_device_table_fn, prof=None, watch=None,
               type_band=None):
    """First half of one rank's work for one tissue: halo exchange (probabilities, and the class map's rows when `type_band` is given) and the
    labelling of halo + band + halo.  -> the labelled BandState."""
    from .launch import null_watch

    watch = watch or null_watch()
    rank, world = dist.get_rank(), dist.get_world_size()
    st = BandState(rank, world, band, y0_global, margin, guard, tissue, ds_factor, type_band=type_band)
    up, down = st.strips()
    above = torch.empty_like(up) if rank > 0 else None
    below = torch.empty_like(down) if rank < world - 1 else None
    t0 = _tock(prof)
    with watch.phase("halo exchange (%s)" % tissue):
        halo_exchange(dist, rank, world, up, down, above, below)
        nb = sum(t.numel() * t.element_size() for t in (above, below) if t is not None)
        if type_band is not None:
            tup, tdown = st.type_strips()
            tabove = torch.empty_like(tup) if rank > 0 else None
            tbelow = torch.empty_like(tdown) if rank < world - 1 else None
            halo_exchange(dist, rank, world, tup, tdown, tabove, tbelow)
            st.set_type_window(tabove, tbelow)
            nb += sum(t.numel() * t.element_size() for t in (tabove, tbelow) if t is not None)
    _tick(prof, "halo_exchange", nb, t0)
    t0 = _tock(prof)
    with watch.phase("band labelling (%s)" % tissue):
        st.n_owned_ = st.label(above, below, label_fn, table_fn)
    _tick(prof, "label_" + tissue, 0, t0)
    return st


def dist_resolve(st, dist, relabel_fn=_device_relabel_fn, prof=None, watch=None):
    """Second half: the count / border-id all-gathers and the relabelling of the band rows.  -> (band labels with global ids, total, info)."""
    from .launch import null_watch

    watch = watch or null_watch()
    rank, world = dist.get_rank(), dist.get_world_size()
    t0 = _tock(prof)
    with watch.phase("instance-count / border-id all-gathers (%s)" % st.tissue):
        out, total, info = _publish_and_resolve(st, st.n_owned_, st.band.device, dist, rank, world, relabel_fn)
    _tick(prof, "ids_" + st.tissue, 0, t0)
    return out, total, info


def run_distributed(band, y0_global, tissue, margin, guard, dist, ds_factor=1.0, label_fn=_device_label_fn,
                    table_fn=_device_table_fn, relabel_fn=_device_relabel_fn, prof=None, watch=None):
    """One rank of the real thing.  `dist` = torch.distributed (initialised).  Returns (band labels with global ids,
    total instance count over all ranks, info dict).  watch: a launch.PhaseWatch -- a collective that does not return ends the rank with
    the name of the phase it was in."""
    st = dist_label(band, y0_global, tissue, margin, guard, dist, ds_factor, label_fn, table_fn, prof, watch)
    return dist_resolve(st, dist, relabel_fn, prof, watch)


def halo_exchange(dist, rank, world, up, down, above, below):
    """Neighbour strips over point-to-point links, in two rounds of disjoint PAIRS: round 0 pairs (0,1), (2,3), ..., round 1 pairs (1,2),
    (3,4), ...  In a round a rank talks to at most one peer and posts its send and its receive in one group, so no rank ever posts a send
    whose matching receive sits behind another blocking operation -- the pattern cannot deadlock whatever the backend's ordering rules
    (RCCL groups, gloo's per-pair queues), and adjacent pairs use different xGMI links at the same time.
    RCCL moves dense buffers only (a strided view is refused at the call, and only with more than one rank -- a fault the one-rank RCCL tests and
    the host-staged gloo ranks cannot show): strided strips are sent from a dense copy and received through one."""
    def dense(t):
        return t if (t is None or t.is_contiguous()) else t.contiguous()

    for rnd in (0, 1):
        ops, back = [], None
        if rank % 2 == rnd and rank < world - 1:      # lower rank of the pair (rank, rank + 1)
            buf = dense(below)
            ops, back = [dist.P2POp(dist.isend, dense(down), rank + 1), dist.P2POp(dist.irecv, buf, rank + 1)], (below, buf)
        elif rank % 2 != rnd and rank > 0:            # upper rank of the pair (rank - 1, rank)
            buf = dense(above)
            ops, back = [dist.P2POp(dist.irecv, buf, rank - 1), dist.P2POp(dist.isend, dense(up), rank - 1)], (above, buf)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if back[0] is not back[1]:
                back[0].copy_(back[1])


def _publish_and_resolve(st, n_owned, dev, dist, rank, world, relabel_fn):
    cnt = torch.tensor([n_owned], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(allc, cnt)
    counts = [int(c.item()) for c in allc]
    offs = np.concatenate([[0], np.cumsum(counts)])
    pub = st.publish(int(offs[rank]))
    # variable-length (key, id) tables: lengths first, then one padded all-gather
    ln = torch.tensor([pub.shape[0]], dtype=torch.int64, device=dev)
    alln = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(alln, ln)
    lens = [int(x.item()) for x in alln]
    mx = max(max(lens), 1)
    buf = torch.zeros((mx, 2), dtype=torch.int64, device=dev)
    if pub.shape[0]:
        buf[: pub.shape[0]] = torch.from_numpy(pub).to(dev)
    allp = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(allp, buf)
    pubs = [allp[r][: lens[r]].cpu().numpy() for r in range(rank)]
    out, info = st.resolve(pubs, relabel_fn)
    info["n_total"] = int(offs[-1])
    return out, int(offs[-1]), info


def gather_streamed_maps(inst, small, slide_hw, out, rank, world, dist, labels=True):
    """Root-side stitch after cerberus_amd.stream_bands on several ranks, where every rank holds ITS rows of the label maps (`inst`) and class
    canvases (`small`).  labels=True (--save_label_maps): everything travels, -> (inst, small) of the whole slide on the root.  labels=False:
    only the quarter-resolution tissue map does ("Patch-Class@0.25"; the bands' own cv2-nearest resizes are the slide's rows when band heights are
    multiples of 8) -> (None, {that map}).  (None, None) off the root."""
    from .tissue import pclass_tissue_map
    from .wsi import SlideGeometry, half_size

    H, W = int(slide_hw[0]), int(slide_hw[1])
    geo = SlideGeometry((H, W), out, out)
    b = geo.bounds(world)
    rows = [max(0, min((b[i + 1] - b[i]) * out, H - b[i] * out)) for i in range(world)]
    if labels:
        g_inst = OrderedDict()
        for t, lab in inst.items():
            half = t != "Nuclei"
            g = _gather_rows(lab, [half_size(r) for r in rows] if half else rows, half_size(W) if half else W, dist, rank, world)
            if rank == 0:
                g_inst[t] = g
        g_small = OrderedDict()
        for k, v in small.items():
            g = _gather_rows(v, rows, W, dist, rank, world)
            if rank == 0:
                g_small[k] = g
        return (g_inst, g_small) if rank == 0 else (None, None)
    res = OrderedDict()
    if "Patch-Class" in small:
        if all(r % 8 == 0 for r in rows[:-1]):
            g = _gather_rows(pclass_tissue_map(small["Patch-Class"]), [int(round(r * 0.25)) for r in rows], int(round(W * 0.25)), dist, rank, world)
            res["Patch-Class@0.25"] = g
        else:
            res["Patch-Class"] = _gather_rows(small["Patch-Class"], rows, W, dist, rank, world)
    return (None, res) if rank == 0 else (None, None)


def assemble(band_labels):
    """Concatenate band label maps (slide order) -- the root-side stitch of the (much smaller) int32 results."""
    return torch.cat(list(band_labels), dim=0) if torch.is_tensor(band_labels[0]) else np.concatenate(list(band_labels), axis=0)


def same_partition(a, b):
    """True iff label maps a and b describe the same instances up to a renaming of the ids."""
    a, b = np.asarray(a).ravel().astype(np.int64), np.asarray(b).ravel().astype(np.int64)
    if not np.array_equal(a > 0, b > 0):
        return False
    pairs = np.unique(np.stack([a, b], axis=1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) and len(np.unique(pairs[:, 1])) == len(pairs)


def local_band_count(rows, cols, max_band_px, margin=0):
    """How many row bands a (rows x cols) canvas is labelled in on ONE GPU so that no labelling CALL exceeds max_band_px pixels (workspace =
    96 B / px, and H*W < 2^31 per call): 1 when it fits.  A local band is labelled together with `margin` halo rows on each side, so the
    rows it may own are `own` = max_band_px / cols - 2 * margin, required to be at least two margins; the canvas is then cut into
    nb = ceil(rows / own) bands of equal height (+- 1 row), each therefore taller than own / 2 >= ONE margin -- the invariant the protocol
    needs (an instance crossing a cut ends inside the neighbour's halo and cannot reach that neighbour's other cut; BandState asserts
    band >= margin).  A canvas so wide that even such a band exceeds the limit cannot be banded by rows: that is an error here, not a
    failure inside the C call."""
    if not max_band_px or rows * cols <= max_band_px:
        return 1
    own = int(max_band_px) // max(1, cols) - 2 * int(margin)
    if own < max(1, 2 * int(margin)):
        raise ValueError("a %d-pixel-wide map cannot be labelled in row bands of at most %d pixels with a %d-row halo on each side: a band of two "
                         "margins plus its halos already has %d pixels (raise max_band_px or lower the margin)"
                         % (cols, int(max_band_px), int(margin), 4 * int(margin) * cols))
    nb = int(-(-rows // own))
    assert rows // nb >= int(margin), "internal: a local band shorter than its margin"
    return nb


def sharded_postprocess(canv, rank, world, dist, wsi_mode=True, margin=512, guard=48, max_band_px=None, prof=None, watch=None, pre=None, arrays=None,
                        type_canv=None, fns=None):
    """Per-rank replacement of WSIRunner.postprocess for band canvases: label maps of THIS rank's band with slide-global
    ids, nothing gathered.  margin: halo rows at full resolution, an int or {tissue: rows, "default": rows} (the reference's
    own nuclei margin is 64 px, infer/wsi.py:906-915; gland clusters need hundreds).  canv: the band canvases of this rank (full-resolution rows of equal count on every rank except
    the last).  Gland / lumen run at x0.5 in wsi_mode (infer/wsi.py:786-804); their margin / guard are halved accordingly.
    max_band_px (world == 1 only): a canvas larger than this is labelled as several row bands one after the other through the
    same halo / ownership / id protocol the ranks use (`run_local`) -- the one-GPU streaming path for slides whose 96 B / px
    labelling workspace would not fit, or that exceed the 2^31-pixel limit of one call.
    arrays (multi-rank path): a dict that receives {tissue: (tab, cnts, pts, has_type, ds_factor)} -- the instance-table rows and contour runs of
    the instances THIS rank owns, in slide coordinates and global-id order (BandState.owned_parts): what rank 0 needs for dat/<slide>.dat
    instead of the label maps (infer/wsi.py:805-853).  type_canv: the band's class canvases ("<Tissue>-TYPE", uint8, full resolution; default:
    the TYPE entries of `canv`) whose halo rows then travel with the probability halos.
    fns (multi-rank path): {"label", "table", "relabel", "arrays", "mask"} overriding the HIP kernels -- the gloo tests inject numpy stand-ins to
    exercise the protocol (tests/test_host_logic.py); the product never passes it."""
    fns = fns or {}
    f_label, f_table = fns.get("label", _device_label_fn), fns.get("table", _device_table_fn)
    f_relabel, f_arrays, f_mask = fns.get("relabel", _device_relabel_fn), fns.get("arrays", _device_arrays_fn), fns.get("mask", _device_mask_fn)
    from .postproc import mask_lumen_by_gland
    from .wsi import downsample2_inst

    inst, info = OrderedDict(), OrderedDict()
    states = OrderedDict()
    type_canv = canv if type_canv is None else type_canv
    rows = int(next(iter(canv.values())).shape[0])
    cnt = torch.tensor([rows], dtype=torch.int64, device=next(iter(canv.values())).device)
    allr = [torch.zeros_like(cnt) for _ in range(world)]
    if dist is not None:
        from .launch import null_watch

        with (watch or null_watch()).phase("band-height all-gather"):
            dist.all_gather(allr, cnt)
    else:
        allr = [cnt]
    y0 = int(sum(int(x.item()) for x in allr[:rank]))
    for t in ("Nuclei", "Gland", "Lumen"):
        key = t + "-INST"
        if key not in canv:
            continue
        half = wsi_mode and t != "Nuclei"
        band = downsample2_inst(canv[key]) if half else canv[key]
        mt = margin.get(t, margin.get("default", 512)) if isinstance(margin, dict) else margin  # per-tissue halo: nuclei need far less than glands
        m, g, yy, ds = (mt // 2, guard // 2, y0 // 2, 0.5) if half else (mt, guard, y0, 1.0)
        if dist is not None:  # also at world == 1 when the caller initialised a process group (bench.py --force-dist, the nccl test)
            tb = None
            if arrays is not None and (t + "-TYPE") in type_canv:
                tb = type_canv[t + "-TYPE"]
                # the half-resolution tissues read the strided sub-sample of the class canvas (cerberus_amd.wsi.build_wsi_inst_info's stated deviation);
                # band row offsets are even, so the band's sub-sample is the band's rows of the slide's sub-sample
                tb = (tb[::2, ::2][: band.shape[0], : band.shape[1]] if half else tb[: band.shape[0], : band.shape[1]]).contiguous()
            states[t] = dist_label(band, yy, t, m, g, dist, ds, f_label, f_table, prof=prof, watch=watch, type_band=tb)
        else:
            t0 = _tock(prof)
            nb = local_band_count(int(band.shape[0]), int(band.shape[1]), max_band_px, m)  # (half-resolution maps: their own pixel count, halved margin)
            if pre and t in pre:  # bands already labelled underneath the inference (IncrementalLocalLabeller): finish the rest, same protocol
                assert pre[t].nb == nb and pre[t].states[0].band.data_ptr() == band.data_ptr(), "the incremental labeller was built for another canvas"
                outs, n, infos = pre[t].finish()
            else:
                cuts = [int(round(i * band.shape[0] / nb)) for i in range(nb + 1)]
                outs, n, infos = run_local([band[cuts[i]:cuts[i + 1]] for i in range(nb)], t, m, g, ds)
            inst[t] = outs[0] if nb == 1 else assemble(outs)
            info[t] = {"n_owned": n, "n_total": n, "n_truncated": sum(i["n_truncated"] for i in infos),
                       "n_unresolved": sum(i["n_unresolved"] for i in infos), "local_bands": nb}
            if pre and t in pre:
                info[t]["bands_labelled_under_inference"] = pre[t].early
            _tick(prof, "label_" + t, 0, t0)
    if states:
        # lumen *= (gland > 0) on the windows, before band rows and instance arrays are taken out of them (a no-op later on the band rows)
        masked_on_windows = False
        if "Lumen" in states and "Gland" in states:
            try:
                states["Lumen"].mask_by(states["Gland"], f_mask)
                masked_on_windows = True
            except ValueError:
                if arrays is not None:  # the owner's table needs the masked window
                    raise
        for t, st in states.items():
            inst[t], _, info[t] = dist_resolve(st, dist, f_relabel, prof=prof, watch=watch)
        if "Lumen" in inst and "Gland" in inst and not masked_on_windows:
            mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
        if arrays is not None:
            t0 = _tock(prof)
            for t, st in states.items():
                half = wsi_mode and t != "Nuclei"
                tab, cnts, pts = st.owned_parts(f_arrays)
                arrays[t] = (tab, cnts, pts, st.type_window is not None, 0.5 if half else 1.0)
            _tick(prof, "tables_and_contours", 0, t0)
    elif "Lumen" in inst and "Gland" in inst:
        mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
    return inst, info


def gather_parts(local, dist, rank, world, dev, prof=None):
    """Per-rank instance arrays -> the root's `parts` list ([(tissue, tab, cnts, pts, offs, has_type, ds_factor)], the format of
    cerberus_amd.wsi.collect_wsi_inst_arrays): per tissue one all-gather of the two lengths, then three padded gathers (table rows, contour
    counts, contour points).  ~0.36 GB for the 885 k instances of a 40000^2 slide against 21 GB of label + class maps.  None off the root."""
    if dist is None or world == 1:  # one rank: its own arrays are the root's (a streamed slide's per-sub-band arrays: no pass over the whole label maps,
        out = []                    # and no 4 - 8 B/px union-find workspace for them -- 77 GB for a 9.7-Gpx slide)
        for t, (tab, cnts, pts, has_type, ds) in local.items():
            cn = np.asarray(cnts, np.int32)
            out.append((t, np.asarray(tab), cn, np.asarray(pts), np.cumsum(cn.astype(np.int64)) - cn, has_type, ds))
        return out
    parts = [] if rank == 0 else None
    moved = 0
    t0 = _tock(prof)
    for t, (tab, cnts, pts, has_type, ds) in local.items():
        ln = torch.tensor([tab.shape[0], pts.shape[0]], dtype=torch.int64, device=dev)
        alln = [torch.zeros_like(ln) for _ in range(world)]
        dist.all_gather(alln, ln)
        lens = [(int(x[0].item()), int(x[1].item())) for x in alln]
        kmax, pmax = max(1, max(k for k, _ in lens)), max(1, max(p_ for _, p_ in lens))
        got = []
        for arr, rows, cols, dt in ((tab, kmax, 16, torch.int64), (cnts.reshape(-1, 1), kmax, 1, torch.int32), (pts, pmax, 2, torch.int32)):
            buf = torch.zeros((rows, cols), dtype=dt, device=dev)
            if arr.shape[0]:
                buf[: arr.shape[0]] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            lst = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, lst, dst=0)
            moved += (world - 1) * buf.numel() * buf.element_size()
            got.append(lst)
        if rank == 0:
            tabs = np.concatenate([got[0][r][: lens[r][0]].cpu().numpy() for r in range(world)], axis=0)
            cn = np.concatenate([got[1][r][: lens[r][0], 0].cpu().numpy() for r in range(world)], axis=0)
            pt = np.concatenate([got[2][r][: lens[r][1]].cpu().numpy() for r in range(world)], axis=0)
            offs = np.cumsum(cn.astype(np.int64)) - cn
            parts.append((t, tabs, cn, pt, offs, has_type, ds))
    _tick(prof, "parts_gather", moved, t0)
    return parts


def band_view(run, H, W, canv=None):
    """The canvases postprocess_bands_and_gather labels: this rank's valid rows and the slide's columns of run.canv (or of `canv`)."""
    src = run.canv if canv is None else canv
    valid = max(0, min(run.band_h, H - run.r0 * run.geo.out))
    return OrderedDict((k, v[:valid, :W]) for k, v in src.items())


def _gather_rows(lab, rows_per_rank, cols, dist, rank, world):
    """Concatenate per-rank label bands (rows_per_rank[r] valid rows each) on the root; None elsewhere."""
    if dist is None:
        return lab[: rows_per_rank[0], :cols]
    hmax = max(rows_per_rank)
    pad = torch.zeros((hmax, cols), dtype=lab.dtype, device=lab.device)
    pad[: lab.shape[0], : min(cols, lab.shape[1])] = lab[:, :cols]
    lst = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, lst, dst=0)
    if rank != 0:
        return None
    return torch.cat([lst[i][: rows_per_rank[i]] for i in range(world)], dim=0)


def postprocess_bands_and_gather(run, H, W, rank, world, dist, margin=512, guard=48, canv=None, max_band_px=None, prof=None, watch=None, pre=None,
                                 parts=None, gather_maps=True):
    """The tail of a slide on 1..N GPUs: band-local label maps with slide-global ids, then only the int32 label bands and the
    uint8 / float class canvases travel to the root (12 + 3 B/px instead of the 36 B/px of raw probability canvases).
    `run` is this rank's WSIRunner after infer_band.  Returns (inst, info, small) -- inst / small are None off the root.
    pre: band_view(run, H, W, canv) labellers from make_incremental that were fed during the inference (one-GPU jobs).
    canv: label these band canvases instead of run.canv (bench.py's structured probability maps); max_band_px: see
    sharded_postprocess; prof: dict collecting bytes / seconds of the halo exchange and the root gather.
    parts (a list; multi-rank path only): every rank builds the instance tables + contours of the instances it OWNS on its halo + band + halo
    window and the root receives the compact arrays -- `parts` is extended there with collect_wsi_inst_arrays' tuples, ready for the .dat
    writer (infer/wsi.py:805-853).  gather_maps=False: the label bands and the class canvases then stay where they are (`inst` is None, `small`
    holds only the quarter-resolution tissue map "Patch-Class@0.25" that tissue/<slide>.mat is written from): ~0.8 GB into the root
    instead of 21 GB for a 40000^2 slide.  run_infer_wsi.py gathers the maps only under --save_label_maps."""
    from .wsi import gather_bands, half_size

    geo = run.geo
    src = run.canv if canv is None else canv
    valid = max(0, min(run.band_h, H - run.r0 * geo.out))
    band = OrderedDict((k, v[:valid, :W]) for k, v in src.items())
    local = OrderedDict() if (parts is not None and dist is not None) else None
    tcanv = OrderedDict((k, v[:valid, :W]) for k, v in run.canv.items() if k.endswith("TYPE"))
    inst_b, info = sharded_postprocess(band, rank, world, dist, wsi_mode=True, margin=margin, guard=guard, max_band_px=max_band_px, prof=prof, watch=watch, pre=pre,
                                       arrays=local, type_canv=tcanv)
    bounds = geo.bounds(world)
    rows = [max(0, min((bounds[i + 1] - bounds[i]) * geo.out, H - bounds[i] * geo.out)) for i in range(world)]
    from .launch import null_watch

    watch = watch or null_watch()
    dev = next(iter(run.canv.values())).device
    if local is not None:
        with watch.phase("instance-array gather to rank 0"):
            got = gather_parts(local, dist, rank, world, dev, prof=prof)
        if rank == 0:
            parts.extend(got)
    if not gather_maps:
        if dist is None:
            raise ValueError("gather_maps=False is the multi-rank path (one rank holds its maps already)")
        small = None
        if "Patch-Class" in run.canv:
            from .tissue import pclass_tissue_map

            t0 = _tock(prof)
            pc = run.canv["Patch-Class"][:valid, :W]
            aligned = all(r % 8 == 0 for r in rows[:-1])  # then the bands' quarter maps are the rows of the slide's (cv2 INTER_NEAREST samples 4 y, 4 x)
            with watch.phase("tissue-map gather to rank 0"):
                if aligned:
                    q = pclass_tissue_map(pc)
                    rq = [int(round(r * 0.25)) for r in rows]
                    g = _gather_rows(q, rq, int(round(W * 0.25)), dist, rank, world)
                    moved = (world - 1) * max(rq) * int(round(W * 0.25)) * 4
                    small = None if rank != 0 else OrderedDict([("Patch-Class@0.25", g)])
                else:  # (patch rows that are not multiples of 8 pixels: the whole class map travels, as before)
                    small = gather_bands(OrderedDict([("Patch-Class", run.canv["Patch-Class"])]), geo, rank, world, dist)
                    moved = (world - 1) * run.canv["Patch-Class"].numel() * 4
            _tick(prof, "root_gather", moved, t0)
        elif rank == 0:
            small = OrderedDict()
        return None, info, small
    inst = OrderedDict() if rank == 0 else None
    t0 = _tock(prof)
    moved = 0
    for t, lab in inst_b.items():
        half = t != "Nuclei"
        rr = [half_size(r) for r in rows] if half else rows
        cc = half_size(W) if half else W
        with watch.phase("label-band gather to rank 0 (%s)" % t):
            g = _gather_rows(lab, rr, cc, dist, rank, world)
        moved += (world - 1) * max(rr) * cc * 4  # what the root receives (every rank pads to the tallest band)
        if rank == 0:
            inst[t] = g
    small_src = OrderedDict((k, v) for k, v in run.canv.items() if not k.endswith("INST"))
    with watch.phase("class-canvas gather to rank 0"):
        small = gather_bands(small_src, geo, rank, world, dist)
    moved += (world - 1) * sum(v.numel() * v.element_size() for v in small_src.values())
    if dist is not None:
        _tick(prof, "root_gather", moved, t0)
    return inst, info, small
