"""Whole-slide sliding-window driver on a device-resident slide (mirror of the hot loops of reference infer/wsi.py).

Reference semantics kept (infer/wsi.py:502-856): sliding window with `patch_input_shape` context and
`patch_output_shape` stride, per-head merged probability maps, nuclei label maps from the full-resolution maps,
gland / lumen from the x0.5 (cv2.resize INTER_LINEAR) maps with ds_factor=0.5 and lumen-inside-gland masking
(infer/wsi.py:786-804), Patch-Class map.

MI355X-first differences (DESIGN.md "WSI driver"):
  * geometry is the in-repo, oracle-checkable tile geometry of infer/tile.py:43-106 generalised to a slide (mirror
    padding, stride = output size), not tiatoolbox's get_coordinates/_get_tile_info (un-vendored);
  * no memmap cache: the six head maps live in HBM (40000^2 x 36 B = 57.6 GB of 288 GB) and are written by the
    head kernels directly (no merge_prediction pass);
  * nuclei are labelled on the whole band in ONE pass instead of 4096^2 tiles + 64 px margin strips -- the strips only
    exist in the reference to approximate the untiled result under a host-memory limit;
  * tiles shard across GPUs by contiguous bands of patch rows: one process per GPU, weights replicated, no data-path
    collective during inference; one RCCL gather stitches the per-head maps on rank 0 (infer/base.py:46 DataParallel
    is the only multi-GPU mechanism the reference has).
Slide file decoding (tiatoolbox WSIReader) is out of scope (synthetic / array slides); tissue masks: cerberus_amd/tissue.py.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch

from .inst_info import _uuid4_hex, write_dat  # noqa: F401

from . import _lib
from .postproc import mask_lumen_by_gland, postproc_device


def band_partition(n_rows, world_size):
    """Contiguous split of patch rows over ranks: rank g owns rows [b[g], b[g+1])."""
    base, rem = divmod(n_rows, world_size)
    b = [0]
    for g in range(world_size):
        b.append(b[-1] + base + (1 if g < rem else 0))
    return b


def band_partition_weighted(weights, world_size):
    """Contiguous split of patch rows with per-row work `weights` (selected patches per row under a tissue mask): cut g is
    the first row at which the running weight reaches g / world of the total; every rank keeps at least one row while rows
    last.  Uniform weights reproduce band_partition up to rounding of the cut points."""
    w = np.asarray(weights, np.float64)
    n = len(w)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    b = [0]
    for g in range(1, world_size):
        cut = int(np.searchsorted(cum, cum[-1] * g / world_size, side="left"))
        cut = max(cut, b[-1] + 1)           # at least one row for the previous rank
        cut = min(cut, n - (world_size - g))  # and one left for each later rank
        b.append(max(min(cut, n), b[-1]))
    b.append(n)
    return b


class SlideGeometry(object):
    """Patch placement for a slide of (H, W): output tiles of `out` px on a regular grid covering the slide, each
    fed by a `win` px input window centred on it (context (win-out)//2, mirror padded at the slide border).
    patch_sel: optional bool [rows, cols] -- the patches that hold tissue (cerberus_amd.tissue.select_patches); bands are
    then balanced by the number of selected patches instead of by rows."""

    def __init__(self, slide_hw, patch_input_shape, patch_output_shape, patch_sel=None):
        self.H, self.W = int(slide_hw[0]), int(slide_hw[1])
        self.win, self.out = int(patch_input_shape), int(patch_output_shape)
        assert self.win >= self.out and (self.win - self.out) % 2 == 0 and self.win % 16 == 0
        self.ctx = (self.win - self.out) // 2
        self.rows = math.ceil(self.H / self.out)
        self.cols = math.ceil(self.W / self.out)
        self.patch_sel = None
        if patch_sel is not None:
            self.patch_sel = np.asarray(patch_sel, bool).reshape(self.rows, self.cols)

    def out_boxes(self):
        """int64 [rows*cols, 2, 2] ((y0, x0), (y1, x1)): the output box of every patch of the grid, row-major"""
        rr, cc = np.meshgrid(np.arange(self.rows), np.arange(self.cols), indexing="ij")
        tl = np.stack([rr.ravel() * self.out, cc.ravel() * self.out], axis=1)
        return np.stack([tl, tl + self.out], axis=1).astype(np.int64)

    def bounds(self, world_size):
        if self.patch_sel is None or world_size == 1:
            return band_partition(self.rows, world_size)
        return band_partition_weighted(self.patch_sel.sum(axis=1) + 1e-3, world_size)

    def band(self, rank, world_size):
        b = self.bounds(world_size)
        return b[rank], b[rank + 1]

    def input_rows(self, r0, r1):
        """Absolute slide rows a band of patch rows reads (before mirror padding), clipped to the slide."""
        y0 = max(0, r0 * self.out - self.ctx)
        y1 = min(self.H, r1 * self.out + self.ctx)
        # mirror padding at the top / bottom edge reads up to ctx rows inside the slide as well
        if r0 == 0:
            y1 = max(y1, min(self.H, self.ctx + 1))
        if r1 == self.rows:
            y0 = min(y0, max(0, self.H - 1 - (r1 * self.out + self.ctx - self.H)))
        return y0, y1


def synth_slide(h, w, y0=0, x0=0, seed=2, device=None):
    """Synthetic uint8 RGB slab [h,w,3] on the GPU; deterministic in absolute coordinates."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        _lib.check(_lib.lib().cerb_synth_slide(out.data_ptr(), h, w, y0, x0, C.c_uint32(seed), C.c_void_p(st)))
    return out


def gather_patches(slab, slab_y0, full_h, tl_y, tl_x, win):
    n = int(tl_y.numel())
    tiles = torch.empty((n, win, win, 3), dtype=torch.uint8, device=slab.device)
    st = torch.cuda.current_stream(slab.device).cuda_stream
    with torch.cuda.device(slab.device):
        _lib.check(_lib.lib().cerb_gather_patches(slab.data_ptr(), slab.shape[0], slab.shape[1], slab_y0, full_h, tl_y.data_ptr(),
                                                  tl_x.data_ptr(), n, win, tiles.data_ptr(), C.c_void_p(st)))
    return tiles


def half_size(n):
    """side of cv2.resize(fx=0.5): cvRound(n * 0.5), half to even (== cerb_half_size)"""
    return int(round(n * 0.5))


def downsample2_inst(inst):
    h, w = int(inst.shape[0]), int(inst.shape[1])
    out = torch.empty((half_size(h), half_size(w), 2), dtype=torch.float32, device=inst.device)
    st = torch.cuda.current_stream(inst.device).cuda_stream
    with torch.cuda.device(inst.device):
        _lib.check(_lib.lib().cerb_downsample2_inst(inst.data_ptr(), inst.stride(0), inst.stride(1), h, w, out.data_ptr(), C.c_void_p(st)))
    return out


def gather_bands(canv, geo, rank, world, dist=None):
    """Bands are contiguous row ranges of the slide canvas, so stitching is a gather + concatenation (no reduction).
    canv: OrderedDict head-key -> this rank's band tensor [(r1-r0)*out, cols*out, ...] (any device).
    Every rank pads its band to the tallest band so one `dist.gather` per head suffices (backend "nccl" = RCCL over
    xGMI on the GPU box, "gloo" in the CPU tests)."""
    bounds = geo.bounds(world)
    if dist is None:
        return OrderedDict((k, v[: geo.H, : geo.W]) for k, v in canv.items())
    max_rows = max(bounds[i + 1] - bounds[i] for i in range(world)) * geo.out
    full = OrderedDict() if rank == 0 else None
    for k, v in canv.items():
        pad = torch.zeros((max_rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        lst = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, lst, dst=0)
        if rank == 0:
            parts = [lst[i][: (bounds[i + 1] - bounds[i]) * geo.out] for i in range(world)]
            full[k] = torch.cat(parts, dim=0)[: geo.H, : geo.W]
    return full


def check_shardable(slide_hw, patch_output_shape, world_size):
    """A rank without a patch row would stay out of the halo exchange and the gathers while the others wait for it: every rank
    evaluates this same condition before the first collective of a slide, so all of them stop together."""
    rows = math.ceil(int(slide_hw[0]) / int(patch_output_shape))
    if rows < int(world_size):
        raise ValueError("slide of %d patch rows cannot be sharded over %d ranks: use at most %d" % (rows, world_size, rows))


class WSIRunner(object):
    """One per process / GPU.  (Single-process simulations of more ranks than patch rows get empty bands; distributed drivers call
    check_shardable first.)"""

    def __init__(self, net, slide_hw, patch_input_shape=256, patch_output_shape=256, batch_size=32, rank=0, world_size=1, patch_sel=None, twin=None, row_range=None):
        """twin: a second handle with the same parameters (NetDesc.twin()); batches then alternate between the two on two side streams.
        row_range: (r0, r1) patch rows instead of rank's band of the world -- the sub-bands of cerberus_amd.stream_bands (slides larger than HBM)."""
        self.net = net
        self.twin = twin
        self._side = None
        self._turn = 0
        self.geo = SlideGeometry(slide_hw, patch_input_shape, patch_output_shape, patch_sel)
        self.batch = int(batch_size)
        self.rank, self.world = int(rank), int(world_size)
        self.r0, self.r1 = self.geo.band(self.rank, self.world) if row_range is None else (int(row_range[0]), int(row_range[1]))
        self.dev = torch.device("cuda", torch.cuda.current_device())
        g = self.geo
        self.band_h = (self.r1 - self.r0) * g.out
        self.canvas_w = g.cols * g.out
        self.canv = OrderedDict()
        for name, hname, och, key in net._decoders:
            if hname == "INST":
                self.canv[key] = torch.zeros((self.band_h, self.canvas_w, 2), dtype=torch.float32, device=self.dev)
            elif hname == "TYPE":
                self.canv[key] = torch.zeros((self.band_h, self.canvas_w), dtype=torch.uint8, device=self.dev)
            else:
                self.canv[key] = torch.zeros((self.band_h, self.canvas_w), dtype=torch.float32, device=self.dev)
        self._outs = [self.canv[d[3]] for d in net._decoders]
        # patch list of this band, row-major
        rr, cc = np.meshgrid(np.arange(self.r0, self.r1), np.arange(g.cols), indexing="ij")
        if g.patch_sel is not None:  # patches without tissue never run; their canvas pixels stay 0 (infer/wsi.py:565-569)
            keep = g.patch_sel[self.r0:self.r1]
            rr, cc = rr[keep], cc[keep]
        self.n_patches = rr.size
        self._tl_y = torch.from_numpy((rr.ravel() * g.out - g.ctx).astype(np.int64)).to(self.dev)
        self._tl_x = torch.from_numpy((cc.ravel() * g.out - g.ctx).astype(np.int64)).to(self.dev)
        self._off = torch.from_numpy(((rr.ravel() - self.r0) * g.out * self.canvas_w + cc.ravel() * g.out).astype(np.int64)).to(self.dev)
        # the data-aware precision guard: one row of cerb_forward_io.logit_absmax words per queued batch (the head kernels raise them; nothing is
        # read on the host until logit_report()).  CERB_LOGIT_GUARD=0 switches the log off, =rerun makes the drivers re-run flagged batches.
        import os

        self.logit_guard = os.environ.get("CERB_LOGIT_GUARD", "count")
        self._logit_rows = []  # (b0, b1) of the batch that owns row i of the log
        self._logit_log = None if self.logit_guard == "0" else torch.zeros((max(1, self.n_patches), len(net._decoders)), dtype=torch.int32, device=self.dev)

    def slab_rows(self):
        return self.geo.input_rows(self.r0, self.r1)

    def infer_band(self, slab, slab_y0, ready=None, progress=None):
        """slab: uint8 [rows, W, 3] holding absolute slide rows [slab_y0, slab_y0+rows) (this rank's band + halo).
        Runs every patch of the band; outputs land in self.canv.  Returns the number of patches.
        ready: optional callable(n_rows) that makes the first n_rows rows of `slab` valid for work queued on the current stream
        (SlabUploader.upload_until: the band is then uploaded chunk by chunk underneath the inference of the rows above)."""
        return self.infer_patches(slab, slab_y0, 0, self.n_patches, ready, progress)

    def rows_final(self, n_patches_done):
        """Canvas rows of this band that are final once the first n patches of the row-major list have run (whole patch rows only)."""
        if self.geo.patch_sel is not None:
            return 0 if n_patches_done < self.n_patches else self.band_h  # (a tissue mask drops patches: no simple row count)
        return min(self.band_h, (int(n_patches_done) // self.geo.cols) * self.geo.out)

    def join(self):
        """Make the caller's stream wait for the side streams (two-handle mode, after infer_patches(..., join=False))."""
        if self._side is not None:
            cur = torch.cuda.current_stream(self.dev)
            for s in self._side:
                cur.wait_stream(s)

    def infer_patches(self, slab, slab_y0, p0, p1, ready=None, progress=None, join=True):
        """Patches [p0, p1) of this band's row-major patch list (bench.py times a slide as K such stripes).
        progress: optional callable(n_done, events) after every queued batch: the first n_done patches of the list are complete once `events`
        (one per stream in use) are -- what shard_postproc.IncrementalLocalLabeller.feed wants.
        join=False (two-handle mode): the caller's stream does not wait for the side streams at the end -- a job cut into many short calls (bench.py's K
        stripes, a few batches each on 8 GPUs) then keeps both streams full across the calls; call join() before anything reads the canvases."""
        g = self.geo
        assert slab.shape[1] == g.W
        p0, p1 = max(0, int(p0)), min(self.n_patches, int(p1))
        tl_y_host = self._tl_y.cpu().numpy() if ready is not None else None

        def one(net, b0):
            b1 = min(p1, b0 + self.batch)
            if ready is not None:  # mirror padding only ever folds back to rows above the window's last in-slide row
                ready(min(int(tl_y_host[b0:b1].max()) + g.win, g.H) - slab_y0)
            tiles = gather_patches(slab, slab_y0, g.H, self._tl_y[b0:b1], self._tl_x[b0:b1], g.win)
            row = None
            if self._logit_log is not None and len(self._logit_rows) < self._logit_log.shape[0]:
                row = self._logit_log[len(self._logit_rows)]
                self._logit_rows.append((b0, b1))
            net._run(tiles, g.out, g.out, self._outs, None, tile_off=self._off[b0:b1], row_stride=self.canvas_w, type_is_u8=True, logit_absmax=row)

        def mark(stream):
            e = torch.cuda.Event()
            e.record(stream)
            return e

        if self.twin is None or (p1 - p0 <= self.batch and join):
            for b0 in range(p0, p1, self.batch):
                one(self.net, b0)
                if progress is not None:
                    progress(min(p1, b0 + self.batch), [mark(torch.cuda.current_stream(self.dev))])
            return p1 - p0
        # two handles, two side streams: everything queued on the caller's stream so far happens before, everything after waits for both
        if self._side is None:
            self._side = [torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)]
        cur = torch.cuda.current_stream(self.dev)
        fork = torch.cuda.Event()
        fork.record(cur)
        nets = (self.net, self.twin)
        for s in self._side:
            s.wait_event(fork)
        last = [fork, fork]
        for b0 in range(p0, p1, self.batch):
            i = self._turn  # the alternation goes on across calls
            self._turn += 1
            k = (i & 1) if self.twin is not None else 0  # (after the second handle was dropped: everything on side stream 0, one handle = one stream)
            try:
                try:
                    with torch.cuda.stream(self._side[k]):
                        one(nets[k], b0)
                except _lib.CerberusHipAllocError:
                    # the library allocates from the driver, and the blocks torch's allocator keeps idle (a previous slide's canvases: they are not
                    # handed back between slides, asking the driver for 200 GB again costs seconds) are invisible to it: release them and try once more
                    torch.cuda.synchronize(self.dev)
                    torch.cuda.empty_cache()
                    with torch.cuda.stream(self._side[k]):
                        one(nets[k], b0)
            except _lib.CerberusHipAllocError as e:
                # the second handle's workspace did not fit after all (the plan is an estimate): go on with one handle instead of losing the slide.
                # (CERB_ERR_ALLOC, not a message match; a failure on the FIRST handle stays fatal: there is nothing left to drop)
                if k != 1:
                    raise
                import logging

                logging.getLogger("cerberus_amd.wsi").warning("second inference handle dropped (%s): continuing on one handle", e)
                torch.cuda.synchronize(self.dev)
                self.twin._release()  # its packed weights and whatever workspace it did get go back to the device BEFORE the retry
                self.twin = None
                nets = (self.net, None)
                torch.cuda.empty_cache()
                k = 0
                with torch.cuda.stream(self._side[0]):
                    one(self.net, b0)
            if progress is not None:
                last[k] = mark(self._side[k])
                progress(min(p1, b0 + self.batch), list(last))
        if join:
            self.join()
        return p1 - p0

    def logit_report(self, threshold=None):
        """What the head kernels saw (host sync): {"batches", "above", "max", "per_head_max", "flagged": [(b0, b1), ..]} -- `above` counts the
        batches whose largest dense |logit| exceeded `threshold` (default NetDesc.LOGIT_SATURATION: the range the F(4x4,3x3) default is
        held to 1e-4 on).  A model prepare() already moved to F(2x2) is never flagged (nothing faster to fall back from)."""
        rep = {"batches": len(self._logit_rows), "above": 0, "max": 0.0, "per_head_max": {}, "flagged": []}
        if self._logit_log is None or not self._logit_rows:
            return rep
        thr = float(self.net.LOGIT_SATURATION if threshold is None else threshold)
        vals = self.net.logit_absmax(words=self._logit_log[: len(self._logit_rows)])
        dense = [i for i, d in enumerate(self.net._decoders) if d[0] != "Patch-Class"]
        if not dense:
            return rep
        per_batch = vals[:, dense].max(axis=1)
        rep["max"] = float(per_batch.max())
        rep["per_head_max"] = {self.net._decoders[i][3]: float(vals[:, i].max()) for i in dense}
        if self.net.precision_decision()["conv_algo"] not in (1, 0):
            hot = np.nonzero(per_batch > thr)[0]
            rep["above"] = int(hot.size)
            rep["flagged"] = [self._logit_rows[i] for i in hot]
        if rep["above"]:
            import logging

            logging.getLogger("cerberus_amd.wsi").warning(
                "%d of %d batches produced logits above %.0f (largest %.0f): the F(4x4,3x3) default is held to 1e-4 below that; "
                "CERB_LOGIT_GUARD=rerun re-runs such batches on F(2x2,3x3)", rep["above"], rep["batches"], thr, rep["max"])
        return rep

    def rerun_flagged(self, slab, slab_y0, flagged):
        """Re-run the batches logit_report() flagged on cerb_net_set_conv_algo(1) (F(2x2,3x3): the algorithm saturated models are held to the
        bar on), on the current stream and the first handle; their canvas windows are overwritten.  The slab must still hold their rows."""
        if not flagged:
            return 0
        g = self.geo
        L, h = _lib.lib(), self.net._ensure_handle()
        was = self.net.precision_decision()["conv_algo"]
        _lib.check(L.cerb_net_set_conv_algo(h, 1))
        try:
            for b0, b1 in flagged:
                tiles = gather_patches(slab, slab_y0, g.H, self._tl_y[b0:b1], self._tl_x[b0:b1], g.win)
                self.net._run(tiles, g.out, g.out, self._outs, None, tile_off=self._off[b0:b1], row_stride=self.canvas_w, type_is_u8=True)
        finally:
            _lib.check(L.cerb_net_set_conv_algo(h, was))
        return len(flagged)

    def gather_to_root(self, dist=None):
        """Stitch the per-head band canvases on rank 0 (one gather per head over RCCL / xGMI).  Returns the full
        canvases on rank 0 (cropped to the slide), None on the other ranks."""
        return gather_bands(self.canv, self.geo, self.rank, self.world, dist)

    @staticmethod
    def postprocess(canv, wsi_mode=True):
        """Label maps from stitched canvases (rank 0).  wsi_mode: gland / lumen at x0.5 with ds_factor 0.5
        (infer/wsi.py:786-804); otherwise tile-mode semantics at full resolution (infer/tile.py:168-191)."""
        inst, info = OrderedDict(), OrderedDict()
        if "Nuclei-INST" in canv:
            inst["Nuclei"], info["Nuclei"] = postproc_device(canv["Nuclei-INST"], "Nuclei", exact_ties=False)
        for t in ("Gland", "Lumen"):
            key = t + "-INST"
            if key not in canv:
                continue
            if wsi_mode:
                from .tissue import half_inst_region

                inst[t], info[t] = postproc_device(half_inst_region(canv[key]), t, 0.5)
            else:
                inst[t], info[t] = postproc_device(canv[key], t, 1.0)
        if "Lumen" in inst and "Gland" in inst:
            mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
        return inst, info


def build_wsi_inst_info(inst, canv, slide_hw, proc_mag, ds_factor=0.5, region_records=None, base_mag=None, base_hw=None, prebuilt=None):
    """The dictionary the reference dumps as dat/<slide>.dat (infer/wsi.py:805-853): per tissue {uuid4 hex -> {'box':
    [x1, y1, x2, y2], 'centroid', 'contour', 'type', 'type_prob'}} plus resolution metadata.  Gland / lumen label maps are
    at x`ds_factor` and their coordinates are scaled back (get_inst_info_dict(..., ds_factor)); nuclei are at full
    resolution.  Per-instance reductions and border following run on the GPU (cerb_inst_table / cerb_inst_contour_*).
    Deviation: the class map handed to the half-resolution tissues is the strided sub-sample of the uint8 class canvas; the
    reference bilinearly resizes class ids together with the probabilities (cv2.resize, infer/wsi.py:786-788).
    region_records: cerberus_amd.tissue.postprocess_regions(...) when the slide has a tissue mask -- the gland / lumen entries then
    come from the per-region dictionaries (already in slide coordinates) and `inst` only supplies the nuclei.
    prebuilt: {tissue: ready dictionary} that replaces the entry computed from `inst[tissue]` (`--reference_tiling`)."""
    import uuid

    from .postproc import get_inst_info_dict

    out = OrderedDict()
    if region_records is not None:
        for rec in region_records:
            for tissue, d in rec["info"].items():
                dst = out.setdefault(tissue, OrderedDict())
                for v in d.values():
                    dst[uuid.uuid4().hex] = v
    for tissue, d in (prebuilt or {}).items():  # dictionaries made elsewhere (cerberus_amd/ref_tiling.py: the reference's tiled nuclei)
        out[tissue] = d
    for tissue, lab in inst.items():
        if (region_records is not None and tissue != "Nuclei") or (prebuilt and tissue in prebuilt):
            continue
        tkey = tissue + "-TYPE"
        half = tuple(lab.shape) != tuple(int(v) for v in slide_hw)
        tmap = canv.get(tkey)
        if tmap is not None and half:
            tmap = tmap[::2, ::2][: lab.shape[0], : lab.shape[1]].contiguous()
        info = get_inst_info_dict(lab.contiguous(), tmap, ds_factor if half else 1.0, flat_box=True)
        out[tissue] = OrderedDict(zip(_uuid4_hex(len(info)), info.values()))
    out["proc_resolution"] = {"resolution": float(proc_mag), "units": "mpp"}
    # scan resolution / baseline size as the reader reports them (infer/wsi.py:529-531); arrays and synthetic slides carry none
    out["base_resolution"] = {"resolution": float(proc_mag if base_mag is None else base_mag), "units": "mpp"}
    out["proc_dimensions"] = np.array([int(slide_hw[0]), int(slide_hw[1])])  # YX
    bh, bw = slide_hw if base_hw is None else base_hw
    out["base_dimensions"] = np.array([int(bh), int(bw)])
    return out


def collect_wsi_inst_arrays(inst, canv, slide_hw, ds_factor=0.5, skip=()):
    """The GPU half of build_wsi_inst_info: per tissue the instance table (cerb_inst_table) and the contour point lists (cerb_inst_contour_*),
    copied to the host as arrays -- what cerberus_amd.inst_info.build_from_parts turns into the dictionary, in this process or in the writer's.
    -> [(tissue, tab, cnts, pts, offs, has_type, ds_factor)]"""
    from .postproc import inst_contours_device, inst_table_device

    parts = []
    for tissue, lab in inst.items():
        if tissue in skip:
            continue
        half = tuple(lab.shape) != tuple(int(v) for v in slide_hw)
        tmap = canv.get(tissue + "-TYPE")
        if tmap is not None and half:
            tmap = tmap[::2, ::2][: lab.shape[0], : lab.shape[1]].contiguous()
        lab = lab.contiguous()
        tab_dev = inst_table_device(lab, None if tmap is None else tmap.contiguous())
        cnts, pts, offs = inst_contours_device(lab, tab_dev)
        parts.append((tissue, tab_dev.cpu().numpy(), cnts, pts, offs, tmap is not None, ds_factor if half else 1.0))
    return parts


def wsi_meta(slide_hw, proc_mag, base_mag=None, base_hw=None):
    """The resolution entries of dat/<slide>.dat (infer/wsi.py:847-851)."""
    meta = OrderedDict()
    meta["proc_resolution"] = {"resolution": float(proc_mag), "units": "mpp"}
    meta["base_resolution"] = {"resolution": float(proc_mag if base_mag is None else base_mag), "units": "mpp"}
    meta["proc_dimensions"] = np.array([int(slide_hw[0]), int(slide_hw[1])])  # YX
    bh, bw = slide_hw if base_hw is None else base_hw
    meta["base_dimensions"] = np.array([int(bh), int(bw)])
    return meta


class DatWriter(object):
    """dat/<slide>.dat written by a CHILD PROCESS (fork) while the parent goes on to the next slide.

    Serialising ~1e6 per-instance dictionaries (three small numpy arrays each) is ~8 s of pure-Python / pickle work for a 40000^2 slide -- as long
    as the slide's whole inference.  On a thread it would share the interpreter lock with the loop that launches the next slide's batches; a forked
    child has the dictionary copy-on-write, touches neither the GPU nor any lock of the parent, writes `<path>.part`, renames it, and leaves through
    os._exit (no atexit handlers, no HIP teardown).  `join()` waits for it and raises if it failed.  Falls back to a thread where fork is not available."""

    def __init__(self, obj, path):
        import os
        import threading

        self.path, self._pid, self._thr, self._err = path, None, None, None
        tmp = path + ".part"

        def work():
            import time

            t0 = time.perf_counter()
            write_dat(obj, tmp)
            os.replace(tmp, path)
            self.seconds = time.perf_counter() - t0
            if os.environ.get("CERB_DAT_WRITER_TIMING"):  # bench.py reads the child's own clock from a side file
                with open(path + ".time", "w") as fh:
                    fh.write("%.6f" % self.seconds)

        if hasattr(os, "fork") and not os.environ.get("CERB_DAT_WRITER_THREAD"):
            pid = os.fork()
            if pid == 0:
                code = 0
                try:
                    work()
                except BaseException:  # the parent reports it
                    code = 1
                os._exit(code)
            self._pid = pid
        else:
            def run():
                try:
                    work()
                except BaseException as e:
                    self._err = e

            self._thr = threading.Thread(target=run)
            self._thr.start()

    @classmethod
    def from_arrays(cls, parts, meta, path, extra=None):
        """The preferred form: the parent hands over ARRAYS (collect_wsi_inst_arrays) and a fresh, torch-free Python process
        (`python -m cerberus_amd.inst_info`) builds the ~1e6 per-instance dictionaries, draws the uuid keys and pickles them.  Nothing of the
        8-9 s a 40000^2 slide's dictionary costs stays in the process that launches the next slide's batches, and -- unlike the forked child
        of __init__, whose copy-on-write image of a 120 GB process slowed the parent's next inference pass by a third -- nothing is shared.
        extra: {key: ready dictionary} merged in (tissue-region records, `--reference_tiling` nuclei), pickled to a side file."""
        import os
        import pickle
        import subprocess
        import sys

        from . import inst_info

        self = cls.__new__(cls)
        self.path, self._pid, self._thr, self._err = path, None, None, None
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        # the hand-over file lives in RAM when the host has a tmpfs with room for it (a 40000^2 slide's arrays: ~0.3 GB; 0.17 s on the output disk, 0.05 s in
        # /dev/shm -- time the parent spends between two slides); CERB_DAT_TMP names another directory, a failed write falls back to the output directory
        src = path + ".parts.npz"
        nbytes = sum(int(getattr(a, "nbytes", 0)) for p_ in parts for a in p_ if hasattr(a, "nbytes"))
        tmpdir = os.environ.get("CERB_DAT_TMP", "/dev/shm")
        cand = None
        try:
            st = os.statvfs(tmpdir)
            if os.path.isdir(tmpdir) and os.access(tmpdir, os.W_OK) and st.f_bavail * st.f_frsize > 2 * nbytes + (64 << 20):
                cand = os.path.join(tmpdir, "cerb_parts_%d_%s.npz" % (os.getpid(), os.path.basename(path)))
                inst_info.save_parts(cand, parts, meta)
                src = cand
        except OSError:
            src = path + ".parts.npz"
            if cand and os.path.exists(cand):
                os.remove(cand)
        if src == path + ".parts.npz":
            inst_info.save_parts(src, parts, meta)
        xtra = ""
        if extra:
            xtra = path + ".extra.pkl"
            with open(xtra, "wb") as fh:
                pickle.dump(extra, fh, protocol=4)
        env = dict(os.environ)
        env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
        env.pop("HIP_VISIBLE_DEVICES", None)
        self._proc = subprocess.Popen([sys.executable, "-m", "cerberus_amd.inst_info", src, path, xtra], env=env)
        self._ram_src = src if src == cand else None  # (the writer removes its input when it is done; a writer that FAILED leaves it -- in RAM: join() removes it)
        return self

    def join(self):
        import os

        proc = getattr(self, "_proc", None)
        if proc is not None:
            rc = proc.wait()
            self._proc = None
            ram = getattr(self, "_ram_src", None)
            if ram and os.path.exists(ram):
                os.remove(ram)
            if rc != 0:
                raise RuntimeError("writing %s failed in the writer process (exit code %d)" % (self.path, rc))
        if self._pid is not None:
            _, status = os.waitpid(self._pid, 0)
            self._pid = None
            if status != 0:
                raise RuntimeError("writing %s failed in the writer process (status %d)" % (self.path, status))
        if self._thr is not None:
            self._thr.join()
            self._thr = None
            if self._err is not None:
                raise self._err


_PINNED_FREE = {}  # shape -> pinned uint8 tensors not in use: page-locking 100 - 400 MB costs 0.05 - 0.2 s, and an uploader is made per slide / sub-band


def _take_pinned(shape):
    free = _PINNED_FREE.get(tuple(shape))
    if free:
        return free.pop()
    return torch.empty(tuple(shape), dtype=torch.uint8).pin_memory()


def _give_pinned(tensors, keep_bytes=2 << 30):
    held = sum(t.numel() for lst in _PINNED_FREE.values() for t in lst)
    for t in tensors:
        if held + t.numel() <= keep_bytes:
            _PINNED_FREE.setdefault(tuple(t.shape), []).append(t)
            held += t.numel()


class SlabUploader(object):
    """Host-resident slide band -> device slab, chunk by chunk through a ring of pinned staging buffers on a copy stream, AHEAD of the inference: a
    producer thread reads / decodes the next chunks (cerberus_amd.reader decodes a chunk's tiles on its thread pool; libjpeg / zlib / the page
    cache all run without the interpreter lock) and issues their copies while the caller's thread queues the batches of the rows already on the
    device -- the role of the reference's 12 persistent DataLoader workers (infer/wsi.py:936-950).  `upload_until(n_rows)` waits until the
    first n_rows rows have been issued and makes the CURRENT stream wait for their copies; chunks are issued in order on one copy stream, so
    waiting for the last one covers all.  CERB_UPLOAD_AHEAD=0: round 5's behaviour (the caller's thread reads each chunk when it is asked for)."""

    def __init__(self, host, y0, y1, device=None, chunk_bytes=24 << 20, buffers=3):
        import os
        import threading

        self.host, self.y0 = host, int(y0)
        self.rows, self.w = int(y1 - y0), int(host.shape[1])
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.slab = torch.empty((self.rows, self.w, 3), dtype=torch.uint8, device=self.dev)
        # chunks end at absolute multiples of the chunk height, itself a multiple of the source's storage tile height (reader._Rows.row_align): a
        # tiled file's tiles are decoded once; the first chunk is the short one, so the first batch starts after a few tile rows, not after 64 MB
        self.align = max(1, int(getattr(host, "row_align", 1)))
        self.chunk = max(self.align, (int(chunk_bytes) // max(1, self.w * 3)) // self.align * self.align)
        self.ahead = os.environ.get("CERB_UPLOAD_AHEAD", "1") != "0"
        nb = max(2, int(buffers)) if self.ahead else 2
        # A slide stored finer than it is processed (a 40x scan read at 0.5 mpp: the common case): the host decodes the STORED level's rows and the
        # device reduces them (cerb_resample_box / cerb_resample_area = reader.read_bounds' bytes) -- the numpy reduction on this one producer
        # thread ran such a slide at 9 Mpx/s.  Chunks are whole storage tile rows of the source level (>= 4 when the factor is not an integer: the
        # tile row a chunk boundary cuts is decoded by both neighbours).  CERB_DEVICE_RESAMPLE=0: the host path.
        self.plan = host.device_plan() if (hasattr(host, "device_plan") and os.environ.get("CERB_DEVICE_RESAMPLE", "1") != "0") else None
        self.stage, self.tabs, self.col_tabs = None, [None] * nb, None
        if self.plan is not None:
            pl = self.plan
            src_row = pl.lw * 3
            self.src_tiles = max(2 if pl.k is not None else 4, (4 * int(chunk_bytes)) // max(1, src_row * pl.tile_rows))
            cap = min((self.src_tiles + 1) * pl.tile_rows + int(np.ceil(pl.rel)) + 2, pl.lh)
            self.pinned = [_take_pinned((cap, pl.lw, 3)) for _ in range(nb)]
            self.stage = [torch.empty((cap, pl.lw, 3), dtype=torch.uint8, device=self.dev) for _ in range(nb)]
            if pl.k is None:
                self.col_tabs = [torch.from_numpy(np.ascontiguousarray(t)).to(self.dev) for t in pl.col_tables()]
        else:
            self.pinned = [_take_pinned((min(self.chunk, max(1, self.rows)), self.w, 3)) for _ in range(nb)]
        self.busy = [None] * nb  # event after which a staging buffer may be overwritten
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.next_row, self.k, self.last_event = 0, 0, None
        self.read_s = 0.0  # seconds the reads / decodes took (bench.py --mode ingest)
        self._cv, self._err, self._thr, self._stop = threading.Condition(), None, None, False
        if self.ahead and self.rows > 0:
            self._thr = threading.Thread(target=self._produce, name="cerb-slab-upload", daemon=True)
            self._thr.start()

    def _issue_one(self):
        import time

        i = self.k % len(self.pinned)
        if self.busy[i] is not None:
            self.busy[i].synchronize()
        a = self.y0 + self.next_row
        if self.plan is not None:
            ev, n = self._issue_resampled(i, a)
        else:
            n = min((a // self.chunk + 1) * self.chunk - a, self.rows - self.next_row)
            t0 = time.perf_counter()
            np.copyto(self.pinned[i][:n].numpy(), self.host[self.y0 + self.next_row: self.y0 + self.next_row + n])
            self.read_s += time.perf_counter() - t0
            with torch.cuda.stream(self.copy_stream):
                self.slab[self.next_row: self.next_row + n].copy_(self.pinned[i][:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
        with self._cv:
            self.busy[i] = self.last_event = ev
            self.next_row += n
            self.k += 1
            self._cv.notify_all()

    def _issue_resampled(self, i, a):
        """output rows [a, a + n) of the slide from the stored level's rows: decode -> pinned -> device staging -> reduction into the slab"""
        import ctypes as C
        import time

        pl = self.plan
        b = pl.out_rows_for_source_tiles(a, 1 if self.k == 0 else self.src_tiles)  # (the first chunk is the short one: the first batch starts early)
        n = min(b - a, self.rows - self.next_row)
        sy0, sy1 = pl.source_rows(a, a + n)
        m = sy1 - sy0
        assert 0 < m <= self.pinned[i].shape[0], (m, self.pinned[i].shape, a, n)
        t0 = time.perf_counter()
        pl.read(sy0, sy1, out=self.pinned[i].numpy())
        self.read_s += time.perf_counter() - t0
        L = _lib.lib()
        dst = self.slab[self.next_row: self.next_row + n]
        with torch.cuda.stream(self.copy_stream):
            self.stage[i][:m].copy_(self.pinned[i][:m], non_blocking=True)
            st = C.c_void_p(self.copy_stream.cuda_stream)
            if pl.k is not None:
                _lib.check(L.cerb_resample_box(self.stage[i].data_ptr(), pl.lw * 3, m, pl.lw, pl.k, dst.data_ptr(), self.w * 3, n, self.w, st))
            else:
                rt = [torch.from_numpy(np.ascontiguousarray(t)).to(self.dev) for t in pl.row_tables(a, a + n, sy0, m)]
                self.tabs[i] = rt  # alive until this buffer's event has passed
                ct = self.col_tabs
                _lib.check(L.cerb_resample_area(self.stage[i].data_ptr(), pl.lw * 3, m, pl.lw, dst.data_ptr(), self.w * 3, n, self.w,
                                                rt[0].data_ptr(), rt[1].data_ptr(), rt[2].data_ptr(), rt[3].data_ptr(), int(rt[0].shape[1]),
                                                ct[0].data_ptr(), ct[1].data_ptr(), ct[2].data_ptr(), ct[3].data_ptr(), int(ct[0].shape[1]), st))
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return ev, n

    def _produce(self):
        try:
            torch.cuda.set_device(self.dev)
            while self.next_row < self.rows and not self._stop:
                self._issue_one()
        except BaseException as e:  # handed to the caller's thread at its next upload_until
            with self._cv:
                self._err = e
                self._cv.notify_all()

    def upload_until(self, n_rows):
        n_rows = min(int(n_rows), self.rows)
        if self._thr is None:
            while self.next_row < n_rows:
                self._issue_one()
        else:
            with self._cv:
                while self.next_row < n_rows and self._err is None:
                    self._cv.wait(0.5)
                if self._err is not None:
                    raise self._err
        with self._cv:
            ev = self.last_event
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

    def close(self):
        """Stop the producer (a caller that does not consume the whole band) and wait for it."""
        self._stop = True
        if self._thr is not None:
            self._thr.join()

    def __del__(self):
        try:  # the staging buffers go back to the pool once nothing is copying out of them
            self._stop = True
            if self._thr is not None and self._thr.is_alive():
                self._thr.join()
            for ev in self.busy:
                if ev is not None:
                    ev.synchronize()
            _give_pinned(self.pinned)
            self.pinned = []
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass
