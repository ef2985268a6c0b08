"""bench.py -- headline benchmark of the Cerberus tiled-inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default (`--mode wsi`): the whole `run_infer_wsi.py` job on ONE synthetic 40000 x 40000 x 3 slide (BASELINE.json north_star /
configs[2-3]; 24,649 tiles of 256 x 256) resident in HBM, sharded over the N ranks by contiguous bands of patch rows (STRONG
scaling: the slide is fixed, every rank owns 1/N of it):
    timed region = K steps, step k = stripe k of this rank's band: on-device patch gather -> forward of all six heads -> fused
                   softmax / crop / argmax -> scatter into the band's canvases (cerberus_amd.wsi.WSIRunner.infer_patches),
                 + the slide's tail: halo exchange with the neighbouring ranks (RCCL send/recv over xGMI), on-GPU labelling of the
                   band (nuclei at full resolution, gland / lumen at x0.5, lumen-in-gland masking), instance tables, slide-global
                   ids (two all-gathers), and the gather of the int32 label bands + class maps onto rank 0 (RCCL gather)
                   (cerberus_amd.shard_postproc.postprocess_bands_and_gather -- the function run_infer_wsi.py calls).
A random-weight network paints slide-sized blobs, so the labelling leg reads seeded STRUCTURED probability maps of the slide's
size instead (600 nuclei / Mpx of radius 4-9 px, glands of 25-150 px; SURVEY.md par.8d cfg 3): same kernels, same band
protocol, realistic instance counts (~1 M nuclei).  `value` = slide pixels / wall time of that whole region (max over ranks);
`config.inference_Mpx_s` is the same slide over the K inference steps alone.
`--mode batch`: the inner loop alone on 32 resident tiles (BASELINE.json configs[1]); `--mode train`: configs[4]; `--mode ingest`: a JPEG-tiled
pyramidal TIFF on disk through reader -> decode pool -> upload-ahead -> inference, beside the resident figure (`--ingest-codec deflate | lzw`: lossless
tiles through libcerberus_host.so's one-call-per-window reader; `--ingest-base-mpp 0.25`: a 40x scan reduced on the device).
Beside `value` the default line carries (all outside the timed region): `config.conv_algo` / `config.precision` (the 3x3 algorithm the headline ran on, the
calibration measurement, what the head kernels' max-|logit| guard saw over the job's batches), `config.other_conv_algos` (the same slide's inference on
F(2x2) and on the direct kernels), `dat` (instance tables + contours + the .dat writer; with several ranks `dat.per_rank_arrays`: tables + contours where
the instances live, compact arrays gathered, bytes into rank 0 either way), `ref_tiling`, `ingest` (a 12288^2 JPEG-tiled TIFF; `ingest_40x`: stored at 0.25 mpp; `ingest_deflate`: lossless tiles through the native reader), `batch_step`, `train_step`,
`dice_vs_reference`, `cpu_baseline`.  `--tail-from-inference`: the tail labels the canvases the timed inference wrote.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BATCH, TILE = 32, 256
WSI_BATCH = int(os.environ.get("CERB_WSI_BATCH", "0"))  # tiles per forward of the slide job; 0 = 64 with two streams (152.4 Mpx/s; 96: 151.9, 48: 150.6), 96 with one (149.9)
MARGIN = 1024  # halo rows exchanged between neighbouring bands (full resolution): above the tallest gland cluster of the structured maps
MARGINS = {"Nuclei": 128, "Gland": MARGIN, "Lumen": 512}  # per tissue: nuclei are < 30 px (the reference's own tile margin is 64)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def _cpu_forward_rate(sd, kw, threads, n_tiles, iters, budget_s):
    """Mpx/s of the oracle's forward + infer_step wrapper at `threads` torch threads on a sample bounded to about budget_s seconds."""
    from oracle import net_ref

    torch.set_num_threads(threads)
    tiles = np.random.RandomState(1).randint(0, 256, (n_tiles, TILE, TILE, 3)).astype(np.uint8)
    t0 = time.perf_counter()
    net_ref.infer_step(sd, tiles[:1], TILE, kw["considered_tasks"], kw["decoder_kwargs"])  # warm-up, also sizes the sample
    t1 = time.perf_counter() - t0
    if t1 > 0.6 * budget_s:  # a single tile already fills the budget: it IS the sample
        return TILE * TILE / t1 / 1e6, 1, 1, True
    if t1 * n_tiles * iters > budget_s:
        iters = 1
        n_tiles = max(1, min(n_tiles, int(0.7 * budget_s / t1)))
        tiles = tiles[:n_tiles]
    t0 = time.perf_counter()
    for _ in range(iters):
        net_ref.infer_step(sd, tiles, TILE, kw["considered_tasks"], kw["decoder_kwargs"])
    dt = time.perf_counter() - t0
    return iters * n_tiles * TILE * TILE / dt / 1e6, iters, n_tiles, False


def cpu_baseline(sd, kw, n_tiles=8, iters=1):
    """Oracle (CPU restatement of the reference path, PyTorch-CPU fp32) on the host cores -- reported, not optimised.  SURVEY par.8(d) asks for the
    host's cores with the count stated; oneDNN's small-batch convolutions do not scale to a 256-thread host (round 4: 16 threads 0.15 Mpx/s, all 256
    threads 0.0009 Mpx/s -- the warm-up tile alone took 70 s), so the thread count is SWEPT (16 / 32 / 64 / 128, each on the same bounded
    sample, ~5 s apiece) and the best one is reported as `value` with its `cores`; `sweep` carries every point, `all_cores` the whole-host figure
    when the sweep has not already collapsed below half of its best by 128 threads."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cand = sorted(set(min(avail, c) for c in (16, 32, 64, 128)))
    sweep, best = [], None
    for c in cand:
        r, it_, nt_, single_ = _cpu_forward_rate(sd, kw, c, n_tiles, iters, 5.5)
        sweep.append({"threads": c, "Mpx_s": round(r, 4), "sample": "%d x %d tiles%s" % (it_, nt_, " (warm-up tile only)" if single_ else "")})
        if best is None or r > best[0]:
            best = (r, it_, nt_, single_, c)
        if single_:  # a single tile already overran the budget at this count: more threads only get slower
            break
    rate, it, nt, single, cores = best
    out_all = None
    if avail > cand[-1]:
        if sweep[-1]["Mpx_s"] >= 0.5 * rate and not sweep[-1]["sample"].endswith("only)"):
            r2, it2, nt2, single2 = _cpu_forward_rate(sd, kw, avail, n_tiles, 1, 6.0)
            out_all = {"value": round(r2, 4), "unit": "Mpx/s", "cores": avail, "sample": "%d x %d tiles%s, %d threads" % (it2, nt2, " (warm-up tile only)" if single2 else "", avail)}
        else:
            out_all = {"value": None, "unit": "Mpx/s", "cores": avail,
                       "sample": "not run: the sweep is already at %.4f Mpx/s with %d threads (best %.4f with %d); round 4 measured 0.0009 Mpx/s with all 256 threads "
                                 "(profiles/r04_bench_wsi_40000.json), one tile = 70 s" % (sweep[-1]["Mpx_s"], sweep[-1]["threads"], rate, cores)}
    torch.set_num_threads(cores)
    # post-processing oracle (C restatement of loader/postproc.py + skimage/scipy, one core) on a 2048^2 structured map
    from oracle import postproc_ref, synth

    pm = synth.nuclei_maps(2048, 2048, 7, 600.0, noise=0.02)
    t0 = time.perf_counter()
    postproc_ref.proc(pm, "Nuclei")
    pp_dt = time.perf_counter() - t0
    return {
        "postproc_nuclei_Mpx_s_1core": round(2048 * 2048 / pp_dt / 1e6, 2),
        "postproc_note": "oracle/postproc_ref.c -- this build's C restatement of loader/postproc.py's nuclei branch on ONE host core, NOT the reference: the "
                         "reference's own skimage / scipy path does ~8 Mpx/s per core (BASELINE.md), about 5x slower than this figure",
        "value": round(rate, 4),
        "unit": "Mpx/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d x %d tiles of %dx%d, all six heads, forward + infer_step wrapper, torch-CPU fp32, %d threads (host has %d)"
        % (it, nt, TILE, TILE, cores, avail),
        "sweep": sweep,
        "all_cores": out_all,
    }


def cpu_baseline_train(sd, kw):
    """The training step on the host cores: oracle/train_step_ref.py (torch-autograd restatement of the reference's train_step) on a
    BOUNDED sample -- 2 tiles of 448 x 448 (BatchNorm in training mode needs more than one sample), one warm-up step + one timed."""
    from oracle import train_step_ref

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)
    torch.set_num_threads(cores)
    heads = {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}
    rs = np.random.RandomState(5)
    n, hw = 2, TRAIN_TILE
    img = rs.randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    tg = {h: (rs.randint(0, c, (n, 1, 1, 1)) if h == "Patch-Class" else (rs.rand(n, hw, hw, 1) < 0.3) * rs.randint(1, c, (n, hw, hw, 1))).astype(np.float32)
          for h, c in heads.items()}
    has = np.ones((n, len(heads)), bool)
    step = train_step_ref.make_step({k: v.numpy() for k, v in sd.items()}, kw["decoder_kwargs"], kw["considered_tasks"])
    t0 = time.perf_counter()
    step(img, tg, has)
    dt = time.perf_counter() - t0
    timed = "first"
    if dt < 15.0:
        t0 = time.perf_counter()
        step(img, tg, has)
        dt = time.perf_counter() - t0
        timed = "second"
    return {"value": round(n / dt, 4), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": "%s of two whole training steps (train-mode forward, six losses, autograd backward, Adam) on %d tiles of %dx%d, torch-CPU fp32, %d threads (host has %d)"
                      % (timed, n, hw, hw, cores, avail)}


TRAIN_BATCH, TRAIN_TILE = 16, 448  # BASELINE.json configs[4]: batch 16; 448 x 448 is the reference's training patch (paramset.yml)


def dice_vs_reference():
    """BASELINE.json metric, second half ("per-head Dice vs ref"): the committed fixture tests/golden/net_cfg2_all.npz holds the REFERENCE's own
    infer_step outputs (oracle/gen_golden_net.py imported /root/reference in the build container: 3 seeded 256^2 tiles, all six heads, fp32).  The same
    tiles go through the HIP path here, outside every timed region; per head: Dice (2 |A & B| / (|A| + |B| + 1e-8), models/run_desc.py:526-531) of the
    INST foreground (probability > 0.5) per channel / of every TYPE class with support, the smallest of them reported, and the largest absolute
    difference of the probability maps (north star: 1e-4)."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.run_desc import infer_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "net_cfg2_all.npz")
    g = np.load(path)
    tasks = [str(t) for t in g["tasks"]]
    kw = default_model_kwargs(tasks)
    m = create_model(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(g["weight_seed"]), kw["decoder_kwargs"], kw["considered_tasks"]).items()}, strict=True)
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
    crops, cs = [(0, 0), (96, 96), (192, 192)], 64  # the fixture keeps three 64 x 64 windows of the large maps (oracle/gen_golden_net.py)

    def dice(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return 2.0 * (a * b).sum() / (a.sum() + b.sum() + 1e-8)

    per_head, worst_err = {}, 0.0
    for k in out[0].keys():
        a = np.stack([out[i][k] for i in range(n)])
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = np.stack([a4[:, y:y + cs, x:x + cs] for (y, x) in crops], axis=1) if key in g else a4
        ds = []
        if k.endswith("-INST"):
            worst_err = max(worst_err, float(np.abs(got.astype(np.float64) - ref).max()))
            ds = [dice(got[..., c] > 0.5, ref[..., c] > 0.5) for c in range(ref.shape[-1]) if (ref[..., c] > 0.5).sum() > 50]
        elif k.endswith("-TYPE"):
            ds = [dice(got == cls, ref == cls) for cls in np.unique(ref) if (ref == cls).sum() > 50]
        elif k == "Patch-Class":
            worst_err = max(worst_err, float(np.abs(got.astype(np.float64) - ref).max()))
        if ds:
            per_head[k] = round(float(min(ds)), 6)
    del m
    return {"fixture": "tests/golden/net_cfg2_all.npz (the reference's own infer_step outputs: %d seeded %dx%d tiles, all six heads)" % (n, hw, hw),
            "per_head_min_dice": per_head, "min": min(per_head.values()) if per_head else None, "max_abs_err_probability_maps": worst_err}


def train_measure(model, dev, dist, world, rank, steps, warmup, backend):
    """K whole training steps (train-mode forward, six losses, backward, bucketed gradient all-reduce over the ranks, Adam, BatchNorm running
    statistics, on-device weight re-pack) on a synthetic batch resident in HBM; every rank has its own batch (weak scaling).  Returns
    (seconds for the K steps -- max over ranks --, last result, roofline of the dominant backward family, per-family rows)."""
    from cerberus_amd.losses import PARAMSET_LOSS
    from cerberus_amd.train import Adam, train_step

    model.train()
    heads = {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}
    n, hw = TRAIN_BATCH, TRAIN_TILE
    g = torch.Generator(device=dev).manual_seed(2000 + rank)
    batch = {"img": torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8, device=dev, generator=g),
             "dummy_target": np.array([list(heads)] * n, dtype=object)}
    for h, c in heads.items():
        if h == "Patch-Class":
            batch[h] = torch.randint(0, c, (n,), device=dev, generator=g).float()
        else:  # sparse foreground, as annotation masks are
            fg = torch.rand((n, hw, hw, 1), device=dev, generator=g) < 0.3
            batch[h] = (fg * torch.randint(1, c, (n, hw, hw, 1), device=dev, generator=g)).float()
    opt = Adam(lr=1.0e-4, betas=(0.9, 0.999))
    info = ({"net": {"desc": model, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None)

    def step():
        return train_step(batch, info, dist=dist, world_size=world)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # roofline leg: per-launch HIP events of ONE more step (cerb_net_profile_*): forward / data-gradient convolutions, the weight-gradient
    # kernels (executed MFMA FLOPs against the fp32 MFMA peak) and the BatchNorm passes (algorithmic bytes against the HBM peak)
    roofline, fam_rows = None, None
    try:
        model.profile(True)
        phases = {}
        train_step(batch, info, dist=dist, world_size=world, timings=phases)
        torch.cuda.synchronize()
        recs = model.profile_records()
        model.profile(False)
        fam = {}
        for _, kern, work, ms in recs:
            f = fam.setdefault(kern, [0.0, 0.0, 0])
            f[0] += work
            f[1] += ms
            f[2] += 1
        fam_rows = []
        for kern, (work, ms, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            row = {"kernel": kern, "launches": cnt, "ms_per_step": round(ms, 3)}
            base = kern[6:] if kern.startswith("dgrad:") else kern
            if kern in HBM_FAMILIES or kern.startswith("bn_"):  # `work` = algorithmic bytes
                row.update(bound="hbm", achieved=round(work / (ms * 1e-3) / 1e9, 1), unit="GB/s", frac=round(work / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            elif work > 0:
                ex = work / (ms * 1e-3) / 1e12 * (0.25 if base.startswith(("conv_wino4", "wgrad_wino4")) else 16.0 / 36.0 if base.startswith("conv_wino") else 1.0)
                row.update(bound="mfma", achieved=round(ex, 2), unit="TFLOP/s", frac=round(ex / PEAK_F32_MFMA_TFLOPS, 4))
            fam_rows.append(row)
        # measured HBM traffic per family and step (profiles/r05_train_pmc_hbm.json: FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950-corrected by
        # scripts/rocprof_summary.py pmc_step); for the bandwidth-bound families beside their algorithmic bytes
        pmc = None
        for cand in ("r06_train_pmc_hbm.json", "r05_train_pmc_hbm.json"):
            pth = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(pth):
                pmc = json.load(open(pth))
                pmc_src = cand
        if pmc:
            for row in fam_rows:
                syms = _TRAIN_SYMBOLS.get(row["kernel"])
                if not syms:
                    continue
                hit = [v for k, v in pmc.items() if any(sname in k for sname in syms)]
                if hit:
                    row["traffic"] = int(sum(v.get("hbm_bytes_per_step", 0.0) for v in hit))
                    row["traffic_unit"] = "HBM bytes per step (PMC passes, profiles/%s)" % pmc_src
                    if row.get("bound") == "hbm" and fam[row["kernel"]][0] > 0:
                        row["traffic_over_algorithmic"] = round(row["traffic"] / fam[row["kernel"]][0], 3)
        # the phases of the step outside the handle's own records (device time between stream events), and what the records leave of the handle's call
        in_handle = sum(r["ms_per_step"] for r in fam_rows)
        fam_rows.append({"kernel": "(inside cerb_net_train_grads, between the records: event gaps, tape bookkeeping)", "launches": 0,
                         "ms_per_step": round(max(phases.get("train_grads", in_handle) - in_handle, 0.0), 3)})
        for name in ("batch_to_device", "allreduce", "adam_and_running_stats", "param_update_and_repack", "raw_payload"):
            if name in phases:
                fam_rows.append({"kernel": "(" + name + ")", "launches": 0, "ms_per_step": round(phases[name], 3)})
        attributed = sum(r["ms_per_step"] for r in fam_rows)
        dom = next((r for r in fam_rows if r["kernel"].startswith("wgrad")), fam_rows[0])
        roofline = dict(dom, peak=PEAK_F32_MFMA_TFLOPS if dom.get("bound") == "mfma" else HBM_PEAK_GBS, traffic=dom.get("traffic"),
                        attributed_ms=round(attributed, 3),
                        note="dominant backward family; `kernels` attributes the whole profiled step: every launch family inside the handle has its own "
                             "per-launch records (forward, data gradients, weight gradients, BatchNorm, pointwise layers, pooling, up-sampling, losses, zero "
                             "fills), the phases around it (upload, all-reduce, Adam + running statistics, parameter copy + re-pack, visualisation "
                             "payload) are timed between stream events; attributed_ms is their sum, to compare with ms_per_step (the profiled step runs "
                             "with an event pair around every launch AND on one stream -- the timed steps queue the convolutions' weight gradients on the handle's side "
                             "stream (CERB_WGRAD_SIDE, default on), where they overlap the BatchNorm backward passes and the data gradients' tails: the "
                             "timed step is ~5 % shorter than attributed_ms)")
    except Exception as e:  # the profile leg never fails the benchmark line
        roofline = {"error": str(e)[:200]}
    return dt, res, roofline, fam_rows


def train_leg(args, model, dev, dist, world, rank, sd=None, kw=None):
    """--mode train: BASELINE.json configs[4] as its own benchmark line."""
    n, hw = TRAIN_BATCH, TRAIN_TILE
    dt, res, roofline, fam_rows = train_measure(model, dev, dist, world, rank, args.steps, args.warmup, args.backend)
    if rank == 0:
        fwd_flops = model.flops(n, hw, hw)
        print(json.dumps({
            "metric": "training tiles/sec (multi-task step, all 6 losses)",
            "value": round(world * args.steps * n / dt, 3),
            "unit": "tiles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "Multi-task training step (train-mode forward + 6 losses + backward + Adam + BN running statistics + weight re-pack), "
                            "batch=%d %dx%dx3 uint8 tiles per GPU, fp32 (BASELINE.json configs[4])" % (n, hw, hw),
                "batch_tiles": n,
                "tile": hw,
                "Mpx_s": round(world * args.steps * n * hw * hw / dt / 1e6, 3),
                "approx_tflops_3x_forward": round(3.0 * fwd_flops / (dt / args.steps) / 1e12 * world, 2),
                "parallelism": "data-parallel x%d, bucketed gradient all-reduce (%s)" % (world, args.backend if world > 1 else "none at 1 GPU"),
                "last_overall_loss": round(float(res["EMA"]["overall_loss"]), 4),
            },
            "roofline": roofline,
            "kernels": fam_rows,
            "multi_gpu": dict(args.identity),
            "cpu_baseline": cpu_baseline_train(sd, kw) if (world == 1 and sd is not None and not args.no_cpu_baseline) else None,
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# training families -> kernel symbols (substrings) whose PMC rows make up the family's HBM traffic
_TRAIN_SYMBOLS = {
    "wgrad_wino4<f4x4>": ["wgrad_wino_kernel", "wgrad_wino_reduce_kernel"],
    "wgrad<ks3,s1>": ["wgrad_kernel<3, 1>"], "wgrad<ks3,s2>": ["wgrad_kernel<3, 2>"],
    "bn_bwd": ["bn_bwd_partial_kernel", "bn_bwd_apply_kernel", "bn_bwd_finalize_kernel"],
    "bn_fwd": ["bn_apply_kernel", "bn_partial_kernel", "bn_finalize_kernel", "bn_partial_fold_kernel"],
    "conv_wino4<f4x4,16x16x2>": ["conv_wino4_kernel<false, 1>"], "dgrad:conv_wino4<f4x4,16x16x2>": ["conv_wino4_kernel<false, 0>", "conv_wino4_kernel<true, 0>", "conv_wino4_kernel<false, 2>"],
    "conv_wino4b<f4x4,16x16>": ["conv_wino4b_kernel<false, 1, false>"], "dgrad:conv_wino4b<f4x4,16x16>": ["conv_wino4b_kernel<false, 0, false>", "conv_wino4b_kernel<true, 0, false>", "conv_wino4b_kernel<false, 2, false>"],
    "conv_wino4b<f4x4,16t>": ["conv_wino4b_kernel<false, 1, true>"], "dgrad:conv_wino4b<f4x4,16t>": ["conv_wino4b_kernel<false, 0, true>", "conv_wino4b_kernel<true, 0, true>", "conv_wino4b_kernel<false, 2, true>"],
    "head_fwd1": ["head_fwd1_kernel"], "head_fwd2": ["head_fwd2_kernel"], "head_bwd1": ["head_bwd1_kernel"], "head_bwd2": ["head_bwd2_kernel"],
    "upadd_bwd": ["upadd_bwd_fused_kernel"], "upsample2_add": ["upsample2_add_kernel"], "maxpool_bwd": ["maxpool_bwd_kernel", "maxpool_bwd_idx_kernel"], "maxpool3x3s2": ["maxpool3x3s2_kernel", "maxpool3x3s2_idx_kernel"],
    "stem_wgrad": ["stem_wgrad_mfma_kernel"], "stem_conv7x7": ["stem_conv7x7_kernel"],
}

# ---- kernel-family table of one batch step (per-launch HIP events on the launch stream, cerb_net_profile_*) ----------------------
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md
HBM_FAMILIES = {"maxpool3x3s2", "maxpool_bwd", "upsample2_add", "upadd_bwd", "crop_gap", "crop_gap_bwd", "bias_colsum", "zero_fill", "head_loss", "head_fwd1", "head_fwd2", "head_bwd1", "head_bwd2"}
XGMI_LINK_GBS = 153.0   # per peer link, SURVEY.md par.8e
_SYMBOL = {"conv_wino4p<f4x4,16x16x2,planar>": "void conv_wino4p_kernel<1>(ConvParams)",
           "conv_wino4p<f4x4,16x16x2,planar,half-res>": "void conv_wino4p_kernel<0>(ConvParams)",
           "conv_wino4b<f4x4,16x16>": "void conv_wino4b_kernel<false, 0, false>(ConvParams)",
           "conv_wino4b<f4x4,16x16,res>": "void conv_wino4b_kernel<true, 0, false>(ConvParams)",
           "conv_wino4b<f4x4,16t>": "void conv_wino4b_kernel<false, 0, true>(ConvParams)",
           "conv_wino4b<f4x4,16t,res>": "void conv_wino4b_kernel<true, 0, true>(ConvParams)",
           "conv_wino4<f4x4,16x16x2>": "void conv_wino4_kernel<false, 0>(ConvParams)",
           "conv_wino4<f4x4,16x16x2,res>": "void conv_wino4_kernel<true, 0>(ConvParams)",
           "conv_wino<f2x2,8x16>": "void conv_wino_kernel<false>(ConvParams)",
           "conv_wino<f2x2,8x16,res>": "void conv_wino_kernel<true>(ConvParams)"}


def _family_bytes(kern, n, launches=4):
    """Algorithmic HBM bytes of one batch step for the families that are bandwidth bound (fp32, batch n of 256^2 tiles)."""
    levels = ((32, 256), (64, 128), (128, 64), (256, 64))  # per level: read prev (5 decoders, a quarter of the output each) + skip (once), write 5 sums
    if kern == "upsample2_add":  # the NHWC launches: all four levels, or the first three when the last level is tile-planar
        tot = 0
        for hw, c in levels[:launches]:
            out = n * hw * hw * c * 4
            tot += 5 * out // 4 + out + 5 * out
        return tot
    if kern == "upsample2_add_planar":  # the last level, and the level below it when both run tile-planar (2 launches)
        tot = 0
        for hw, c in levels[4 - launches:]:
            out = n * hw * hw * c * 4
            tot += 5 * out // 4 + out + 5 * out
        return tot
    if kern == "maxpool3x3s2":
        return n * 256 * 256 * 64 * 4 + n * 128 * 128 * 64 * 4
    return None


def kernel_table(model, step, n_tiles):
    model.profile(True)
    step()
    torch.cuda.synchronize()
    recs = model.profile_records()
    model.profile(False)
    fam = {}
    for name, kern, fl, ms in recs:
        f = fam.setdefault(kern, [0.0, 0.0, 0])
        f[0] += fl
        f[1] += ms
        f[2] += 1
    total_ms = sum(v[1] for v in fam.values())
    rows = []
    for kern, (fl, ms, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        row = {"kernel": kern, "launches": cnt, "ms_per_step": round(ms, 4), "share": round(ms / total_ms, 4)}
        by = _family_bytes(kern, n_tiles, cnt)
        if by is not None:
            gbs = by / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=round(gbs, 1), unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), algorithmic_bytes=by)
        elif fl > 0:
            alg = fl / (ms * 1e-3) / 1e12
            # F(2x2,3x3) executes 16 of the 36 multiplies of 4 outputs, F(4x4,3x3) 36 of the 144 multiplies of 16 outputs
            ex = alg * (0.25 if kern.startswith("conv_wino4") else 16.0 / 36.0 if kern.startswith("conv_wino") else 1.0)
            row.update(bound="mfma", achieved=round(ex, 2), unit="TFLOP/s", frac=round(ex / PEAK_F32_MFMA_TFLOPS, 4), algorithmic_tflops=round(alg, 2))
        rows.append(row)
    dom = rows[0]
    fl, ms, cnt = fam[dom["kernel"]]
    traffic = None
    for cand in ("r06_bench_pmc_hbm.json", "r05_bench_pmc_hbm.json", "r04_bench_pmc_hbm.json", "r03_bench_pmc_hbm.json", "r02_bench_pmc_hbm.json", "r01_bench_pmc_hbm.json"):
        pth = os.path.join(ROOT, "profiles", cand)
        sym = _SYMBOL.get(dom["kernel"])
        if sym and os.path.exists(pth):
            traffic = json.load(open(pth)).get(sym, {}).get("hbm_bytes_per_launch")
            traffic_src = cand
            break
    # `achieved` / `frac`: the MFMA FLOPs the kernel EXECUTES per second against the dense fp32 MFMA peak (the share of the matrix
    # pipe it fills).  The algorithmic (direct-convolution, SURVEY par.8d) rate is 4x that for the F(4x4,3x3) kernel (36/16 for
    # F(2x2,3x3)) and is carried beside it: it can exceed the peak because Winograd skips 3/4 (5/9) of the multiplies.
    roofline = {
        "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": dom["frac"], "algorithmic_tflops": dom.get("algorithmic_tflops"),
        "algorithmic_frac": round(dom.get("algorithmic_tflops", 0.0) / PEAK_F32_MFMA_TFLOPS, 4),
        "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC passes, profiles/%s)" % (traffic_src if traffic else "-"),
        "algorithmic_gflop_per_launch": round(fl / cnt / 1e9, 3), "launches_per_step": cnt, "avg_launch_ms": round(ms / cnt, 4),
        "kernel_share_of_step": dom["share"],
        "whole_step": {"ms_sum_of_kernels": round(total_ms, 3),
                       "executed_mfma_frac": round(sum(r["achieved"] * r["ms_per_step"] for r in rows if r.get("bound") == "mfma")
                                                   / total_ms / PEAK_F32_MFMA_TFLOPS, 4)},
    }
    return roofline, rows


CONV_ALGO_NAMES = {6: "Winograd F(4x4,3x3) on maps of 16x16 and more, F(2x2,3x3) below (default)", 5: "Winograd F(4x4,3x3), conv_wino4", 7: "Winograd F(4x4,3x3), conv_wino4b",
                   1: "Winograd F(2x2,3x3)", 0: "direct implicit GEMM"}


def px_all(h, w):
    return float(h) * float(w)


def batch_loop(model, dev, rank, steps, warmup, dist, backend):
    """configs[1]: 32 resident tiles, forward + fused output wrapper + scatter into a 4 x 8-tile canvas.  Returns (dt, step, n_tiles)."""
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    tiles = torch.randint(0, 256, (BATCH, TILE, TILE, 3), dtype=torch.uint8, device=dev, generator=g)
    gy, gx = 4, 8
    Wc = gx * TILE
    canvas = {
        "Lumen": torch.zeros((gy * TILE, Wc, 2), dtype=torch.float32, device=dev),
        "Gland": torch.zeros((gy * TILE, Wc, 2), dtype=torch.float32, device=dev),
        "Nuclei": torch.zeros((gy * TILE, Wc, 2), dtype=torch.float32, device=dev),
        "Nuclei#TYPE": torch.zeros((gy * TILE, Wc), dtype=torch.uint8, device=dev),
        "Gland#TYPE": torch.zeros((gy * TILE, Wc), dtype=torch.uint8, device=dev),
        "Patch-Class": torch.zeros((gy * TILE, Wc), dtype=torch.float32, device=dev),
    }
    outs = [canvas[d[0]] for d in model._decoders]
    off = torch.tensor([(i // gx) * TILE * Wc + (i % gx) * TILE for i in range(BATCH)], dtype=torch.int64, device=dev)

    def step():
        model._run(tiles, TILE, TILE, outs, None, tile_off=off, row_stride=Wc, type_is_u8=True)

    for _ in range(warmup):
        step()
    dt = _timed(lambda: [step() for _ in range(steps)], dev, dist, backend)
    return dt, step, BATCH


def _timed(fn, dev, dist, backend):
    """barrier + synchronize on both sides, max over ranks"""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def structured_band(dev, y0, rows, W, period=4096):
    """Seeded structured probability maps for canvas rows [y0, y0 + rows) x [0, W): a `period`^2 tile of each kind (numpy generator,
    cerberus_amd/synth_maps.py) repeated over the slide in absolute coordinates, so every rank sees the same slide."""
    from cerberus_amd import synth_maps as synth

    yi = (torch.arange(y0, y0 + rows, device=dev) % period)
    xi = (torch.arange(0, W, device=dev) % period)
    out = {}
    nuc = torch.from_numpy(synth.nuclei_maps(period, period, 7, 600.0, noise=0.02)).to(dev)
    out["Nuclei-INST"] = nuc.index_select(0, yi).index_select(1, xi)
    del nuc
    gl = torch.from_numpy(synth.gland_maps(period, period, 9, n=90, noise=0.02, holes=0.3)).to(dev)  # ~13 % gland area
    out["Gland-INST"] = gl.index_select(0, yi).index_select(1, xi)
    del gl
    lu = torch.from_numpy(synth.blob_maps(period, period, 11, 260, 8.0, 40.0, rim=2.0, sharp=1.0, noise=0.02)).to(dev)
    out["Lumen-INST"] = lu.index_select(0, yi).index_select(1, xi)
    return out


def ingest_leg(model, dev, side, batch, streams, tmpdir=None, sweep=(1, 2, 4, 8, 16, 32, 64), base_mpp=0.5, codec="jpeg"):
    """Real-slide ingest (SURVEY par.8f rank 2): a JPEG-tiled pyramidal TIFF of side^2 pixels (256-pixel tiles of stain-field
    texture, quality 80, written with cerberus_amd.reader.write_tiled_tiff) through the path run_infer_wsi.py takes for a slide on disk --
    reader rows -> tile decode on the reader's thread pool -> pinned chunks -> copy stream (wsi.SlabUploader, a producer thread ahead of the
    inference) -> gather / forward / scatter (WSIRunner) -- against the same pixels resident in HBM.  Reports decode Mpx/s per thread count
    (host only), upload GB/s of decoded rows, inference Mpx/s resident and from the file, and the thread count that saturates this GPU.
    base_mpp = 0.25: a 40x scan -- the file holds (2 side)^2 pixels and is read at the 0.5 mpp the network runs on: the stored level's tiles are
    decoded by worker processes (threads stop at ~350 Mpx/s of decode, a quarter of what this needs) and reduced x2 on the device
    (cerb_resample_box); the canvases must equal those of the host-reduced rows."""
    import io
    import tempfile

    from PIL import Image

    from cerberus_amd import reader as rd
    from cerberus_amd.synth_tiles import stain_field
    from cerberus_amd.wsi import SlabUploader, WSIRunner

    H = W = int(side)
    kf = int(round(0.5 / base_mpp))
    assert kf >= 1 and abs(0.5 / base_mpp - kf) < 1e-9
    fH, fW = H * kf, W * kf
    t0 = time.perf_counter()
    rs = np.random.RandomState(17)
    atlas = [np.clip(stain_field(TILE, 100 + i).astype(np.int16) + rs.randint(-10, 11, (TILE, TILE, 3)), 0, 255).astype(np.uint8) for i in range(64)]
    ny, nx = -(-fH // TILE), -(-fW // TILE)
    pick = rs.randint(0, 64, (ny, nx))
    img = np.empty((ny * TILE, nx * TILE, 3), np.uint8)
    for ty in range(ny):
        for tx in range(nx):
            img[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = atlas[pick[ty, tx]]
    img = img[:fH, :fW]
    cache = {}

    def enc(t):
        key = t.tobytes()
        if key not in cache and codec == "jp2k":
            # Aperio compression 33005: a raw JPEG 2000 codestream per tile, R, G, B components, 9-7 wavelet at 15:1 as scanners write them (OpenJPEG
            # decodes such a tile at ~7 Mpx/s per core -- a lossless one at 1.7 --: this leg is bound by the decode processes, ~22 of them per GPU)
            b = io.BytesIO()
            Image.fromarray(t).save(b, format="JPEG2000", no_jp2=True, irreversible=True, quality_mode="rates", quality_layers=[15])
            cache[key] = b.getvalue()
        if key not in cache and codec != "jpeg":
            # lossless tiles (generic tiled TIFFs: bioformats / libvips exports): decoded by libcerberus_host.so, one native call per window
            import zlib

            # (LZW with the horizontal predictor, as such files are written: write_tiled_tiff(predictor=2) hands over the differenced tile -- still
            #  one of 64 distinct byte strings, so the plain-Python encoder runs 64 times)
            cache[key] = rd.tiff_lzw_encode(key) if codec == "lzw" else zlib.compress(key, 6)
        if key not in cache:
            # Aperio-style: the R, G, B planes ARE the stream's three components (PhotometricInterpretation RGB, no chroma subsampling, no JFIF marker)
            b = io.BytesIO()
            Image.merge("YCbCr", [Image.fromarray(t[..., i]) for i in range(3)]).save(b, format="JPEG", quality=80, subsampling=0)
            raw = b.getvalue()
            n = (raw[4] << 8) | raw[5]
            cache[key] = raw[:2] + raw[4 + n:] if raw[2:4] == b"\xff\xe0" else raw
        return cache[key]

    td = tempfile.mkdtemp(dir=tmpdir)
    path = os.path.join(td, "slide.tif")
    rd.write_tiled_tiff(path, [img, np.ascontiguousarray(img[::4, ::4])], tile=TILE, mpp=base_mpp, encode=(enc, {"jpeg": 7, "deflate": 8, "lzw": 5, "jp2k": 33005}[codec]),
                        predictor=2 if codec == "lzw" else 1)
    src_band = None if codec in ("jpeg", "jp2k") else np.ascontiguousarray(img[: min(fH, 512)])  # (lossless codecs: the decoded pixels are held to these)
    del img
    build_s = time.perf_counter() - t0
    res = {"slide": [H, W], "stored": {"pixels": [fH, fW], "mpp": base_mpp, "read_at_mpp": 0.5}, "file": {"format": ("pyramidal TIFF, %d x %d JPEG tiles (Aperio-style: RGB components, 4:4:4, quality 80) + a x4 level" % (TILE, TILE)) if codec == "jpeg" else
                                    ("pyramidal TIFF, %d x %d JPEG 2000 tiles (Aperio compression 33005, 9-7 wavelet at 15:1) + a x4 level (decoded by OpenJPEG behind PIL, on threads / worker processes)" % (TILE, TILE)) if codec == "jp2k" else
                                    ("pyramidal TIFF, %d x %d %s tiles + a x4 level (decoded by libcerberus_host.so: one native call per window)" % (TILE, TILE, {"deflate": "deflate", "lzw": "LZW + horizontal-predictor"}[codec])), "MB": round(os.path.getsize(path) / 1e6, 1),
                                    "tiles": int(ny * nx), "build_s": round(build_s, 1)}}
    try:
        reader = rd.WSIReader.open(input_img=path)
        rows = reader.rows(0.5, "mpp")
        assert tuple(rows.shape) == (H, W, 3)
        # (1) decode alone, host only: 2048 stored rows per point (the first point also warms the page cache) -- on threads, then on worker processes
        span = min(fH, 2048)
        old = os.environ.get("CERB_DECODE_THREADS")
        old_procs = os.environ.get("CERB_DECODE_PROCS")
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        dec = []
        os.environ["CERB_DECODE_PROCS"] = "0"
        reader._read_level(0, 0, 0, fW, span)
        for n in [c for c in sweep if c <= max(1, avail)]:
            os.environ["CERB_DECODE_THREADS"] = str(n)
            t0 = time.perf_counter()
            a = reader._read_level(0, 0, 0, fW, span)
            dt = time.perf_counter() - t0
            dec.append({"threads": n, "Mpx_s": round(span * fW / dt / 1e6, 1)})
        ref_px = a
        if src_band is not None:  # lossless: the decoded pixels ARE the source's
            assert np.array_equal(ref_px[: src_band.shape[0]], src_band), "the reader returned other pixels than the file was written from"
        proc_counts = [c for c in (2, 4, 8, 16, 32) if c <= max(1, avail // 2)] if codec in ("jpeg", "jp2k") else []  # (worker processes decode the tiles that go through PIL: JPEG, JPEG 2000)
        for n in proc_counts:
            os.environ["CERB_DECODE_PROCS"] = str(n)
            reader._read_level(0, 0, 0, fW, min(span, 512))  # the workers start here
            t0 = time.perf_counter()
            a = reader._read_level(0, 0, 0, fW, span)
            dt = time.perf_counter() - t0
            dec.append({"processes": n, "Mpx_s": round(span * fW / dt / 1e6, 1)})
            assert np.array_equal(a, ref_px)
        res["decode"] = {"host_threads_available": avail, "sweep": dec, "unit": "Mpx/s of STORED pixels"}
        # the decoded pixels are what the writer's JPEG holds (not bit-equal to the source: lossy), identical whatever the thread count
        os.environ["CERB_DECODE_THREADS"], os.environ["CERB_DECODE_PROCS"] = "1", "0"
        assert np.array_equal(reader._read_level(0, 0, 0, fW, min(span, 512)), ref_px[:min(span, 512)])
        if proc_counts:
            os.environ["CERB_DECODE_PROCS"] = str(proc_counts[-1])
        # (2) resident: the same pixels already in HBM
        run = WSIRunner(model, (H, W), TILE, TILE, batch)
        if streams == 2:
            run.twin = model.twin()
        y0, y1 = run.slab_rows()
        os.environ["CERB_DECODE_THREADS"] = str(max(d["threads"] for d in dec if "threads" in d))
        if kf > 1:
            # a stored level finer than 0.5 mpp: the slab comes from the device reduction, and a 1024-row band of it is held to the HOST statement
            # (reader.read_bounds, 9 Mpx/s on one thread: the whole slide that way would take longer than everything else in this leg)
            up0 = SlabUploader(rows, y0, y1)
            up0.upload_until(y1 - y0)
            torch.cuda.synchronize()
            slab = up0.slab
            band = min(1024, y1 - y0)
            assert np.array_equal(slab[:band].cpu().numpy(), rows[y0:y0 + band]), "the device reduction returned other bytes than reader.read_bounds"
            host = slab.cpu().numpy()
            del up0
        else:
            host = np.ascontiguousarray(rows[y0:y1])
            slab = torch.from_numpy(host).to(dev)
        run.infer_patches(slab, y0, 0, min(run.n_patches, 4 * batch))  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.infer_band(slab, y0)
        torch.cuda.synchronize()
        resident_s = time.perf_counter() - t0
        want = {k: float(v.double().sum().item()) if v.is_floating_point() else int(v.long().sum().item()) for k, v in run.canv.items()}
        del slab
        res["inference_resident"] = {"s": round(resident_s, 3), "Mpx_s": round(H * W / resident_s / 1e6, 2)}
        # (3) upload alone: decoded rows in RAM -> pinned chunks -> HBM
        t0 = time.perf_counter()
        up = SlabUploader(host, y0, y1)
        up.upload_until(y1 - y0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["upload"] = {"GB_s": round(host.nbytes / dt / 1e9, 2), "s": round(dt, 3), "note": "pageable host rows -> %d pinned chunks of %d rows -> copy stream" % (len(up.pinned), up.chunk)}
        del up, host
        # (4) end to end from the file, per thread count (ahead = the producer thread; the last row: round 5's caller-thread reads)
        e2e = []
        trial = [c for c in (1, 4, 8, 16, 32, 64) if c <= max(1, avail)]
        for n, ahead, np_ in [(c, "1", 0) for c in trial] + [(trial[-1], "0", 0)] + [(1, "1", c) for c in proc_counts if c >= 4]:
            os.environ["CERB_DECODE_THREADS"], os.environ["CERB_UPLOAD_AHEAD"], os.environ["CERB_DECODE_PROCS"] = str(n), ahead, str(np_)
            for v in run.canv.values():
                v.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            up = SlabUploader(rows, y0, y1)
            run.infer_band(up.slab, y0, ready=up.upload_until)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            got = {k: float(v.double().sum().item()) if v.is_floating_point() else int(v.long().sum().item()) for k, v in run.canv.items()}
            assert got == want, "the file-fed run wrote other canvases than the resident run"
            e2e.append({"decode_threads": n, "decode_processes": np_, "upload_ahead": ahead == "1", "reduced_on_device": up.plan is not None, "s": round(dt, 3), "Mpx_s": round(H * W / dt / 1e6, 2), "decode_s_in_producer": round(up.read_s, 3),
                        "of_resident": round(resident_s / dt, 3)})
            del up
        os.environ.pop("CERB_UPLOAD_AHEAD", None)
        res["end_to_end_from_file"] = e2e
        best = max((e for e in e2e if e["upload_ahead"]), key=lambda e: e["Mpx_s"])
        sat = next((e for e in e2e if e["upload_ahead"] and e["Mpx_s"] >= 0.99 * best["Mpx_s"]), best)
        res["best"] = {"Mpx_s": best["Mpx_s"], "of_resident": best["of_resident"], "decode_threads": best["decode_threads"], "decode_processes": best["decode_processes"],
                       "first_that_saturates_this_gpu": {"decode_threads": sat["decode_threads"], "decode_processes": sat["decode_processes"]}}
        for name, val in (("CERB_DECODE_THREADS", old), ("CERB_DECODE_PROCS", old_procs)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
        del run
    finally:
        try:
            os.remove(path)
            os.rmdir(td)
        except OSError:
            pass
    return res


def sparse_foreground_weights(model, sd, dev, q=0.03):
    """`--tail-from-inference`: the seeded test weights paint slide-sized blobs (99 % of the nuclei map is foreground), which no labelling window can
    hold.  The same weights with the BACKGROUND logit's bias of every INST head raised by the (1 - q) quantile of `logit_inner - logsumexp(the
    others)` over the calibration tile: about q of a noise slide's pixels are then foreground, in islands as wide as the network's own spatial
    correlation -- maps the timed inference itself writes, with instances a band protocol can own.  A function of the weights alone (every rank
    derives the same shift)."""
    tile = torch.from_numpy(np.random.RandomState(20240229).randint(0, 256, (1, TILE, TILE, 3)).astype(np.uint8)).to(dev)
    lg = model(tile)
    new = {k: v.clone() for k, v in sd.items()}
    shifts = {}
    for name, hname, och, key in model._decoders:
        if hname != "INST":
            continue
        v = lg[key][0]  # (3, H, W)
        margin = v[1] - torch.logsumexp(torch.stack([v[0], v[2]]), 0)
        d = float(torch.quantile(margin.flatten().float(), 1.0 - q))
        bkey = "output_head.%s.INST.x.1.conv.bias" % name
        new[bkey][0] += d
        shifts[key] = round(d, 4)
    model.load_state_dict(new, strict=True)
    model.prepare(dev)
    return new, shifts


def _warm_handles(run, slab, y0, p0, p1):
    """Untimed warm-up of the slide runner on patches [p0, p1): two unjoined calls, so that the batch alternation reaches BOTH handles even when the
    range is a single batch (a 3072^2 test slide: a joined one-batch call stays on the first handle, and the second one then allocated the
    workspace of its algorithm inside the next timed region -- 0.5 s of hipMalloc read as 15 Mpx/s).  Recomputes, caches nothing but memory."""
    for _ in range(2):
        run.infer_patches(slab, y0, p0, p1, join=False)
    run.join()


def wsi_leg(args, model, dev, dist, world, rank, sd, kw):
    from collections import OrderedDict

    from cerberus_amd.shard_postproc import postprocess_bands_and_gather
    from cerberus_amd.wsi import WSIRunner, band_partition, check_shardable, synth_slide

    free = torch.cuda.mem_get_info(dev)[0]
    if dist is not None:  # one decision for all ranks: the smallest free HBM among them
        t = torch.tensor([float(free)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        free = float(t.item())
    side = args.slide
    if side <= 0:  # 40000^2 needs ~130 GB on one GPU (slide, 36 B/px canvases, structured maps, labels, banded workspace)
        side = 40000 if free > (140e9 if world == 1 else 270e9 / world + 8e9) + (42e9 if args.streams == 2 else 0.0) else 20000  # (a twin handle's workspace at batch 64: ~40 GB)
    H = W = side
    K = args.steps
    global WSI_BATCH
    WSI_BATCH = WSI_BATCH or (64 if args.streams == 2 else 96)
    check_shardable((H, W), TILE, world)
    fg_shifts = None
    if args.tail_from_inference:
        sd, fg_shifts = sparse_foreground_weights(model, sd, dev, q=0.02)
        # (a random network's islands are as wide as its 16x-downsampled encoder correlates -- hundreds of pixels, not a nucleus's 20: every tissue
        #  gets the gland's halo)
        global MARGINS
        MARGINS = {"Nuclei": MARGIN, "Gland": MARGIN, "Lumen": MARGIN}
    run = WSIRunner(model, (H, W), TILE, TILE, WSI_BATCH, rank, world)
    y0, y1 = run.slab_rows()
    slab = synth_slide(y1 - y0, W, y0=y0, seed=3)
    valid = max(0, min(run.band_h, H - run.r0 * TILE))
    if args.tail_from_inference:  # the labelling reads what the inference writes: views of the runner's own canvases
        struct = OrderedDict((k, v[:valid, :W]) for k, v in run.canv.items() if k.endswith("INST"))
    else:
        struct = structured_band(dev, run.r0 * TILE, valid, W)
    # stripes: this rank's patches in K contiguous pieces, cut at batch boundaries
    nb = -(-run.n_patches // WSI_BATCH)
    cuts = [min(run.n_patches, c * WSI_BATCH) for c in band_partition(nb, K)]
    max_band_px = int(args.max_band_mpx * 1e6)
    # warm-up: W stripes' worth of batches + one small labelling call per tissue (allocations, code objects, RCCL channels)
    for k in range(min(args.warmup, K)):
        run.infer_patches(slab, y0, cuts[k], min(cuts[k] + 2 * WSI_BATCH, cuts[k + 1]))
    from cerberus_amd.postproc import _workspace, postproc_device

    hw = min(valid * W, max_band_px) + 2 * (MARGIN + 64) * W if world == 1 else (valid + 2 * MARGIN + 64) * W
    side_ws = int(hw ** 0.5) + 1
    _workspace(dev, side_ws, side_ws)  # the labelling workspace of the largest call, allocated outside the timed region
    for t in ("Nuclei", "Gland", "Lumen"):
        postproc_device(struct[t + "-INST"][:512, :512], t, exact_ties=False)
    if args.streams == 2:
        # the second handle (NetDesc.twin) only when its workspace (~0.65 GB per tile of the batch) fits beside everything allocated so far AND what the
        # tail / dat / ref_tiling legs still take; otherwise the job runs on one handle and says so in config.streams
        torch.cuda.synchronize()
        need = 0.7e9 * WSI_BATCH + 0.04e9 * (valid * W / 1e6) * 1e0 + 8e9
        if torch.cuda.mem_get_info(dev)[0] > need:
            run.twin = model.twin()
            for k in range(min(max(args.warmup, 1), K)):
                _warm_handles(run, slab, y0, cuts[k], min(cuts[k] + 2 * WSI_BATCH, cuts[-1]))
            torch.cuda.synchronize()
        else:
            args.streams = 1
    if dist is not None:  # first RCCL send/recv + gather open their channels outside the timed region
        x = torch.zeros(1024, device=dev)
        lst = [torch.zeros_like(x) for _ in range(world)] if rank == 0 else None
        dist.gather(x, lst, dst=0)
        from cerberus_amd.shard_postproc import halo_exchange

        with args.watch.phase("warm-up halo exchange (opens the neighbour channels)"):
            halo_exchange(dist, rank, world, x, x.clone(), torch.zeros_like(x) if rank > 0 else None, torch.zeros_like(x) if rank < world - 1 else None)
    # ... and one untimed full-size tail: the first labelling of a slide-sized map pays for the allocator's first touch of the label /
    # table buffers (hipMalloc of several GB: 1.1 s against 0.34 s for every later slide of a run_infer_wsi.py session); like the W warm-up
    # stripes it computes everything again in the timed region -- nothing is cached but the memory blocks
    postprocess_bands_and_gather(run, H, W, rank, world, dist, margin=MARGINS, guard=48, canv=OrderedDict(struct), max_band_px=max_band_px, prof={})
    torch.cuda.synchronize()
    prof = {}
    phase = {}
    res = {}

    def job():
        from cerberus_amd.shard_postproc import band_view, make_incremental

        t0 = time.perf_counter()
        canv = OrderedDict(struct)
        # One GPU, several local labelling bands: a band is labelled on a side stream as soon as the stripes that cover its rows and its halo have
        # run (shard_postproc.IncrementalLocalLabeller; run_infer_wsi.py does the same on the canvases the inference writes).  The structured maps
        # the labelling reads here exist beforehand, so the dependency is imposed: band b waits for the completion event of the stripe that
        # finished its rows, and is only started after the NEXT stripe has been queued (the host then has a stripe of device work ahead of it).
        pre = make_incremental(band_view(run, H, W, canv), dist, margin=MARGINS, guard=48, max_band_px=max_band_px) if args.overlap_tail else {}
        evs = []
        for k in range(K):
            # (no join of the two handles' streams between the stripes unless the overlapped labelling needs a per-stripe completion event)
            run.infer_patches(slab, y0, cuts[k], cuts[k + 1], join=bool(pre))
            if pre:
                e = torch.cuda.Event()
                e.record()
                evs.append(e)
                if k >= 1:
                    for lab in pre.values():
                        lab.feed(run.rows_final(cuts[k]), [evs[k - 1]])
        run.join()
        torch.cuda.synchronize()
        phase["inference_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        inst, info, small = postprocess_bands_and_gather(run, H, W, rank, world, dist, margin=MARGINS, guard=48, canv=canv, max_band_px=max_band_px,
                                                         prof=prof, watch=args.watch, pre=pre)
        torch.cuda.synchronize()
        phase["tail_s"] = time.perf_counter() - t1
        res.update(inst=inst, info=info, small=small)

    if os.environ.get("CERB_BENCH_PROBE_REPEAT"):  # developer probe: is the first full-size tail slower than the second (allocator warm-up)?
        for _ in range(int(os.environ["CERB_BENCH_PROBE_REPEAT"])):
            dtp = _timed(job, dev, dist, args.backend)
            print("probe: job %.3f s inference %.3f tail %.3f nuclei %.3f" % (dtp, phase["inference_s"], phase["tail_s"], prof.get("label_Nuclei", {}).get("s", -1)),
                  file=sys.stderr, flush=True)
            prof.clear()
            res.clear()
    dt = _timed(job, dev, dist, args.backend)
    if dist is not None:  # slowest rank's phases
        for key in ("inference_s", "tail_s"):
            t = torch.tensor([phase[key]], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            phase[key] = float(t.item())
    info = res["info"]
    # proof that the tail consumed what THIS job's inference produced: checksums of the class maps the timed tail returned (taken here, outside the
    # timed region) against the canvases the timed inference wrote
    def _csum(v):
        return float(v.sum(dtype=torch.float64 if v.is_floating_point() else torch.int64).item())

    small_sum = {k: _csum(v) for k, v in res["small"].items()} if res.get("small") is not None else None
    if small_sum is not None and world == 1:  # one rank: the gathered class maps ARE the inference's canvases, cropped to the slide
        for k, v in small_sum.items():
            assert v == _csum(run.canv[k][:valid, :W]), "the tail's %s map is not the inference's canvas" % k
    n_inst = {t: int(i.get("n_total", 0)) for t, i in info.items()}
    checks = {t: {"n_truncated": int(i.get("n_truncated", 0)), "n_unresolved": int(i.get("n_unresolved", 0)), "local_bands": int(i.get("local_bands", 1)),
                  "bands_labelled_under_inference": int(i.get("bands_labelled_under_inference", 0))}
              for t, i in info.items()}
    # The reference's actual WSI output is the instance DICTIONARY (infer/wsi.py:805-853: get_inst_info_dict per tissue -> uuid keys ->
    # joblib.dump): contour tracing on the GPU (cerb_inst_contour_*), the per-instance dictionaries on the host, and the .dat file.  Timed here,
    # AFTER the headline region (the metric counts pixels inferred and labelled; `end_to_end_Mpx_s` = the same slide over job + dictionary + file).
    dat = None
    if rank == 0 and not args.no_dat:
        import tempfile

        from cerberus_amd.wsi import DatWriter, collect_wsi_inst_arrays, wsi_meta

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parts = collect_wsi_inst_arrays(res["inst"], res["small"], (H, W))  # GPU: instance tables + border following; arrays to the host
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # The ~1e6 per-instance dictionaries, the uuid keys and the pickle are built the way run_infer_wsi.py builds them: by a separate, torch-free
        # writer process (cerberus_amd.wsi.DatWriter.from_arrays) UNDERNEATH the next slide's inference -- here a second pass over the same K
        # stripes, untimed by `value`, so that the line shows what the overlap costs the inference (`inference_pass_under_writer_s` against
        # config.inference_s) and how long the parent still waits afterwards (`writer_wait_s`).
        os.environ["CERB_DAT_WRITER_TIMING"] = "1"
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, "slide.dat")
            wr = DatWriter.from_arrays(parts, wsi_meta((H, W), 0.5), pth)
            t2 = time.perf_counter()
            for k in range(K):
                run.infer_patches(slab, y0, cuts[k], cuts[k + 1])
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            wr.join()
            t4 = time.perf_counter()
            nbytes = os.path.getsize(pth)
            build_s, write_s = [float(v) for v in open(pth + ".time").read().split()] if os.path.exists(pth + ".time") else (0.0, t4 - t2)
        dat = {"tables_and_contours_s": round(t1 - t0, 3), "hand_over_s": round(t2 - t1, 3), "writer_build_s": round(build_s, 3), "writer_pickle_s": round(write_s, 3),
               "dat_s": round(t2 - t0 + build_s + write_s, 3), "dat_MB": round(nbytes / 1e6, 1),
               "entries": {p_[0]: int(((p_[1][:, 0] > 0) & (p_[2] >= 3)).sum()) for p_ in parts},
               "inference_pass_under_writer_s": round(t3 - t2, 3), "writer_wait_s": round(t4 - t3, 3),
               "note": "tables_and_contours_s = cerb_inst_table + cerb_inst_contour_* + copies to the host, hand_over_s = the arrays written for the writer "
                       "process (both in the parent, between two slides); writer_build_s / writer_pickle_s = reading them back / the protocol-4 pickle joblib.load "
                       "reads, assembled as byte matrices without a Python object per instance (inst_info.write_dat_fast), on the writer's own clock, "
                       "underneath the next slide; end_to_end_Mpx_s counts all four "
                       "serially, as a one-slide run pays them"}
        del parts
    # The same dictionary the way run_infer_wsi.py builds it on SEVERAL ranks (round 6): every rank computes the tables + contours of
    # the instances it owns on its halo + band + halo window and rank 0 receives the compact arrays and the quarter-resolution tissue map -- the
    # label bands and class canvases (15 B/px) stay on their ranks.  The whole tail again with that hand-over, untimed by `value` (which keeps the
    # map stitch north_star names): seconds of the per-rank tables + contours (slowest rank), bytes into the root either way.
    per_rank = None
    if dist is not None and not args.no_dat:
        parts2, prof2 = [], {}

        def arr_job():
            postprocess_bands_and_gather(run, H, W, rank, world, dist, margin=MARGINS, guard=48, canv=OrderedDict(struct), max_band_px=max_band_px, prof=prof2,
                                         watch=args.watch, parts=parts2, gather_maps=False)

        adt = _timed(arr_job, dev, dist, args.backend)
        t = torch.tensor([prof2.get("tables_and_contours", {}).get("s", 0.0)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            arr_bytes = prof2.get("parts_gather", {}).get("bytes", 0) + prof2.get("root_gather", {}).get("bytes", 0)
            map_bytes = prof.get("root_gather", {}).get("bytes", 0)
            per_rank = {"tail_with_array_hand_over_s": round(adt, 4), "tables_and_contours_s_slowest_rank": round(float(t.item()), 4),
                        # the whole job if the tail ended with what run_infer_wsi.py hands to rank 0 (arrays) instead of the map stitch `value` keeps timing
                        "whole_job_Mpx_s_with_array_hand_over": round(H * W / (phase["inference_s"] + adt) / 1e6, 3),
                        "bytes_into_rank0": {"instance_arrays_and_quarter_map": int(arr_bytes), "label_bands_and_class_maps": int(map_bytes),
                                             "ratio": round(map_bytes / max(1, arr_bytes), 1)},
                        "entries": {p_[0]: int(((p_[1][:, 0] > 0) & (p_[2] >= 3)).sum()) for p_ in parts2}}
            if dat is not None:
                dat["per_rank_arrays"] = per_rank
        del parts2
    res.clear()
    # The REFERENCE's nuclei scheme over the same maps (infer/wsi.py:81-268, 642-684: 4096-px tiles, 64-px margins, strips, cross sections; every
    # tile labelled with skimage's tie order), tiles sharded over the ranks (cerberus_amd/ref_tiling.py) -- what `run_infer_wsi.py
    # --reference_tiling` runs instead of the band scheme.  `value` uses the band scheme; this leg is timed beside it.
    rt = None
    if not args.no_ref_tiling:
        from cerberus_amd.ref_tiling import reference_tiled_nuclei_sharded

        tprof = {}
        tmap = run.canv["Nuclei-TYPE"][:valid] if "Nuclei-TYPE" in run.canv else None

        def ref_job():
            res["ref"] = reference_tiled_nuclei_sharded(struct["Nuclei-INST"], tmap, run.r0 * TILE, (H, W), rank, world, dist, tile_shape=4096, margin=64,
                                                        patch_output_shape=TILE, exact_ties=True, watch=args.watch, prof=tprof, as_part=True)

        rdt = _timed(ref_job, dev, dist, args.backend)
        if rank == 0:
            n_ref, n_band = int(len(res["ref"][1])), n_inst.get("Nuclei", 0)
            rt = {"ref_tiling_s": round(rdt, 3), "Mpx_s": round(H * W / rdt / 1e6, 1), "instances": n_ref, "band_scheme_instances": n_band,
                  "instances_lost_by_the_reference_scheme": round(1.0 - n_ref / max(1, n_band), 5),
                  "rank0": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tprof.items()},
                  "note": "nuclei instance tables + contours (the arrays the .dat writer takes: boxes, centroid sums, contour runs, type votes, tile origins) by "
                          "the reference's own tile sets, no Python object per instance; compare with dat.tables_and_contours_s + postproc.Nuclei.s of the "
                          "band scheme"}
        res.clear()
    # secondary: the inner loop alone (configs[1]) + the per-kernel table of one batch step
    bdt, bstep, bn = batch_loop(model, dev, rank, 20, 3, None, args.backend)
    roofline, rows = kernel_table(model, bstep, bn)
    # what the head kernels saw during the timed job (cerb_forward_io.logit_absmax, one row per batch) and the algorithm the headline ran on
    guard = run.logit_report()
    precision = dict(model.precision_decision(), name=CONV_ALGO_NAMES.get(model.precision_decision()["conv_algo"]),
                     logit_saturation=model.LOGIT_SATURATION,
                     logit_guard={"batches": guard["batches"], "batches_above": guard["above"], "largest_abs_logit": round(guard["max"], 3)})
    # The headline's precondition made explicit: the same slide's inference on the algorithms a model that trips the
    # calibration probe (conv_algo 1) or a caller's explicit choice (0) would run -- 4 of the K stripes each, same runner, same handles, same
    # streams, barrier + synchronize on both sides; `value_with_default_tail` = slide pixels / (that inference rate over the whole slide + the
    # default run's measured tail): an estimate built from two measurements, said so.
    algo_values = {}
    try:
        ks = min(4, K)
        for algo in (1, 0):
            for h in (model, run.twin):
                if h is not None:
                    h.set_conv_algo(algo)
            _warm_handles(run, slab, y0, cuts[0], min(cuts[0] + 2 * WSI_BATCH, cuts[-1]))  # warm-up (workspaces of this algorithm, on BOTH handles)
            torch.cuda.synchronize()

            def part():
                for k in range(ks):
                    run.infer_patches(slab, y0, cuts[k], cuts[k + 1], join=False)
                run.join()

            adt = _timed(part, dev, dist, args.backend)
            apx = (cuts[ks] - cuts[0]) * TILE * TILE * world  # (every rank runs the same number of its own patches, up to one batch)
            abdt, _, _ = batch_loop(model, dev, rank, 4, 1, None, args.backend)
            inf_s = px_all(H, W) / (apx / adt)
            algo_values[str(algo)] = {"name": CONV_ALGO_NAMES[algo], "stripes": ks, "inference_Mpx_s": round(apx / adt / 1e6, 3),
                                      "value_with_default_tail": round(px_all(H, W) / (inf_s + phase["tail_s"]) / 1e6, 3),
                                      "batch_step_ms": round(abdt / 4 * 1e3, 3)}
    except Exception as e:  # never fails the headline
        algo_values["error"] = str(e)[:300]
    finally:
        for h in (model, run.twin):
            if h is not None:
                h.set_conv_algo(6)
    if rank != 0:
        return
    px = H * W
    n_tiles = run.geo.rows * run.geo.cols
    flops_tile = model.flops(1, TILE, TILE)
    pp = {}
    for t in ("Nuclei", "Gland", "Lumen"):
        e = prof.get("label_" + t)
        if e:
            pp[t] = {"s": round(e["s"], 4), "n_inst": n_inst.get(t), "band_px": (valid * W) if t == "Nuclei" else (valid // 2) * (W // 2)}
            pp[t]["Gpx_s"] = round(pp[t]["band_px"] / e["s"] / 1e9, 3)
            pp[t]["hbm_frac_of_8TBs_at_12B_px"] = round(12.0 * pp[t]["band_px"] / e["s"] / 8e12, 5)
            pp[t].update(checks.get(t, {}))
            if pp[t].get("bands_labelled_under_inference"):  # `s` then only covers what was left for the tail (last band(s), id protocol, relabelling)
                pp[t]["s_covers"] = "the tail only: %d of %d bands were labelled on a side stream during the inference" % (pp[t]["bands_labelled_under_inference"], pp[t]["local_bands"])
                for key in ("Gpx_s", "hbm_frac_of_8TBs_at_12B_px"):
                    pp[t].pop(key, None)
    mg = dict(args.identity)
    if dist is not None:
        mg.update({"peak_GB_s_per_xgmi_link": XGMI_LINK_GBS, "rank": 0})
        for key in ("halo_exchange", "root_gather"):
            e = prof.get(key)
            if e:
                mg[key] = {"bytes_into_rank0": e["bytes"], "s": round(e["s"], 5), "GB_s": round(e["bytes"] / max(e["s"], 1e-9) / 1e9, 2)}
        if "root_gather" in mg:
            mg["root_gather"]["GB_s_per_link"] = round(mg["root_gather"]["GB_s"] / max(1, world - 1), 2)
    line = {
        "metric": "Mpx/sec WSI tiled inference (all heads)",
        "value": round(px / dt / 1e6, 3),
        "unit": "Mpx/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(dt / K * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "Synthetic %dx%dx3 uint8 WSI resident in HBM (BASELINE.json north_star / configs[%d]), sliding window 256-in / 256-out = %d tiles, "
                        "full Cerberus (ResNet34 + 6 heads, fp32) with on-device patch gather + canvas scatter, then on-GPU post-processing of "
                        "slide-sized structured maps (nuclei watershed at x1, gland / lumen at x0.5, instance tables, global ids) and the label "
                        "stitch on rank 0; a step = 1/%d of every rank's band, the tail is inside the timed region" % (H, W, 3 if side >= 40000 else 2, n_tiles, K),
            "slide": [H, W],
            "tiles": n_tiles,
            "batch_tiles": WSI_BATCH, "streams": args.streams, "overlap_tail": int(bool(args.overlap_tail and dist is None)),
            "conv_algo": precision["conv_algo"], "precision": precision, "other_conv_algos": algo_values,
            "inference_s": round(phase["inference_s"], 3),
            "inference_Mpx_s": round(px / phase["inference_s"] / 1e6, 3),
            "postproc_and_stitch_s": round(phase["tail_s"], 3),
            "whole_job_s": round(dt, 3),
            "tail_inputs": {"INST probability maps": ("the canvases this job's timed inference wrote (--tail-from-inference: the seeded weights with the INST heads' "
                                                      "background bias raised so that ~3 %% of the pixels are foreground; shifts %s)" % fg_shifts) if args.tail_from_inference else
                                                     "seeded structured maps of the slide's size (random weights make no instances; cerberus_amd/synth_maps.py)",
                            "TYPE and Patch-Class maps": "the canvases this job's inference wrote (majority types of the instance tables, class-map gather)",
                            "class_canvas_checksum": small_sum},
            "nuclei_scheme": "exact band ownership (every instance of the whole-slide labelling once; cerberus_amd/shard_postproc.py) -- the reference's own "
                             "4096-px tile sets + 64-px margins (`run_infer_wsi.py --reference_tiling`) are timed beside it as `ref_tiling`",
            "inference_algorithmic_tflops_per_gpu": round(n_tiles * flops_tile / phase["inference_s"] / 1e12 / world, 2),
            "gflop_per_tile": round(flops_tile / 1e9, 3),
            "parallelism": "band-sharded x%d (contiguous patch rows), no collective during inference; halo send/recv + 2 all-gathers + 1 gather per "
                           "label map in the tail (%s)" % (world, args.backend if world > 1 else "single GPU"),
        },
        "roofline": roofline,
        "kernels": rows,
        "postproc": pp,
        "dat": dat if dat is not None else ({"per_rank_arrays": per_rank} if per_rank else None),
        "ref_tiling": rt,
        "end_to_end_Mpx_s": round(px / (dt + dat["dat_s"]) / 1e6, 3) if dat else None,
        "multi_gpu": mg,
        "batch_step": {"workload": "configs[1]: batch=32 256x256 tiles, inner loop only", "ms_per_step": round(bdt / 20 * 1e3, 3),
                       "Mpx_s": round(20 * BATCH * TILE * TILE / bdt / 1e6, 2)},
    }
    if world == 1 and not args.no_train_leg:
        # BASELINE.json configs[4] rides along in the default line (3 steps, batch 16 x 448 x 448 on this GPU; `--mode train` is the full leg):
        # the slide job's buffers are released first, the training handle is a second, train-packed copy of the same weights
        try:
            del run, slab, struct
            torch.cuda.empty_cache()
            from cerberus_amd.net_desc import create_model

            tm = create_model(**kw)
            tm.load_state_dict(sd, strict=True)
            tdt, tres, troof, trows = train_measure(tm, dev, None, 1, 0, 3, 1, args.backend)
            line["train_step"] = {"workload": "configs[4] on one GPU: whole multi-task training step (train-mode forward, 6 losses, backward, Adam, BN running "
                                              "statistics, device re-pack), batch %d x %dx%dx3 uint8, fp32; 1 warm-up + 3 timed steps" % (TRAIN_BATCH, TRAIN_TILE, TRAIN_TILE),
                                  "ms_per_step": round(tdt / 3 * 1e3, 3), "tiles_s": round(3 * TRAIN_BATCH / tdt, 3),
                                  "last_overall_loss": round(float(tres["EMA"]["overall_loss"]), 4), "roofline": troof, "kernels": (trows or [])[:8]}
            del tm
        except Exception as e:  # never fails the headline
            line["train_step"] = {"error": str(e)[:300]}
    if world == 1 and not args.no_ingest_leg:
        # a slide on DISK (JPEG-tiled pyramidal TIFF) through the reader / decode pool / upload-ahead path, beside the resident figure (`--mode ingest`: 20000^2)
        try:
            line["ingest"] = ingest_leg(model, dev, 12288, WSI_BATCH, 1)
        except Exception as e:  # never fails the headline
            line["ingest"] = {"error": str(e)[:300]}
        # the same for a 40x scan: 24576^2 stored pixels at 0.25 mpp read at 0.5 mpp (decode processes + the x2 reduction on the device)
        try:
            ing40 = ingest_leg(model, dev, 12288, WSI_BATCH, 1, sweep=(1, 8), base_mpp=0.25)
            line["ingest_40x"] = {k: ing40[k] for k in ("slide", "stored", "file", "decode", "inference_resident", "end_to_end_from_file", "best")}
        except Exception as e:  # never fails the headline
            line["ingest_40x"] = {"error": str(e)[:300]}
        # ... and for a generic tiled TIFF with lossless (deflate) tiles: one native call per window (libcerberus_host.so: pread + zlib + placement on pthreads)
        try:
            ingd = ingest_leg(model, dev, 8192, WSI_BATCH, 1, sweep=(1, 4, 16, 64), codec="deflate")
            line["ingest_deflate"] = {k: ingd[k] for k in ("slide", "file", "decode", "inference_resident", "end_to_end_from_file", "best")}
        except Exception as e:  # never fails the headline
            line["ingest_deflate"] = {"error": str(e)[:300]}
    if world == 1:  # per-head Dice against the reference's own outputs (the metric's second half), outside the timed region
        try:
            line["dice_vs_reference"] = dice_vs_reference()
        except Exception as e:  # never fails the headline
            line["dice_vs_reference"] = {"error": str(e)[:300]}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd, kw)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dat", action="store_true", help="skip the untimed-by-`value` instance-dictionary leg (`dat`: contours + dictionary + .dat file)")
    ap.add_argument("--no-ref-tiling", action="store_true", help="skip the `ref_tiling` leg (the reference's own nuclei tile scheme over the same maps)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the short configs[4] training leg that the default (wsi, 1 GPU) line carries as `train_step`")
    ap.add_argument("--no-ingest-leg", action="store_true", help="skip the `ingest` leg of the default line (a 12288^2 JPEG-tiled TIFF through reader -> decode pool -> upload-ahead -> inference)")
    ap.add_argument("--mode", default="wsi", choices=["wsi", "batch", "infer", "train", "ingest"],
                    help='"wsi" (default): the headline, whole-slide job of north_star / configs[2-3]; "batch" (= "infer"): configs[1] inner loop; '
                         '"train": the multi-task training step of configs[4]')
    ap.add_argument("--tail-from-inference", action="store_true",
                    help="slide job: the labelling reads the INST canvases the timed inference wrote instead of the seeded structured maps (the data dependency "
                         "inference -> labelling at slide scale); the seeded weights get a sparse-foreground bias calibration so that those maps hold instances")
    ap.add_argument("--ingest-base-mpp", type=float, default=0.5, help="--mode ingest: microns per pixel the file is STORED at (0.25 = a 40x scan: (2 x slide)^2 pixels on disk, read at 0.5 mpp)")
    ap.add_argument("--ingest-codec", default="jpeg", choices=["jpeg", "deflate", "lzw", "jp2k"],
                    help="--mode ingest: the tiles' compression (jpeg: Aperio-style, decoded by libjpeg on threads / worker processes; deflate, lzw (+ horizontal predictor): "
                         "generic tiled TIFFs, decoded by libcerberus_host.so -- one native call per window on CERB_DECODE_THREADS pthreads)")
    ap.add_argument("--slide", type=int, default=0, help="slide side in pixels (default: 40000, or 20000 when HBM is short)")
    ap.add_argument("--max-band-mpx", type=float, default=220.0, help="largest labelling call on one GPU, in Mpx (96 B/px of workspace)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and take the collective code path even at world size 1 (RCCL accepts one rank): "
                         "the nccl test of tests/test_cli_gpu.py")
    ap.add_argument("--planar", type=int, default=1, help="last decoder level in the tile-planar layout (1, default: conv_wino4p.hip) or NHWC (0: conv_wino4.hip, round 2's path) -- A/B")
    ap.add_argument("--overlap-tail", type=int, default=0, choices=[0, 1], help="one-GPU slide job: 1 = the nuclei bands are labelled on a side stream as soon as their "
                                                                                       "rows are inferred (only the last band and the id protocol remain in the tail), 0 (default) = everything after the "
                                                                                       "inference.  Measured: the tail shrinks by 0.18 s and the inference grows by 0.25 s (149.6 against 150.6 Mpx/s) -- the "
                                                                                       "flood kernels' LDS keeps persistent convolution workgroups off their CUs; it pays when the inference waits for a "
                                                                                       "slide's decode, not on resident data")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2], help="slide job: 2 (default) = batches alternate between two handles on two streams "
                                                                               "(NetDesc.twin: the ramps / tails / sub-chip launches of one batch overlap the other's), 1 = one handle")
    ap.add_argument("--backend", default="nccl", help='torch.distributed backend ("nccl" = RCCL over xGMI; "gloo" only for plumbing tests)')
    ap.add_argument("--oversubscribe", action="store_true",
                    help="with --backend gloo only: let the N self-spawned ranks time-share fewer than N devices (plumbing tests on a one-GPU box); "
                         "RCCL needs one device per rank and never oversubscribes")
    args = ap.parse_args()

    # `--gpus N` without a launcher: re-execute N ranks (one per device, rendezvous on 127.0.0.1) -- or exit non-zero when N devices are not
    # there.  Under torch.distributed.run WORLD_SIZE must agree with --gpus.  A line with n_gpus < --gpus is never printed.
    from cerberus_amd import launch

    launch.ensure_world(args.gpus, args.backend, oversubscribe=args.oversubscribe)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:  # (ensure_world has already refused; kept as the last line of defence)
        sys.exit("error: --gpus %d but world size %d" % (args.gpus, world))
    dist = None
    args.watch = launch.null_watch()
    if world > 1 or args.force_dist:
        n_dev = max(1, torch.cuda.device_count())
        if args.backend == "nccl" and world > n_dev:
            sys.exit("error: %d ranks but %d visible GPU(s): RCCL needs one device per rank" % (world, n_dev))
        local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)
        dist = launch.init_dist(args.backend, local_rank)
        args.watch = launch.PhaseWatch(rank)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    with args.watch.phase("rank identity all-gather (first collective on the communicator)"):
        args.identity = launch.rank_identity(dist, dev, args.backend)
    if dist is not None and args.identity["world"] != world:
        sys.exit("error: the communicator spans %d ranks, the launcher said %d" % (args.identity["world"], world))

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    kw = default_model_kwargs()
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
    model = create_model(**kw)
    model.load_state_dict(sd, strict=True)
    if args.planar != 1:
        model.set_planar(args.planar)
    if args.mode != "train":
        model.prepare(dev)  # the precision decision at load time, outside every timed region (CERB_AUTO_PRECISION=0: no calibration launch, for profiling runs)
    if args.mode == "train":
        return train_leg(args, model, dev, dist, world, rank, sd, kw)
    if args.mode == "ingest":
        side = args.slide if args.slide > 0 else 20000
        ing = ingest_leg(model, dev, side, 64 if args.streams == 2 else 96, args.streams, base_mpp=args.ingest_base_mpp, codec=args.ingest_codec)
        codec_name = {"jpeg": "JPEG", "deflate": "deflate", "lzw": "LZW", "jp2k": "JPEG 2000"}[args.ingest_codec]
        workload = ("%dx%d slide from a %s-tiled pyramidal TIFF stored at %.4g mpp (%dx%d pixels) -> reader (%s%s) -> pinned chunks ahead of the inference -> full "
                    "Cerberus forward into device canvases; inference only, no labelling tail" % (
                        side, side, codec_name, args.ingest_base_mpp, ing["stored"]["pixels"][0], ing["stored"]["pixels"][1],
                        "tile decode on threads / worker processes" if args.ingest_codec in ("jpeg", "jp2k") else "one native call per window: libcerberus_host.so, pthreads",
                        ", reduced to 0.5 mpp on the device" if args.ingest_base_mpp < 0.5 else ""))
        print(json.dumps({"metric": "Mpx/sec WSI tiled inference (all heads) from a %s-tiled pyramidal TIFF on disk" % codec_name, "value": ing["best"]["Mpx_s"], "unit": "Mpx/s",
                          "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": round(side * side / ing["best"]["Mpx_s"] / 1e3, 1), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": workload, "streams": args.streams, "conv_algo": model.precision_decision()["conv_algo"], "tile_codec": args.ingest_codec},
                          "ingest": ing}), flush=True)
        return
    if args.mode == "wsi":
        wsi_leg(args, model, dev, dist, world, rank, sd, kw)
    else:
        dt, step, n = batch_loop(model, dev, rank, args.steps, args.warmup, dist, args.backend)
        roofline, rows = kernel_table(model, step, n)
        if rank == 0:
            flops_step = model.flops(BATCH, TILE, TILE)
            line = {
                "metric": "Mpx/sec WSI tiled inference (all heads)", "value": round(world * args.steps * BATCH * TILE * TILE / dt / 1e6, 3), "unit": "Mpx/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "Full Cerberus (ResNet34 encoder + 6 decoder heads) batch=32 256x256x3 uint8 tiles, fp32, all heads + fused "
                                       "softmax/crop/argmax scattered into a device-resident canvas (BASELINE.json configs[1]; inner loop of the slide job)",
                           "batch_tiles": BATCH, "tile": TILE, "gflop_per_tile": round(flops_step / BATCH / 1e9, 3),
                           "conv_algo": model.precision_decision()["conv_algo"], "precision": model.precision_decision(),
                           "whole_step_tflops": round(flops_step / (dt / args.steps) / 1e12 * world, 2),
                           "parallelism": "tile-sharded x%d, no data-path collective" % world},
                "roofline": roofline, "kernels": rows, "multi_gpu": dict(args.identity),
            }
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(sd, kw)
            print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
