"""bench.py -- headline benchmark of the Cerberus tiled-inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of 32 synthetic 256x256x3 uint8 tiles (BASELINE.json
configs[1]: full Cerberus, ResNet34 + 6 heads, fp32): forward of all heads, fused softmax / crop / argmax, outputs
scattered straight into this rank's device-resident canvas.  Inputs are resident in HBM before the timed region.
Multi-GPU: tiles shard embarrassingly -> one process per GPU, every rank runs its own batches, no
data-path collective; weak scaling.  Rank 0 prints ONE JSON line.
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
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def cpu_baseline(sd, kw, n_tiles=8, iters=2):
    """Oracle (CPU restatement of the reference path, PyTorch-CPU fp32) on the host cores -- reported, not optimised."""
    from oracle import net_ref

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 16)  # oneDNN with one thread per logical CPU of a 256-thread host is pathologically slow (10 s/tile)
    torch.set_num_threads(cores)
    tiles = np.random.RandomState(1).randint(0, 256, (n_tiles, TILE, TILE, 3)).astype(np.uint8)
    t0 = time.perf_counter()
    net_ref.infer_step(sd, tiles[:1], TILE, kw["considered_tasks"], kw["decoder_kwargs"])  # warm-up, also sizes the sample
    t1 = time.perf_counter() - t0
    if t1 * n_tiles * iters > 30.0:  # keep the CPU leg to ~10-30 s
        iters = 1
        n_tiles = max(1, min(n_tiles, int(20.0 / t1)))
        tiles = tiles[:n_tiles]
    t0 = time.perf_counter()
    for _ in range(iters):
        net_ref.infer_step(sd, tiles, TILE, kw["considered_tasks"], kw["decoder_kwargs"])
    dt = time.perf_counter() - t0
    # post-processing oracle (C restatement of loader/postproc.py + skimage/scipy, one core) on a 2048^2 structured map
    from oracle import postproc_ref, synth

    pm = synth.nuclei_maps(2048, 2048, 7, 600.0, noise=0.02)
    t0 = time.perf_counter()
    postproc_ref.proc(pm, "Nuclei")
    pp_dt = time.perf_counter() - t0
    return {
        "postproc_nuclei_Mpx_s_1core": round(2048 * 2048 / pp_dt / 1e6, 2),
        "value": round(iters * n_tiles * TILE * TILE / dt / 1e6, 4),
        "unit": "Mpx/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d x %d tiles of %dx%d, all six heads, forward + infer_step wrapper, torch-CPU fp32, %d threads (host has %d)"
        % (iters, n_tiles, TILE, TILE, cores, avail),
    }


TRAIN_BATCH, TRAIN_TILE = 16, 448  # BASELINE.json configs[4]: batch 16; 448 x 448 is the reference's training patch (paramset.yml)


def train_leg(args, model, dev, dist, world, rank):
    """--mode train: K whole training steps (train-mode forward, six losses, backward, bucketed gradient all-reduce over the ranks, Adam,
    BatchNorm running statistics, on-device weight re-pack) on a synthetic batch resident in HBM; every rank has its own batch (weak)."""
    import numpy as np

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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
            "roofline": None,
            "cpu_baseline": None,
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help='"infer" (default): the headline, BASELINE.json configs[1]; "train": the multi-task training step of configs[4]')
    ap.add_argument("--backend", default="nccl", help='torch.distributed backend ("nccl" = RCCL over xGMI; "gloo" only for plumbing tests)')
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    kw = default_model_kwargs()
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
    model = create_model(**kw)
    model.load_state_dict(sd, strict=True)
    if args.mode == "train":
        return train_leg(args, model, dev, dist, world, rank)

    # synthetic slide strip resident in HBM: this rank's batches (seeded per rank), and its output canvas
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    tiles = torch.randint(0, 256, (BATCH, TILE, TILE, 3), dtype=torch.uint8, device=dev, generator=g)
    gy, gx = 4, 8  # canvas of 4 x 8 tiles
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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline leg: per-launch HIP-event timing of ONE more step (outside the timed region) -------------
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
    dom = max(fam.items(), key=lambda kv: kv[1][1])
    dom_name, (dom_fl, dom_ms, dom_cnt) = dom
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12
    total_ms = sum(r[3] for r in recs)
    # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of this same command; gfx950 read-side x2 correction applied by scripts/rocprof_summary.py)
    traffic = None
    sym = {"conv_wino<f2x2,8x16>": "void conv_wino_kernel<false>(ConvParams)",
           "conv_wino<f2x2,8x16,res>": "void conv_wino_kernel<true>(ConvParams)",
           "conv_igemm<ks3,s1,mode0,8x32>": "void conv_igemm_kernel<3, 1, 8, 32, 32, 0>(ConvParams)",
           "conv_igemm<ks3,s1,mode1,8x32>": "void conv_igemm_kernel<3, 1, 8, 32, 32, 1>(ConvParams)"}.get(dom_name)
    pmc_path = os.path.join(ROOT, "profiles", "r01_bench_pmc_hbm.json")
    if sym and os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get(sym, {}).get("hbm_bytes_per_launch")
    # `achieved` is ALGORITHMIC work (direct-convolution FLOPs, SURVEY par.8d) per second.  The Winograd F(2x2,3x3) kernel
    # executes 16 multiplies per 2x2 output patch and input channel instead of 36, so its matrix pipe runs 4/9 of those FLOPs:
    # `frac` can exceed 1; `executed_frac` is the share of the fp32 MFMA roof the kernel's own instructions fill.
    exec_ratio = 16.0 / 36.0 if dom_name.startswith("conv_wino") else 1.0
    roofline = {
        "bound": "mfma",
        "kernel": dom_name,
        "achieved": round(achieved, 2),
        "peak": PEAK_F32_MFMA_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
        "executed_mfma_tflops": round(achieved * exec_ratio, 2),
        "executed_frac": round(achieved * exec_ratio / PEAK_F32_MFMA_TFLOPS, 4),
        "traffic": traffic,
        "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_bench_pmc_hbm.json)",
        "flop_per_launch": round(dom_fl / dom_cnt / 1e9, 3),
        "launches": dom_cnt,
        "avg_launch_ms": round(dom_ms / dom_cnt, 4),
        "kernel_share_of_step": round(dom_ms / total_ms, 4),
    }

    # ---- post-processing leg (reported beside the headline; P1-P5 of SURVEY.md par.8a): on-GPU label maps from a seeded
    # structured probability map (600 nuclei / Mpx, radius 4-9 px; glands 14-60 px) and from this step's own canvas -------
    postproc = None
    if rank == 0:
        from cerberus_amd.postproc import postproc_device
        from cerberus_amd import synth_maps as synth  # structured synthetic INPUT generator (numpy)

        PH = 2048
        maps = {
            "Nuclei": torch.from_numpy(synth.nuclei_maps(PH, PH, 7, 600.0, noise=0.02)).to(dev),
            "Gland": torch.from_numpy(synth.blob_maps(PH, PH, 9, 120, 14.0, 60.0, rim=4.0, sharp=1.0, noise=0.02, holes=0.3)).to(dev),
        }
        maps["Lumen"] = maps["Gland"]
        postproc = {"map": "%dx%d structured synthetic" % (PH, PH), "unit": "Mpx/s", "algorithmic_bytes_per_px": 12}
        for t in ("Nuclei", "Gland", "Lumen"):
            postproc_device(maps[t], t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                lab, info = postproc_device(maps[t], t)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            postproc[t] = {"ms": round(ms, 3), "Mpx_s": round(PH * PH / ms / 1e3, 1), "n_inst": int(info["n_inst"].item()),
                           "hbm_frac_of_8TBs": round(12.0 * PH * PH / (ms * 1e-3) / 8e12, 5)}
        # the three label maps of this step's own 1024 x 2048 canvas (random-weight network => degenerate maps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ("Nuclei", "Gland", "Lumen"):
            postproc_device(canvas[t], t)
        torch.cuda.synchronize()
        postproc["own_canvas_all_tissues_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        postproc["own_canvas_note"] = "random-weight network => degenerate maps (canvas-sized components): the flood's serial worst case"

    if rank == 0:
        px = world * args.steps * BATCH * TILE * TILE
        flops_step = model.flops(BATCH, TILE, TILE)
        line = {
            "metric": "Mpx/sec WSI tiled inference (all heads)",
            "value": round(px / dt / 1e6, 3),
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "Full Cerberus (ResNet34 encoder + 6 decoder heads) batch=32 256x256x3 uint8 tiles, fp32, all heads + fused "
                            "softmax/crop/argmax scattered into a device-resident canvas (BASELINE.json configs[1])",
                "batch_tiles": BATCH,
                "tile": TILE,
                "gflop_per_tile": round(flops_step / BATCH / 1e9, 3),
                "whole_step_tflops": round(flops_step / (dt / args.steps) / 1e12 * world, 2),
                "parallelism": "tile-sharded x%d, no data-path collective" % world,
            },
            "roofline": roofline,
            "postproc": postproc,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, kw)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
