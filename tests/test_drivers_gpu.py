"""Tile / WSI drivers on the GPU: stitched maps vs the CPU oracle, tile-mode vs WSI-mode self-consistency, sharded vs
unsharded equivalence (SURVEY.md par.4 'multi-GPU' row), and label maps bit-exact given identical probability maps."""
import os

import numpy as np
import pytest
import torch

from cerberus_amd.tile import InferManager, _prepare_patching
from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs, make_state_dict
from cerberus_amd.wsi import WSIRunner, synth_slide
from oracle import net_ref
from oracle import postproc_ref as pr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def manager():
    kw = default_model_kwargs()
    return InferManager(checkpoint_path=None, decoder_dict=dict(DEFAULT_REQ_TARGET_CODE), model_args=kw)


def _oracle_stitch(img, win, out, kw, sd):
    padded, info, pos = _prepare_patching(img, win, out, 0)
    uniq = info[: info.shape[0] // 2]
    hw = np.max(info[:, 1, 1], axis=0)
    canv = {}
    tiles = np.stack([padded[i[0, 0, 0]:i[0, 1, 0], i[0, 0, 1]:i[0, 1, 1]] for i in uniq])
    outs = net_ref.infer_step(sd, tiles, out, kw["considered_tasks"], kw["decoder_kwargs"])
    for o, i in zip(outs, uniq):
        for k, v in o.items():
            c = canv.setdefault(k, np.zeros((hw[0], hw[1]) + v.shape[2:], v.dtype))
            c[i[1, 0, 0]:i[1, 1, 0], i[1, 0, 1]:i[1, 1, 1]] = v
    y0, x0 = pos
    return {k: v[y0:y0 + img.shape[0], x0:x0 + img.shape[1]] for k, v in canv.items()}


@pytest.mark.parametrize("win,out,hw", [(256, 256, (300, 421)), (448, 144, (200, 310))])
def test_tile_manager_stitched_maps_vs_oracle(manager, win, out, hw):
    kw = manager.model_args
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}
    img = np.random.RandomState(77).randint(0, 256, hw + (3,)).astype(np.uint8)
    res = manager.infer_image(img, win, out, batch_size=5)
    ref = _oracle_stitch(img, win, out, kw, sd)
    for k, r in ref.items():
        a = res["raw"][k].cpu().numpy()
        assert a.shape == r.shape, k
        if r.dtype == np.float32:
            assert np.abs(a - r).max() < 1e-4, k
        else:
            assert (a != r).mean() < 1e-4, k
    # label maps: bit-exact given identical probability maps (feed the GPU's own maps to the C oracle)
    for t in ("Nuclei", "Gland", "Lumen"):
        m = res["raw"][t + "-INST"].cpu().numpy()
        exp = pr.proc(np.ascontiguousarray(m), t).astype(np.int32)
        if t == "Lumen":
            g = pr.proc(np.ascontiguousarray(res["raw"]["Gland-INST"].cpu().numpy()), "Gland")
            exp = exp * (g > 0)
        amb = int(res["info"][t]["n_ambiguous"].item())
        if amb == 0:
            assert np.array_equal(res["inst"][t].cpu().numpy(), exp), t


@pytest.mark.parametrize("win,out", [(256, 256), (448, 144)])
def test_wsi_runner_equals_tile_manager_and_sharding(manager, win, out):
    H, W = 600, 700
    slide = synth_slide(H, W, seed=5)
    img = slide.cpu().numpy()
    assert img.std() > 50  # the counter-based generator is not degenerate
    a = synth_slide(100, W, y0=250, seed=5)
    assert torch.equal(a, slide[250:350])  # value depends only on absolute coordinates
    res = manager.infer_image(img, win, out, batch_size=7)
    one = WSIRunner(manager.net, (H, W), win, out, batch_size=6)
    one.infer_band(slide, 0)
    full = one.gather_to_root()
    for k, v in full.items():
        assert torch.equal(v, res["raw"][k]), k  # same kernels, same patches -> bitwise identical
    # 3-way sharding: every rank sees only its band + halo rows of the slide
    parts = {}
    for r in range(3):
        run = WSIRunner(manager.net, (H, W), win, out, batch_size=4, rank=r, world_size=3)
        y0, y1 = run.slab_rows()
        run.infer_band(slide[y0:y1].contiguous(), y0)
        for k, v in run.canv.items():
            parts.setdefault(k, []).append(v.clone())
    for k, v in full.items():
        assert torch.equal(torch.cat(parts[k], 0)[:H, :W], v), k
    inst, info = WSIRunner.postprocess(full, wsi_mode=True)
    assert inst["Nuclei"].shape == (H, W) and inst["Gland"].shape == (H // 2, W // 2)
    m = full["Gland-INST"].cpu().numpy()
    ds = (m[0::2, 0::2] * 0.5 + m[0::2, 1::2] * 0.5) * 0.5 + (m[1::2, 0::2] * 0.5 + m[1::2, 1::2] * 0.5) * 0.5
    assert np.array_equal(inst["Gland"].cpu().numpy(), pr.proc(np.ascontiguousarray(ds.astype(np.float32)), "Gland", 0.5).astype(np.int32))


def test_wsi_runner_two_handles_on_two_streams_is_bitwise_the_one_handle_run(manager):
    """WSIRunner(twin=NetDesc.twin()): batches alternate between two handles on two side streams (bench.py --streams 2, run_infer_wsi.py).  Same
    kernels, same patches, disjoint canvas tiles: every canvas is bit-identical to the one-handle run, with work queued on the caller's stream
    before (the slab) and after (a reduction over the canvases) ordered against the side streams; the twin carries the parent's switches."""
    H, W = 1100, 1300
    one = WSIRunner(manager.net, (H, W), 256, 256, batch_size=3)
    slide = synth_slide(H, W, seed=11)
    one.infer_band(slide, 0)
    want = {k: v.clone() for k, v in one.canv.items()}
    manager.net.set_planar(1)
    twin = manager.net.twin()
    assert twin is not manager.net and twin._switches["set_planar"] == 1 and twin.handle_value() != manager.net.handle_value()
    two = WSIRunner(manager.net, (H, W), 256, 256, batch_size=3, twin=twin)
    for rep in range(3):  # repeated: stream-ordering mistakes are timing dependent
        for v in two.canv.values():
            v.zero_()
        slide2 = synth_slide(H, W, seed=11)  # produced on the caller's stream right before the fork
        if rep < 2:
            n = two.infer_band(slide2, 0)
        else:  # the job cut into short calls that leave the side streams unjoined (bench.py's stripes), one explicit join at the end
            n = sum(two.infer_patches(slide2, 0, a, b, join=False) for a, b in ((0, 4), (4, 5), (5, 17), (17, 30)))
            two.join()
        sums = {k: v.double().sum().item() for k, v in two.canv.items()}  # consumed on the caller's stream right after the join
        assert n == two.n_patches == 5 * 6
        for k, v in two.canv.items():
            assert torch.equal(v, want[k]), (rep, k)
            assert sums[k] == want[k].double().sum().item(), (rep, k)


def test_nuclei_bands_labelled_under_the_inference_equal_the_tail_only_run(manager):
    """run_infer_wsi.py / bench.py on one GPU: WSIRunner.infer_band(progress=...) feeds shard_postproc.IncrementalLocalLabeller, which labels a local
    nuclei band on a side stream as soon as its rows and the halo below are final -- while both inference streams go on writing the rows further
    down.  The tail then finishes the remaining band(s) and the id protocol.  Label maps, counts and checks must equal the run that labels
    everything after the inference, bit for bit (repeated: an ordering mistake between the three streams is timing dependent)."""
    from cerberus_amd.shard_postproc import band_view, make_incremental, postprocess_bands_and_gather

    H, W, margin = 1500, 1300, 64
    max_px = (300 + 2 * margin) * W  # five local bands
    slide = synth_slide(H, W, seed=21)
    twin = manager.net.twin()
    ref_run = WSIRunner(manager.net, (H, W), 256, 256, batch_size=4)
    ref_run.infer_band(slide, 0)
    want, want_info, _ = postprocess_bands_and_gather(ref_run, H, W, 0, 1, None, margin=margin, guard=16, max_band_px=max_px)
    assert want_info["Nuclei"]["local_bands"] == 5
    for rep in range(3):
        run = WSIRunner(manager.net, (H, W), 256, 256, batch_size=4, twin=twin if rep else None)
        pre = make_incremental(band_view(run, H, W), None, margin=margin, guard=16, max_band_px=max_px)
        assert list(pre) == ["Nuclei"] and pre["Nuclei"].nb == 5
        seen = []

        def progress(n_done, events):
            pre["Nuclei"].feed(run.rows_final(n_done), events)
            seen.append(pre["Nuclei"].done)

        run.infer_band(slide, 0, progress=progress)
        got, info, _ = postprocess_bands_and_gather(run, H, W, 0, 1, None, margin=margin, guard=16, max_band_px=max_px, pre=pre)
        assert seen[-1] >= 3 and seen == sorted(seen) and info["Nuclei"]["bands_labelled_under_inference"] == seen[-1]
        for t in want:
            assert torch.equal(got[t], want[t]), (rep, t)
        for key in ("n_total", "n_truncated", "n_unresolved", "local_bands"):
            assert info["Nuclei"][key] == want_info["Nuclei"][key], (rep, key)


def test_band_postprocess_and_gather_single_rank_equals_root_path(manager):
    """postprocess_bands_and_gather (the multi-GPU tail of run_infer_wsi.py) at world 1: same instances as the root-side
    WSIRunner.postprocess on a slide whose size is not a multiple of the patch (canvas rows / columns beyond the slide are cropped)."""
    from cerberus_amd.shard_postproc import postprocess_bands_and_gather, same_partition

    H, W = 650, 730
    slide = synth_slide(H, W, seed=9)
    run = WSIRunner(manager.net, (H, W), 256, 256, batch_size=6)
    run.infer_band(slide, 0)
    inst_a, info_a = WSIRunner.postprocess(run.gather_to_root(), wsi_mode=True)
    inst_b, info_b, small = postprocess_bands_and_gather(run, H, W, 0, 1, None)
    assert set(small.keys()) == {"Nuclei-TYPE", "Gland-TYPE", "Patch-Class"} and small["Nuclei-TYPE"].shape == (H, W)
    for t in ("Nuclei", "Gland", "Lumen"):
        a, b = inst_a[t].cpu().numpy(), inst_b[t].cpu().numpy()
        assert a.shape == b.shape, t
        assert same_partition(a, b), t
        assert info_b[t]["n_truncated"] == 0 and info_b[t]["n_unresolved"] == 0


def test_pipelined_band_upload_equals_resident_slab(manager):
    """SlabUploader: a host-resident band uploaded in chunks on a copy stream underneath the inference gives the same canvases as the
    band uploaded up front (many small chunks here; 448 -> 144 so that windows reach well below their output rows)."""
    from cerberus_amd.wsi import SlabUploader

    H, W = 1300, 1000
    host = np.random.RandomState(4).randint(0, 256, (H, W, 3)).astype(np.uint8)
    for win, out, batch in ((448, 144, 5), (256, 256, 7)):
        a = WSIRunner(manager.net, (H, W), win, out, batch_size=batch)
        a.infer_band(torch.from_numpy(host).cuda(), 0)
        ref = {k: v.clone() for k, v in a.canv.items()}
        b = WSIRunner(manager.net, (H, W), win, out, batch_size=batch)
        up = SlabUploader(host, 0, H, chunk_bytes=90 * W * 3)
        assert up.chunk == 90
        b.infer_band(up.slab, 0, ready=up.upload_until)
        torch.cuda.synchronize()
        assert up.next_row == H
        for k in ref:
            assert torch.equal(ref[k], b.canv[k]), (win, k)
    # a band in the middle of the slide (rank 1 of 3): rows are relative to the slab
    r = WSIRunner(manager.net, (H, W), 256, 256, batch_size=4, rank=1, world_size=3)
    y0, y1 = r.slab_rows()
    r.infer_band(torch.from_numpy(host[y0:y1]).cuda(), y0)
    ref = {k: v.clone() for k, v in r.canv.items()}
    r2 = WSIRunner(manager.net, (H, W), 256, 256, batch_size=4, rank=1, world_size=3)
    up = SlabUploader(host, y0, y1, chunk_bytes=64 * W * 3)
    r2.infer_band(up.slab, y0, ready=up.upload_until)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(ref[k], r2.canv[k]), k


@pytest.mark.parametrize("base_mpp, levels", [(0.25, (1,)), (0.2431, (1,)), (0.5 / 3, (1,)), (0.125, (1, 4)), (0.2528, (1, 4))])
def test_slab_of_a_slide_stored_finer_than_processed_is_reduced_on_the_device(tmp_path, base_mpp, levels):
    """A 40x scan (0.25 mpp -- or the 0.2431 / 0.2528 real scanners write -- with pyramid levels x1, x4, ..) read at the 0.5 mpp the network runs on:
    SlabUploader uploads the stored level's decoded rows and the device reduces them (cerb_resample_box: integer factors; cerb_resample_area: the
    reader's area means through its own float32 tables).  The slab must hold the bytes reader.read_bounds returns -- whole slide, a band in the
    middle, ragged last tiles, the slide's end inside the last output row / column -- and the host path (CERB_DEVICE_RESAMPLE=0) the same."""
    from cerberus_amd import reader as rd
    from cerberus_amd.wsi import SlabUploader

    rs = np.random.RandomState(int(base_mpp * 1e4))
    H, W = 1493, 2101
    base = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    base[200:500, 300:900] = np.linspace(0, 255, 600).astype(np.uint8)[None, :, None]  # a ramp: rounding cases that noise rarely hits
    lv = [base]
    for d in levels[1:]:
        hh, ww = -(-H // d), -(-W // d)
        pad = np.zeros((hh * d, ww * d, 3), np.uint8)
        pad[:H, :W] = base
        lv.append(pad.reshape(hh, d, ww, d, 3).mean(axis=(1, 3)).astype(np.uint8))
    path = str(tmp_path / "s.tif")
    rd.write_tiled_tiff(path, lv, tile=256, mpp=base_mpp)
    reader = rd.WSIReader.open(input_img=path)
    rows = reader.rows(0.5, "mpp")
    plan = rows.device_plan()
    assert plan is not None and plan.rel >= 1.0
    want = 0.5 / base_mpp
    rel = want / max(d for d in reader.info.level_downsamples if d <= want * (1 + 1e-6))  # (a level's downsample is its width ratio: 2101 / 526, not 4)
    assert abs(plan.rel - rel) < 1e-6 and (plan.k is not None) == (abs(rel - round(rel)) < 1e-9)
    assert plan.lvl == (1 if len(levels) > 1 and base_mpp < 0.13 else 0)
    oh, ow = rows.shape[:2]
    whole = rows[0:oh]
    for a, b in ((0, oh), (oh // 3, oh - 57), (oh - 5, oh)):
        up = SlabUploader(rows, a, b, chunk_bytes=1 << 16)  # small chunks: several per band, boundaries inside tile rows
        assert up.plan is not None
        up.upload_until(b - a)
        torch.cuda.synchronize()
        got = up.slab.cpu().numpy()
        assert got.shape == (b - a, ow, 3)
        bad = np.argwhere(got != whole[a:b])
        assert bad.size == 0, (base_mpp, a, b, bad[:5], got[tuple(bad[0])] if bad.size else None)
        assert up.k >= (2 if b - a > 300 else 1)
    os.environ["CERB_DEVICE_RESAMPLE"] = "0"
    try:
        up = SlabUploader(rows, 0, oh)
        assert up.plan is None
        up.upload_until(oh)
        torch.cuda.synchronize()
        assert np.array_equal(up.slab.cpu().numpy(), whole)
    finally:
        del os.environ["CERB_DEVICE_RESAMPLE"]


def test_images_sharing_batches_equal_images_alone(manager):
    """infer_images: several files of different sizes through shared batches (one canvas buffer per head, row blocks per image) give,
    image by image, bit-identical maps and labels to infer_image on its own -- and a folder of 256^2 tiles no longer runs at batch 1."""
    rs = np.random.RandomState(12)
    imgs = [rs.randint(0, 256, hw + (3,)).astype(np.uint8) for hw in ((256, 256), (300, 421), (256, 256), (97, 530), (512, 256))]
    for win, out in ((256, 256), (448, 144)):
        together = manager.infer_images(imgs, win, out, batch_size=6)
        for img, res in zip(imgs, together):
            alone = manager.infer_image(img, win, out, batch_size=6)
            for k in alone["raw"]:
                assert torch.equal(alone["raw"][k], res["raw"][k]), (win, k)
            for t in alone["inst"]:
                assert torch.equal(alone["inst"][t], res["inst"][t]), (win, t)
            assert torch.equal(alone["pclass"], res["pclass"])


def _gpu_shard_worker(rank, world, port, tissue, ret):
    import os

    import torch.distributed as dist

    from cerberus_amd import shard_postproc as sp
    from cerberus_amd import synth_maps as synth
    from cerberus_amd.hostdist import HostStagedDist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 1536, 1024
    m = synth.nuclei_maps(H, W, 31, 600.0, noise=0.02) if tissue == "Nuclei" else synth.blob_maps(H, W, 9, 60, 8.0, 30.0, rim=3.0, sharp=1.0, noise=0.02, holes=0.3)
    bounds = [0, 700, H] if world == 2 else [0, 500, 1010, H]
    band = torch.from_numpy(m[bounds[rank]:bounds[rank + 1]].copy()).cuda()
    out, n_total, info = sp.run_distributed(band, bounds[rank], tissue, 192, 24, HostStagedDist(dist), 1.0 if tissue == "Nuclei" else 0.5)
    ret.put((rank, out.cpu().numpy(), int(n_total), {k: int(v) for k, v in info.items() if k in ("n_truncated", "n_unresolved")}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tissue", [(2, "Nuclei"), (3, "Gland")])
def test_sharded_postproc_on_device_with_staged_collectives(world, tissue):
    """shard_postproc.run_distributed with the DEVICE label / table / relabel kernels in every rank (the ranks share this box's GPU;
    halo strips, counts and crossing-instance tables travel through cerberus_amd.hostdist) against the whole-map post-processing."""
    import socket

    import torch.multiprocessing as mp

    from cerberus_amd import synth_maps as synth
    from cerberus_amd.postproc import postproc_device
    from cerberus_amd.shard_postproc import same_partition

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_gpu_shard_worker, args=(r, world, port, tissue, ret)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    got, t_end = [], time.time() + 300
    while len(got) < world:
        try:
            got.append(ret.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > t_end:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    got.sort(key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    H, W = 1536, 1024
    m = synth.nuclei_maps(H, W, 31, 600.0, noise=0.02) if tissue == "Nuclei" else synth.blob_maps(H, W, 9, 60, 8.0, 30.0, rim=3.0, sharp=1.0, noise=0.02, holes=0.3)
    ref, info = postproc_device(torch.from_numpy(m).cuda(), tissue, 1.0 if tissue == "Nuclei" else 0.5)
    ref = ref.cpu().numpy()
    lab = np.concatenate([g[1] for g in got], axis=0)
    assert all(g[3]["n_truncated"] == 0 and g[3]["n_unresolved"] == 0 for g in got), [g[3] for g in got]
    n_ref = int(ref.max())
    assert n_ref > 20 and got[0][2] == n_ref and lab.shape == ref.shape
    assert same_partition(ref, lab)


def _gpu_gather_worker(rank, world, port, ret):
    import os

    import torch.distributed as dist

    from cerberus_amd.hostdist import HostStagedDist
    from cerberus_amd.wsi import SlideGeometry, gather_bands

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hd = HostStagedDist(dist)
    geo = SlideGeometry((1000, 640), 256, 128)  # 8 patch rows of 128 px, the last one ragged (1000 = 7 * 128 + 104)
    b = geo.bounds(world)
    rows = (b[rank + 1] - b[rank]) * geo.out
    g = torch.Generator().manual_seed(77)
    full_inst = torch.rand((geo.rows * geo.out, geo.cols * geo.out, 2), generator=g)
    full_type = torch.randint(0, 7, (geo.rows * geo.out, geo.cols * geo.out), generator=g, dtype=torch.uint8)
    y0 = b[rank] * geo.out
    canv = {"Nuclei-INST": full_inst[y0:y0 + rows].cuda(), "Nuclei-TYPE": full_type[y0:y0 + rows].cuda()}
    # the halo exchange of shard_postproc.run_distributed, spelled out: 64 rows to each neighbour, received into CUDA buffers
    up, down = canv["Nuclei-INST"][:64].contiguous(), canv["Nuclei-INST"][-64:].contiguous()
    above = torch.empty_like(up) if rank > 0 else None
    below = torch.empty_like(down) if rank < world - 1 else None
    ops = []
    if rank > 0:
        ops += [hd.P2POp(hd.isend, up, rank - 1), hd.P2POp(hd.irecv, above, rank - 1)]
    if rank < world - 1:
        ops += [hd.P2POp(hd.isend, down, rank + 1), hd.P2POp(hd.irecv, below, rank + 1)]
    for req in hd.batch_isend_irecv(ops):
        req.wait()
    halo_ok = True
    if above is not None:
        halo_ok &= bool(torch.equal(above.cpu(), full_inst[y0 - 64:y0]))
    if below is not None:
        halo_ok &= bool(torch.equal(below.cpu(), full_inst[y0 + rows:y0 + rows + 64]))
    full = gather_bands(canv, geo, rank, world, hd)
    ok = None
    if rank == 0:
        ok = bool(full["Nuclei-INST"].is_cuda and torch.equal(full["Nuclei-INST"].cpu(), full_inst[:1000, :640]) and
                  torch.equal(full["Nuclei-TYPE"].cpu(), full_type[:1000, :640]))
    ret.put((rank, halo_ok, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_bands_and_halo_exchange_of_cuda_tensors_through_hostdist(world):
    """wsi.gather_bands (one dist.gather per head, bands padded to the tallest, ragged last patch row cropped on the root) and the
    batched send / recv of the band halos, with CUDA tensors on both sides, through cerberus_amd.hostdist -- the calls "nccl" (RCCL over
    xGMI) moves device to device on a multi-GPU node; here the ranks share this box's GPU and the bytes are staged through the host."""
    import queue
    import socket
    import time

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_gpu_gather_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, t_end = [], time.time() + 300
    while len(got) < world:
        try:
            got.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > t_end:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert all(g[1] for g in got), got
    assert got[0][2] is True


# ---- the reference's nuclei tile scheme, sharded over ranks (cerberus_amd/ref_tiling.py, `run_infer_wsi.py --reference_tiling`) --------------
def _ref_tiling_case():
    from cerberus_amd import synth_maps as synth

    H, W = 3000, 2500
    t = synth.nuclei_maps(1024, 1024, 31, 700.0, noise=0.02)
    m = np.tile(t, (3, 3, 1))[:H, :W].copy()
    tmap = ((np.arange(H)[:, None] // 37 + np.arange(W)[None, :] // 53) % 6).astype(np.uint8)
    return H, W, m, tmap


def _ref_tiling_worker(rank, world, port, bounds, ret):
    import os

    import torch.distributed as dist

    from cerberus_amd import ref_tiling as rt
    from cerberus_amd.hostdist import HostStagedDist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, m, tmap = _ref_tiling_case()
    band = torch.from_numpy(m[bounds[rank]:bounds[rank + 1]].copy()).cuda()
    tband = torch.from_numpy(tmap[bounds[rank]:bounds[rank + 1]].copy()).cuda()
    prof = {}
    got = rt.reference_tiled_nuclei_sharded(band, tband, bounds[rank], (H, W), rank, world, HostStagedDist(dist), tile_shape=1024, margin=32,
                                            patch_output_shape=16, prof=prof)
    if rank == 0:
        ret.put((0, sorted((tuple(int(v) for v in d["box"]), int(d["type"])) for d in got.values()), prof.get("tiles")))
    else:
        assert got is None
        ret.put((rank, None, prof.get("tiles")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bounds", [(2, [0, 1700, 3000]), (3, [0, 900, 2100, 3000]), (4, [0, 400, 800, 2000, 3000])])
def test_reference_tiling_sharded_over_ranks_equals_one_rank(world, bounds):
    """`--reference_tiling` band-sharded over ranks.  Every rank labels the tiles that START in its band (rows below it are
    fetched from the ranks that hold them -- with 400-row bands and 1024-row tiles a rank reaches into the next TWO bands), rank 0 merges in
    the reference's order and applies the cross sections' evictions: the instance set (boxes + majority types) is exactly the one-rank result,
    whatever the band cuts (none of them tile-aligned)."""
    import queue
    import socket
    import time

    import torch.multiprocessing as mp

    from cerberus_amd import ref_tiling as rt

    H, W, m, tmap = _ref_tiling_case()
    one = rt.reference_tiled_nuclei(torch.from_numpy(m).cuda(), torch.from_numpy(tmap).cuda(), tile_shape=1024, margin=32, patch_output_shape=16)
    ref = sorted((tuple(int(v) for v in d["box"]), int(d["type"])) for d in one.values())
    assert len(ref) > 3000 and len(set(b for b, _ in ref)) == len(ref)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_ref_tiling_worker, args=(r, world, port, bounds, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, t_end = [], time.time() + 600
    while len(got) < world:
        try:
            got.append(ret.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > t_end:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][1] == ref
    # every tile is labelled by exactly one rank; more than one rank has tiles (a rank whose band starts below the last tile row has none)
    assert sum(g[2] for g in got) == sum(len(b) for b, _ in rt.get_tile_info((W, H), [1024, 1024], 32, [16, 16])) and sum(g[2] > 0 for g in got) >= 2


class _PixelNet(object):
    """Stand-in for NetDesc in the streaming test: every head is a per-pixel function of the tile (win == out), so that the slide's pixels can carry
    STRUCTURED probability maps (the seeded test weights make one slide-sized nuclei blob, which no band protocol can cut exactly).
    R / G = nuclei inner / contour, B = gland inner; lumen = the gland's core."""
    _decoders = [("Lumen", "INST", 3, "Lumen-INST"), ("Gland", "INST", 3, "Gland-INST"), ("Nuclei", "INST", 3, "Nuclei-INST"),
                 ("Nuclei#TYPE", "TYPE", 7, "Nuclei-TYPE"), ("Gland#TYPE", "TYPE", 3, "Gland-TYPE"), ("Patch-Class", "OUT", 9, "Patch-Class")]

    def _run(self, tiles, oh, ow, outs, _, tile_off=None, row_stride=None, type_is_u8=True, logit_absmax=None):
        t = tiles.float() / 255.0
        zero = torch.zeros_like(t[..., 0])
        vals = [torch.stack([(t[..., 2] - 0.6).clamp(0, 1) * 2.5, zero], -1), torch.stack([t[..., 2], zero], -1), t[..., 0:2].contiguous(),
                (tiles[..., 0] >> 6).to(torch.uint8), (tiles[..., 2] >> 7).to(torch.uint8), (tiles[..., 1] & 7).float()]
        idx = (tile_off[:, None, None] + torch.arange(oh, device=tiles.device)[None, :, None] * row_stride + torch.arange(ow, device=tiles.device)[None, None, :]).reshape(-1)
        for o, v in zip(outs, vals):
            flat = o.view(-1, 2) if o.dim() == 3 else o.view(-1)
            flat[idx] = v.reshape((-1, 2) if o.dim() == 3 else (-1,))


def test_slide_streamed_in_sub_bands_equals_the_resident_run(manager):
    """cerberus_amd.stream_bands (slides whose canvases exceed HBM): the band walked as sequential sub-bands -- a sub-band's probability
    canvases live only until the sub-band below has been inferred, labelling / ownership / global ids are the band protocol run in band order -- must
    give the resident run's label maps and class maps BIT FOR BIT (ids come out in first-pixel raster order whatever the cuts are), with no
    instance cut by a window (n_truncated == 0) and the same counts.  Structured maps through a per-pixel stand-in network; then the real network
    (whose test weights give slide-sized blobs: its class maps must be equal, its label maps wherever the protocol reports them exact)."""
    from cerberus_amd import synth_maps
    from cerberus_amd.shard_postproc import postprocess_bands_and_gather, same_partition
    from cerberus_amd.stream_bands import infer_and_label_streamed

    H, W, margin = 1500, 1300, 256  # (the tallest gland cluster of these maps: 140 rows)
    nuc, gl = synth_maps.nuclei_maps(H, W, 5, 1500.0), synth_maps.blob_maps(H, W, 6, 40, 24.0, 50.0, rim=4.0, sharp=1.0)
    slide = torch.from_numpy(np.stack([nuc[..., 0], nuc[..., 1], gl[..., 0]], -1).clip(0, 1) * 255.0).to(torch.uint8).cuda()
    for net, strict in ((_PixelNet(), True), (manager.net, False)):
        run = WSIRunner(net, (H, W), 256, 256, batch_size=4)
        run.infer_band(slide, 0)
        want, want_info, want_small = postprocess_bands_and_gather(run, H, W, 0, 1, None, margin=margin, guard=16)
        del run
        if strict:
            assert int(want["Nuclei"].max()) > 500 and int(want["Gland"].max()) > 5
        for nb, twin in ((3, None), (2, None if strict else manager.net.twin())):
            calls = []

            def source(y0, y1):
                calls.append((y0, y1))
                return slide[y0:y1].contiguous()

            prof, own = {}, []
            got, info, small = infer_and_label_streamed(net, source, (H, W), 256, 256, 4, nb, margin=margin, guard=16, twin=twin, prof=prof, parts=own)
            assert len(calls) == nb and calls[0][0] == 0 and calls[-1][1] == H and "stream_infer_s" in prof
            if strict:  # the arrays taken from the sub-bands' windows (what run_infer_wsi.py hands the dictionary writer) = the arrays of the whole maps
                from cerberus_amd.shard_postproc import gather_parts
                from cerberus_amd.wsi import collect_wsi_inst_arrays

                mine, whole = _entries(gather_parts(own[0], None, 0, 1, None)), _entries(collect_wsi_inst_arrays(got, small, (H, W)))
                assert set(mine) == set(whole) == set(want)
                for t_ in whole:
                    assert mine[t_] == whole[t_] and len(whole[t_]) > 5, (nb, t_, len(mine[t_]), len(whole[t_]))
            assert set(got) == set(want) and set(small) == set(want_small)
            for k in want_small:
                assert torch.equal(small[k], want_small[k]), (nb, k)
            for t in want:
                assert got[t].shape == want[t].shape and info[t]["local_bands"] == nb, (nb, t, info[t])
                if strict:
                    assert info[t]["n_truncated"] == 0 and info[t]["n_unresolved"] == 0, (nb, t, info[t])
                    assert info[t]["n_total"] == want_info[t]["n_total"], (nb, t)
                    assert torch.equal(got[t], want[t]), (nb, t)
                elif info[t]["n_truncated"] == 0 and info[t]["n_unresolved"] == 0:  # (the test weights' slide-sized blobs are cut by any window: then the
                    assert same_partition(got[t].cpu().numpy(), want[t].cpu().numpy()), (nb, t)  # protocol itself reports that it is not exact)


def _stream_slide(H, W):
    from cerberus_amd import synth_maps

    nuc, gl = synth_maps.nuclei_maps(H, W, 5, 1500.0), synth_maps.blob_maps(H, W, 6, max(4, int(round(80 * H * W / 3.9e6))), 24.0, 50.0, rim=4.0, sharp=1.0)
    return torch.from_numpy(np.stack([nuc[..., 0], nuc[..., 1], gl[..., 0]], -1).clip(0, 1) * 255.0).to(torch.uint8)


def _gpu_stream_worker(rank, world, port, subs, ret, H=3000, W=1300):
    import os

    import torch.distributed as dist

    from cerberus_amd.hostdist import HostStagedDist
    from cerberus_amd.shard_postproc import gather_parts, gather_streamed_maps
    from cerberus_amd.stream_bands import infer_and_label_streamed

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hd = HostStagedDist(dist)
    slide = _stream_slide(H, W).cuda()
    calls, parts = [], []

    def source(y0, y1):
        calls.append((y0, y1))
        return slide[y0:y1].contiguous()

    inst, info, small = infer_and_label_streamed(_PixelNet(), source, (H, W), 256, 256, 4, subs[rank], margin=256, guard=16, rank=rank, world=world, dist=hd, parts=parts)
    root_parts = gather_parts(parts[0], hd, rank, world, torch.device("cuda", 0))
    g_inst, g_small = gather_streamed_maps(inst, small, (H, W), 256, rank, world, hd, labels=True)
    _, q = gather_streamed_maps(inst, small, (H, W), 256, rank, world, hd, labels=False)
    ret.put((rank, None if g_inst is None else {k: v.cpu().numpy() for k, v in g_inst.items()}, None if g_small is None else {k: v.cpu().numpy() for k, v in g_small.items()},
             {t: {k: int(v) for k, v in i.items() if not isinstance(v, bool)} for t, i in info.items()}, root_parts, len(calls),
             None if q is None else {k: v.cpu().numpy() for k, v in q.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,subs", [(2, (3, 2)), (3, (2, 1, 2))])
def test_slide_streamed_in_sub_bands_on_several_ranks_equals_the_resident_run(world, subs):
    _check_streamed_ranks(world, subs, 3000, 1300)


def _check_streamed_ranks(world, subs, H, W):
    """cerberus_amd.stream_bands on N ranks: every rank walks its own band in sequential sub-bands -- one early halo exchange
    at the rank boundaries (a rank infers its last patch row ahead for the neighbour below), ids offset after the walks, the instances a rank's
    first sub-band sees but the rank above owns named from the border all-gather -- and the ranks' rows, stacked, are the resident ONE-rank run's
    label maps and class maps BIT FOR BIT; the per-rank instance arrays make the one-GPU dictionary entry for entry.  The ranks share this
    GPU; collectives through cerberus_amd.hostdist (gloo); structured maps through the per-pixel stand-in network."""
    import queue
    import socket
    import time

    import torch.multiprocessing as mp

    from cerberus_amd.shard_postproc import postprocess_bands_and_gather
    from cerberus_amd.tissue import pclass_tissue_map
    from cerberus_amd.wsi import collect_wsi_inst_arrays

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_gpu_stream_worker, args=(r, world, port, subs, ret, H, W)) for r in range(world)]
    for p in procs:
        p.start()
    got, t_end = [], time.time() + 500
    while len(got) < world:
        try:
            got.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > t_end:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    got.sort(key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    slide = _stream_slide(H, W).cuda()
    run = WSIRunner(_PixelNet(), (H, W), 256, 256, batch_size=4)
    run.infer_band(slide, 0)
    want, want_info, want_small = postprocess_bands_and_gather(run, H, W, 0, 1, None, margin=256, guard=16)
    assert int(want["Nuclei"].max()) > 0.0004 * H * W and int(want["Gland"].max()) >= 3
    inst, small, infos = got[0][1], got[0][2], [g[3] for g in got]
    for k in want_small:
        assert np.array_equal(small[k], want_small[k].cpu().numpy()), k
    for t in want:
        assert all(i[t]["n_truncated"] == 0 and i[t]["n_unresolved"] == 0 for i in infos), (t, [i[t] for i in infos])
        assert all(i[t]["n_total"] == want_info[t]["n_total"] for i in infos) and sum(i[t]["n_owned"] for i in infos) == want_info[t]["n_total"], t
        assert np.array_equal(inst[t], want[t].cpu().numpy()), t
    have, ref = _entries(got[0][4]), _entries(collect_wsi_inst_arrays(want, want_small, (H, W)))
    for t in want:
        assert have[t] == ref[t] and len(ref[t]) >= 3, (t, len(have[t]), len(ref[t]))
    assert np.array_equal(got[0][6]["Patch-Class@0.25"], pclass_tissue_map(want_small["Patch-Class"]).cpu().numpy())
    # every rank read its sub-bands once, plus -- all but the last -- the patch row it infers ahead for the neighbour below
    assert [g[5] for g in got] == [subs[r] + (1 if r < world - 1 else 0) for r in range(world)]
    # instances cross the rank boundaries (the placeholder path ran)
    from cerberus_amd.wsi import SlideGeometry

    cuts = [c * 256 for c in SlideGeometry((H, W), 256, 256).bounds(world)[1:-1]]
    assert all(len((set(np.unique(inst["Nuclei"][c - 1])) & set(np.unique(inst["Nuclei"][c]))) - {0}) > 0 for c in cuts)


# ---- per-rank instance tables + contours, arrays gathered instead of label maps ------------------------------------------
def _parts_maps(H, W):
    from cerberus_amd import synth_maps as synth

    maps = {"Nuclei-INST": synth.nuclei_maps(H, W, 31, 600.0, noise=0.02),
            "Gland-INST": synth.blob_maps(H, W, 9, 40, 20.0, 70.0, rim=4.0, sharp=1.0, noise=0.02, holes=0.3),
            "Lumen-INST": synth.blob_maps(H, W, 11, 90, 8.0, 30.0, rim=2.0, sharp=1.0, noise=0.02)}
    rs = np.random.RandomState(12)
    # class maps in coarse random blocks (so that an instance's majority vote depends on ALL of its pixels, the ones in the halo included)
    blk = lambda k: np.kron(rs.randint(0, k, (H // 16, W // 16)), np.ones((16, 16), np.int64)).astype(np.uint8)  # noqa: E731
    types = {"Nuclei-TYPE": blk(7), "Gland-TYPE": blk(3), "Patch-Class": np.kron(rs.randint(0, 9, (H // 256, W // 256)), np.ones((256, 256))).astype(np.float32)}
    return maps, types


class _BandRun(object):
    """What postprocess_bands_and_gather reads of a WSIRunner: geometry, band position, the band's canvases."""

    def __init__(self, geo, rank, world, canv):
        self.geo = geo
        self.r0, self.r1 = geo.band(rank, world)
        self.band_h = (self.r1 - self.r0) * geo.out
        self.canv = canv


def _gpu_parts_worker(rank, world, port, backend, ret):
    import os

    import torch.distributed as dist

    from cerberus_amd.hostdist import HostStagedDist
    from cerberus_amd.shard_postproc import postprocess_bands_and_gather
    from cerberus_amd.wsi import SlideGeometry

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    d = dist if backend == "nccl" else HostStagedDist(dist)
    H, W = 1500, 1024  # 6 patch rows of 256, the last one ragged
    geo = SlideGeometry((H, W), 256, 256)
    maps, types = _parts_maps(geo.rows * 256, W)
    r0, r1 = geo.band(rank, world)
    canv = {k: torch.from_numpy(v[r0 * 256:r1 * 256].copy()).cuda() for k, v in list(maps.items()) + list(types.items())}
    run = _BandRun(geo, rank, world, canv)
    parts, prof = [], {}
    inst, info, small = postprocess_bands_and_gather(run, H, W, rank, world, d, margin={"Nuclei": 128, "Gland": 448, "Lumen": 256}, guard=48, parts=parts,
                                                     gather_maps=False, prof=prof)
    assert inst is None
    ret.put((rank, parts if rank == 0 else None, {t: {k: int(v) for k, v in i.items()} for t, i in info.items()},
             None if small is None else {k: v.cpu().numpy() for k, v in small.items()},
             {k: (int(v["bytes"]), float(v["s"])) for k, v in prof.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _entries(parts):
    """The dictionary entries of a parts list as sortable tuples (uuid keys aside)."""
    from cerberus_amd.inst_info import info_from_table

    out = {}
    for part in parts:
        tissue, tab, cnts, pts, offs, has_type, ds = part[:7]
        info = info_from_table(tab, cnts, pts, offs, bool(has_type), float(ds), flat_box=True)
        out[tissue] = sorted((tuple(int(v) for v in d["box"]), tuple(float(v) for v in np.asarray(d["centroid"], np.float64)), d["contour"].astype(np.int64).tobytes(),
                              d.get("type"), None if "type_prob" not in d else round(float(d["type_prob"]), 12)) for d in info.values())
    return out


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")])
def test_per_rank_instance_tables_and_contours_equal_the_one_gpu_dictionary(world, backend):
    """postprocess_bands_and_gather(parts=..., gather_maps=False) with the DEVICE kernels in every rank (gloo: the ranks share this GPU and stage
    their collectives through the host; nccl: a one-rank RCCL communicator): the entries built from the gathered arrays -- box, centroid, contour,
    type, type_prob -- are the entries of the one-GPU run over the whole maps, one for one; the root receives arrays and the quarter-resolution
    tissue map only."""
    import queue
    import socket
    import time

    import torch.multiprocessing as mp

    from cerberus_amd.tissue import pclass_tissue_map
    from cerberus_amd.wsi import WSIRunner, collect_wsi_inst_arrays

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_gpu_parts_worker, args=(r, world, port, backend, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, t_end = [], time.time() + 400
    while len(got) < world:
        try:
            got.append(ret.get(timeout=2))
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() > t_end:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    got.sort(key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    H, W = 1500, 1024
    maps, types = _parts_maps(1536, W)
    canv = {k: torch.from_numpy(v[:H]).cuda() for k, v in list(maps.items()) + list(types.items())}
    inst, _ = WSIRunner.postprocess(canv, wsi_mode=True)
    want = _entries(collect_wsi_inst_arrays(inst, canv, (H, W)))
    have = _entries(got[0][1])
    for t in ("Nuclei", "Gland", "Lumen"):
        assert all(g[2][t]["n_truncated"] == 0 and g[2][t]["n_unresolved"] == 0 for g in got), (t, [g[2][t] for g in got])
        assert len(want[t]) > (300 if t == "Nuclei" else 5), (t, len(want[t]))
        assert have[t] == want[t], (t, len(have[t]), len(want[t]))
    assert any(e[3] not in (None, 0) for e in have["Nuclei"]) and any(e[3] is not None for e in have["Gland"]) and all(e[3] is None for e in have["Lumen"])
    # the root got the quarter-resolution tissue map, stitched from the bands' own resizes
    assert list(got[0][3].keys()) == ["Patch-Class@0.25"]
    assert np.array_equal(got[0][3]["Patch-Class@0.25"], pclass_tissue_map(canv["Patch-Class"]).cpu().numpy())
    if world > 1:  # bytes into the root: arrays + the small map, against 15 B / px of label + class maps
        moved = got[0][4]["parts_gather"][0] + got[0][4]["root_gather"][0]
        assert moved * 5 < 15 * H * W * (world - 1) / world, moved  # (a 1.5 Mpx toy slide with padded buffers: already 5x below the maps; bench.py reports the slide-scale figure)
