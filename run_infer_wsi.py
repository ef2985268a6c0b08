"""Slide-mode inference driver (the role of the reference's run_infer_wsi.py).  Slides open through cerberus_amd.reader.WSIReader
(the interface of tiatoolbox's reader, infer/wsi.py:521-531): tiled / pyramidal TIFF and JPEG-tiled `.svs`, `.npy` uint8 [H, W, 3]
arrays, PNG / JPG images, or a `.txt` file holding `synthetic:<H>x<W>:<seed>`; pixels are read at `--wsi_proc_mag` microns per
pixel when the file records its scan resolution.  The slide band of this rank lives in HBM; patches are gathered, inferred and scattered on the device, the band is
labelled on the device, and only the instance dictionary is written -- `dat/<slide>.dat` in the reference's joblib format
(infer/wsi.py:844-853 there).  Flag names and defaults are the reference's (cerberus_amd/cli.py); the tiling / cache flags it
parses and then overrides with constants are accepted and ignored.  `--msk_dir` tissue masks are honoured: patches without
tissue do not run, the Patch-Class map is masked, gland / lumen are labelled per tissue region (cerberus_amd/tissue.py); every
slide also gets `tissue/<slide>.mat` (infer/wsi.py:688-716).

Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N run_infer_wsi.py ...`; patch rows shard into one
contiguous band per rank, inference needs no collective, post-processing is band-local (cerberus_amd/shard_postproc.py)."""
import glob
from collections import OrderedDict
import os
import sys
import threading
import time

import numpy as np

from cerberus_amd.cli import WSI_OPTIONS, parse, require_model


# largest map labelled in one call on one GPU (400 Mpx = 38 GB of workspace); larger slides are labelled in row bands (CERB_ONE_CALL_MPX moves the line)
ONE_CALL_PX = int(float(os.environ.get("CERB_ONE_CALL_MPX", "400")) * 1e6)


def _release_device_memory():
    """Between two slides: the labelling workspace (up to 96 B / px of the largest call) goes back to the allocator and everything unreachable is
    collected, so that stream_bands.plan_slide prices the next slide against the HBM it can really have."""
    import gc

    import torch

    from cerberus_amd import postproc

    postproc._ws_cache.clear()
    gc.collect()
    # (the allocator keeps the freed blocks: hbm_budget counts them as free, a slide of the same size gets them back at once, and handing ~200 GB
    #  to the driver and asking for it again cost 4.5 s per 3.2-Gpx slide; WSIRunner empties the cache when a handle's own allocation fails)


def _basename(path, ext):
    base = os.path.basename(path)
    return base[: -len(ext)] if ext else base


def _slide_logger(log_dir, base):
    """`<logging_dir>/<slide>_<dd-mm-YYYY_HH:MM:SS>_std.log` with the reference's record format (infer/wsi.py:957-967): phase timings of
    one slide.  A dedicated logger (the reference attaches the file to the root logger and clears it after the slide)."""
    import logging
    from datetime import datetime

    if not log_dir:
        return None
    os.makedirs(log_dir, exist_ok=True)
    lg = logging.getLogger("cerberus_amd.wsi.%s" % base)
    lg.handlers.clear()
    lg.propagate = False
    fh = logging.FileHandler(filename=os.path.join(log_dir, "%s_%s_std.log" % (base, datetime.now().strftime("%d-%m-%Y_%H:%M:%S"))), mode="w")
    fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    lg.addHandler(fh)
    lg.setLevel(logging.DEBUG)
    return lg


def _close_logger(lg):
    for h in list(lg.handlers):
        h.close()
    lg.handlers.clear()


def _open_slide(path, proc_mpp):
    """-> (row source or None, H, W, seed, reader): rows of the slide at the processing resolution (`--wsi_proc_mag` microns per
    pixel, infer/wsi.py:521-527) through cerberus_amd.reader.WSIReader; None means a synthetic slide generated on the device."""
    from cerberus_amd.reader import SyntheticReader, WSIReader

    reader = WSIReader.open(input_img=path)
    w, h = [int(v) for v in reader.slide_dimensions(resolution=proc_mpp, units="mpp")]
    if isinstance(reader, SyntheticReader):
        return None, h, w, reader.seed, reader
    return reader.rows(proc_mpp, "mpp"), h, w, 0, reader


def main(argv=None):
    args = parse("run_infer_wsi.py", WSI_OPTIONS, argv, version="CoBi Gland Inference")
    require_model(args)
    from cerberus_amd import launch

    backend = os.environ.get("CERB_DIST_BACKEND", "nccl")
    if args["--gpu"] and "WORLD_SIZE" not in os.environ:
        # the reference's `--gpu=0,1` drives both devices from one process (DataParallel, infer/base.py:46-47); here every listed device gets
        # its own rank: the driver re-executes itself once per id (cerberus_amd/launch.py) -- or refuses when the devices are not there
        if not os.environ.get("CERB_OVERSUBSCRIBE"):  # (plumbing tests list more ids than the box has devices: the ranks then time-share device 0)
            os.environ["HIP_VISIBLE_DEVICES"] = args["--gpu"]
        ids = [g for g in args["--gpu"].split(",") if g.strip() != ""]
        if len(ids) > 1:
            launch.ensure_world(len(ids), backend, argv=[os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv),
                                oversubscribe=bool(os.environ.get("CERB_OVERSUBSCRIBE")))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    import torch

    from cerberus_amd.tile import InferManager
    from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs
    from cerberus_amd.wsi import DatWriter, SlabUploader, WSIRunner, check_shardable, collect_wsi_inst_arrays, synth_slide, wsi_meta

    dist, watch = None, launch.null_watch()
    n_dev = max(1, torch.cuda.device_count())
    if world > 1 and backend == "nccl" and world > n_dev:
        raise SystemExit("%d ranks but %d visible GPU(s): RCCL needs one device per rank" % (world, n_dev))
    local = local % n_dev
    torch.cuda.set_device(local)
    # CERB_FORCE_DIST=1: open the communicator even for ONE rank, so that every collective of the N-rank path (skip-flag broadcast, barriers, the band
    # protocol's all-gathers and gathers, --reference_tiling's gather_object) runs over RCCL on a one-GPU box (tests/test_cli_gpu.py)
    if world > 1 or os.environ.get("CERB_FORCE_DIST"):  # "nccl" = RCCL over xGMI, one process per GPU; "gloo" = host-staged collectives (cerberus_amd/hostdist.py)
        dist = launch.init_dist(backend, local)
        watch = launch.PhaseWatch(rank)
        with watch.phase("rank identity all-gather (first collective on the communicator)"):
            ident = launch.rank_identity(dist, torch.device("cuda", local), backend)
        if rank == 0:
            print("ranks: %d over %s on %d distinct device(s): %s" % (ident["world"], ident["backend"], ident["distinct_devices"],
                                                                     ", ".join("%d:cuda%d[%s]" % (r["rank"], r["device"], (r["uuid"] or "")[:13]) for r in ident["ranks"])))
    out_dir = args["--output_dir"]
    os.makedirs(out_dir, exist_ok=True)

    checkpoint, decoders, model_args = None, dict(DEFAULT_REQ_TARGET_CODE), default_model_kwargs()
    if args["--model"]:
        import yaml

        checkpoint = os.path.join(args["--model"], "weights.tar")
        with open(os.path.join(args["--model"], "settings.yml")) as fh:
            settings = yaml.full_load(fh)
        decoders, model_args = settings["dataset_kwargs"]["req_target_code"], settings["model_kwargs"]
    if (args["--wsi_file_ext"] or "").lower() in (".tif", ".tiff", ".svs"):  # tiled files: their tile-decode worker processes start underneath the model's loading
        from cerberus_amd.reader import warm_decode_workers

        warm_decode_workers()
    manager = InferManager(checkpoint_path=checkpoint, decoder_dict=decoders, model_args=model_args)
    # a second handle with the same weights (WSIRunner alternates batches between the two on two streams: +2 .. 3 %; CERB_WSI_STREAMS=1: one handle)
    # is made per slide, and only when cerberus_amd.stream_bands.plan_slide finds room for its workspace beside everything else
    want_twin = os.environ.get("CERB_WSI_STREAMS", "2") == "2"
    twin = None

    ext = args["--wsi_file_ext"]
    slides = sorted(glob.glob(os.path.join(args["--input_dir"], "*" + ext))) if args["--input_dir"] else []
    step, bulk = int(args["--wsi_proc_step"]), int(args["--wsi_bulk_idx"])  # the reference's batch-of-slides window
    msk_dir = args["--msk_dir"]
    if msk_dir:  # only slides that have a mask are considered (run_infer_wsi.py:76-83 there)
        slides = [p for p in slides if os.path.isfile(os.path.join(msk_dir, _basename(p, ext) + ".png"))]
    slides = slides[(bulk - 1) * step: bulk * step]
    print("Number of WSIs in list:", len(slides))
    win, out, batch = int(args["--patch_input_shape"]), int(args["--patch_output_shape"]), int(args["--batch_size"])
    writer = tissue_writer = None
    tissue_err = []
    log_dir = args["--logging_dir"]
    for path in slides:
        base = _basename(path, ext)
        dat_path = os.path.join(out_dir, "dat", base + ".dat")
        done = os.path.exists(dat_path)  # a finished slide is skipped on re-runs (infer/wsi.py:969-978)
        if dist is not None:  # rank 0's view decides for everybody: ranks that disagreed (laggy / unshared file system) would deadlock
            flag = torch.tensor([1 if done else 0], dtype=torch.int64, device="cuda")
            with watch.phase("skip-flag broadcast (%s)" % base):
                dist.broadcast(flag, src=0)
            done = bool(int(flag.item()))
        log = _slide_logger(log_dir, base) if rank == 0 else None  # one log file per slide and run (infer/wsi.py:957-980)
        if done:
            if log:
                log.warning("Skip %s- already processed!" % base)
                _close_logger(log)
            continue
        if log:
            log.info("Processing %s ..." % base)
        # nothing of the previous slide may still hold HBM when this one is priced: its runner and canvases, label maps, closures over them
        # (a directory of slides is the reference's normal job; the second 40000^2 slide used to be planned against what the first had left)
        # (every name the loop body binds to something slide-sized, on the device or the host: three 3.2-Gpx slides in a row showed `nuc_only` and
        #  `pmap` -- the label maps and the tissue map of the previous slide, 21 GB -- still alive here)
        run = maps = inst = nuc_only = nb = lab = v = d = dst = rec = records = pmap = pre = progress = pending = up = slab_dev = source = None  # noqa: F841
        regions = ref_nuclei = parts = extra = prebuilt = rank_parts = own_parts = host = reader = mask = sel = guard = flag = None  # noqa: F841
        _release_device_memory()
        t0 = time.perf_counter()
        host, H, W, seed, reader = _open_slide(path, float(args["--wsi_proc_mag"]))
        if min(H, W) < 4:  # (the quarter-resolution tissue map and the half-resolution gland maps have no pixels: the reference dies in cv2.resize there)
            raise ValueError("%s is %d x %d pixels at the processing resolution: nothing to segment" % (base, H, W))
        mask, sel, regions = None, None, None
        if msk_dir:
            from cerberus_amd.tissue import TissueRegions, load_mask, select_patches
            from cerberus_amd.wsi import SlideGeometry

            mask = load_mask(os.path.join(msk_dir, base + ".png"))
            sel = select_patches(mask, SlideGeometry((H, W), win, out).out_boxes(), (H, W))
            if rank == 0 and args["--save_mask"]:
                from PIL import Image

                os.makedirs(os.path.join(out_dir, "mask"), exist_ok=True)
                Image.fromarray(mask * 255).save(os.path.join(out_dir, "mask", base + ".png"))
        check_shardable((H, W), out, world)
        # Capacity check BEFORE anything slide-sized is allocated (the reference streams any slide through 15000^2 tiles + memmaps,
        # infer/wsi.py:551-556, 899): resident band, resident without the second handle, or -- one rank -- sequential sub-bands
        # (cerberus_amd/stream_bands.py); a band that fits neither way ends here with the numbers instead of inside torch.zeros.
        from cerberus_amd.stream_bands import agree_on_plan, infer_and_label_streamed, plan_slide

        plan = plan_slide(manager.net, (H, W), win, out, batch, rank, world, want_twin=want_twin, max_band_px=ONE_CALL_PX if world == 1 else None,
                          allow_stream=(mask is None and not args["--reference_tiling"]))
        if dist is not None:  # the streamed and the resident tail use different collectives: one mode for all ranks
            with watch.phase("memory-plan agreement (%s)" % base):
                plan = agree_on_plan(plan, dist, torch.device("cuda", local) if backend == "nccl" else "cpu")
        if plan.twin and twin is None:
            twin = manager.net.twin()
        if log:
            log.info("Memory plan: {0}".format(plan))
        if getattr(plan, "over_budget", None):
            print("warning:", plan.over_budget)
        if plan.mode == "streamed":
            t_prep = time.perf_counter()
            if host is None:
                source = lambda a, b: synth_slide(b - a, W, y0=a, seed=seed)  # noqa: E731
            elif isinstance(host, np.ndarray) and not isinstance(host, np.memmap):
                source = lambda a, b: torch.from_numpy(np.ascontiguousarray(host[a:b])).cuda()  # noqa: E731
            else:
                def source(a, b):
                    up = SlabUploader(host, a, b)
                    return up.slab, up.upload_until
            pprof, records, ref_nuclei, rank_parts = {}, None, None, None
            several = dist is not None and world > 1
            # the instance arrays are taken from every sub-band's window while it is in HBM -- on one rank as well (round 6: the whole-slide table +
            # contour pass over a 9.7-Gpx slide's label maps wanted a 77 GB union-find workspace and 5 s); --save_label_maps keeps the whole-map pass
            own_parts = [] if not args["--save_label_maps"] else None
            inst, _, maps = infer_and_label_streamed(manager.net, source, (H, W), win, out, batch, plan.sub_bands, prof=pprof, rank=rank, world=world,
                                                     dist=dist if several else None, parts=own_parts, watch=watch)
            torch.cuda.synchronize()
            if several:
                # every rank holds its own rows: the root gets the instance arrays (+ the quarter-resolution tissue map), or -- --save_label_maps -- the maps
                from cerberus_amd.shard_postproc import gather_parts, gather_streamed_maps

                dev_ = torch.device("cuda", local)
                if own_parts is not None:
                    with watch.phase("instance-array gather to rank 0 (%s)" % base):
                        rank_parts = gather_parts(own_parts[0], dist, rank, world, dev_, prof=pprof)
                with watch.phase("map gather to rank 0 (%s)" % base):
                    inst, maps = gather_streamed_maps(inst, maps, (H, W), out, rank, world, dist, labels=own_parts is None)
            elif own_parts is not None:
                from cerberus_amd.shard_postproc import gather_parts

                rank_parts = gather_parts(own_parts[0], None, 0, 1, None)
            t1 = t2 = time.perf_counter()
            if log:
                log.info("Inference Time: {0} ({1} sub-bands streamed through HBM)".format(pprof.get("stream_infer_s"), plan.sub_bands))
                log.info("Nuclei, Gland & Lumen Labelling Time (inside the stream): {0}".format(pprof.get("stream_label_s")))
        else:
            run = WSIRunner(manager.net, (H, W), win, out, batch, rank, world, patch_sel=sel, twin=twin if plan.twin else None)
            y0, y1 = run.slab_rows()  # this rank's band + context halo
            t_prep = time.perf_counter()
            if log:
                log.info("Preparing Input Output Placement: {0}".format(t_prep - t0))
            # CERB_WSI_OVERLAP_TAIL=1 (one GPU, a slide labelled in several local bands): the nuclei bands are labelled on a side stream while the rows
            # below are still being inferred (shard_postproc.IncrementalLocalLabeller); a band is started a few batches after its rows became final, so
            # that the host's waits inside the labelling call have queued inference to hide behind.  Off by default: on resident data the labelling's
            # kernels cost the saturated inference more than the tail they remove (bench.py --overlap-tail: 149.6 against 150.6 Mpx/s).
            pre, progress = {}, None
            if mask is None and world == 1 and H * W > ONE_CALL_PX and os.environ.get("CERB_WSI_OVERLAP_TAIL", "0") == "1":
                from collections import deque

                from cerberus_amd.shard_postproc import band_view, make_incremental

                pre = make_incremental(band_view(run, H, W), dist, max_band_px=ONE_CALL_PX)
                if pre:
                    pending = deque()

                    def progress(n_done, events, pending=pending, pre=pre, run=run):
                        pending.append((run.rows_final(n_done), events))
                        if len(pending) > 6:
                            rows_final, evs = pending.popleft()
                            for lab in pre.values():
                                lab.feed(rows_final, evs)
            if host is None:
                slab_dev = synth_slide(y1 - y0, W, y0=y0, seed=seed)
                run.infer_band(slab_dev, y0, progress=progress)
            elif isinstance(host, np.ndarray) and not isinstance(host, np.memmap):  # already in RAM: one 50 GB/s copy, nothing to hide
                slab_dev = torch.from_numpy(np.ascontiguousarray(host[y0:y1])).cuda()
                run.infer_band(slab_dev, y0, progress=progress)
            else:  # a slide on disk (memory-mapped array, tiled TIFF / .svs pyramid): read / decode + upload chunk by chunk on a copy
                up = SlabUploader(host, y0, y1)  # stream underneath the inference of the rows above
                slab_dev = up.slab
                run.infer_band(slab_dev, y0, ready=up.upload_until, progress=progress)
            torch.cuda.synchronize()
            # what the head kernels saw, batch by batch (cerb_forward_io.logit_absmax): batches whose logits left the range the F(4x4,3x3) default is
            # held to 1e-4 on are counted in the slide's log; CERB_LOGIT_GUARD=rerun re-runs them on F(2x2,3x3) while the slab is still in HBM
            guard = run.logit_report()
            if guard["above"] and run.logit_guard == "rerun":
                run.rerun_flagged(slab_dev, y0, guard["flagged"])
                torch.cuda.synchronize()
            del slab_dev
            if log:
                log.info("Logit guard: {0} of {1} batches above {2:.0f} (largest |logit| {3:.1f}; conv_algo {4}{5})".format(
                    guard["above"], guard["batches"], manager.net.LOGIT_SATURATION, guard["max"], manager.net.precision_decision()["conv_algo"],
                    ", flagged batches re-run on F(2x2,3x3)" if guard["above"] and run.logit_guard == "rerun" else ""))
            t1 = time.perf_counter()
            if dist is not None:
                with watch.phase("end-of-inference barrier (%s)" % base):
                    dist.barrier()
            if log:
                log.info("Inference Time: {0}".format(t1 - t_prep))
            records = None
            pprof = {}
            rank_parts = None
            if mask is None and (dist is not None or H * W > ONE_CALL_PX):
                # band-local labelling with slide-global ids; only int32 label bands and the class maps travel to the root.  On ONE GPU a
                # slide too large for a single labelling call (96 B / px of workspace, 2^31 px) streams through the same protocol band by band
                from cerberus_amd.shard_postproc import postprocess_bands_and_gather

                # Several ranks: every rank builds the instance tables + contours of the instances it owns (on its halo + band + halo window) and rank 0
                # receives those compact arrays (~0.4 GB for a 40000^2 slide) plus the quarter-resolution tissue map; the int32 label bands and the
                # class canvases (15 B / px = 21 GB) only travel when the maps themselves are asked for (--save_label_maps)
                rank_parts = [] if (dist is not None and not args["--save_label_maps"]) else None
                inst, _, maps = postprocess_bands_and_gather(run, H, W, rank, world, dist, max_band_px=ONE_CALL_PX if world == 1 else None, prof=pprof, watch=watch, pre=pre,
                                                             parts=rank_parts, gather_maps=rank_parts is None)
            else:  # with a mask gland / lumen are labelled per tissue region (infer/wsi.py:730-835), on the root
                with watch.phase("canvas gather to rank 0 (%s)" % base):
                    maps = run.gather_to_root(dist)
                if rank == 0 and mask is not None:
                    from cerberus_amd.postproc import postproc_device
                    from cerberus_amd.tissue import postprocess_regions

                    regions = TissueRegions(torch.from_numpy(mask).cuda())
                    inst = {}
                    if "Nuclei-INST" in maps and H * W > ONE_CALL_PX:
                        # (nuclei do not care about tissue regions -- unselected patches left zeros -- so a slide too large for one labelling call, or past
                        #  2^31 pixels, goes through the row bands of the mask-less path: exact ownership, slide-global ids)
                        from cerberus_amd.shard_postproc import sharded_postprocess

                        nb, ninfo = sharded_postprocess(OrderedDict([("Nuclei-INST", maps["Nuclei-INST"][:H, :W])]), 0, 1, None, wsi_mode=True, max_band_px=ONE_CALL_PX, prof=pprof)
                        inst["Nuclei"] = nb["Nuclei"]
                        if log:
                            bad = ninfo["Nuclei"].get("n_truncated") or ninfo["Nuclei"].get("n_unresolved")
                            (log.warning if bad else log.info)("Nuclei labelled in {0} row bands under the tissue mask: {1}".format(ninfo["Nuclei"].get("local_bands"), ninfo["Nuclei"]))
                    elif "Nuclei-INST" in maps:
                        inst["Nuclei"] = postproc_device(maps["Nuclei-INST"], "Nuclei", exact_ties=False)[0]
                    records = postprocess_regions(maps, (H, W), regions)
                elif rank == 0:
                    inst, _ = WSIRunner.postprocess(maps, wsi_mode=True)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ref_nuclei = None
            if args["--reference_tiling"] and "Nuclei-INST" in run.canv:
                # the reference's own tile sets and margin rules (infer/wsi.py:81-268, 642-684) over the band canvases, every rank labelling the tiles
                # that start in its band (rows below it fetched from its neighbours), each tile with skimage's tie order: the reference's instance
                # set exactly, seam losses included; merged on rank 0
                from cerberus_amd.ref_tiling import reference_tiled_nuclei_sharded

                valid = max(0, min(run.band_h, H - run.r0 * out))
                tprof = {}
                ref_nuclei = reference_tiled_nuclei_sharded(run.canv["Nuclei-INST"][:valid], None if "Nuclei-TYPE" not in run.canv else run.canv["Nuclei-TYPE"][:valid],
                                                            run.r0 * out, (H, W), rank, world, dist, tile_shape=4096, margin=64, patch_output_shape=out, watch=watch,
                                                            prof=tprof, as_part=True)  # arrays: the writer process builds the entries
                if log:
                    log.info("Reference-Tiled Nuclei Time: {0} ({1} tiles on rank 0)".format(time.perf_counter() - t2, tprof.get("tiles")))
            if twin is not None and plan.twin and run.twin is None:
                twin = None  # its workspace did not fit after all (WSIRunner dropped it): release the handle
        if rank != 0:
            continue
        # the reference times nuclei, the tissue map and gland + lumen as separate phases (infer/wsi.py:684, 719, 856); here the three tissues
        # are labelled in one call: the per-tissue split comes from the labelling calls' own clocks when the band protocol ran
        t_nuc = pprof.get("label_Nuclei", {}).get("s")
        t_gl = sum(pprof.get(k, {}).get("s", 0.0) for k in ("label_Gland", "label_Lumen")) if t_nuc is not None else None
        if log:
            if t_nuc is not None:
                log.info("Nuclei Post Proc Time: {0}".format(t_nuc))
            else:
                log.info("Nuclei, Gland & Lumen Labelling Time: {0}".format(t2 - t1))
        if "Patch-Class" in maps or "Patch-Class@0.25" in maps:  # tissue-region map (infer/wsi.py:688-716)
            import scipy.io as sio

            from cerberus_amd.tissue import pclass_tissue_map

            os.makedirs(os.path.join(out_dir, "tissue"), exist_ok=True)
            # (several ranks without --save_label_maps: every rank resized its own band, the root holds the stitched quarter-resolution map)
            pmap = maps["Patch-Class@0.25"] if "Patch-Class@0.25" in maps else pclass_tissue_map(maps["Patch-Class"], None if regions is None else regions.mask)
            # (0.8 GB for a 3.2-Gpx slide: 1.1 s of file writing -- on a thread, underneath the dictionary and the next slide; one file in flight)
            if tissue_writer is not None:
                tissue_writer.join()
            if tissue_err:
                raise tissue_err[0]

            def _write_tissue(dst, arr):
                try:
                    sio.savemat(dst + ".part", {"pclass": arr}, appendmat=False)
                    os.replace(dst + ".part", dst)
                except BaseException as e:  # handed to the main thread at the next join
                    tissue_err.append(e)

            tissue_writer = threading.Thread(target=_write_tissue, args=(os.path.join(out_dir, "tissue", base + ".mat"), pmap.cpu().numpy()), name="cerb-tissue-mat")
            tissue_writer.start()
        if log:
            log.info("Tissue Region Post Proc Time: {0}".format(time.perf_counter() - t2))
        if args["--save_label_maps"]:  # the reference keeps only the instance dictionary; the maps are for tests / inspection
            if records is not None:  # per-region maps live on their own half-resolution grids
                for i, rec in enumerate(records):
                    for k, v in rec["inst"].items():
                        inst["%s_region%d" % (k, i)] = v
                    inst["topleft_region%d" % i] = torch.tensor(rec["topleft"])
            np.savez_compressed(os.path.join(out_dir, base + ".npz"), **{k: v.cpu().numpy() for k, v in inst.items()},
                                **{"type_" + k: v.cpu().numpy() for k, v in maps.items() if k.endswith("TYPE")},
                                **({"pclass": maps["Patch-Class"].cpu().numpy()[::4, ::4]} if "Patch-Class" in maps else {}))
        t3 = time.perf_counter()
        os.makedirs(os.path.dirname(dat_path), exist_ok=True)
        nuc_only = None if inst is None else ({k: v for k, v in inst.items() if k == "Nuclei"} if records is not None else inst)
        bw, bh = reader.info.slide_dimensions
        prebuilt = None
        # The dictionary's GPU half (tables, contours) here; its ~1e6 per-instance Python objects, the uuid keys and the pickle in a separate,
        # torch-free writer process underneath the next slide (cerberus_amd.wsi.DatWriter.from_arrays); a finished dat/<slide>.dat is always
        # complete (written to a temporary name and renamed) -- the resume-by-skip above relies on that.
        extra = OrderedDict()
        if records is not None:  # tissue regions: per-region dictionaries are already built (slide coordinates)
            import uuid

            for rec in records:
                for tissue, d in rec["info"].items():
                    dst = extra.setdefault(tissue, OrderedDict())
                    for v in d.values():
                        dst[uuid.uuid4().hex] = v
        for tissue, d in (prebuilt or {}).items():
            extra[tissue] = d
        skip = tuple(extra.keys()) + (("Nuclei",) if ref_nuclei is not None else ())
        if rank_parts is not None:  # built by the ranks that own the instances, gathered as arrays
            parts = [p_ for p_ in rank_parts if p_[0] not in skip]
        else:
            parts = collect_wsi_inst_arrays(nuc_only, maps, (H, W), skip=skip)
        if ref_nuclei is not None:  # the reference-tiled nuclei replace the band scheme's: same kind of arrays (+ per-instance tile origins)
            parts.insert(0, ref_nuclei)
        meta = wsi_meta((H, W), float(args["--wsi_proc_mag"]), base_mag=None if reader.info.mpp is None else float(reader.info.mpp[0]), base_hw=(bh, bw))
        if writer is not None:
            writer.join()
        writer = DatWriter.from_arrays(parts, meta, dat_path, extra=extra)
        t4 = time.perf_counter()
        print("%s: Inference Time: %.3f  Post Proc Time: %.3f  Instance Table Time: %.3f  (%.1f Mpx/s inference)" % (
            base, t1 - t0, t2 - t1, t4 - t3, H * W / (t1 - t0) / 1e6))
        if log:
            # gland + lumen labelling (when timed apart) + contours + the instance dictionary: the reference's last phase ends at joblib.dump
            log.info("Gland & Lumen Post Proc Time: {0}".format((t_gl or 0.0) + (t4 - t3)))
            log.info("Instance Dictionary Time: {0} (dat/%s.dat is serialised on a writer thread underneath the next slide)".format(t4 - t3) % base)
            log.info("Overall Time: {0}".format(t4 - t0))
            log.info("Finish")
            _close_logger(log)
    if writer is not None:
        writer.join()
    if tissue_writer is not None:
        tissue_writer.join()
    if tissue_err:
        raise tissue_err[0]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
