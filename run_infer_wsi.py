"""run_infer_wsi.py

Usage:
  run_infer_wsi.py [--gpu=<id>] [--model=<path>] [--nr_inference_workers=<n>] \
            [--nr_post_proc_workers=<n>] [--batch_size=<n>] [--tile_shape=<n>] [--chunk_shape=<n>] \
            [--ambiguous_size=<int>] [--wsi_proc_mag=<n>] [--wsi_file_ext=<str>] [--cache_path=<path>] \
            [--logging_dir=<path>] [--input_dir=<path>] [--msk_dir=<path>] [--output_dir=<path>] [--patch_input_shape=<n>] \
            [--patch_output_shape=<n>] [--wsi_bulk_idx=<n>] [--wsi_proc_step=<n>] [--save_thumb] [--save_mask] [--save_label_maps]
  run_infer_wsi.py (-h | --help)
  run_infer_wsi.py --version

Options:
  -h --help                   Show this string.
  --version                   Show version.
  --gpu=<id>                  GPU list. [default: 0]
  --model=<path>              Path to saved checkpoint.
  --nr_inference_workers=<n>  Number of workers during inference. [default: 0]
  --nr_post_proc_workers=<n>  Number of workers during post-processing. [default: 0]
  --batch_size=<n>            Batch size. [default: 30]
  --tile_shape=<n>            Shape of tile for processing. [default: 2048]
  --chunk_shape=<n>           Shape of tile for processing. [default: 15000]
  --ambiguous_size=<int>      Define ambiguous region along tiling grid to perform re-post processing. [default: 64]
  --wsi_proc_mag=<n>          Microns per pixel used for WSI processing. [default: 0.5]
  --wsi_file_ext=<str>        File extension of WSIs to process. [default: .svs]
  --cache_path=<path>         Path for cache. Should be placed on SSD with at least 100GB. [default: cache/]
  --logging_dir=<path>        Path for python logging. [default: logging/]
  --input_dir=<path>          Path to input data directory. Assumes the files are not nested within directory.
  --msk_dir=<path>            Path to directory containing tissue masks. Should have the same name as corresponding WSIs.
  --output_dir=<path>         Path to output data directory. Will create automtically if doesn't exist. [default: output/]
  --patch_input_shape=<n>     Shape of input patch to the network- Assume square shape. [default: 448]
  --patch_output_shape=<n>    Shape of network output- Assume square shape. [default: 144]
  --wsi_bulk_idx=<n>          Index for batch processing. Indexing is from 0 to n-1. [default: 1]
  --wsi_proc_step=<n>         Increments for batch WSI processing. [default: 10]
  --save_thumb                Whether to save the slide thumbnail
  --save_mask                 Whether to save the slide mask
  --save_label_maps           (not in the reference) also dump the label / class maps as <output_dir>/<slide>.npz

"""
# Same command line as the reference's run_infer_wsi.py (flags verbatim, :4-35).  Slide-file decoding (tiatoolbox
# WSIReader) is out of scope: slides are `.npy` uint8 [H,W,3] arrays, PNG/JPG images, or `synthetic:<H>x<W>:<seed>`
# names listed in a text file.  The tiling / cache flags the reference parses and then overrides with constants
# (infer/wsi.py:885-915) are accepted and ignored: the maps live in HBM, there is no cache and no chunking.
# Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N`; tiles shard by bands of patch rows.
import glob
import os
import time

import numpy as np
import yaml

from cerberus_amd.cli import parse

if __name__ == "__main__":
    args = parse(__doc__, version="CoBi Gland Inference")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if args["--gpu"] and world == 1:
        os.environ["HIP_VISIBLE_DEVICES"] = args["--gpu"]
    import torch

    from cerberus_amd.tile import InferManager
    from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs
    from cerberus_amd.wsi import WSIRunner, synth_slide

    dist = None
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    os.makedirs(args["--output_dir"], exist_ok=True)
    if args["--model"]:
        checkpoint_path = "%s/weights.tar" % args["--model"]
        with open("%s/settings.yml" % args["--model"]) as fptr:
            run_paramset = yaml.full_load(fptr)
        decoder_dict, model_args = run_paramset["dataset_kwargs"]["req_target_code"], run_paramset["model_kwargs"]
    else:
        checkpoint_path, decoder_dict, model_args = None, dict(DEFAULT_REQ_TARGET_CODE), default_model_kwargs()
    mgr = InferManager(checkpoint_path=checkpoint_path, decoder_dict=decoder_dict, model_args=model_args)
    ext = args["--wsi_file_ext"]
    names = sorted(glob.glob("%s/*%s" % (args["--input_dir"], ext))) if args["--input_dir"] else []
    step = int(args["--wsi_proc_step"])
    names = names[(int(args["--wsi_bulk_idx"]) - 1) * step: int(args["--wsi_bulk_idx"]) * step]
    print("Number of WSIs in list:", len(names))
    for path in names:
        base = os.path.basename(path)[: -len(ext)] if ext else os.path.basename(path)
        if os.path.exists("%s/dat/%s.dat" % (args["--output_dir"], base)):  # resume-by-skip (infer/wsi.py:969-978)
            continue
        t0 = time.perf_counter()
        if path.endswith(".npy"):
            host = np.load(path, mmap_mode="r")
            H, W = host.shape[:2]
        elif path.endswith(".txt"):
            _, dims, seed = open(path).read().strip().split(":")
            H, W = [int(v) for v in dims.split("x")]
            host = None
        else:
            from PIL import Image

            host = np.array(Image.open(path).convert("RGB"))
            H, W = host.shape[:2]
        run = WSIRunner(mgr.net, (H, W), int(args["--patch_input_shape"]), int(args["--patch_output_shape"]), int(args["--batch_size"]), rank, world)
        y0, y1 = run.slab_rows()
        slab = synth_slide(y1 - y0, W, y0=y0, seed=int(seed)) if host is None else torch.from_numpy(np.ascontiguousarray(host[y0:y1])).cuda()
        run.infer_band(slab, y0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if world > 1:
            # band-local labelling with slide-global ids; only int32 label bands and the class maps travel to the root
            from cerberus_amd.shard_postproc import postprocess_bands_and_gather

            inst, info, full = postprocess_bands_and_gather(run, H, W, rank, world, dist)
        else:
            full = run.gather_to_root(dist)
            inst, info = WSIRunner.postprocess(full, wsi_mode=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rank == 0:
            if args["--save_label_maps"]:  # the reference keeps only the instance dictionary; the maps are for tests / inspection
                np.savez_compressed("%s/%s.npz" % (args["--output_dir"], base), **{k: v.cpu().numpy() for k, v in inst.items()},
                                    **{"type_" + k: v.cpu().numpy() for k, v in full.items() if k.endswith("TYPE")},
                                    pclass=full.get("Patch-Class").cpu().numpy()[::4, ::4])
            t_dump = time.perf_counter()
            # instance dictionary in the reference's wire format (joblib, infer/wsi.py:844-853)
            import joblib

            from cerberus_amd.wsi import build_wsi_inst_info

            os.makedirs("%s/dat" % args["--output_dir"], exist_ok=True)
            wsi_info = build_wsi_inst_info(inst, full, (H, W), float(args["--wsi_proc_mag"]))
            joblib.dump(wsi_info, "%s/dat/%s.dat" % (args["--output_dir"], base))
            t3 = time.perf_counter()
            print("%s: Inference Time: %.3f  Post Proc Time: %.3f  Instance Table Time: %.3f  (%.1f Mpx/s inference)" % (
                base, t1 - t0, t2 - t1, t3 - t_dump, H * W / (t1 - t0) / 1e6))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
