"""Tile-mode inference driver (the role of the reference's run_infer_tile.py): every .png / .jpg under --input_dir through
the MI355X-native InferManager -- HIP forward, on-GPU post-processing, instance table -- writing <tissue>_mat/<name>.mat,
pclass_mat/<name>.mat and overlay/<name>.jpg.  Flag names and defaults are the reference's (cerberus_amd/cli.py);
<model>/settings.yml + weights.tar are read as the reference reads them (run_infer_tile.py:47-49 there); --synthetic
selects the package's seeded test weights instead (this image has no network to fetch a checkpoint)."""
import os
import sys

from cerberus_amd.cli import TILE_OPTIONS, parse, require_model


def main(argv=None):
    args = parse("run_infer_tile.py", TILE_OPTIONS, argv, version="CoBi Gland Inference")
    require_model(args)
    if args["--gpu"] and "WORLD_SIZE" not in os.environ:
        # `--gpu=0,1`: the reference drives both devices from one process (DataParallel, infer/base.py:46-47); here one rank per listed device,
        # self-spawned (cerberus_amd/launch.py), each taking every w-th file -- or a non-zero exit when the devices are not there
        if not os.environ.get("CERB_OVERSUBSCRIBE"):  # (plumbing tests list more ids than the box has devices: the ranks then time-share device 0)
            os.environ["HIP_VISIBLE_DEVICES"] = args["--gpu"]
        ids = [g for g in args["--gpu"].split(",") if g.strip() != ""]
        if len(ids) > 1:
            from cerberus_amd import launch

            launch.ensure_world(len(ids), "gloo" if os.environ.get("CERB_OVERSUBSCRIBE") else "nccl",
                                argv=[os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), oversubscribe=bool(os.environ.get("CERB_OVERSUBSCRIBE")))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    os.makedirs(args["--output_dir"], exist_ok=True)
    dist = None
    if world > 1 or os.environ.get("CERB_FORCE_DIST"):
        import torch

        from cerberus_amd import launch

        n_dev = max(1, torch.cuda.device_count())
        if world > n_dev and not os.environ.get("CERB_OVERSUBSCRIBE"):
            raise SystemExit("%d ranks but %d visible GPU(s)" % (world, n_dev))
        torch.cuda.set_device(local % n_dev)
        # Tile mode shards FILES (rank r takes every world-th image): no data-path collective.  The communicator is opened all the same -- the rank
        # identities are gathered over it (a launcher that started fewer ranks than it said, or two ranks on one device, shows here and not as a
        # silently missing third of the output) and the ranks leave together through a barrier.  CERB_FORCE_DIST=1: also with one rank (tests).
        backend = os.environ.get("CERB_DIST_BACKEND", "gloo" if os.environ.get("CERB_OVERSUBSCRIBE") else "nccl")
        dist = launch.init_dist(backend, local % n_dev)
        ident = launch.rank_identity(dist, torch.device("cuda", local % n_dev), backend)
        if ident["world"] != world:
            raise SystemExit("the communicator spans %d ranks, the launcher said %d" % (ident["world"], world))
        if rank == 0:
            print("ranks: %d over %s on %d distinct device(s)" % (ident["world"], ident["backend"], ident["distinct_devices"]))
    from cerberus_amd.tile import InferManager
    from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs

    checkpoint, decoders, model_args = None, dict(DEFAULT_REQ_TARGET_CODE), default_model_kwargs()
    if args["--model"]:
        import yaml

        checkpoint = os.path.join(args["--model"], "weights.tar")
        with open(os.path.join(args["--model"], "settings.yml")) as fh:
            settings = yaml.full_load(fh)
        decoders, model_args = settings["dataset_kwargs"]["req_target_code"], settings["model_kwargs"]
    manager = InferManager(checkpoint_path=checkpoint, decoder_dict=decoders, model_args=model_args)
    manager.process_file_list({
        "input_dir": args["--input_dir"],
        "output_dir": args["--output_dir"],
        "batch_size": int(args["--batch_size"]),
        "patch_input_shape": int(args["--patch_input_shape"]),
        "patch_output_shape": int(args["--patch_output_shape"]),
        "patch_output_overlap": 0,
        "nr_inference_workers": int(args["--nr_inference_workers"]),
        "nr_post_proc_workers": int(args["--nr_post_proc_workers"]),
        "postproc_list": ["gland", "lumen", "nuclei", "patch-class"],
        "rank": rank,
        "world_size": world,
    })
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
