"""run_infer_tile.py

Usage:
  run_infer_tile.py [--gpu=<id>] [--model=<path>] [--nr_inference_workers=<n>] \
            [--nr_post_proc_workers=<n>] [--batch_size=<n>] [--input_dir=<path>] \
            [--output_dir=<path>] [--patch_input_shape=<n>] [--patch_output_shape=<n>]
  run_infer_tile.py (-h | --help)
  run_infer_tile.py --version

Options:
  -h --help                   Show this string.
  --version                   Show version.
  --gpu=<id>                  GPU list. [default: 0]
  --model=<path>              Path to saved checkpoint.
  --nr_inference_workers=<n>  Number of workers during inference. [default: 0]
  --nr_post_proc_workers=<n>  Number of workers during post-processing. [default: 0]
  --batch_size=<n>            Batch size. [default: 10]
  --input_dir=<path>          Path to input data directory. Assumes the files are not nested within directory.
  --output_dir=<path>         Path to output data directory. Will create automtically if doesn't exist. [default: output/]
  --patch_input_shape=<n>     Shape of input patch to the network- Assume square shape. [default: 448]
  --patch_output_shape=<n>    Shape of network output- Assume square shape. [default: 144]

"""
# Same command line as the reference's run_infer_tile.py (flags verbatim, :4-21); the managers behind it are the
# MI355X-native ones of cerberus_amd (HIP kernels behind libcerberus_hip.so).  <model>/settings.yml + weights.tar as in
# the reference (:47-49); without --model a seeded synthetic checkpoint is used (no network here to fetch weights).
import os

import yaml

from cerberus_amd.cli import parse

if __name__ == "__main__":
    args = parse(__doc__, version="CoBi Gland Inference")
    if args["--gpu"]:
        os.environ["HIP_VISIBLE_DEVICES"] = args["--gpu"]
    output_dir = args["--output_dir"]
    os.makedirs(output_dir, exist_ok=True)
    from cerberus_amd.tile import InferManager
    from cerberus_amd.weights import DEFAULT_REQ_TARGET_CODE, default_model_kwargs

    if args["--model"]:
        checkpoint_path = "%s/weights.tar" % args["--model"]
        with open("%s/settings.yml" % args["--model"]) as fptr:
            run_paramset = yaml.full_load(fptr)
        decoder_dict, model_args = run_paramset["dataset_kwargs"]["req_target_code"], run_paramset["model_kwargs"]
    else:
        checkpoint_path, decoder_dict, model_args = None, dict(DEFAULT_REQ_TARGET_CODE), default_model_kwargs()
    run_args = {
        "nr_inference_workers": int(args["--nr_inference_workers"]),
        "nr_post_proc_workers": int(args["--nr_post_proc_workers"]),
        "batch_size": int(args["--batch_size"]),
        "input_dir": args["--input_dir"],
        "output_dir": output_dir,
        "patch_input_shape": int(args["--patch_input_shape"]),
        "patch_output_shape": int(args["--patch_output_shape"]),
        "patch_output_overlap": 0,
        "postproc_list": ["gland", "lumen", "nuclei", "patch-class"],
    }
    infer = InferManager(checkpoint_path=checkpoint_path, decoder_dict=decoder_dict, model_args=model_args)
    infer.process_file_list(run_args)
