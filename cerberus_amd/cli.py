"""Command lines of the two inference drivers.  The reference parses docopt usage strings (run_infer_tile.py, run_infer_wsi.py);
docopt is not installed here and the drivers behind the flags are different programs, so the interface is a table: flag NAMES and
DEFAULTS are the reference's (a maintainer's existing command lines keep working, `--flag=value` or `--flag value`), the help
text and everything behind it are this package's."""
import sys

# (flag, takes a value, default, help)
_COMMON = [
    ("--gpu", True, "0", "GPU id(s), exported as HIP_VISIBLE_DEVICES when not launched under torch.distributed.run; several ids (`0,1,2`) = one rank per "
                          "device, self-spawned (cerberus_amd/launch.py): slide bands / the tile file list are sharded over them"),
    ("--model", True, None, "directory holding settings.yml + weights.tar (required unless --synthetic)"),
    ("--synthetic", False, False, "(not in the reference) run on the package's seeded synthetic test weights instead of a checkpoint"),
    ("--nr_inference_workers", True, "0", "accepted for compatibility: tiles are gathered on the device, there is no loader pool"),
    ("--nr_post_proc_workers", True, "0", "accepted for compatibility: post-processing runs on the GPU in the main process"),
]
TILE_OPTIONS = _COMMON + [
    ("--batch_size", True, "10", "tiles per forward"),
    ("--input_dir", True, None, "directory of .png / .jpg tiles (searched recursively)"),
    ("--output_dir", True, "output/", "where <tissue>_mat/, pclass_mat/ and overlay/ are written"),
    ("--patch_input_shape", True, "448", "network input window (square)"),
    ("--patch_output_shape", True, "144", "centre crop kept from every window (square)"),
]
WSI_OPTIONS = _COMMON + [
    ("--batch_size", True, "30", "patches per forward"),
    ("--tile_shape", True, "2048", "accepted and ignored (the reference overrides it too): maps stay in HBM, nothing is tiled"),
    ("--chunk_shape", True, "15000", "accepted and ignored"),
    ("--ambiguous_size", True, "64", "accepted and ignored: bands are labelled in one pass, seams are exact (shard_postproc)"),
    ("--wsi_proc_mag", True, "0.5", "microns per pixel recorded in the instance dictionary"),
    ("--wsi_file_ext", True, ".svs", "slide extension: .npy arrays, .png/.jpg images or .txt `synthetic:<H>x<W>:<seed>` specs"),
    ("--cache_path", True, "cache/", "accepted and ignored: there is no memmap cache"),
    ("--logging_dir", True, "logging/", "one <slide>_<date>_std.log per slide and run with its phase timings (rank 0)"),
    ("--input_dir", True, None, "directory of slides (not searched recursively)"),
    ("--msk_dir", True, None, "directory of tissue masks <slide>.png (any resolution); when given, only slides that have a mask are processed"),
    ("--output_dir", True, "output/", "where dat/<slide>.dat, tissue/<slide>.mat (and <slide>.npz with --save_label_maps) are written"),
    ("--patch_input_shape", True, "448", "network input window (square)"),
    ("--patch_output_shape", True, "144", "centre crop kept from every window (square)"),
    ("--wsi_bulk_idx", True, "1", "accepted for compatibility"),
    ("--wsi_proc_step", True, "10", "accepted for compatibility"),
    ("--save_thumb", False, False, "accepted and ignored"),
    ("--save_mask", False, False, "write the tissue mask used as <output_dir>/mask/<slide>.png"),
    ("--save_label_maps", False, False, "(not in the reference) also dump the label / class maps as <output_dir>/<slide>.npz"),
    ("--reference_tiling", False, False, "(not in the reference) nuclei instances through the reference's own 4096 x 4096 tiles, 64-px margins, strips and "
                                         "cross sections (infer/wsi.py:81-268, 642-684; cerberus_amd/ref_tiling.py) instead of exact band ownership: the "
                                         "reference's instance set, including the few seam instances its scheme drops; tiles are sharded over the ranks"),
]


def usage(prog, options):
    lines = ["usage: %s [options]" % prog, "", "options:", "  -h, --help", "  --version"]
    for flag, has_val, default, text in options:
        left = flag + ("=<value>" if has_val else "")
        lines.append("  %-30s %s%s" % (left, text, "" if default in (None, False) else "  [default: %s]" % default))
    return "\n".join(lines)


def require_model(args):
    """A forgotten --model must not produce plausible-looking label maps from synthetic weights."""
    if not args.get("--model") and not args.get("--synthetic"):
        sys.exit("--model=<dir with settings.yml + weights.tar> is required (or --synthetic for the seeded test weights)")


def parse(prog, options, argv=None, version=None):
    """Returns {flag: value}: strings for valued flags (None when absent and without default), booleans for switches."""
    argv = list(sys.argv[1:] if argv is None else argv)
    opts = {flag: default for flag, _, default, _ in options}
    valued = {flag: has_val for flag, has_val, _, _ in options}
    if "-h" in argv or "--help" in argv:
        print(usage(prog, options))
        sys.exit(0)
    if "--version" in argv:
        print(version)
        sys.exit(0)
    i = 0
    while i < len(argv):
        key, eq, val = argv[i].partition("=")
        if key not in opts:
            print(usage(prog, options))
            sys.exit("unknown option %s" % argv[i])
        if not valued[key]:
            opts[key] = True
        elif eq:
            opts[key] = val
        else:
            i += 1
            if i >= len(argv):
                sys.exit("option %s needs a value" % key)
            opts[key] = argv[i]
        i += 1
    return opts
