"""Generate tests/golden/ref_tiling_8192.npz: the instance set (sorted boxes) that oracle/wsi_tiles_ref.py -- the CPU restatement of the reference's
tiled nuclei post-processing (infer/wsi.py:81-268, 642-684; the C oracle as the per-tile labeller) -- keeps on the seeded 8192 x 8192 structured map
at the reference's own geometry (4096-pixel tiles, 64-pixel margins, 256-pixel output patches).

    python oracle/gen_golden_ref_tiling.py          (~12 minutes of one core; the GPU test then compares in seconds)

This is a cached ORACLE answer (the oracle itself is unpinned for this scheme: tiatoolbox's _get_tile_info is restated from memory), stored because
running the oracle at this size inside the GPU suite took 756 s of the suite's 19 minutes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth, wsi_tiles_ref as wt  # noqa: E402


def case_map():
    t = synth.nuclei_maps(2048, 2048, 17, 600.0, noise=0.02)
    return np.tile(t, (4, 4, 1))


if __name__ == "__main__":
    m = case_map()
    ref = wt.reference_tiled_nuclei(m, tile_shape=4096, margin=64, patch_output_shape=256)
    boxes = np.array(ref, dtype=np.int32).reshape(-1, 4)
    path = os.path.join(ROOT, "tests", "golden", "ref_tiling_8192.npz")
    np.savez_compressed(path, boxes=boxes, map_sha1=np.array(__import__("hashlib").sha1(m.tobytes()).hexdigest()))
    print("wrote", path, len(boxes), "instances", os.path.getsize(path) // 1024, "KiB")
