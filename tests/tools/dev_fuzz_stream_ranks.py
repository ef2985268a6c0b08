"""Developer fuzz (GPU box): cerberus_amd.stream_bands on N ranks against the resident one-rank run for random slide sizes, rank counts and sub-band
counts -- tests/test_drivers_gpu.py::_check_streamed_ranks (bit-equal label / class maps, equal dictionary entries) on geometries the suite does not fix:
ragged last patch rows, odd widths, one-sub-band ranks next to many-sub-band ranks.
    python tests/tools/dev_fuzz_stream_ranks.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import test_drivers_gpu as T  # noqa: E402
from cerberus_amd.wsi import SlideGeometry  # noqa: E402

def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    bad = 0
    for i in range(n_cases):
        world = int(rs.choice([2, 3, 4]))
        rows_per_rank = int(rs.randint(2, 7))
        H = 256 * rows_per_rank * world - int(rs.choice([0, 0, 37, 100, 200]))  # ragged last patch row
        W = int(rs.choice([1024, 1300, 1110, 900]))
        b = SlideGeometry((H, W), 256, 256).bounds(world)
        subs = tuple(int(rs.randint(1, max(1, (b[r + 1] - b[r]) // 2) + 1)) for r in range(world))  # sub-bands of at least two patch rows (two margins of 256)
        try:
            T._check_streamed_ranks(world, subs, H, W)
            print("case %d ok: %d x %d, world %d, sub-bands %s" % (i, H, W, world, subs), flush=True)
        except AssertionError as e:
            bad += 1
            print("case %d FAILED: %d x %d, world %d, sub-bands %s: %s" % (i, H, W, world, subs, str(e)[:400]), flush=True)
    print("fuzz: %d cases, %d failures" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":  # (the ranks are spawned: they import this module again)
    main()
