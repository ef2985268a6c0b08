"""Generate tests/golden/tile_patching.npz from the reference's infer/tile.py::_prepare_patching (numpy-only code;
its module-level imports cv2 / torch / termcolor / viz are inert stubs here).
Run: /opt/conda/bin/python3.9 oracle/gen_golden_tile.py   (numpy 1.x: the reference calls np.lib.pad, gone in numpy 2)"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
for m in ["cv2", "skimage", "skimage.filters", "skimage.morphology", "skimage.segmentation", "termcolor", "matplotlib", "matplotlib.pyplot",
          "matplotlib.cm", "matplotlib.colors", "tqdm", "pandas", "joblib"]:
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
import types  # noqa: E402

if "torch" not in sys.modules:
    try:
        import torch  # noqa: F401
    except Exception:
        tud = types.ModuleType("torch.utils.data")
        tud.IterableDataset = object
        tud.Dataset = object
        tud.DataLoader = object
        tud.get_worker_info = lambda: None
        tu = types.ModuleType("torch.utils")
        tu.data = tud
        t = MagicMock()
        t.utils = tu
        sys.modules["torch"] = t
        sys.modules["torch.utils"] = tu
        sys.modules["torch.utils.data"] = tud
sys.modules["misc.viz_utils"] = MagicMock()  # drawing helpers only (needs scipy.interp, removed from modern scipy)
from infer.tile import _prepare_patching  # noqa: E402  (reference)

store = {}
cases = [(300, 421, 256, 256, 0), (1000, 777, 448, 144, 0), (144, 144, 448, 144, 0), (513, 257, 256, 256, 0), (95, 130, 448, 144, 0)]
_rs = np.random.RandomState(2024)  # + random geometries (window a multiple of 16, even window - output gap, a few overlaps)
for _ in range(24):
    _win = 16 * int(_rs.randint(6, 30))
    _out = int(_rs.randint(_win // 3, _win + 1))
    _out -= (_win - _out) % 2
    _ovl = int(_rs.choice([0, 0, 0, 8, 32])) if _out > 64 else 0
    cases.append((int(_rs.randint(20, 1200)), int(_rs.randint(20, 1200)), _win, _out, _ovl))
for i, (h, w, win, out, ovl) in enumerate(cases):
    img = np.random.RandomState(100 + i).randint(0, 256, (h, w, 3)).astype(np.uint8)
    padded, info, pos = _prepare_patching(img, win, out, ovl)
    store["case%d/args" % i] = np.array([h, w, win, out, ovl, 100 + i])
    store["case%d/padded_shape" % i] = np.array(padded.shape)
    store["case%d/padded_sum" % i] = np.int64(padded.astype(np.int64).sum())
    store["case%d/padded_crc" % i] = np.int64((padded.astype(np.int64) * (np.arange(padded.size).reshape(padded.shape) % 9973)).sum())
    store["case%d/info" % i] = info
    store["case%d/pos" % i] = np.array(pos)
    print(i, (h, w, win, out), padded.shape, info.shape, pos)
store["n"] = len(cases)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tile_patching.npz"), **store)
