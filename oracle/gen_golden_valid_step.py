"""Generate tests/golden/valid_step.npz by running the REFERENCE's own `valid_step` (models/run_desc.py:332-436) on CPU in this container.

Run (py3.10 + torch):  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_valid_step.py
Needs /root/reference (read-only); never runs on the GPU box.  What is stored is DATA: the seeded uint8 batch, the targets / target
flags fed in, and result["raw"]["pred"] / ["true"] per head exactly as valid_step returned them (numpy).  Two batches: one whose
samples carry Patch-Class targets (the branch that sends every head through F.interpolate) and one without a Patch-Class head target.
Inert stubs / shims as in gen_golden_train_loss.py."""
import os
import sys
from collections import OrderedDict
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
for m in ["cv2", "skimage", "skimage.filters", "skimage.morphology", "termcolor", "matplotlib", "matplotlib.pyplot", "tensorboardX", "imgaug",
          "imgaug.augmenters"]:
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    if a and a[0] == "cuda":
        return self
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to

from models.net_desc import create_model  # noqa: E402  (reference)
from models.run_desc import valid_step  # noqa: E402  (reference)

from cerberus_amd.weights import default_model_kwargs, make_state_dict  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = create_model(**default_model_kwargs())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    net = torch.nn.DataParallel(model)
    heads = OrderedDict([("Lumen-INST", 3), ("Gland-INST", 3), ("Nuclei-INST", 3), ("Nuclei-TYPE", 7), ("Gland-TYPE", 3), ("Patch-Class", 9)])
    store = {"weight_seed": 0, "heads": np.array(list(heads))}
    for case, (N, H, with_pc) in {"pc/": (2, 64, True), "nopc/": (3, 96, False)}.items():
        rs = np.random.RandomState(5 + N)
        batch = {"img": torch.from_numpy(rs.randint(0, 256, (N, H, H, 3)).astype(np.uint8))}
        for h, c in heads.items():
            t = rs.randint(0, c, (N, 1, 1, 1)) if h == "Patch-Class" else (rs.rand(N, H, H, 1) < 0.4) * rs.randint(1, c, (N, H, H, 1))
            batch[h] = torch.from_numpy(t.astype(np.float32))
            store[case + "target/" + h] = t.astype(np.float32)
        has = np.full((N, len(heads)), None, dtype=object)
        for j, h in enumerate(heads):
            if h == "Patch-Class" and not with_pc:
                continue
            has[:, j] = h
        has[0, 1] = None  # one dummy target, as mixed datasets produce
        batch["dummy_target"] = has
        store[case + "img"] = batch["img"].numpy().copy()
        store[case + "has_target"] = np.array([[x is not None for x in row] for row in has])
        res = valid_step(dict(batch), ({"net": {"desc": net}}, None))["raw"]
        assert np.array_equal(res["img"], store[case + "img"])
        for h in heads:
            store[case + "pred/" + h] = np.asarray(res["pred"][h])
            store[case + "true/" + h] = np.asarray(res["true"][h])
            print(case, h, "pred", store[case + "pred/" + h].shape, store[case + "pred/" + h].dtype, "true", store[case + "true/" + h].shape)
    path = os.path.join(ROOT, "tests", "golden", "valid_step.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
