"""cerb_head_loss against the values the reference's own train_step produced (tests/golden/train_loss.npz) and against the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from cerberus_amd.losses import PARAMSET_LOSS, head_loss
from conftest import ROOT
from oracle import train_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "train_loss.npz"))


@pytest.mark.parametrize("case", ["paramset/", "typew1/"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_head_loss_and_gradient_vs_reference_train_step(gold, case, channels_last):
    opts = copy.deepcopy(PARAMSET_LOSS)
    if case == "typew1/":
        opts["loss_info"]["Nuclei-TYPE"]["weight"] = 1.0
    total = 0.0
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        lg = torch.from_numpy(gold["logits/" + h]).cuda()
        if channels_last:
            lg = lg.permute(0, 2, 3, 1).contiguous()
        tgt = torch.from_numpy(gold["target/" + h][..., 0]).cuda()
        flag = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
        loss, dl = head_loss(h, lg, tgt, flag, opts, channels_last=channels_last)
        exp = float(gold[case + "loss/" + h])
        assert abs(float(loss) - exp) <= 1e-4 * max(1.0, abs(exp)), (h, float(loss), exp)  # the bar of north_star: 1e-4
        assert abs(float(loss) - exp) <= 5e-6 * max(1.0, abs(exp)), (h, float(loss), exp)  # what it actually reaches
        gkey = case + "dlogits/" + h if case + "dlogits/" + h in gold.files else "paramset/dlogits/" + h
        g = gold[gkey]
        got = dl.permute(0, 3, 1, 2).cpu().numpy() if channels_last else dl.cpu().numpy()
        assert np.abs(got - g).max() <= 2e-5 * max(1e-3, np.abs(g).max()) + 1e-9, (h, np.abs(got - g).max(), np.abs(g).max())
        total += float(loss)
    assert abs(total - float(gold[case + "overall_loss"])) < 1e-4


def test_head_loss_is_bitwise_reproducible_and_matches_the_oracle_on_larger_maps():
    rs = np.random.RandomState(3)
    n, c, hw = 4, 7, 200
    lg = (rs.randn(n, c, hw, hw) * 2).astype(np.float32)
    tgt = ((rs.rand(n, hw, hw) < 0.4) * rs.randint(1, c, (n, hw, hw))).astype(np.float32)
    flag = np.array([1, 0, 1, 1], np.float32)
    opts = copy.deepcopy(PARAMSET_LOSS)
    opts["loss_info"]["Nuclei-TYPE"]["weight"] = 0.7
    a = head_loss("Nuclei-TYPE", torch.from_numpy(lg).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(flag).cuda(), opts)
    b = head_loss("Nuclei-TYPE", torch.from_numpy(lg).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(flag).cuda(), opts)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    exp, g = train_ref.head_loss("Nuclei-TYPE", lg, tgt[..., None], flag, opts)
    assert abs(float(a[0]) - exp) <= 1e-5 * abs(exp)
    assert np.abs(a[1].cpu().numpy() - g).max() <= 2e-5 * np.abs(g).max()
