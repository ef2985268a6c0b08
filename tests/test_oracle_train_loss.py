"""The training-loss oracle against the values the reference's own train_step produced (tests/golden/train_loss.npz,
oracle/gen_golden_train_loss.py): per-head loss as reported, d(overall loss)/d(logits), for paramset.yml's weights and with the
Nuclei-TYPE head switched on.  (BASELINE configs[4] groundwork: the loss is the first pinned piece of the training step.)"""
import copy
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import train_ref


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "train_loss.npz"))


@pytest.mark.parametrize("case", ["paramset/", "typew1/", "wmap/"])
def test_head_losses_and_logit_gradients_match_the_reference_train_step(gold, case):
    opts = copy.deepcopy(train_ref.PARAMSET_LOSS)
    if case == "typew1/":
        opts["loss_info"]["Nuclei-TYPE"]["weight"] = 1.0
    total = 0.0
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        assert gold[case + "loss_weight"][j] == opts["loss_info"][h]["weight"]
        wm = gold["wmap/weight_map/" + h] if case == "wmap/" and "wmap/weight_map/" + h in gold.files else None  # "<head>#WEIGHT-MAP" targets
        loss, grad = train_ref.head_loss(h, gold["logits/" + h], gold["target/" + h], gold["has_target"][:, j], opts, weight_map_nhw1=wm)
        exp = float(gold[case + "loss/" + h])
        assert abs(loss - exp) <= 2e-6 * max(1.0, abs(exp)), (h, loss, exp)
        gkey = case + "dlogits/" + h if case + "dlogits/" + h in gold.files else "paramset/dlogits/" + h
        g = gold[gkey]
        assert grad.shape == g.shape and np.abs(grad - g).max() <= 1e-6 * max(1e-3, np.abs(g).max()) + 1e-9, h
        total += loss
    assert abs(total - float(gold[case + "overall_loss"])) < 1e-4
