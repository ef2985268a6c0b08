"""cerb_head_loss against the values the reference's own train_step produced (tests/golden/train_loss.npz) and against the oracle."""
import copy
import os

import numpy as np
import pytest

from conftest import dev_switches
import torch

from cerberus_amd.losses import PARAMSET_LOSS, head_loss
from conftest import ROOT
from oracle import train_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "train_loss.npz"))


@pytest.mark.parametrize("case", ["paramset/", "typew1/", "wmap/"])
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
        wm = None
        if case == "wmap/" and "wmap/weight_map/" + h in gold.files:  # the head's "#WEIGHT-MAP" target of the reference's batch
            wm = torch.from_numpy(gold["wmap/weight_map/" + h][..., 0]).cuda()
        loss, dl = head_loss(h, lg, tgt, flag, opts, channels_last=channels_last, pixel_weight=wm)
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


def test_train_mode_forward_vs_reference_train_step(gold):
    """cerb_net_forward_train (BatchNorm with the batch's statistics, raw conv weights, the Patch-Class dropout mask of that step) against
    the logits the reference's network produced inside its own train_step (forward hooks, oracle/gen_golden_train_loss.py), and the
    losses computed from OUR logits against the losses train_step reported."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    kw = default_model_kwargs()
    m = create_model(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
    m.train()
    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    out = m.forward_train(tiles, keep)
    total = 0.0
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        ref = gold["logits/" + h]                      # NCHW
        got = out[h].cpu().numpy()
        got = got.reshape(ref.shape) if h == "Patch-Class" else got.transpose(0, 3, 1, 2)
        err = np.abs(got - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (h, err, np.abs(ref).max())
        lg = out[h].reshape(int(gold["N"]), -1, 1, 1) if h == "Patch-Class" else out[h]
        loss, _ = head_loss(h, lg, torch.from_numpy(gold["target/" + h][..., 0]).cuda(),
                            torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda(), channels_last=(h != "Patch-Class"))
        exp = float(gold["paramset/loss/" + h])
        assert abs(float(loss) - exp) <= 1e-4 * max(1.0, abs(exp)), (h, float(loss), exp)  # north_star's bar on the loss
        total += float(loss)
    assert abs(total - float(gold["paramset/overall_loss"])) <= 1e-4 * float(gold["paramset/overall_loss"])
    # an inference-packed network refuses the train-mode entry point and vice versa
    from cerberus_amd._lib import CerberusHipError

    with pytest.raises(CerberusHipError):
        m.infer_tiles(tiles, 64)
    e = create_model(**kw)
    e.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    e.infer_tiles(tiles, 64)
    with pytest.raises(CerberusHipError):
        e.train()


def test_backward_pass_vs_reference_train_step(gold):
    """cerb_net_train_grads: the gradient of EVERY parameter (309 tensors) after train-mode forward + losses + backward against what
    `all_loss.backward()` left in the reference's own train_step -- per tensor the sum, the absolute sum and three sampled elements
    (oracle/gen_golden_train_loss.py stores those statistics instead of 113 MB of gradients)."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    losses, grads = m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
    for h in targets:
        exp = float(gold["paramset/loss/" + h])
        assert abs(losses[h] - exp) <= 1e-4 * max(1.0, abs(exp)), (h, losses[h], exp)
    names = [str(x) for x in gold["step/param_names"]]
    stats = gold["step/grad_stats"]
    # fc exists but is never called: no gradient; under the BatchNorm buffers' keys train_grads returns the step's batch statistics
    assert set(n for n in names if not n.startswith("backbone.fc.")) == set(k for k in grads if not k.endswith(("running_mean", "running_var")))
    worst = 0.0
    for k, (s_sum, s_abs, e0, em, e1) in zip(names, stats):
        if k.startswith("backbone.fc."):
            assert s_abs == 0.0
            continue
        g = grads[k].double().flatten().cpu().numpy()
        # yardsticks: the tensor's absolute sum, with a floor for gradients that are mathematically zero -- the bias of a conv in front
        # of a BatchNorm -- where both sides hold rounding noise (1e-6 .. 1e-7 in the reference, less here)
        floor_ = 1e-3 * g.size ** 0.5
        ref_abs = max(s_abs, floor_)
        per_el = ref_abs / g.size
        err = max(abs(g.sum() - s_sum) / ref_abs, abs(np.abs(g).sum() - s_abs) / ref_abs, abs(g[0] - e0) / (50 * per_el + abs(e0)),
                  abs(g[g.size // 2] - em) / (50 * per_el + abs(em)), abs(g[-1] - e1) / (50 * per_el + abs(e1)))
        worst = max(worst, err)
        assert err < 2e-3, (k, err, g.sum(), s_sum, np.abs(g).sum(), s_abs, g[0], e0)
    # the reference's train_decoder_list quirk: the #TYPE decoders get gradients only inside their last block and head
    assert float(grads["decoder_head.Gland#TYPE.0.block.0.conv.weight"].abs().sum()) == 0.0
    assert float(grads["decoder_head.Gland#TYPE.3.block.0.conv.weight"].abs().sum()) > 0.0
    print("worst relative gradient-statistic error over %d tensors: %.2e" % (len(names), worst))
    # FULL tensors, element by element, for one layer or more of every backward kernel family (3x3 weight gradient downstream of the
    # Winograd data gradient, 3x3 stride 2, 1x1 stride 2, plain 1x1, the 7x7 stem, the heads' pointwise layers, BatchNorm gamma / beta,
    # conv biases, Patch-Class): a permutation or a sign error inside a tensor cannot hide behind matching sums here.
    # Errors are taken per OUTPUT CHANNEL relative to the tensor's largest element.  Two bars:
    #   * the 90th percentile over channels <= max(2e-3, 3 x the reference's own inter-backend fp32 noise on the tensor)
    #     (measured: 1e-6 .. 3e-3 -- decoder tensors sit at 2e-6, the backbone collects ~1e-3 of summation-order noise through ~40
    #     batch-normalised layers of a 3-sample batch);
    #   * every channel <= 2e-2 and cosine > 0.9999.  Isolated channels deviate by up to ~1.3e-2 where ONE pixel's pre-activation is
    #     within rounding of zero and the ReLU masks of two fp32 implementations differ on it (tests/tools/dev_grad_diff.py: all
    #     channels of decoder_head.Gland#TYPE.3.block.1.conv.weight at 2e-6 except channel 32 at 1.3e-2, identical under both conv
    #     algorithms; with other weights and data the outlier moves to another channel, the reference's own float64 replay agrees
    #     with its fp32 run to 7e-5 there).  A kernel defect shows up in many channels or at O(1).
    full_names = [str(x) for x in gold["step/grad_full_names"]]
    assert len(full_names) >= 30
    worst_p90 = worst_max = 0.0
    for k in full_names:
        ref = gold["step/grad_full/" + k].astype(np.float64)
        got = grads[k].double().cpu().numpy().reshape(ref.shape)
        scale = max(float(np.abs(ref).max()), 1e-30)
        if float(np.abs(ref).sum()) <= 1e-3 * ref.size ** 0.5:  # mathematically zero (bias in front of a BatchNorm): rounding noise both sides
            assert float(np.abs(got).max()) < 1e-4, k
            continue
        per_co = np.abs(got - ref).reshape(ref.shape[0], -1).max(axis=1) / scale
        p90, mx = float(np.percentile(per_co, 90)), float(per_co.max())
        cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
        bar = max(2e-3, 3.0 * float(gold["step/grad_full_noise/" + k]))
        worst_p90, worst_max = max(worst_p90, p90), max(worst_max, mx)
        # every channel: one flipped ReLU pixel (1.3e-2 in the REFERENCE's own fp32-vs-fp32 comparison, see above) with margin -- a bar from the
        # reference's behaviour, not from what this build happens to achieve (the achieved figures are printed below)
        assert p90 < bar and mx < 2e-2 and cos > 0.9999, (k, p90, bar, mx, cos)
    print("element-wise gradients over %d full tensors: worst 90th-percentile channel error %.2e, worst channel %.2e (of the tensor's largest element)"
          % (len(full_names), worst_p90, worst_max))


def test_whole_train_step_vs_reference(gold):
    """cerberus_amd.train.train_step (the reference's protocol: batch dict + run_info) for one step: reported losses, then the Adam update
    of every parameter and the BatchNorm running statistics against what optimizer.step() / the train-mode forward left in the reference
    model (per tensor: sum after the step, and the absolute sum / three sampled elements of the change).  Not compared: the biases of convs
    in front of a BatchNorm -- their gradient is mathematically zero, Adam turns the rounding noise both sides hold there into +-lr steps."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import Adam, train_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    m = create_model(**default_model_kwargs())
    sd0 = {k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}
    m.load_state_dict(sd0, strict=True)
    heads = [str(h) for h in gold["heads"]]
    has = np.full((int(gold["N"]), len(heads)), None, dtype=object)
    for j, h in enumerate(heads):
        for n in range(int(gold["N"])):
            if gold["has_target"][n, j]:
                has[n, j] = h
    batch = {"img": torch.from_numpy(gold["img"]), "dummy_target": has}
    for h in heads:
        batch[h] = torch.from_numpy(gold["target/" + h])
    opt = Adam(lr=1.0e-3, betas=(0.9, 0.999))
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    res = train_step(batch, ({"net": {"desc": m, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
    assert abs(res["EMA"]["overall_loss"] - float(gold["paramset/overall_loss"])) <= 1e-4 * float(gold["paramset/overall_loss"])
    new = m.state_dict()
    n_cmp = 0
    for k, (p_sum, d_abs, d0, dm, d1) in zip([str(x) for x in gold["step/param_names"]], gold["step/update_stats"]):
        if k.startswith("backbone.fc."):
            continue
        d = (new[k].double() - sd0[k].double()).flatten().numpy()
        if ".block." in k and k.endswith(".conv.bias") or k.endswith("Patch-Class.conv1.bias"):
            continue  # a bias in front of a BatchNorm (ConvBlock layers, Patch-Class conv1): zero gradient, noise-driven +-lr steps on both sides
        n_cmp += 1
        # first step of Adam: |update| = lr for every element whose gradient is not tiny; a sign flip of a near-zero gradient costs 2 lr
        flips = abs(np.abs(d).sum() - d_abs) / max(d_abs, 1e-12)
        assert flips < 2e-2, (k, np.abs(d).sum(), d_abs)
        # + 4.2e-3: two sign flips (2 lr each) in a small tensor -- between the F(2x2) and the F(4x4) convolutions of this package the
        # elements whose first step changes sign all have gradients below 2e-3 of the tensor's largest (tests/tools/dev_train_flip.py)
        assert abs(new[k].double().sum().item() - p_sum) <= 2e-2 * d_abs + 4.2e-3 + 1e-6 * abs(p_sum), (k, new[k].double().sum().item(), p_sum, d_abs)
    assert n_cmp > 250
    for k, (b_sum, b_dabs, b0) in zip([str(x) for x in gold["step/bn_names"]], gold["step/bn_stats"]):
        got = new[k].double()
        assert abs(got.sum().item() - b_sum) <= 1e-4 * max(1.0, abs(b_sum)) + 1e-3 * b_dabs, (k, got.sum().item(), b_sum)
        assert abs(got.flatten()[0].item() - b0) <= 1e-4 * max(1.0, abs(b0)), k
    # a second step runs on the re-packed weights
    res2 = train_step(batch, ({"net": {"desc": m, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
    assert np.isfinite(res2["EMA"]["overall_loss"]) and res2["EMA"]["overall_loss"] != res["EMA"]["overall_loss"]


def test_train_step_with_weight_maps_reports_the_reference_losses(gold):
    """The reference's batch protocol with "<head>#WEIGHT-MAP" entries (loader/targets.py, models/run_desc.py:111-117) through
    cerberus_amd.train.train_step: every reported loss against the reference's own train_step on the same batch."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import Adam, train_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
    heads = [str(h) for h in gold["heads"]]
    has = np.full(gold["has_target"].shape, None, dtype=object)
    for j, h in enumerate(heads):
        has[gold["has_target"][:, j], j] = h
    batch = {"img": torch.from_numpy(gold["img"]), "dummy_target": has}
    for h in heads:
        batch[h] = torch.from_numpy(gold["target/" + h])
        if "wmap/weight_map/" + h in gold.files:
            batch[h + "#WEIGHT-MAP"] = torch.from_numpy(gold["wmap/weight_map/" + h])
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    res = train_step(batch, ({"net": {"desc": m, "optimizer": Adam(lr=1.0e-3), "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
    for h in heads:
        exp = float(gold["wmap/loss/" + h])
        assert abs(res["EMA"][h + "_loss"] - exp) <= 1e-4 * max(1.0, abs(exp)), (h, res["EMA"][h + "_loss"], exp)
    assert abs(res["EMA"]["overall_loss"] - float(gold["wmap/overall_loss"])) <= 2e-4
    assert abs(float(gold["wmap/overall_loss"]) - float(gold["paramset/overall_loss"])) > 0.1  # the maps matter


@pytest.mark.parametrize("case", ["pc/", "nopc/"])
def test_valid_step_vs_reference(case):
    """cerberus_amd.train.valid_step against the reference's own valid_step (tests/golden/valid_step.npz, oracle/gen_golden_valid_step.py):
    per-head predictions (probabilities to 1e-4, class maps equal except where the two best logits tie within float noise) and the 'true'
    arrays with the reference's shapes, including the [N, H, H, W] one that F.interpolate makes of NHWC targets when Patch-Class is present."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import valid_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    g = np.load(os.path.join(ROOT, "tests", "golden", "valid_step.npz"))
    heads = [str(h) for h in g["heads"]]
    has = np.full(g[case + "has_target"].shape, None, dtype=object)
    for j, h in enumerate(heads):
        has[g[case + "has_target"][:, j], j] = h
    batch = {"img": torch.from_numpy(g[case + "img"]), "dummy_target": has}
    for h in heads:
        batch[h] = torch.from_numpy(g[case + "target/" + h])
    for mode in ("eval", "train"):  # a model in training mode validates through its inference twin
        m = create_model(**default_model_kwargs())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(g["weight_seed"])).items()}, strict=True)
        if mode == "train":
            m.train()
            m.forward_train(torch.from_numpy(g[case + "img"]).cuda())
        raw = valid_step(dict(batch), ({"net": {"desc": m}}, None))["raw"]
        assert np.array_equal(raw["img"], g[case + "img"]) and raw["dummy"] is not None
        assert list(raw["channel_info"].keys()) == list(m.decoder_info_list.keys())
        for h in heads:
            exp, got = g[case + "pred/" + h], raw["pred"][h]
            assert got.shape == exp.shape, (h, got.shape, exp.shape)
            if h.endswith("INST"):
                assert got.dtype == np.float32 and np.abs(got - exp).max() <= 1e-4, (h, np.abs(got - exp).max())
            else:
                assert (got != exp).mean() <= 1e-3, (h, (got != exp).mean())
            te, tg = g[case + "true/" + h], raw["true"][h]
            assert tg.shape == te.shape and np.array_equal(tg, te), (h, tg.shape, te.shape)


def test_device_side_parameter_update_equals_a_rebuilt_handle(gold):
    """Three optimiser steps with the handle updated device to device (cerb_net_update_params: every tensor the train-mode kernels read
    + the packing kernels) against the same steps with the handle thrown away and rebuilt from the host state dict after each one:
    identical losses at every step and an identical final state dict -- a tensor missed by the device path would leave stale weights."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import Adam, train_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    heads = [str(h) for h in gold["heads"]]
    has = np.full(gold["has_target"].shape, None, dtype=object)
    for j, h in enumerate(heads):
        has[gold["has_target"][:, j], j] = h
    batch = {"img": torch.from_numpy(gold["img"]), "dummy_target": has}
    for h in heads:
        batch[h] = torch.from_numpy(gold["target/" + h])
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    runs = []
    for rebuild in (False, True):
        m = create_model(**default_model_kwargs())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
        opt = Adam(lr=1.0e-3, betas=(0.9, 0.999))
        losses = []
        for _ in range(3):
            res = train_step(dict(batch), ({"net": {"desc": m, "optimizer": opt, "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
            losses.append([res["EMA"][h + "_loss"] for h in heads])
            if rebuild:
                m._sync_state_dict()
                m._release()  # the next step creates a fresh handle from the host state dict
        runs.append((losses, m.state_dict()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][0][0] != runs[0][0][2]  # the parameters did move
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k


def _gold_batch(g):
    heads = [str(h) for h in g["heads"]]
    has = np.full((int(g["N"]), len(heads)), None, dtype=object)
    for j, h in enumerate(heads):
        for n in range(int(g["N"])):
            if g["has_target"][n, j]:
                has[n, j] = h
    batch = {"img": torch.from_numpy(g["img"]), "dummy_target": has}
    for h in heads:
        batch[h] = torch.from_numpy(g["target/" + h])
    return heads, batch


def test_train_step_raw_payload_vs_reference_logits(gold):
    """train_step's 'raw' entry (models/run_desc.py:172-230): two random samples -- images, targets and the train-mode predictions read out per
    head -- against the same read-outs of the REFERENCE's train-mode logits (tests/golden/train_loss.npz), drawn with the same indices."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import Adam, train_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
    heads, batch = _gold_batch(gold)
    n, hw = int(gold["N"]), int(gold["H"])
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(n, 512)).cuda()
    torch.manual_seed(123)
    res = train_step(batch, ({"net": {"desc": m, "optimizer": Adam(lr=1.0e-3), "extra_info": {"loss": PARAMSET_LOSS}}}, None), dropout_keep=keep)
    torch.manual_seed(123)
    idx = torch.randint(0, n, (2,))
    raw = res["raw"]
    assert set(raw.keys()) == {"img", "true", "pred"} and raw["img"].dtype == np.uint8
    assert np.array_equal(raw["img"], gold["img"][idx.numpy()])
    assert list(raw["pred"].keys()) == heads and list(raw["true"].keys()) == heads
    for h in heads:
        lg = torch.from_numpy(gold["logits/" + h])[idx]  # NCHW from the reference
        if h == "Patch-Class":
            want = torch.argmax(lg.reshape(2, -1), dim=-1).reshape(2, 1, 1).expand(2, hw, hw).numpy()
            assert raw["pred"][h].shape == (2, hw, hw) and np.array_equal(raw["pred"][h], want.astype(raw["pred"][h].dtype))
            assert np.array_equal(raw["true"][h], np.broadcast_to(gold["target/" + h][idx.numpy()].reshape(2, 1, 1), (2, hw, hw)))
            continue
        sm = torch.softmax(lg.permute(0, 2, 3, 1), -1)
        if h.endswith("TYPE"):
            want, got = torch.argmax(sm, -1).numpy(), raw["pred"][h]
            top = torch.topk(sm, 2, dim=-1).values
            bad = got != want
            assert got.shape == (2, hw, hw) and (not bad.any() or float((top[..., 0] - top[..., 1]).numpy()[bad].max()) < 1e-4), h
        else:
            assert raw["pred"][h].shape == (2, hw, hw, 2) and np.abs(raw["pred"][h] - sm[..., 1:].numpy()).max() < 1e-4, h
        assert np.array_equal(raw["true"][h], gold["target/" + h][idx.numpy()][..., 0]), h


def test_subtype_fine_tune_step_vs_reference():
    """subtype_nuclei=True (the frozen-backbone sub-typing fine-tune, models/net_desc.py:105-142 + 160-170, run_desc.py:83-84) for one step,
    against the REFERENCE's own train_step in that configuration (oracle/gen_golden_subtype.py -> tests/golden/train_subtype.npz):
      * the train-mode forward normalises the frozen modules with their RUNNING statistics (eval-mode BatchNorm): reported losses agree;
      * Adam moves exactly the 14 tensors the reference moves -- the last level of the 'Nuclei#TYPE' decoder and its output head -- by the
        same amounts, every other parameter is bit-for-bit what it was;
      * running statistics change in the 9 BatchNorm layers of that decoder / head only (and num_batches_tracked advances there only)."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.train import Adam, train_step
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    g = np.load(os.path.join(ROOT, "tests", "golden", "train_subtype.npz"))
    kw = default_model_kwargs()
    kw["subtype_nuclei"] = True
    m = create_model(**kw)
    sd0 = {k: torch.from_numpy(v) for k, v in make_state_dict(int(g["weight_seed"])).items()}
    m.load_state_dict(sd0, strict=True)
    heads, batch = _gold_batch(g)
    loss = copy.deepcopy(PARAMSET_LOSS)
    loss["loss_info"]["Nuclei-TYPE"]["weight"] = 1.0
    keep = torch.from_numpy(g["dropout_mask"].reshape(int(g["N"]), 512)).cuda()
    opt = Adam(lr=1.0e-3, betas=(0.9, 0.999))
    res = train_step(batch, ({"net": {"desc": m, "optimizer": opt, "extra_info": {"loss": loss}}}, None), dropout_keep=keep)
    for h in heads:
        want = float(g["loss/" + h])
        assert abs(res["EMA"]["%s_loss" % h] - want) <= 2e-4 * max(1.0, abs(want)), (h, res["EMA"]["%s_loss" % h], want)
    assert abs(res["EMA"]["overall_loss"] - float(g["overall_loss"])) <= 1e-4 * float(g["overall_loss"])
    new = m.state_dict()
    moved_ref = set(str(k) for k in g["moved"])
    assert len(moved_ref) == 14
    moved = set()
    for k, (p_sum, d_abs, d0, dm, d1) in zip([str(x) for x in g["param_names"]], g["update_stats"]):
        if k.startswith("backbone.fc."):
            continue
        d = (new[k].double() - sd0[k].double()).flatten().numpy()
        if np.abs(d).sum() > 0:
            moved.add(k)
        if k not in moved_ref:
            assert np.abs(d).sum() == 0.0, "frozen parameter moved: " + k
            continue
        if ".block." in k and k.endswith(".conv.bias"):
            continue  # a bias in front of a BatchNorm: zero gradient, noise-driven +-lr steps on both sides
        assert abs(np.abs(d).sum() - d_abs) / max(d_abs, 1e-12) < 2e-2, (k, np.abs(d).sum(), d_abs)
        assert abs(new[k].double().sum().item() - p_sum) <= 2e-2 * d_abs + 4.2e-3 + 1e-6 * abs(p_sum), k
    assert moved - moved_ref == set() and all(k in moved or k.endswith(".conv.bias") for k in moved_ref), (sorted(moved ^ moved_ref))
    n_changed = 0
    for k, (b_sum, b_dabs, b0) in zip([str(x) for x in g["bn_names"]], g["bn_stats"]):
        got = new[k].double()
        if b_dabs == 0:
            assert torch.equal(new[k], sd0[k]), "running statistics of a frozen BatchNorm moved: " + k
            continue
        n_changed += 1
        assert abs(got.sum().item() - b_sum) <= 1e-4 * max(1.0, abs(b_sum)) + 1e-3 * b_dabs, (k, got.sum().item(), b_sum)
        assert abs(got.flatten()[0].item() - b0) <= 1e-4 * max(1.0, abs(b0)), k
    assert n_changed == 18
    tracked = set(str(k) for k in g["tracked_moved"])
    for k, v in new.items():
        if k.endswith("num_batches_tracked"):
            assert (int(v) != int(sd0[k])) == (k in tracked), k


def test_data_gradients_upstream_of_an_eval_mode_batchnorm():
    """a BatchNorm put in eval mode by cerb_net_set_bn_eval normalises with CONSTANTS, so its backward is dy = dz * gamma * rstd
    with no batch-mean / xhat-projection terms.  Round 3 ran the batch-statistics backward there (train.py only hid it by deleting the frozen
    gradients).  Here the frozen configuration (subtype_nuclei) keeps every gradient the tape produces and compares the ones UPSTREAM of
    frozen BatchNorm layers -- backbone, conv_map, an INST decoder, the frozen BatchNorm's own gamma / beta -- with torch autograd over the oracle
    network run the same way (eval-mode BatchNorm under the frozen prefixes, the #TYPE decoders with block-local gradients, same dropout)."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict
    from oracle import net_ref

    kw = default_model_kwargs()
    kw["subtype_nuclei"] = True
    m = create_model(**kw)
    sd_np = make_state_dict(0)
    rs = np.random.RandomState(7)
    for k, v in sd_np.items():  # running statistics away from (0, 1), so that eval mode differs visibly from batch statistics
        if k.endswith("running_mean"):
            sd_np[k] = (v + 0.05 * rs.randn(*v.shape)).astype(np.float32)
        elif k.endswith("running_var"):
            sd_np[k] = (v * (0.8 + 0.4 * rs.rand(*v.shape))).astype(np.float32)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    n, hw = 2, 96
    tiles = rs.randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    heads = {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}
    targets, flags, tnp = {}, {}, {}
    for h, c in heads.items():
        t = rs.randint(0, c, (n,) if h == "Patch-Class" else (n, hw, hw)).astype(np.float32)
        tnp[h] = t.reshape(n, 1, 1, 1) if h == "Patch-Class" else t[..., None]
        targets[h] = torch.from_numpy(t).cuda()
        flags[h] = torch.ones(n).cuda()
    loss = copy.deepcopy(PARAMSET_LOSS)
    loss["loss_info"]["Nuclei-TYPE"]["weight"] = 1.0
    keep = torch.from_numpy(rs.rand(n, 512) >= 0.3)
    losses, grads = m.train_grads(torch.from_numpy(tiles).cuda(), targets, flags, loss, keep.cuda())
    # the oracle under autograd
    params = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v))
        if t.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) and not k.startswith("backbone.fc"):
            t.requires_grad_(True)
        params[k] = t
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    lg = net_ref.net_forward(params, x, kw["decoder_kwargs"], kw["considered_tasks"], training=True, eval_bn_prefixes=m.frozen_prefixes(),
                             block_local_grads=("Nuclei#TYPE", "Gland#TYPE"), dropout_scale=keep.float() / 0.7)
    total = 0
    for h in heads:
        one = train_ref.head_loss_tensor(h, lg[h], tnp[h], np.ones(n, bool), loss, n_classes=heads[h])
        assert abs(float(one) - losses[h]) <= 2e-4 * max(1.0, abs(float(one))), (h, float(one), losses[h])
        total = total + one
    total.backward()
    checked = 0
    for k in ("backbone.conv1.weight", "backbone.layer1.1.conv2.weight", "backbone.layer2.0.downsample.0.weight", "backbone.layer3.2.conv1.weight",
              "backbone.layer4.2.conv2.weight", "backbone.layer3.4.bn2.weight", "backbone.layer2.1.bn1.bias", "conv_map.weight",
              "decoder_head.Lumen.1.block.0.conv.weight", "decoder_head.Gland.3.block.1.conv.weight", "decoder_head.Gland.3.block.1.bn.weight",
              "output_head.Nuclei.INST.x.0.block.0.conv.weight", "decoder_head.Nuclei#TYPE.3.block.1.conv.weight"):
        ref = params[k].grad.double().numpy()
        got = grads[k].double().cpu().numpy().reshape(ref.shape)
        scale = float(np.abs(ref).max())
        assert scale > 0, k
        cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        err = float(np.abs(got - ref).max()) / scale
        assert cos > 0.9999 and err < 2e-2, (k, cos, err)
        checked += 1
    assert checked == 13


@pytest.mark.gpu
@dev_switches
def test_batchnorm_statistics_from_the_conv_output_stage_equal_the_separate_pass(gold):
    """Training forward: the F(4x4) convolutions (conv_wino4 / conv_wino4b STATS instantiations) and the heads' 64 -> 96 pointwise layer leave
    per-block (sum, sum of squares) partials from their own output stage and the BatchNorm behind them finalises its batch statistics from those;
    CERB_BN_STATS_PASS=1 keeps the separate statistics pass over the layer's output.  Both ways every published batch mean / unbiased variance
    (what the running statistics are updated with) must agree to float rounding of sums taken in a different order, and the losses and
    gradients -- everything downstream of the normalisation -- with them."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    out = {}
    for mode in ("fused", "pass"):
        if mode == "pass":
            os.environ["CERB_BN_STATS_PASS"] = "1"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            losses, grads = m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
            out[mode] = (losses, {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v)) for k, v in grads.items()})
        finally:
            os.environ.pop("CERB_BN_STATS_PASS", None)
    (la, ga), (lb, gb) = out["fused"], out["pass"]
    stats = [k for k in ga if k.endswith(("running_mean", "running_var"))]
    assert len(stats) >= 2 * 51 and set(ga) == set(gb)
    for k in stats:
        a, b = ga[k].astype(np.float64), gb[k].astype(np.float64)
        scale = max(float(np.abs(b).max()), 1e-6)
        # observed: 2.1e-6 of a layer's largest value right behind a convolution, 1.7e-5 at the Patch-Class branch (statistics over the batch's
        # few rows, downstream of 36 re-ordered sums)
        assert float(np.abs(a - b).max()) <= 1e-4 * scale + 1e-9, (k, float(np.abs(a - b).max()), scale)
    for h in la:
        assert abs(la[h] - lb[h]) <= 1e-5 * max(1.0, abs(lb[h])), (h, la[h], lb[h])
    for k in ga:
        if k in stats:
            continue
        a, b = ga[k].astype(np.float64).ravel(), gb[k].astype(np.float64).ravel()
        den = float(np.linalg.norm(a) * np.linalg.norm(b))
        if float(np.abs(b).max()) < 1e-7:  # a bias in front of a train-mode BatchNorm has no gradient: what is there (1e-11) is rounding noise
            continue
        if den > 0:
            assert float(a @ b) / den > 0.9999, k  # (observed worst: 0.99999 -- re-ordered sums flip a few ReLU masks at values next to zero; the bar of the reference comparison)


@pytest.mark.gpu
@dev_switches
def test_fused_output_heads_equal_the_separate_passes(gold):
    """Round 5: an output head's train-mode chain (1x1 64->96 -> BatchNorm -> ReLU -> 1x1 96->out) stores only its hidden map and reads it three times
    (csrc/head_train.hip: forward 2, backward 1, backward 2) instead of running seven separate passes; CERB_HEAD_UNFUSED=1 keeps round 4's passes.
    Losses, logits and EVERY published gradient / batch statistic must agree to the rounding of re-ordered fp32 sums."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    out = {}
    for mode in ("fused", "separate", "fused_own_bn_passes"):
        if mode == "separate":
            os.environ["CERB_HEAD_UNFUSED"] = "1"
        if mode == "fused_own_bn_passes":  # fused heads, but the last decoder BatchNorm applied / reduced by its own passes (no deferral to the heads' loads)
            os.environ["CERB_HEAD_BN_APPLY_PASS"] = "1"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            logits = {}
            losses, grads = m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep, logits_out=logits)
            m.profile(True)
            m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
            fams = set(r[1] for r in m.profile_records())
            m.profile(False)
            out[mode] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in grads.items()}, {k: v.detach().cpu().numpy().copy() for k, v in logits.items()}, fams)
        finally:
            os.environ.pop("CERB_HEAD_UNFUSED", None)
            os.environ.pop("CERB_HEAD_BN_APPLY_PASS", None)
    # the deferred BatchNorm (normalised on the heads' loads, backward sums from head_bwd2's epilogue) against the same fused heads behind its own passes
    (lc, gc, zc, fc) = out["fused_own_bn_passes"]
    for k in out["fused"][1]:
        a, c = out["fused"][1][k].astype(np.float64).ravel(), gc[k].astype(np.float64).ravel()
        if float(np.abs(c).max()) < 1e-7:
            continue
        assert float(np.abs(a - c).max()) / float(np.abs(c).max()) < (2e-4 if k.startswith(("output_head.", "decoder_head.")) else 1e-2), k
    (la, ga, za, fa), (lb, gb, zb, fb) = out["fused"], out["separate"]
    assert {"head_fwd2", "head_bwd1", "head_bwd2"} <= fa and not ({"head_fwd2", "head_bwd1", "head_bwd2"} & fb), (fa, fb)  # the A/B really ran both ways
    assert set(ga) == set(gb) and set(za) == set(zb)
    for h in la:
        assert abs(la[h] - lb[h]) <= 1e-5 * max(1.0, abs(lb[h])), (h, la[h], lb[h])
    for k in za:
        assert float(np.abs(za[k] - zb[k]).max()) <= 2e-5 * max(1.0, float(np.abs(zb[k]).max())), k
    worst = 0.0
    for k in ga:
        a, b = ga[k].astype(np.float64).ravel(), gb[k].astype(np.float64).ravel()
        if float(np.abs(b).max()) < 1e-7:  # a bias in front of a train-mode BatchNorm: rounding noise both ways
            assert float(np.abs(a).max()) < 1e-5, k
            continue
        err = float(np.abs(a - b).max()) / float(np.abs(b).max())
        cos = float(a @ b) / float(np.linalg.norm(a) * np.linalg.norm(b))
        worst = max(worst, err)
        # head tensors: re-ordered sums only (1e-5); upstream tensors additionally see the few ReLU masks that flip next to zero when the
        # decoder gradient moves in its last bits (the bar of the reference comparison)
        assert cos > 0.9999 and err < (2e-4 if k.startswith("output_head.") else 1e-2), (k, err, cos)
    print("fused vs separate heads: worst element error %.2e of a tensor's largest value over %d tensors" % (worst, len(ga)))


@pytest.mark.gpu
@dev_switches
def test_winograd_domain_weight_gradients_equal_the_direct_kernel(gold):
    """Round 5: the weight gradients of the 3x3 stride-1 convolutions are accumulated in the Winograd domain (csrc/conv_wgrad_wino.hip:
    dU = sum_tiles (A dY A^T) .* (B^T d B), dg = G^T dU G -- a quarter of the direct form's matrix instructions); CERB_WGRAD_DIRECT=1 keeps round 4's
    direct split-K kernel.  Same data gradients either way, so every OTHER tensor must be bit-identical and every 3x3 weight / conv-bias gradient
    must agree with the direct kernel to fp32 rounding of a differently ordered sum (F(4x4) transforms included)."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    out = {}
    for mode in ("wino", "direct"):
        if mode == "direct":
            os.environ["CERB_WGRAD_DIRECT"] = "1"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            losses, grads = m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
            m.profile(True)
            m.train_grads(tiles, targets, flags, PARAMSET_LOSS, keep)
            fams = [r[1] for r in m.profile_records()]
            m.profile(False)
            out[mode] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in grads.items()}, fams)
        finally:
            os.environ.pop("CERB_WGRAD_DIRECT", None)
    (la, ga, fa), (lb, gb, fb) = out["wino"], out["direct"]
    n_wino = sum(f.startswith("wgrad_wino4") for f in fa)
    assert n_wino >= 20 and not any(f.startswith("wgrad_wino4") for f in fb), (n_wino, sorted(set(fb)))
    assert la == lb and set(ga) == set(gb)
    worst, n3 = 0.0, 0
    for k in ga:
        a, b = ga[k].astype(np.float64), gb[k].astype(np.float64)
        conv3 = a.ndim == 4 and a.shape[2:] == (3, 3)
        if not conv3 and not k.endswith("conv.bias"):
            assert np.array_equal(a, b), k  # nothing but the weight-gradient kernel changed
            continue
        if float(np.abs(b).max()) < 1e-7:  # a bias in front of a train-mode BatchNorm: rounding noise both ways
            assert float(np.abs(a).max()) < 1e-5, k
            continue
        err = float(np.abs(a - b).max()) / float(np.abs(b).max())
        worst = max(worst, err)
        n3 += conv3
        assert err < 2e-4, (k, err)
    assert n3 >= 50
    print("Winograd-domain vs direct weight gradients: worst element error %.2e of a tensor's largest value over %d 3x3 tensors" % (worst, n3))


@pytest.mark.gpu
def test_training_step_with_packed_items_equals_block_items():
    """Round 5: the training step's forward and data-gradient convolutions on the 56^2 / 28^2 maps of a 448-pixel batch take packed work items
    (conv_wino4b.hip: 16 consecutive tiles, no padding tiles); cerb_net_set_packed_items(0) keeps the 16 x 16 blocks.  Convolution outputs and data
    gradients are bitwise the same either way; the BatchNorm partial sums behind the convolutions are grouped by item, so batch statistics, losses and
    gradients may move by the rounding of re-ordered fp32 sums and no more."""
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    n, win = 3, 448
    g = torch.Generator(device="cuda").manual_seed(77)
    heads = {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}
    batch = {"tiles": torch.randint(0, 256, (n, win, win, 3), dtype=torch.uint8, device="cuda", generator=g), "targets": {}, "flags": {},
             "keep": torch.rand((n, 512), device="cuda", generator=g) < 0.7}
    for h, c in heads.items():
        if h == "Patch-Class":
            batch["targets"][h] = torch.randint(0, c, (n,), device="cuda", generator=g).float()
        else:
            fg = torch.rand((n, win, win), device="cuda", generator=g) < 0.3
            batch["targets"][h] = (fg * torch.randint(1, c, (n, win, win), device="cuda", generator=g)).float()
        batch["flags"][h] = torch.ones(n, device="cuda")
    out = {}
    for mode in ("packed", "blocks"):
        m = create_model(**default_model_kwargs())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(5).items()}, strict=True)
        m.train()  # the handle is created by the first switch: it has to be packed for training
        m.set_packed_items(mode == "packed")
        losses, grads = m.train_grads(batch["tiles"], batch["targets"], batch["flags"], PARAMSET_LOSS, batch["keep"])
        m.profile(True)
        m.train_grads(batch["tiles"], batch["targets"], batch["flags"], PARAMSET_LOSS, batch["keep"])
        fams = [r[1] for r in m.profile_records()]
        m.profile(False)
        out[mode] = (losses, {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v)) for k, v in grads.items()}, fams)
    (la, ga, fa), (lb, gb, fb) = out["packed"], out["blocks"]
    n_fwd = sum(f.startswith("conv_wino4b<f4x4,16t") for f in fa), sum(f.startswith("dgrad:conv_wino4b<f4x4,16t") for f in fa)
    assert n_fwd[0] >= 16 and n_fwd[1] >= 16 and not any(",16t" in f for f in fb), (n_fwd, sorted(set(fa)))
    assert set(ga) == set(gb)
    for h in la:
        assert abs(la[h] - lb[h]) <= 1e-5 * max(1.0, abs(lb[h])), (h, la[h], lb[h])
    for k in ga:
        a, b = ga[k].astype(np.float64).ravel(), gb[k].astype(np.float64).ravel()
        if k.endswith(("running_mean", "running_var")):
            assert float(np.abs(a - b).max()) <= 1e-4 * max(float(np.abs(b).max()), 1e-6) + 1e-9, k
            continue
        if float(np.abs(b).max()) < 1e-6:  # a bias in front of a train-mode BatchNorm has no gradient: what is there (1e-9 at this batch size) is rounding noise
            continue
        den = float(np.linalg.norm(a) * np.linalg.norm(b))
        if den > 0:
            assert float(a @ b) / den > 0.9999, k


@pytest.mark.gpu
@dev_switches
def test_maxpool_backward_by_recorded_positions_equals_the_scan(gold):
    """Round 5: the training forward's max-pool records which window position held each first maximum (one byte per element) and the backward pass routes
    the gradients by it; CERB_MAXPOOL_SCAN=1 keeps round 4's backward, which re-finds the first maxima from the stem's output and the pooled map
    (torch.nn.functional.max_pool2d's rule, held to the reference's train_step by the tests above).  Same rule, same additions: EVERY gradient, batch
    statistic and loss must be bit-identical -- also on a batch whose maps are not multiples of the window stride's pattern (66 x 66: odd pooled rows)."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    # a second batch with many ties: tiles of a few grey levels only (equal maxima inside a window are where the two rules could differ)
    g = torch.Generator(device="cuda").manual_seed(5)
    flat = (torch.randint(0, 3, tuple(tiles.shape), device="cuda", generator=g) * 100).to(torch.uint8)
    out = {}
    for mode in ("positions", "scan"):
        if mode == "scan":
            os.environ["CERB_MAXPOOL_SCAN"] = "1"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            res = []
            for tl in (tiles, flat):
                losses, grads = m.train_grads(tl, targets, flags, PARAMSET_LOSS, keep)
                res.append((losses, {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v)) for k, v in grads.items()}))
            out[mode] = res
        finally:
            os.environ.pop("CERB_MAXPOOL_SCAN", None)
    for (la, ga), (lb, gb) in zip(out["positions"], out["scan"]):
        assert la == lb and set(ga) == set(gb)
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), k


@pytest.mark.gpu
@dev_switches
def test_weight_gradients_on_the_side_stream_equal_the_single_stream_step(gold):
    """Round 5: the backward pass queues the weight gradients of the 3x3 / 1x1 convolutions on a side stream of the handle (forked when a layer's output gradient is
    final, joined at the end of cerb_net_train_grads) so that they overlap the BatchNorm backward passes; CERB_WGRAD_SIDE=0 keeps everything on the caller's stream.
    Same kernels on the same data: every gradient, statistic and loss must be BITWISE equal, step after step (the second step re-uses the tape arena and the side
    stream's workspace while nothing may still be in flight), on the 64-pixel fixture batch and on a 448-pixel batch."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    n2, win = 2, 448
    big = {"tiles": torch.randint(0, 256, (n2, win, win, 3), dtype=torch.uint8, device="cuda", generator=g), "targets": {}, "flags": {},
           "keep": torch.rand((n2, 512), device="cuda", generator=g) < 0.7}
    for h, c in {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}.items():
        big["targets"][h] = torch.randint(0, c, (n2,), device="cuda", generator=g).float() if h == "Patch-Class" else \
            ((torch.rand((n2, win, win), device="cuda", generator=g) < 0.3) * torch.randint(1, c, (n2, win, win), device="cuda", generator=g)).float()
        big["flags"][h] = torch.ones(n2, device="cuda")
    out = {}
    for mode in ("side", "single"):
        if mode == "single":
            os.environ["CERB_WGRAD_SIDE"] = "0"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            res = []
            for rep in range(2):
                for (tl, tg, fl, kp) in ((tiles, targets, flags, keep), (big["tiles"], big["targets"], big["flags"], big["keep"])):
                    losses, grads = m.train_grads(tl, tg, fl, PARAMSET_LOSS, kp)
                    res.append((losses, {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v)) for k, v in grads.items()}))
            out[mode] = res
        finally:
            os.environ.pop("CERB_WGRAD_SIDE", None)
    assert len(out["side"]) == 4
    for (la, ga), (lb, gb) in zip(out["side"], out["single"]):
        assert la == lb and set(ga) == set(gb)
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), k


@pytest.mark.gpu
@dev_switches
def test_batchnorm_backward_sums_from_the_data_gradient_equal_the_reduction_pass(gold):
    """Round 5: where a data gradient (conv_wino4 / conv_wino4b on rotated weights) is the only writer of the gradient behind a train-mode BatchNorm + ReLU -- the first
    BatchNorm of every BasicBlock and of every decoder level -- its output stage reads the BatchNorm's input at its own pixels and leaves that BatchNorm's backward sums
    (sum dz', sum dz' xhat per block, STATS 2); CERB_BN_BWD_PASS1=1 keeps the BatchNorm's own reduction pass over dz and y.  Same masks (one bn_out expression), the sums
    taken in another order and in float per 256-pixel block instead of double: losses identical, every gradient within the rounding of re-ordered sums, on the 64-pixel
    fixture batch (block form with partial blocks) and on a 448-pixel batch (packed items)."""
    import os

    from cerberus_amd.net_desc import create_model
    from cerberus_amd.weights import default_model_kwargs, make_state_dict

    tiles = torch.from_numpy(gold["img"]).cuda()
    keep = torch.from_numpy(gold["step/dropout_mask"].reshape(int(gold["N"]), 512)).cuda()
    targets, flags = {}, {}
    for j, h in enumerate(gold["heads"]):
        h = str(h)
        t = gold["target/" + h][..., 0]
        targets[h] = torch.from_numpy(t.reshape(t.shape[0]) if h == "Patch-Class" else t).cuda()
        flags[h] = torch.from_numpy(gold["has_target"][:, j].astype(np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(13)
    n2, win = 2, 448
    big = {"tiles": torch.randint(0, 256, (n2, win, win, 3), dtype=torch.uint8, device="cuda", generator=g), "targets": {}, "flags": {},
           "keep": torch.rand((n2, 512), device="cuda", generator=g) < 0.7}
    for h, c in {"Lumen-INST": 3, "Gland-INST": 3, "Nuclei-INST": 3, "Nuclei-TYPE": 7, "Gland-TYPE": 3, "Patch-Class": 9}.items():
        big["targets"][h] = torch.randint(0, c, (n2,), device="cuda", generator=g).float() if h == "Patch-Class" else \
            ((torch.rand((n2, win, win), device="cuda", generator=g) < 0.3) * torch.randint(1, c, (n2, win, win), device="cuda", generator=g)).float()
        big["flags"][h] = torch.ones(n2, device="cuda")
    out = {}
    for mode in ("fused", "pass"):
        if mode == "pass":
            os.environ["CERB_BN_BWD_PASS1"] = "1"
        try:
            m = create_model(**default_model_kwargs())
            m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(gold["weight_seed"])).items()}, strict=True)
            res = []
            for (tl, tg, fl, kp) in ((tiles, targets, flags, keep), (big["tiles"], big["targets"], big["flags"], big["keep"])):
                losses, grads = m.train_grads(tl, tg, fl, PARAMSET_LOSS, kp)
                m.profile(True)
                m.train_grads(tl, tg, fl, PARAMSET_LOSS, kp)
                recs = m.profile_records()
                m.profile(False)
                res.append((losses, {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v)) for k, v in grads.items()},
                            sum(r[2] for r in recs if r[1] == "bn_bwd")))
            out[mode] = res
        finally:
            os.environ.pop("CERB_BN_BWD_PASS1", None)
    for (la, ga, ba), (lb, gb, bb) in zip(out["fused"], out["pass"]):
        assert la == lb and set(ga) == set(gb)  # the forward pass is the same code
        assert ba < 0.93 * bb, (ba, bb)         # the BatchNorm backward family moves fewer bytes: the fused path is the one that ran
        for k in ga:
            a, b = ga[k].astype(np.float64).ravel(), gb[k].astype(np.float64).ravel()
            if k.endswith(("running_mean", "running_var")):
                assert np.array_equal(a, b), k
                continue
            if float(np.abs(b).max()) < 1e-6:
                continue
            den = float(np.linalg.norm(a) * np.linalg.norm(b))
            assert den > 0 and float(a @ b) / den > 0.99999, k
            assert float(np.abs(a - b).max()) <= 2e-4 * float(np.abs(b).max()), (k, float(np.abs(a - b).max()), float(np.abs(b).max()))
