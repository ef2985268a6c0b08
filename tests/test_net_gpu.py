"""Parity of the HIP network path (through the C ABI) with the CPU oracle and the reference's golden vectors.
Tolerance: 1e-4 absolute on float probability maps (BASELINE.json north_star); integer maps must agree except where
the oracle's own top-2 softmax margin is below 1e-5 (fp32 rounding-order ties)."""
import os

import numpy as np
import pytest
import torch

from cerberus_amd.net_desc import create_model
from cerberus_amd.run_desc import infer_step
from cerberus_amd.weights import default_model_kwargs, make_state_dict, reference_init_state_dict
from oracle import net_ref

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-4
DEFAULT_ALGO = 6  # cerb_net_set_conv_algo: F(4x4,3x3) for maps >= 16 x 16 (conv_wino4b up to 64 x 64, conv_wino4 above), F(2x2,3x3) below
CROPS = [(0, 0), (96, 96), (192, 192)]
CS = 64


def _crops(a):
    return np.stack([a[:, y:y + CS, x:x + CS] for (y, x) in CROPS], axis=1)


def _model(tasks, seed=0, family="seeded", head_scale=None, decoder_kwargs=None):
    """family "seeded": cerberus_amd.weights.make_state_dict; "refinit": the reference's default initialisation (weights_init_cnn,
    models/net_desc.py:89-103) from a seeded generator; "scaled": the seeded recipe with every dense head's last 1x1 multiplied by the
    fixture's float32 factors (calibration logits at 30 / 80) -- exactly what oracle/gen_golden_net.py loaded into the reference."""
    kw = default_model_kwargs(tasks)
    if decoder_kwargs is not None:
        from collections import OrderedDict

        kw["decoder_kwargs"] = OrderedDict((k, OrderedDict(v)) for k, v in decoder_kwargs)
    if family == "refinit":
        sd_np = reference_init_state_dict(kw["decoder_kwargs"], kw["considered_tasks"], generator=torch.Generator().manual_seed(seed))
    else:
        sd_np = make_state_dict(seed, kw["decoder_kwargs"], kw["considered_tasks"], head_logit_scale=head_scale)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    m = create_model(**kw)
    m.load_state_dict(sd, strict=True)
    return m, sd, kw


def _golden_model(g):
    tasks = [str(t) for t in g["tasks"]]
    fam = str(g["weight_family"]) if "weight_family" in g else "seeded"
    scale = None
    if fam == "scaled":
        scale = {str(k): np.float32(v) for k, v in zip(g["head_scale_names"], g["head_scale_values"])}
    dk = None
    if "decoder_kwargs_json" in g:  # another head layout than paramset.yml's (several heads over one decoder)
        import json

        dk = [(k, [tuple(h) for h in v]) for k, v in json.loads(str(g["decoder_kwargs_json"]))]
    m, sd, kw = _model(tasks, int(g["weight_seed"]), fam, scale, dk)
    if "head_name_list" in g:
        tasks = [str(t) for t in g["head_name_list"]]  # what infer_step is called with (models/run_desc.py:475-476)
    from cerberus_amd.weights import state_dict_sha256
    assert state_dict_sha256({k: v.numpy() for k, v in sd.items()}) == str(g["weights_sha256"]), "the fixture's weights were not rebuilt bit for bit"
    return m, sd, kw, tasks


def _golden_tiles(g):
    """The fixture's input tiles, rebuilt from its seed: uniform noise (rounds 1-5) or the structured set of cerberus_amd.synth_tiles (stain field,
    half glass, all white, all black), checked against the sha256 the generator stored."""
    n, hw = int(g["n"]), int(g["hw"])
    if "tiles_kind" in g and str(g["tiles_kind"]) == "structured":
        from cerberus_amd.synth_tiles import structured_tiles

        tiles = structured_tiles(hw, int(g["tile_seed"]))
    else:
        tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    if "tiles_sha256" in g:
        from cerberus_amd.synth_tiles import tiles_sha256

        assert tiles_sha256(tiles) == str(g["tiles_sha256"]), "the fixture's tiles were not rebuilt bit for bit"
    assert tiles.shape[0] == n
    return tiles


# The bars (north_star: "within 1e-4 on float probability maps ... bit-exact on integer maps given identical seeds"):
#   * probabilities: 1e-4 absolute -- OR, where the reference's own float32 evaluation is further than that from its float64 evaluation
#     (noise/<head> in the fixture, measured by oracle/gen_golden_net.py with model.double(): up to 4.6e-4 under the reference's default
#     initialisation, whose logits run into the thousands), 3x that noise: two faithful fp32 evaluations of one network cannot be asked to
#     agree more closely than each agrees with the exact result.  The relative logit error is checked at 1e-5 beside it.
#   * integer (argmax) maps: EVERY mismatching pixel must sit where the reference's own top-1 / top-2 softmax margin (margin/<head>) is
#     below twice the probability bar of that head -- i.e. only genuine rounding-order ties may flip; a systematic argmax error fails.
def _prob_bar(g, head):
    noise = float(g["noise/" + head]) if ("noise/" + head) in g else 0.0
    return max(PROB_TOL, 3.0 * noise)


def _margin_bar(noise):
    return max(2e-5, 6.0 * noise)


def _check_type_map(g, head, got, ref):
    bad = got != ref
    if not bad.any():
        return
    assert ("margin/" + head) in g, "fixture without margins: " + head
    mg = g["margin/" + head]
    assert mg.shape == bad.shape, (mg.shape, bad.shape)
    noise = float(g["noise/" + head]) if ("noise/" + head) in g else 0.0
    worst = float(mg[bad].max())
    assert worst < _margin_bar(noise), "%s: an argmax differs where the reference's top-2 margin is %.3e (bar %.1e), %d pixels differ" % (head, worst, _margin_bar(noise), int(bad.sum()))


def _oracle_type_margins(sd, tiles, out_shape, kw):
    """top-1 minus top-2 softmax probability of the oracle's own logits inside the kept window, per TYPE head: (N, h, w)."""
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    lg = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"])
    oh, ow = (out_shape, out_shape) if np.isscalar(out_shape) else out_shape
    res = {}
    for k, v in lg.items():
        if not k.endswith("TYPE"):
            continue
        top = torch.topk(torch.softmax(v, 1), 2, dim=1).values
        mg = (top[:, 0] - top[:, 1]).numpy()
        y0, x0 = int((mg.shape[1] - oh) * 0.5), int((mg.shape[2] - ow) * 0.5)
        res[k] = mg[:, y0:y0 + oh, x0:x0 + ow]
    return res


def _assert_type_equal_up_to_ties(a, b, margin, k):
    bad = a != b
    if bad.any():
        worst = float(margin[bad].max())
        assert worst < _margin_bar(0.0), "%s: an argmax differs where the oracle's top-2 margin is %.3e, %d pixels differ" % (k, worst, int(bad.sum()))


@pytest.fixture(scope="module")
def full_model():
    return _model(None)


def test_native_library_is_loaded():
    from cerberus_amd import _lib

    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "libcerberus_hip.so" in maps


def test_encoder_and_logits_vs_oracle(full_model):
    m, sd, kw = full_model
    tiles = np.random.RandomState(11).randint(0, 256, (2, 256, 256, 3)).astype(np.uint8)
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    ref, feats, bottom = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"], return_feats=True)
    got = m.encoder_features(torch.from_numpy(tiles).cuda())
    for a, b in zip(got, feats[:4] + [feats[4], bottom]):
        assert (a.cpu().permute(0, 3, 1, 2) - b).abs().max().item() < 1e-4
    out = m(x.cuda())  # reference-style forward: float NCHW 0..255 -> logits NCHW
    assert list(out.keys()) == ["Lumen-INST", "Gland-INST", "Nuclei-INST", "Nuclei-TYPE", "Gland-TYPE", "Patch-Class"]
    for k, v in out.items():
        assert v.shape == ref[k].shape
        assert (v.cpu() - ref[k]).abs().max().item() < 2e-4, k


def test_forward_accepts_any_float_input(full_model):
    """NetDesc.forward on float NCHW values that are NOT whole numbers in 0..255 (models/net_desc.py:144-147 divides whatever it gets by 255):
    the float-input instantiation of the stem against the oracle on the same floats -- fractional, negative and above 255 -- and, on
    integer-valued floats, bit-identical to the uint8 path."""
    m, sd, kw = full_model
    rs = np.random.RandomState(12)
    x = torch.from_numpy((rs.rand(2, 3, 96, 96) * 300.0 - 20.0).astype(np.float32))
    ref = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"])
    out = m(x.cuda())
    assert list(out.keys()) == list(ref.keys())
    for k, v in out.items():
        assert v.shape == ref[k].shape
        assert (v.cpu() - ref[k]).abs().max().item() < 2e-4, k
    tiles = rs.randint(0, 256, (2, 96, 96, 3)).astype(np.uint8)
    a = m(torch.from_numpy(tiles).cuda())                                          # uint8 NHWC
    b = m(torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous())        # whole numbers as float NCHW: routed to the uint8 stem
    xf = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous().cuda()
    xf[0, 0, 0, 0] += 0.5                                                          # one fractional value: the float stem; every other input equal
    c = m(xf)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        # one input value moved by 0.5 / 255: the float stem does the same arithmetic on all the others (a wrong channel order, a missing /255 or a
        # transposed read would be O(1) here), and tile 1 does not see the change at all
        assert (a[k] - c[k]).abs().max().item() < 1e-2 * max(1.0, a[k].abs().max().item()), k
        assert torch.equal(a[k][1], c[k][1]), k


def test_head_w2_on_4x4_matrix_instructions_vs_padded_16_row_instruction(full_model):
    """cerb_net_set_head_algo(1) (default: 96 -> 3 / 7 logits on v_mfma_f32_4x4x1) against 2 (round 3: zero-padded 16 x 16 x 4) and 0 (one
    launch per head): the same products summed in another order -- logits within 2e-6 of their scale, probabilities within 2e-6, argmax
    maps equal wherever the top-2 margin is not rounding-sized; 0 and 2 stay bit-identical to each other."""
    m, sd, kw = full_model
    tiles = torch.from_numpy(np.random.RandomState(31).randint(0, 256, (3, 256, 256, 3)).astype(np.uint8)).cuda()
    res, lgs = {}, {}
    try:
        for algo in (1, 2, 0):
            m.set_head_algo(algo)
            res[algo] = {k: v.clone() for k, v in m.infer_tiles(tiles, [200, 232]).items()}
            lgs[algo] = {k: v.clone() for k, v in m(tiles).items()}
    finally:
        m.set_head_algo(1)
    for k in res[1]:
        assert torch.equal(res[2][k], res[0][k]), k
        if res[1][k].dtype == torch.float32:
            assert (res[1][k] - res[2][k]).abs().max().item() < 2e-6, k
        else:
            assert (res[1][k] != res[2][k]).float().mean().item() < 1e-4, k
    for k in lgs[1]:
        scale = max(1.0, lgs[2][k].abs().max().item())
        assert (lgs[1][k] - lgs[2][k]).abs().max().item() < 2e-6 * scale, k


GOLDEN_TAGS = ["cfg1_nuclei", "cfg2_all", "g448_all", "small96_all", "seed1_all", "refinit_all",
               # round 6: "a confident trained model" -- every dense head's calibration logits at 30 / 80 -- and structured inputs
               "logit30_all", "logit80_all", "struct_all", "struct80_all",
               # several output heads over one decoder (models/net_desc.py:81-87: {"Gland": {"INST": 3, "TYPE": 3}}): the trunk runs once, both heads read it
               "multihead"]


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_infer_step_vs_reference_golden(golden_dir, tag):
    """The default path against what the REFERENCE's own NetDesc / infer_step produced for the same seeded weights and tiles: two draws of the
    non-saturating recipe, four geometries, and the reference's DEFAULT initialisation (refinit_all: logits in the thousands, saturated softmax).
    ONE bar for every family, anchored on the reference's float64 evaluation (fixture p64_*, oracle/gen_golden_net.py with model.double()):

        |got - ref_fp64|  <=  |ref_fp32 - ref_fp64| + 1e-4          (north_star's 1e-4, measured from the exact result)

    None of the bars below comes from this build's own measured errors: they are the north star's constant and the reference's own
    fp32-vs-fp64 noise.  refinit_all meets the bar because the handle probes its weights (NetDesc._auto_precision: calibration logits above 100
    -> F(2x2,3x3) instead of F(4x4,3x3)); it is also held to 1.5x the DIRECT convolution's distance from float64 -- the algorithm the
    reference's cuDNN may pick can never be much closer to the exact result than this path."""
    g = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    m, sd, kw, tasks = _golden_model(g)
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = _golden_tiles(g)
    dec = m.prepare()  # the load-time decision (run_infer_*.py log it)
    m.watch_logits()
    out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
    assert isinstance(out, list) and len(out) == n
    saturated = tag == "refinit_all"
    assert (m.calibration_logit_absmax > m.LOGIT_SATURATION) == saturated and bool(getattr(m, "_algo_is_auto", False)) == saturated, m.calibration_logit_absmax
    assert dec["conv_algo"] == (1 if saturated else DEFAULT_ALGO) and dec["auto"] == saturated and dec["probed"], dec
    if "logit_target" in g:  # the scaled families sit where the generator put them: the calibration tile's logits at the target on every dense head
        assert abs(m.calibration_logit_absmax - float(g["logit_target"])) < 0.02 * float(g["logit_target"]), m.calibration_logit_absmax
    # the data-aware guard: the head kernels' own maximum over THIS batch equals the largest |logit| of the reference's float64 evaluation
    seen = m.logit_absmax()
    for k, v in seen.items():
        want = float(g["logit_absmax/" + k])  # (over the whole map; a cropped forward evaluates the kept window only)
        assert (abs(v - want) if osz == hw else max(0.0, v - want)) <= 2e-3 * max(1.0, want), (k, v, want)
    direct = None
    if saturated:  # the same network on the direct implicit-GEMM convolution (conv_algo 0)
        md, _, _, _ = _golden_model(g)
        md.set_conv_algo(0)
        direct = infer_step(torch.from_numpy(tiles), md, osz, tasks)
    lg = m(torch.from_numpy(tiles))
    for k in out[0].keys():
        a = np.stack([out[i][k] for i in range(n)])
        assert str(a.dtype) == str(g["out_dtype/" + k]), k  # float32 / int64 protocol of run_desc.py:439-502
        assert a.shape[1:3] == (osz, osz)
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = _crops(a4) if key in g else a4
        if a.dtype == np.float32:
            err = np.abs(got - ref).max()
            assert err < _prob_bar(g, k), (k, err, _prob_bar(g, k))  # against the reference's fp32 output: 1e-4, or 3x the reference's own fp32 noise
            if not k.endswith("INST"):  # Patch-Class: class ids carried in a float map, compared above
                continue
            p64 = g["p64_crops/" + k] if ("p64_crops/" + k) in g else g["p64_full/" + k]
            e64, r64 = float(np.abs(got - p64).max()), float(np.abs(ref - p64).max())
            assert e64 <= r64 + PROB_TOL, (k, e64, r64)
            msg = "achieved %s %s: |got - ref32| %.3e  |got - ref64| %.3e  (|ref32 - ref64| %.3e)" % (tag, k, err, e64, r64)
            if direct is not None:
                d = np.stack([direct[i][k] for i in range(n)])
                d = _crops(d) if key in g else d
                d64 = float(np.abs(d - p64).max())
                assert e64 <= 1.5 * d64 + 1e-5, (k, e64, d64)
                msg += "  direct convolution |.. - ref64| %.3e" % d64
            print(msg)
        elif k != "Patch-Class":
            _check_type_map(g, k, got, ref)
        else:
            assert np.array_equal(got, ref), k
    for k, v in lg.items():
        a = v.permute(0, 2, 3, 1).contiguous().cpu().numpy()
        key = "logits_crops/" + k
        ref = g[key] if key in g else g["logits_full/" + k]
        got = _crops(a) if key in g else a
        scale = max(1.0, float(g["logit_absmax/" + k]) if ("logit_absmax/" + k) in g else 1.0)
        assert np.abs(got - ref).max() < 3e-4 * max(1.0, scale / 30.0), (k, np.abs(got - ref).max(), scale)


def test_infer_step_full_tensor_vs_oracle(full_model):
    m, sd, kw = full_model
    tiles = np.random.RandomState(5).randint(0, 256, (3, 256, 256, 3)).astype(np.uint8)
    got = infer_step(torch.from_numpy(tiles), m, 256, kw["considered_tasks"])
    ref = net_ref.infer_step(sd, tiles, 256, kw["considered_tasks"], kw["decoder_kwargs"])
    mgs = _oracle_type_margins(sd, tiles, 256, kw)
    for i in range(3):
        assert list(got[i].keys()) == list(ref[i].keys())
        for k in ref[i]:
            a, b = got[i][k], ref[i][k]
            assert a.shape == b.shape and a.dtype == b.dtype, k
            if a.dtype == np.float32:
                assert np.abs(a - b).max() < PROB_TOL, k
            elif k in mgs:
                _assert_type_equal_up_to_ties(a, b, mgs[k][i], k)
            else:
                assert np.array_equal(a, b), k


def test_ragged_batch_sizes_and_crop(full_model):
    """batch 1 (the squeeze/unsqueeze special case of run_desc.py:484-487) and a non-multiple-of-32 tile size."""
    m, sd, kw = full_model
    tiles = np.random.RandomState(6).randint(0, 256, (1, 304, 304, 3)).astype(np.uint8)
    got = infer_step(torch.from_numpy(tiles), m, [144, 160], kw["considered_tasks"])
    ref = net_ref.infer_step(sd, tiles, [144, 160], kw["considered_tasks"], kw["decoder_kwargs"])
    mgs = _oracle_type_margins(sd, tiles, [144, 160], kw)
    for k in ref[0]:
        a, b = got[0][k], ref[0][k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype == np.float32:
            assert np.abs(a - b).max() < PROB_TOL, k
        elif k in mgs:
            _assert_type_equal_up_to_ties(a, b, mgs[k][0], k)
        else:
            assert np.array_equal(a, b), k


def test_linearity_property_of_identical_tiles(full_model):
    """Size-independent property: a batch of identical tiles gives identical outputs per sample, and outputs do not
    depend on the batch they were computed in (eval-mode BN, no cross-tile state -- SURVEY.md par.8e)."""
    m, sd, kw = full_model
    t = torch.randint(0, 256, (1, 256, 256, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    other = torch.randint(0, 256, (6, 256, 256, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(4))
    a = m.infer_tiles(t.cuda(), 256)
    b = m.infer_tiles(torch.cat([other[:3], t, other[3:]]).cuda(), 256)
    for k in a:
        assert torch.equal(a[k][0], b[k][3]), k


def test_scatter_into_canvas(full_model):
    """tile_off/row_stride addressing == the stitching of infer/tile.py:141-163 for non-overlapping patches."""
    m, sd, kw = full_model
    tiles = torch.randint(0, 256, (4, 256, 256, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(8)).cuda()
    dense = m.infer_tiles(tiles, 256)
    W = 512
    canvas = torch.zeros((512, W, 2), dtype=torch.float32, device="cuda")
    tmap = torch.zeros((512, W), dtype=torch.uint8, device="cuda")
    off = torch.tensor([0, 256, 256 * W, 256 * W + 256], dtype=torch.int64, device="cuda")
    outs = [None] * 6
    outs[2] = canvas
    outs[3] = tmap
    m._run(tiles, 256, 256, outs, None, tile_off=off, row_stride=W, type_is_u8=True)
    torch.cuda.synchronize()
    for i, (y, x) in enumerate([(0, 0), (0, 256), (256, 0), (256, 256)]):
        assert torch.equal(canvas[y:y + 256, x:x + 256], dense["Nuclei-INST"][i])
        assert torch.equal(tmap[y:y + 256, x:x + 256].long(), dense["Nuclei-TYPE"][i])


@pytest.mark.parametrize("tag", ["cfg2_all", "g448_all"])
def test_direct_conv_algo_vs_reference_golden(golden_dir, tag):
    """cerb_net_set_conv_algo(0): the direct implicit-GEMM kernels (conv_igemm.hip, fused upsample+skip staging) meet the same
    golden vectors as the default Winograd path, and the two algorithms agree to 5e-5 on every probability map."""
    g = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    tasks = [str(t) for t in g["tasks"]]
    m, sd, kw = _model(tasks, int(g["weight_seed"]))
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    wino = infer_step(torch.from_numpy(tiles), m, osz, tasks)
    m.set_conv_algo(0)
    try:
        direct = infer_step(torch.from_numpy(tiles), m, osz, tasks)
        m.profile(True)
        m.infer_tiles(torch.from_numpy(tiles).cuda(), osz)
        torch.cuda.synchronize()
        kernels = {r[1] for r in m.profile_records()}
        m.profile(False)
    finally:
        m.set_conv_algo(DEFAULT_ALGO)
    assert not any(k.startswith("conv_wino") for k in kernels) and any("mode1" in k for k in kernels)
    for k in direct[0].keys():
        a = np.stack([direct[i][k] for i in range(n)])
        b = np.stack([wino[i][k] for i in range(n)])
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = _crops(a4) if key in g else a4
        if a.dtype == np.float32:
            assert np.abs(got - ref).max() < _prob_bar(g, k), k
            assert np.abs(a - b).max() < (5e-5 if tag != "refinit_all" else 2.0 * _prob_bar(g, k)), k
            if not k.endswith("INST"):
                continue
            p64 = g["p64_crops/" + k] if ("p64_crops/" + k) in g else g["p64_full/" + k]
            if tag != "refinit_all":
                assert float(np.abs(got - p64).max()) <= float(np.abs(ref - p64).max()) + PROB_TOL, k
            else:
                assert float(np.abs(got - p64).max()) <= 3.0 * float(g["noise/" + k]) + PROB_TOL, k
        elif tag == "refinit_all" and k != "Patch-Class":
            _check_type_map(g, k, got, ref)
        else:
            assert (got != ref).mean() < 1e-4, k
            assert (a != b).mean() < 1e-4, k


def _dice(pred, ref):
    """models/run_desc.py:526-531,608-617: 2*sum(pred & ref) / (sum(pred) + sum(ref) + 1e-8)"""
    pred, ref = pred.astype(np.float64), ref.astype(np.float64)
    return 2.0 * (pred * ref).sum() / (pred.sum() + ref.sum() + 1e-8)


def test_dice_vs_reference_golden(golden_dir):
    """SURVEY par.8d: per-head Dice of the HIP path against the reference's own outputs -- INST foreground (> 0.5) per channel
    and every TYPE class with support in the fixture."""
    g = np.load(os.path.join(golden_dir, "net_cfg2_all.npz"))
    tasks = [str(t) for t in g["tasks"]]
    m, sd, kw = _model(tasks, int(g["weight_seed"]))
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
    checked = 0
    for k in out[0].keys():
        a = np.stack([out[i][k] for i in range(n)])
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = _crops(a4) if key in g else a4
        if k.endswith("-INST"):
            for c in range(2):
                r, p = ref[..., c] > 0.5, got[..., c] > 0.5
                if r.sum() > 50:
                    assert _dice(p, r) > 0.9995, (k, c)
                    checked += 1
        elif k.endswith("-TYPE"):
            for cls in np.unique(ref):
                r, p = ref == cls, got == cls
                if r.sum() > 50:
                    assert _dice(p, r) > 0.9995, (k, int(cls))
                    checked += 1
    assert checked >= 6


def test_large_odd_tile_guard_band():
    """One 1040 x 1040 tile (levels 1040/520/260/130/65: partial Winograd / MFMA tiles at every deep level, rows hanging past the
    last image row) -- exercises the size of the zero guard band around the activation buffers.  Nuclei decoder only."""
    m, sd, kw = _model(["Nuclei"])
    tiles = np.random.RandomState(13).randint(0, 256, (1, 1040, 1040, 3)).astype(np.uint8)
    got = infer_step(torch.from_numpy(tiles), m, 1040, kw["considered_tasks"])
    ref = net_ref.infer_step(sd, tiles, 1040, kw["considered_tasks"], kw["decoder_kwargs"])
    for k in ref[0]:
        a, b = got[0][k], ref[0][k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype == np.float32:
            assert np.abs(a - b).max() < PROB_TOL, k
        else:
            assert (a != b).mean() < 1e-4, k
    m.set_conv_algo(0)
    try:
        direct = infer_step(torch.from_numpy(tiles), m, 1040, kw["considered_tasks"])
    finally:
        m.set_conv_algo(DEFAULT_ALGO)
    for k in ref[0]:
        if ref[0][k].dtype == np.float32:
            assert np.abs(direct[0][k] - ref[0][k]).max() < PROB_TOL, k


@pytest.mark.parametrize("hw", [(96, 256), (64, 64), (128, 112)])
def test_small_feature_maps_patch_class_crop(full_model, hw):
    """Bottom feature maps smaller than 9 x 9: cropping_center's Python slice starts at a NEGATIVE index (misc_utils.py:22-24), e.g.
    a 6-row map keeps only its last row.  All heads, logits and wrapper outputs."""
    m, sd, kw = full_model
    tiles = np.random.RandomState(hw[0]).randint(0, 256, (2,) + hw + (3,)).astype(np.uint8)
    got = infer_step(torch.from_numpy(tiles), m, list(hw), kw["considered_tasks"])
    ref = net_ref.infer_step(sd, tiles, list(hw), kw["considered_tasks"], kw["decoder_kwargs"])
    for i in range(2):
        for k in ref[i]:
            a, b = got[i][k], ref[i][k]
            assert a.shape == b.shape and a.dtype == b.dtype, k
            if a.dtype == np.float32:
                assert np.abs(a - b).max() < PROB_TOL, k
            else:
                assert (a != b).mean() < 1e-4, k
    x = torch.from_numpy(tiles).float().permute(0, 3, 1, 2).contiguous()
    lg = m(x.cuda())["Patch-Class"].cpu()
    rl = net_ref.net_forward(sd, x, kw["decoder_kwargs"], kw["considered_tasks"])["Patch-Class"]
    assert (lg - rl).abs().max().item() < 2e-4


@pytest.mark.parametrize("algo", [6, 1, 0, 5, 7])
def test_forward_is_bitwise_reproducible(full_model, algo):
    """Races and un-padded hardware hazards show up as run-to-run differences long before they show up as large errors (the
    gfx950 buffer_store hazard of DESIGN par.4.1 did): the same batch through the same handle must give identical bits."""
    m, sd, kw = full_model
    tiles = torch.from_numpy(np.random.RandomState(21).randint(0, 256, (12, 256, 256, 3)).astype(np.uint8)).cuda()
    m.set_conv_algo(algo)
    try:
        ref = None
        for _ in range(4):
            outs = m.infer_tiles(tiles, 256)
            torch.cuda.synchronize()
            cur = {k: o.clone() for k, o in outs.items()}
            if ref is None:
                ref = cur
            else:
                for k in ref:
                    assert torch.equal(ref[k], cur[k]), k
    finally:
        m.set_conv_algo(DEFAULT_ALGO)


@pytest.mark.parametrize("algo", [5, 7])
@pytest.mark.parametrize("tag", ["cfg2_all", "g448_all", "small96_all", "refinit_all"])
def test_wino4_algo_vs_reference_golden(golden_dir, tag, algo):
    """cerb_net_set_conv_algo(5 / 7): Winograd F(4x4,3x3) (conv_wino4.hip / conv_wino4b.hip: 36 products per 16 outputs, transform points
    0, +-1, +-2, inf) against the reference's golden vectors at the same 1e-4 bar -- plain, residual, grouped, cropped (region-of-interest blocks),
    odd-sized (blocks hanging over the image, odd block counts) launches -- and against algorithm 1 (F(2x2)): the two differ by the
    transforms' rounding only, a few 1e-6 on the probability maps (tests/tools/dev_wino4_numerics.py)."""
    g = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    m, sd, kw, tasks = _golden_model(g)  # (refinit_all: the reference's default initialisation -- logits in the thousands, the stress case)
    n, hw, osz = int(g["n"]), int(g["hw"]), int(g["out_shape"])
    tiles = np.random.RandomState(int(g["tile_seed"])).randint(0, 256, (n, hw, hw, 3)).astype(np.uint8)
    m.set_conv_algo(1)
    wino = infer_step(torch.from_numpy(tiles), m, osz, tasks)
    m.set_conv_algo(algo)
    try:
        out = infer_step(torch.from_numpy(tiles), m, osz, tasks)
        m.profile(True)
        m.infer_tiles(torch.from_numpy(tiles).cuda(), osz)
        torch.cuda.synchronize()
        kernels = {r[1] for r in m.profile_records()}
        m.profile(False)
    finally:
        m.set_conv_algo(DEFAULT_ALGO)
    assert any(k.startswith("conv_wino4") for k in kernels) and not any(k.startswith("conv_wino<") for k in kernels)
    for k in out[0].keys():
        a = np.stack([out[i][k] for i in range(n)])
        b = np.stack([wino[i][k] for i in range(n)])
        a4 = a[..., None] if a.ndim == 3 else a
        key = "out_crops/" + k
        ref = g[key] if key in g else g["out_full/" + k]
        got = _crops(a4) if key in g else a4
        if a.dtype == np.float32:
            assert np.abs(got - ref).max() < _prob_bar(g, k), k
            assert np.abs(a - b).max() < (5e-5 if tag != "refinit_all" else 2.0 * _prob_bar(g, k)), k
            if k.endswith("INST"):  # (Patch-Class is a float map of class ids: no fp64 anchor)
                p64 = g["p64_crops/" + k] if ("p64_crops/" + k) in g else g["p64_full/" + k]
                if tag != "refinit_all":
                    assert float(np.abs(got - p64).max()) <= float(np.abs(ref - p64).max()) + PROB_TOL, k
                else:
                    assert float(np.abs(got - p64).max()) <= 3.0 * float(g["noise/" + k]) + PROB_TOL, k
        elif tag == "refinit_all" and k != "Patch-Class":
            _check_type_map(g, k, got, ref)
        else:
            assert (got != ref).mean() < 1e-4, k
            assert (a != b).mean() < 1e-4, k


@pytest.mark.parametrize("hw,out", [((448, 448), (144, 144)), ((256, 256), (144, 144)), ((304, 272), (100, 36)), ((96, 128), (32, 128)),
                                    ((448, 448), (447, 3)), ((256, 256), (2, 2))])
def test_crop_region_of_interest_is_bit_identical_to_the_full_computation(full_model, hw, out):
    """cerb_net_set_crop_roi: with a centre crop the decoders and heads compute only what the kept window depends on (3x3 conv: one
    pixel per layer, bilinear x2: one source pixel, item granularity on top).  Same kernels, same arithmetic for every kept pixel:
    the outputs must be bit-identical to the full computation, whatever stale values the skipped regions of the buffers hold."""
    m = full_model[0]
    rs = np.random.RandomState(hw[0] + out[1])
    tiles = torch.from_numpy(rs.randint(0, 256, (3,) + hw + (3,)).astype(np.uint8)).cuda()
    other = torch.from_numpy(rs.randint(0, 256, (3,) + hw + (3,)).astype(np.uint8)).cuda()
    try:
        m.set_crop_roi(False)
        full = {k: v.clone() for k, v in m.infer_tiles(tiles, list(out)).items()}
        m.infer_tiles(other, list(hw))  # leave a different image's activations in every workspace buffer
        m.set_crop_roi(True)
        roi = m.infer_tiles(tiles, list(out))
        for k in full:
            assert torch.equal(full[k], roi[k]), k
        # a second pass over the stale region-of-interest state of the first
        roi2 = m.infer_tiles(tiles, list(out))
        for k in full:
            assert torch.equal(full[k], roi2[k]), k
    finally:
        m.set_crop_roi(True)


@pytest.mark.parametrize("win,osz,n", [(144, 144, 3), (176, 80, 5), (304, 144, 3), (208, 208, 2)])
def test_winograd_partial_tiles_and_odd_blocks_vs_direct(full_model, win, osz, n):
    """Map sizes that are not multiples of the 4x4 Winograd tile (a 144-pixel tile has an 18 x 18 level, a 176-pixel one 22 x 22 / 11 x 11),
    odd numbers of 16 x 16 blocks (conv_wino4's two-block items: the last pair repeats a block) and blocks hanging over the map, with
    and without a centre crop: every F(4x4) path (5: conv_wino4, 7: conv_wino4b, 6: the default mix) agrees with the direct implicit
    GEMM (conv_algo 0, no tiles at all) to 2e-5 on every probability map, is bitwise reproducible, and the TYPE maps differ on < 1e-4 of
    the pixels."""
    m, sd, kw = full_model
    tiles = torch.from_numpy(np.random.RandomState(100 + win).randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
    try:
        m.set_conv_algo(0)
        ref = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        for algo in (5, 7, 6):
            m.set_conv_algo(algo)
            a = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
            b = m.infer_tiles(tiles, osz)
            torch.cuda.synchronize()
            for k in ref:
                assert torch.equal(a[k], b[k]), (algo, k)
                if ref[k].dtype.is_floating_point:
                    assert (a[k] - ref[k]).abs().max().item() < 2e-5, (algo, k)
                else:
                    assert (a[k] != ref[k]).float().mean().item() < 1e-4, (algo, k)
    finally:
        m.set_conv_algo(DEFAULT_ALGO)


@pytest.mark.parametrize("win,osz,n", [(448, 448, 3), (448, 144, 2), (224, 224, 5), (208, 80, 2), (240, 240, 3), (160, 160, 7), (176, 176, 5), (112, 112, 9), (256, 256, 2)])
def test_packed_items_are_bit_identical_to_block_items(full_model, win, osz, n):
    """cerb_net_set_packed_items(1) (default): on maps whose sides are multiples of 4 but not of 16 -- the 56^2 / 28^2 maps of the reference's 448-pixel
    patch, 52^2 of a 208-pixel one, 60^2 of 240, 40^2 / 20^2 of 160, 44^2 of 176, 28^2 of 112 -- conv_wino4b takes 16 CONSECUTIVE tiles of the batch per work item instead of a
    16 x 16-pixel block with padding tiles (3 x 49 tiles = 9 items + 3 tiles: the last item runs with 13 empty tiles).  A tile's arithmetic does not
    depend on the item it rides in: every head's output must be BITWISE equal to the block form's -- residual layers, grouped decoder launches,
    centre crops (the cropped decoder launches stay on blocks) and a different batch in between included.  256-pixel tiles have no such map: the
    switch must not change which kernels run there."""
    m, sd, kw = full_model
    rs = np.random.RandomState(900 + win + osz)
    tiles = torch.from_numpy(rs.randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
    other = torch.from_numpy(rs.randint(0, 256, (n + 2, win, win, 3)).astype(np.uint8)).cuda()
    try:
        m.set_packed_items(False)
        m.profile(True)
        ref = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        torch.cuda.synchronize()
        kern_blk = [r[1] for r in m.profile_records()]
        m.profile(False)
        m.set_packed_items(True)
        m.infer_tiles(other, osz)
        m.profile(True)
        got = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        torch.cuda.synchronize()
        kern_pk = [r[1] for r in m.profile_records()]
        m.profile(False)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (k, (got[k].float() - ref[k].float()).abs().max().item())
    finally:
        m.profile(False)
        m.set_packed_items(True)
    n_pk = sum(k.startswith("conv_wino4b<f4x4,16t") for k in kern_pk)
    assert not any(k.startswith("conv_wino4b<f4x4,16t") for k in kern_blk), kern_blk
    assert len(kern_pk) == len(kern_blk)
    if win == 256:
        assert n_pk == 0 and kern_pk == kern_blk
    else:
        assert n_pk >= 5, (n_pk, sorted(set(kern_pk)))
        assert [k.replace(",16t", ",16x16") for k in kern_pk] == kern_blk


@pytest.mark.parametrize("win,osz,n", [(256, 256, 3), (448, 144, 2), (272, 272, 2), (304, 144, 3), (96, 96, 5), (208, 80, 2), (256, 256, 40)])  # 40 tiles: planar tensors beyond 4 GiB
def test_planar_last_level_is_bit_identical_to_nhwc(full_model, win, osz, n):
    """cerb_net_set_planar(1) (default): the last decoder level in the tile-planar layout (upsample2_add_planar -> conv_wino4p x2 -> heads
    reading planar features) against cerb_net_set_planar(0) (round 2's NHWC path through conv_wino4): same arithmetic in the same order,
    so every head's output must be BITWISE equal -- whole tiles, centre crops (region-of-interest launches), odd block counts, a 96-pixel
    tile whose last level falls back to NHWC, and a different batch in between (stale planar workspaces).  Also checks that the planar
    kernels are the ones that ran."""
    m, sd, kw = full_model
    rs = np.random.RandomState(500 + win + osz)
    tiles = torch.from_numpy(rs.randint(0, 256, (n, win, win, 3)).astype(np.uint8)).cuda()
    other = torch.from_numpy(rs.randint(0, 256, (n + 1, win, win, 3)).astype(np.uint8)).cuda()
    try:
        m.set_planar(False)
        ref = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        m.set_planar(True)
        m.infer_tiles(other, osz)  # leaves another batch's values in every planar buffer
        m.profile(True)
        got = {k: v.clone() for k, v in m.infer_tiles(tiles, osz).items()}
        torch.cuda.synchronize()
        kernels = [r[1] for r in m.profile_records()]
        m.profile(False)
        again = m.infer_tiles(tiles, osz)
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(got[k], ref[k]), (k, (got[k].float() - ref[k].float()).abs().max().item())
            assert torch.equal(again[k], ref[k]), k
    finally:
        m.profile(False)
        m.set_planar(True)
    if win * win > 4096 * 4:  # the two last levels are above 64 x 64: the planar kernels carry both
        want = 4
        assert sum(k.startswith("conv_wino4p") for k in kernels) == want and kernels.count("upsample2_add_planar") == want // 2, kernels


def test_logit_guard_words_are_the_batch_maximum_and_follow_the_data(golden_dir):
    """cerb_forward_io.logit_absmax: the head kernels' per-head word after a forward IS the largest |logit| of that forward
    (bit for bit: the same kernel writes the logits), it rises with later batches and never falls (atomic max), words of a batch of stain-field /
    white / black tiles stay below those of a noise batch on the logit-80 family, and the Patch-Class word is left alone."""
    g = np.load(os.path.join(golden_dir, "net_struct80_all.npz"))
    m, sd, kw, tasks = _golden_model(g)
    m.prepare()
    smooth = torch.from_numpy(_golden_tiles(g)).cuda()
    noise = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (3, 256, 256, 3)).astype(np.uint8)).cuda()
    nd = len(m._decoders)
    rows = torch.zeros((3, nd), dtype=torch.int32, device="cuda")
    rows[:, nd - 1] = 12345  # Patch-Class: untouched
    outs = [None] * nd

    def run(tiles, row):
        lg = [torch.empty((tiles.shape[0], 256, 256, d[2]), dtype=torch.float32, device="cuda") if d[0] != "Patch-Class" else None for d in m._decoders]
        m._run(tiles, 256, 256, outs, lg, logit_absmax=row)
        torch.cuda.synchronize()
        return [None if t is None else float(t.abs().max()) for t in lg]

    lg_smooth = run(smooth, rows[0])
    lg_noise = run(noise, rows[1])
    run(smooth, rows[2])
    run(noise, rows[2])  # two batches into one row: the maximum of both
    vals = m.logit_absmax(words=rows)
    for i, d in enumerate(m._decoders):
        if d[0] == "Patch-Class":
            assert (rows[:, i].cpu().numpy() == 12345).all()
            continue
        assert vals[0, i] == np.float32(lg_smooth[i]) and vals[1, i] == np.float32(lg_noise[i]), (d[3], vals[:, i], lg_smooth[i], lg_noise[i])
        assert vals[2, i] == max(vals[0, i], vals[1, i])
        assert vals[0, i] < vals[1, i], d[3]  # smooth tiles drive this family's heads less hard than texture does
    # the cropped forward (heads evaluate the kept window only) reports the window's maximum: never above the whole map's
    crop = torch.zeros(nd, dtype=torch.int32, device="cuda")
    m.infer_tiles(noise, 144)  # no row: nothing written, nothing read
    res = [torch.empty((3, 144, 144, 2), dtype=torch.float32, device="cuda") if d[1] == "INST" else
           (torch.empty((3, 144, 144), dtype=torch.uint8, device="cuda") if d[1] == "TYPE" else torch.empty((3, 144, 144), dtype=torch.float32, device="cuda")) for d in m._decoders]
    m._run(noise, 144, 144, res, None, type_is_u8=True, logit_absmax=crop)
    cv = m.logit_absmax(words=crop)
    for i, d in enumerate(m._decoders):
        if d[0] != "Patch-Class":
            assert 0.0 < cv[i] <= vals[1, i], d[3]


def test_wsi_runner_counts_and_reruns_a_planted_high_logit_batch(golden_dir):
    """The slide loop's use of the guard: a slab of smooth stain fields with one patch of texture in it, the logit-80 model, the bar put between
    the hardest-driven batch and the next -- logit_report() gives every batch the largest |logit| a plain forward of its tiles has, flags exactly
    the batch above the bar, and rerun_flagged() leaves that batch's canvas windows equal to a run of the same tiles on conv_algo 1 while every
    other window keeps its F(4x4) bytes."""
    from cerberus_amd.synth_tiles import stain_field
    from cerberus_amd.wsi import WSIRunner, gather_patches

    g = np.load(os.path.join(golden_dir, "net_logit80_all.npz"))
    m, sd, kw, tasks = _golden_model(g)
    m.prepare()
    H, W, win, batch = 512, 1536, 256, 4  # 2 x 6 patches, 3 batches of 4
    slab = np.concatenate([np.concatenate([stain_field(256, 20 + 6 * r + c) for c in range(6)], axis=1) for r in range(2)], axis=0)
    slab[256:512, 512:768] = np.random.RandomState(5).randint(0, 256, (256, 256, 3))  # patch (1, 2) = index 8 -> batch 2 (patches 8 .. 11)
    slab_dev = torch.from_numpy(slab).cuda()
    run = WSIRunner(m, (H, W), win, win, batch)
    run.infer_band(slab_dev, 0)
    rep = run.logit_report(threshold=1e9)
    assert rep["batches"] == 3 and rep["above"] == 0
    words = m.logit_absmax(words=run._logit_log[:3])
    dense = [i for i, d in enumerate(m._decoders) if d[0] != "Patch-Class"]
    per_batch = words[:, dense].max(axis=1)
    truth = []
    for b in range(3):  # what a plain forward of the batch's tiles gives
        tiles = gather_patches(slab_dev, 0, H, run._tl_y[4 * b:4 * b + 4], run._tl_x[4 * b:4 * b + 4], win)
        truth.append(max(float(v.abs().max()) for k, v in m(tiles).items() if k != "Patch-Class"))
    assert np.array_equal(per_batch, np.array(truth, np.float32)), (per_batch, truth)
    order = np.argsort(per_batch)
    hot = int(order[-1])
    assert per_batch[hot] > per_batch[order[-2]]
    assert abs(per_batch[2] - per_batch[:2].max()) > 1.0, per_batch  # texture and stain field drive the heads differently: the maximum follows the data
    m.LOGIT_SATURATION = float(0.5 * (per_batch[hot] + per_batch[order[-2]]))
    rep = run.logit_report()
    assert rep["above"] == 1 and rep["flagged"] == [(4 * hot, 4 * hot + 4)] and abs(rep["max"] - per_batch[hot]) < 1e-6
    before = {k: v.clone() for k, v in run.canv.items()}
    assert run.rerun_flagged(slab_dev, 0, rep["flagged"]) == 1
    torch.cuda.synchronize()
    assert m.precision_decision()["conv_algo"] == DEFAULT_ALGO  # the handle is back on its default
    tiles = gather_patches(slab_dev, 0, H, run._tl_y[4 * hot:4 * hot + 4], run._tl_x[4 * hot:4 * hot + 4], win)
    m.set_conv_algo(1)
    want = m.infer_tiles(tiles, win, type_dtype=torch.uint8)
    m.set_conv_algo(DEFAULT_ALGO)
    changed = 0
    for k, v in run.canv.items():
        keep = torch.ones(v.shape[:2], dtype=torch.bool, device="cuda")
        for j, p in enumerate(range(4 * hot, 4 * hot + 4)):
            r, c = divmod(p, 6)
            assert torch.equal(v[r * 256:(r + 1) * 256, c * 256:(c + 1) * 256], want[k][j]), (k, p)
            keep[r * 256:(r + 1) * 256, c * 256:(c + 1) * 256] = False
        assert torch.equal(v[keep], before[k][keep]), k
        changed += int((v != before[k]).any())
    assert changed > 0  # F(2x2) and F(4x4) differ in the last bits somewhere
