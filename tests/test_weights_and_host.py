"""Host logic that needs no GPU: state-dict schema, seeded weights, strict loading, model-construction errors."""
import numpy as np
import pytest
import torch

from cerberus_amd.net_desc import NetDesc, create_model
from cerberus_amd.weights import default_model_kwargs, make_state_dict, state_dict_schema, state_dict_sha256


def test_schema_has_558_reference_keys():
    sch = state_dict_schema()
    assert len(sch) == 558  # 558 state-dict keys of the six-head model
    keys = [k for k, _, _ in sch]
    assert len(set(keys)) == 558
    for k in ["backbone.conv1.weight", "backbone.layer4.2.bn2.running_var", "backbone.layer2.0.downsample.0.weight",
              "backbone.fc.bias", "conv_map.weight", "decoder_head.Nuclei#TYPE.3.block.1.conv.bias",
              "decoder_head.Patch-Class.conv2.weight", "output_head.Gland#TYPE.TYPE.x.1.conv.weight",
              "output_head.Lumen.INST.x.0.block.0.bn.num_batches_tracked"]:
        assert k in keys, k
    n_params = sum(int(np.prod(s)) for k, s, kind in sch if not kind.startswith("bn_m") and not kind.startswith("bn_v") and kind != "bn_n")
    assert n_params == 28377284  # SURVEY.md par.2b parameter count


def test_seeded_weights_are_deterministic(golden_dir):
    import os

    a, b = make_state_dict(0), make_state_dict(0)
    assert state_dict_sha256(a) == state_dict_sha256(b)
    assert state_dict_sha256(a) != state_dict_sha256(make_state_dict(1))
    g = np.load(os.path.join(golden_dir, "net_cfg2_all.npz"))
    assert str(g["weights_sha256"]) == state_dict_sha256(a)  # the fixtures were made with exactly these weights


def test_nuclei_only_schema():
    kw = default_model_kwargs(["Nuclei"])
    sch = state_dict_schema(kw["decoder_kwargs"], kw["considered_tasks"])
    keys = [k for k, _, _ in sch]
    assert any(k.startswith("decoder_head.Nuclei.") for k in keys)
    assert not any("Lumen" in k or "Patch-Class" in k or "#TYPE" in k for k in keys)


def test_load_state_dict_strict_errors():
    m = create_model(**default_model_kwargs())
    sd = m.state_dict()
    assert len(sd) == 558
    m.load_state_dict(sd, strict=True)
    bad = dict(sd)
    del bad["conv_map.weight"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["module.extra"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["conv_map.weight"] = torch.zeros(256, 512, 3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad, strict=True)


def test_unsupported_backbone_and_train_mode():
    kw = default_model_kwargs()
    kw["encoder_backbone_name"] = "densenet121"
    with pytest.raises(NotImplementedError):
        create_model(**kw)
    m = create_model(**default_model_kwargs(["Nuclei"]))
    assert isinstance(m, NetDesc) and isinstance(m, torch.nn.Module)
    # train() selects the training packing of a handle that does not exist yet (forward half of the training step); eval() the default
    assert m.train() is m and m.training and m.eval() is m and not m.training


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cerberus_amd._lib import CerberusHipError

    m = create_model(**default_model_kwargs(["Nuclei"]))
    with pytest.raises(CerberusHipError, match="no CPU fallback"):
        m.infer_tiles(torch.zeros((1, 256, 256, 3), dtype=torch.uint8), 256)


def test_custom_ops_register_without_a_gpu_and_refuse_cpu_tensors():
    """cerberus_amd/ops.py: schemas + FakeTensor shape inference work anywhere; a CPU tensor is refused by the dispatcher (CUDA-only kernels)."""
    import pytest
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from cerberus_amd import ops  # noqa: F401

    assert "Tensor[]" in str(torch.ops.cerberus_amd.infer_tiles.default._schema)
    with FakeTensorMode():
        outs = torch.ops.cerberus_amd.infer_tiles(torch.empty((2, 256, 256, 3), dtype=torch.uint8), 0, 256, 256, "Nuclei-INST,Nuclei-TYPE")
        assert [tuple(o.shape) for o in outs] == [(2, 256, 256, 2), (2, 256, 256)] and outs[1].dtype == torch.int64
    with pytest.raises(NotImplementedError):
        torch.ops.cerberus_amd.postproc(torch.zeros((8, 8, 2)), "Nuclei", 1.0, True)
