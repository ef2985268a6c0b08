"""torch.ops.cerberus_amd.* (cerberus_amd/ops.py): the torch-level operators over the C ABI give exactly what the host mirrors give."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_custom_ops_equal_the_mirrors_bit_for_bit():
    from cerberus_amd import ops  # noqa: F401  (registers the operators)
    from cerberus_amd.net_desc import create_model
    from cerberus_amd.postproc import inst_table_device, postproc_device
    from cerberus_amd.weights import default_model_kwargs, make_state_dict
    from oracle import synth

    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    tiles = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (3, 256, 256, 3)).astype(np.uint8)).cuda()
    h = m.handle_value()
    want = ["Nuclei-INST", "Nuclei-TYPE", "Patch-Class", "Gland-INST"]
    got = torch.ops.cerberus_amd.infer_tiles(tiles, h, 200, 216, ",".join(want))
    ref = m.infer_tiles(tiles, [200, 216])
    assert len(got) == 4
    for k, t in zip(want, got):
        assert t.dtype == ref[k].dtype and torch.equal(t, ref[k]), k
    everything = torch.ops.cerberus_amd.infer_tiles(tiles, h, 256, 256, "")
    assert len(everything) == 6 and everything[0].shape == (3, 256, 256, 2)
    # the dispatcher refuses CPU tensors: no CPU implementation exists
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.cerberus_amd.infer_tiles(tiles.cpu(), h, 256, 256, "")
    mp = torch.from_numpy(synth.nuclei_maps(320, 288, 3, 1500.0, noise=0.02)).cuda()
    lab = torch.ops.cerberus_amd.postproc(mp, "Nuclei", 1.0, True)
    lab_ref, info = postproc_device(mp, "Nuclei")
    assert lab.dtype == torch.int32 and torch.equal(lab, lab_ref)
    n = int(lab.max())
    tmap = (torch.arange(320 * 288, device="cuda").view(320, 288) % 5).to(torch.uint8)
    assert torch.equal(torch.ops.cerberus_amd.inst_table(lab, tmap, n), inst_table_device(lab, tmap, n))
    assert torch.equal(torch.ops.cerberus_amd.inst_table(lab, torch.empty(0, dtype=torch.uint8, device="cuda"), n), inst_table_device(lab, None, n))


def test_custom_op_schemas_and_fake_shapes():
    """The operators carry schemas the dispatcher can introspect and FakeTensor implementations (shape inference without a launch)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    from cerberus_amd import ops  # noqa: F401

    s = str(torch.ops.cerberus_amd.infer_tiles.default._schema)
    assert "Tensor tiles" in s and "handle" in s and "Tensor[]" in s
    with FakeTensorMode():
        t = torch.empty((5, 448, 448, 3), dtype=torch.uint8, device="cuda")
        outs = torch.ops.cerberus_amd.infer_tiles(t, 0, 144, 144, "Lumen-INST,Gland-TYPE,Patch-Class")
        assert [tuple(o.shape) for o in outs] == [(5, 144, 144, 2), (5, 144, 144), (5, 144, 144)]
        assert outs[1].dtype == torch.int64
        lab = torch.ops.cerberus_amd.postproc(torch.empty((100, 120, 2), device="cuda"), "Gland", 0.5, False)
        assert tuple(lab.shape) == (100, 120) and lab.dtype == torch.int32
