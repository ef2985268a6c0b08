"""PyTorch custom operators over the C ABI (`torch.ops.cerberus_amd.*`).

north_star: "Python host code ... calls hand-written HIP kernels (PyTorch-ROCm custom ops over a thin C-ABI)".  The C ABI
(include/cerberus_hip.h) is the boundary; these operators are the torch-visible names for its three entry points on the inference path, so
that a caller holding torch tensors reaches the kernels through the dispatcher (schemas, device checks, `torch.ops` discoverability,
FakeTensor shape inference for tracing tools) instead of through ctypes by hand:

    torch.ops.cerberus_amd.infer_tiles(tiles, handle, out_h, out_w, heads)   uint8 [N, H, W, 3] -> the F7 outputs of infer_step, on the device
                                                                            (models/run_desc.py:439-502), in `heads` order
    torch.ops.cerberus_amd.postproc(inst, tissue, ds_factor, exact_ties)     [H, W, 2] float -> int32 label map (loader/postproc.py:269-381)
    torch.ops.cerberus_amd.inst_table(labels, type_map, n_inst)              label map -> int64 [n_inst, 16] table (loader/postproc.py:12-98)

`handle` is the integer value of a finalized `cerb_net*` (NetDesc.handle_value()); a Python object cannot cross an operator schema.  Each
operator only has a CUDA implementation: on CPU tensors the dispatcher raises NotImplementedError -- there is no CPU fallback here either.
The NetDesc / postproc mirrors keep calling the C ABI directly (one ctypes call per batch; the dispatcher would add nothing); the
operators are the public torch-level surface and tests/test_ops_gpu.py holds them to the mirrors bit for bit."""
import ctypes as C
import weakref

import torch

from . import _lib

_HEAD_KIND = {"INST": 0, "TYPE": 1, "OUT": 2}
_nets = weakref.WeakValueDictionary()  # handle value -> NetDesc, filled by NetDesc.handle_value; weak: a dropped model frees its handle and workspace, and a
# recycled handle value can never resolve to a dead model's decoder list


def register_net(model):
    h = model._ensure_handle()
    _nets[int(h.value)] = model
    return int(h.value)


@torch.library.custom_op("cerberus_amd::infer_tiles", mutates_args=(), device_types="cuda")
def infer_tiles(tiles: torch.Tensor, handle: int, out_h: int, out_w: int, heads: str) -> list[torch.Tensor]:
    """heads: comma-separated head keys ('Nuclei-INST,Nuclei-TYPE'), '' = all in model order.  INST -> float32 [N, oh, ow, 2],
    TYPE -> int64 [N, oh, ow], Patch-Class -> float32 [N, oh, ow]."""
    m = _nets.get(int(handle))
    if m is None:
        raise _lib.CerberusHipError("cerberus_amd::infer_tiles: unknown handle (use NetDesc.handle_value())")
    if tiles.dtype != torch.uint8 or tiles.dim() != 4 or tiles.shape[3] != 3:
        raise ValueError("cerberus_amd::infer_tiles: tiles must be uint8 [N, H, W, 3]")
    want = [h for h in heads.split(",") if h] or [d[3] for d in m._decoders]
    keys = [d[3] for d in m._decoders]
    for h in want:
        if h not in keys:
            raise ValueError("cerberus_amd::infer_tiles: the model has no head %r" % h)
    n = int(tiles.shape[0])
    outs, res = [], {}
    for name, hname, och, key in m._decoders:
        if key not in want:
            outs.append(None)
            continue
        t = torch.empty((n, out_h, out_w, 2) if hname == "INST" else (n, out_h, out_w), dtype=torch.int64 if hname == "TYPE" else torch.float32,
                        device=tiles.device)
        outs.append(t)
        res[key] = t
    m._run(tiles, int(out_h), int(out_w), outs, None)
    return [res[h] for h in want]


@infer_tiles.register_fake
def _(tiles, handle, out_h, out_w, heads):
    m = _nets.get(int(handle))
    want = [h for h in heads.split(",") if h] or ([d[3] for d in m._decoders] if m is not None else [])
    n = tiles.shape[0]
    out = []
    for h in want:
        if h.endswith("INST"):
            out.append(tiles.new_empty((n, out_h, out_w, 2), dtype=torch.float32))
        elif h.endswith("TYPE"):
            out.append(tiles.new_empty((n, out_h, out_w), dtype=torch.int64))
        else:
            out.append(tiles.new_empty((n, out_h, out_w), dtype=torch.float32))
    return out


@torch.library.custom_op("cerberus_amd::postproc", mutates_args=(), device_types="cuda")
def postproc(inst: torch.Tensor, tissue: str, ds_factor: float, exact_ties: bool) -> torch.Tensor:
    from .postproc import postproc_device

    labels, _ = postproc_device(inst, tissue, float(ds_factor), exact_ties=bool(exact_ties))
    return labels


@postproc.register_fake
def _(inst, tissue, ds_factor, exact_ties):
    return inst.new_empty((inst.shape[0], inst.shape[1]), dtype=torch.int32)


@torch.library.custom_op("cerberus_amd::inst_table", mutates_args=(), device_types="cuda")
def inst_table(labels: torch.Tensor, type_map: torch.Tensor, n_inst: int) -> torch.Tensor:
    """type_map: uint8 [H, W], or an EMPTY tensor for 'no type map' (operator schemas have no optional-by-None tensors with defaults here)."""
    from .postproc import inst_table_device

    return inst_table_device(labels, type_map if type_map.numel() else None, int(n_inst))


@inst_table.register_fake
def _(labels, type_map, n_inst):
    return labels.new_empty((max(int(n_inst), 0), 16), dtype=torch.int64)
