"""Per-head training loss of the reference's train_step on the GPU (cerb_head_loss; models/run_desc.py:88-170 there) -- the loss
piece of BASELINE configs[4]; the backward pass and the optimiser step that consume its gradients are in cerberus_amd/train.py."""
import ctypes as C

import torch

from . import _lib

PARAMSET_LOSS = {  # the reference's models/paramset.yml:13-31 (configuration data)
    "loss_info": {"Lumen-INST": {"weight": 1.5, "loss": {"ce": 1}}, "Gland-INST": {"weight": 1.4, "loss": {"ce": 1}},
                  "Nuclei-INST": {"weight": 1, "loss": {"ce": 1}}, "Nuclei-TYPE": {"weight": 0, "loss": {"ce": 1, "dice": 1}},
                  "Gland-TYPE": {"weight": 1, "loss": {"ce": 1, "dice": 1}}, "Patch-Class": {"weight": 0.4, "loss": {"ce": 1}}},
    "class_weight": {"Gland-TYPE": {1: 1, 2: 1}, "Nuclei-TYPE": {1: 12, 2: 1, 3: 2, 4: 6, 5: 12, 6: 2}},
}


def head_loss(head_name, logits, target, has_target, loss_opts=PARAMSET_LOSS, channels_last=False, with_grad=True, pixel_weight=None):
    """logits: CUDA float32 [N, C, H, W] (or [N, H, W, C] with channels_last); target: CUDA float32 [N, H, W] class ids;
    has_target: CUDA float32 [N]; pixel_weight: the head's '#WEIGHT-MAP' target, CUDA float32 [N, H, W], or None (ones).
    -> (loss: 0-d CUDA float32, dlogits like logits or None)"""
    if not torch.cuda.is_available():
        raise _lib.CerberusHipError("cerberus_amd needs a ROCm GPU; there is no CPU fallback")
    L = _lib.lib()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4
    if channels_last:
        n, h, w, c = [int(v) for v in logits.shape]
        sn, sy, sx, sc = logits.stride()
    else:
        n, c, h, w = [int(v) for v in logits.shape]
        sn, sc, sy, sx = logits.stride()
    target = target.reshape(n, h, w).contiguous().float()
    has_target = has_target.contiguous().float()
    info = loss_opts["loss_info"][head_name]
    cw = None
    if head_name in loss_opts.get("class_weight", {}):
        cw = torch.arange(c, dtype=torch.float32)  # classes that are not listed keep their own id as weight (get_class_wmap)
        for k, v in loss_opts["class_weight"][head_name].items():
            cw[int(k)] = float(v)
        cw = cw.to(logits.device)
    pw = None if pixel_weight is None else pixel_weight.reshape(n, h, w).contiguous().float()
    dl = torch.empty_like(logits) if with_grad else None
    loss = torch.zeros((), dtype=torch.float32, device=logits.device)
    ws = torch.empty(int(L.cerb_head_loss_workspace_bytes(n, h, w)), dtype=torch.uint8, device=logits.device)
    st = torch.cuda.current_stream(logits.device).cuda_stream
    with torch.cuda.device(logits.device):
        _lib.check(L.cerb_head_loss_wmap(logits.data_ptr(), sn, sc, sy, sx, target.data_ptr(), has_target.data_ptr(), n, h, w, c,
                                         None if cw is None else cw.data_ptr(), None if pw is None else pw.data_ptr(), float(info["loss"].get("ce", 0)),
                                         float(info["loss"].get("dice", 0)), float(info["weight"]), int(head_name == "Patch-Class"), loss.data_ptr(),
                                         None if dl is None else dl.data_ptr(), ws.data_ptr(), ws.numel(), C.c_void_p(st)))
    return loss, dl
