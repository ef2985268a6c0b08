"""Mirror of the reference's inference run step (models/run_desc.py:439-502)."""
from collections import OrderedDict

import torch

from .net_desc import HEAD_NAME_MAP  # noqa: F401  (re-exported: the reference keeps head_name_map next to infer_step, run_desc.py:466-473)


def infer_step(img_list, model, output_shape, head_name_list):
    """Same signature / return protocol as the reference's infer_step:

    img_list        uint8 NHWC tensor (N,H,W,3) on host or device
    model           cerberus_amd.net_desc.NetDesc
    output_shape    int or [h, w] -- centre crop applied to every dense head (cropping_center, misc/utils.py:94-104)
    head_name_list  decoder names (`model_args["considered_tasks"]`, infer/base.py:52)
    returns         list (one per sample) of dict head-key -> numpy array:
                    '*-INST' (oh,ow,2) float32, '*-TYPE' (oh,ow) int64, 'Patch-Class' (oh,ow) float32

    The whole of forward + softmax + channel slice + crop + argmax runs in the HIP kernels; the only host work
    left is the final .cpu().numpy() the reference also does (run_desc.py:492).
    """
    if not isinstance(img_list, torch.Tensor):
        img_list = torch.as_tensor(img_list)
    if img_list.dtype != torch.uint8:
        raise TypeError("infer_step expects the uint8 NHWC batch the reference's DataLoader yields, got %s" % img_list.dtype)
    img_list = img_list.to("cuda")
    model.eval()
    with torch.no_grad():
        dev = model.infer_tiles(img_list, output_shape, head_name_list)
    sub_pred_dict = OrderedDict((k, v.cpu().numpy()) for k, v in dev.items())
    batch_output_list = []
    for sample_idx in range(img_list.shape[0]):
        batch_output_list.append({k: v[sample_idx] for k, v in sub_pred_dict.items()})
    return batch_output_list
