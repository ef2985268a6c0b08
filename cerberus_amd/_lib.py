"""ctypes binding of libcerberus_hip.so (include/cerberus_hip.h).  Fails loudly: there is NO CPU / eager fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcerberus_hip.so")

EXPORTS = [
    "cerb_version", "cerb_last_error", "cerb_net_create", "cerb_net_destroy", "cerb_net_load_tensor",
    "cerb_net_finalize", "cerb_net_forward", "cerb_net_flops", "cerb_device_bytes_held", "cerb_pp_workspace_bytes", "cerb_postproc_nuclei",
    "cerb_postproc_gland", "cerb_postproc_lumen", "cerb_mask_lumen_by_gland", "cerb_event_create",
    "cerb_event_record", "cerb_event_elapsed_ms", "cerb_event_destroy", "cerb_net_profile_enable",
    "cerb_net_profile_count", "cerb_net_profile_get", "cerb_net_set_conv_algo", "cerb_net_set_head_algo", "cerb_net_set_planar", "cerb_net_set_packed_items", "cerb_net_set_crop_roi", "cerb_net_set_fold_bn", "cerb_net_set_bn_eval", "cerb_net_begin_reload", "cerb_net_update_params", "cerb_net_forward_train", "cerb_net_train_grads", "cerb_net_grad_lookup", "cerb_copy_d2d", "cerb_adam_step", "cerb_adam_step_multi", "cerb_synth_slide", "cerb_gather_patches", "cerb_downsample2_inst", "cerb_half_size", "cerb_downsample2_inst_region", "cerb_pclass_tissue_map", "cerb_resample_box", "cerb_resample_area", "cerb_label_mask", "cerb_inst_table", "cerb_relabel", "cerb_head_loss_workspace_bytes", "cerb_head_loss", "cerb_head_loss_wmap", "cerb_inst_contour_start", "cerb_inst_contour_start_workspace_bytes", "cerb_inst_contour_count", "cerb_inst_contour_points",
]


class ForwardIO(C.Structure):
    _fields_ = [
        ("tiles", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int),
        ("out_h", C.c_int), ("out_w", C.c_int),
        ("out", C.POINTER(C.c_void_p)),
        ("logits", C.POINTER(C.c_void_p)),
        ("tile_off", C.c_void_p),
        ("tile_stride", C.c_longlong),
        ("row_stride", C.c_longlong),
        ("type_is_u8", C.c_int),
        ("feats", C.POINTER(C.c_void_p)),
        ("tiles_f32", C.c_void_p),
        ("logit_absmax", C.c_void_p),
    ]


class TrainIO(C.Structure):
    _fields_ = [
        ("tiles", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int),
        ("dropout_scale", C.c_void_p),
        ("logits", C.POINTER(C.c_void_p)),
    ]


class TrainStepIO(C.Structure):
    _fields_ = [
        ("tiles", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int),
        ("dropout_scale", C.c_void_p),
        ("target", C.POINTER(C.c_void_p)),
        ("has_target", C.POINTER(C.c_void_p)),
        ("class_weight", C.POINTER(C.c_void_p)),
        ("ce_w", C.POINTER(C.c_float)),
        ("dice_w", C.POINTER(C.c_float)),
        ("head_w", C.POINTER(C.c_float)),
        ("loss_out", C.c_void_p),
        ("logits", C.POINTER(C.c_void_p)),
        ("decoder_trained", C.POINTER(C.c_int)),
        ("pixel_weight", C.POINTER(C.c_void_p)),
    ]


class CerberusHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (building is `python -m cerberus_amd.build` / __graft_entry__.build())."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("CERB_DEV_LIB") == "1":
        # the developers' build of the same sources (-DCERB_DEV_SWITCHES: the CERB_* A/B environment switches inside the schedules exist only there);
        # the A/B tests run in a child process that sets this (tests/conftest.py: dev_switches) -- the product library carries none of them
        LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libcerberus_hip_dev.so")
    if not os.path.exists(LIB_PATH):
        raise CerberusHipError(
            "libcerberus_hip.so not found at %s -- run `python -m cerberus_amd.build` (needs hipcc). "
            "There is no CPU fallback for the Cerberus HIP path." % LIB_PATH
        )
    # PyTorch-ROCm ships its own copy of the HIP runtime.  Whichever libamdhip64 is loaded FIRST serves every later user of that
    # soname; if this library came first it would bring /opt/rocm's runtime in beside torch's, and the second runtime in a process
    # does not get the device ("no ROCm-capable device is detected" from hipMalloc while torch.cuda works).  Device memory, streams
    # and torch.distributed are torch's here, so torch's runtime is the one to share: import it before the dlopen.
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    L.cerb_last_error.restype = C.c_char_p
    L.cerb_net_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.cerb_net_destroy.argtypes = [C.c_void_p]
    L.cerb_net_destroy.restype = None
    L.cerb_net_load_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    L.cerb_net_finalize.argtypes = [C.c_void_p]
    L.cerb_net_forward.argtypes = [C.c_void_p, C.POINTER(ForwardIO), C.c_void_p]
    L.cerb_net_flops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.cerb_net_flops.restype = C.c_double
    L.cerb_device_bytes_held.argtypes = []
    L.cerb_device_bytes_held.restype = C.c_size_t
    L.cerb_resample_box.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
    L.cerb_resample_area.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + \
        [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    L.cerb_pp_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.cerb_pp_workspace_bytes.restype = C.c_size_t
    L.cerb_postproc_nuclei.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    for f in (L.cerb_postproc_gland, L.cerb_postproc_lumen):
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_size_t, C.c_void_p]
    L.cerb_mask_lumen_by_gland.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    L.cerb_net_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_profile_count.argtypes = [C.c_void_p]
    L.cerb_net_set_conv_algo.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_set_planar.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_set_packed_items.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_set_bn_eval.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    L.cerb_net_set_head_algo.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_set_crop_roi.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_set_fold_bn.argtypes = [C.c_void_p, C.c_int]
    L.cerb_net_begin_reload.argtypes = [C.c_void_p]
    L.cerb_net_update_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.c_void_p]
    L.cerb_net_forward_train.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.cerb_net_train_grads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.cerb_adam_step_multi.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong),
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.cerb_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.cerb_copy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cerb_net_grad_lookup.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]
    L.cerb_net_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.cerb_synth_slide.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_uint32, C.c_void_p]
    L.cerb_gather_patches.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p]
    L.cerb_downsample2_inst.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.cerb_half_size.argtypes = [C.c_int]
    L.cerb_downsample2_inst_region.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]
    L.cerb_pclass_tissue_map.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.cerb_label_mask.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cerb_head_loss_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.cerb_head_loss_workspace_bytes.restype = C.c_size_t
    L.cerb_head_loss.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cerb_head_loss_wmap.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cerb_inst_contour_start_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.cerb_inst_contour_start_workspace_bytes.restype = C.c_size_t
    L.cerb_inst_contour_start.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cerb_inst_table.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.cerb_relabel.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    L.cerb_inst_contour_count.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cerb_inst_contour_points.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cerb_event_create.argtypes = [C.POINTER(C.c_void_p)]
    L.cerb_event_record.argtypes = [C.c_void_p, C.c_void_p]
    L.cerb_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.cerb_event_destroy.argtypes = [C.c_void_p]
    _lib = L
    return L


class CerberusHipAllocError(CerberusHipError):
    """CERB_ERR_ALLOC (return code 2): a workspace / tape allocation did not fit the device."""


def check(rc):
    if rc != 0:
        msg = lib().cerb_last_error().decode("utf-8", "replace")
        raise (CerberusHipAllocError if rc == 2 else CerberusHipError)(msg)
