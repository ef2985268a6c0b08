"""Host-side mirror of the reference's decoder-head API (models/net_desc.py) on top of libcerberus_hip.so.

    create_model(**model_kwargs) -> NetDesc            (reference models/net_desc.py:203-204)
    NetDesc.forward(imgs NCHW float 0..255) -> OrderedDict[str, Tensor NCHW logits]   (reference :144-200)
    NetDesc.state_dict() / load_state_dict(sd, strict=True)  -- the reference's 558 key names (infer/base.py:28-45)

plus the device-resident fast path used by infer_step and the WSI driver:

    NetDesc.infer_tiles(tiles uint8 NHWC cuda, output_shape, head_name_list, ...) -> dict of CUDA tensors

All arithmetic happens in hand-written gfx950 kernels behind the C ABI; PyTorch only provides device memory and
the stream.  There is no CPU path: without a GPU / the built library every compute call raises.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .weights import reference_init_state_dict, DEFAULT_DECODER_KWARGS, make_state_dict, state_dict_schema

HEAD_NAME_MAP = {  # reference models/run_desc.py:466-473
    "Gland": "Gland-INST",
    "Gland#TYPE": "Gland-TYPE",
    "Lumen": "Lumen-INST",
    "Nuclei": "Nuclei-INST",
    "Nuclei#TYPE": "Nuclei-TYPE",
    "Patch-Class": "Patch-Class",
}


class _DeviceArray(object):
    """A float32 device buffer owned by the library, exposed to torch without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


class NetDesc(torch.nn.Module):
    """U-Net style network with a shared ResNet34 encoder and per-task decoders (reference net_desc.py:16-103)."""

    def __init__(
        self,
        encoder_backbone_name=None,
        backbone_imagenet_pretrained=False,
        fullnet_custom_pretrained=False,
        decoder_kwargs={},
        considered_tasks=[],
        subtype_gland=False,
        subtype_nuclei=False,
    ):
        super().__init__()
        if encoder_backbone_name != "resnet34":
            # the other backbones are out of scope of the MI355X path
            raise NotImplementedError("cerberus_amd implements encoder_backbone_name='resnet34' only, got %r" % (encoder_backbone_name,))
        self._init_kwargs = dict(encoder_backbone_name=encoder_backbone_name, backbone_imagenet_pretrained=backbone_imagenet_pretrained,
                                 fullnet_custom_pretrained=fullnet_custom_pretrained, decoder_kwargs=decoder_kwargs, considered_tasks=list(considered_tasks),
                                 subtype_gland=subtype_gland, subtype_nuclei=subtype_nuclei)
        self._param_version = 0  # bumped whenever the parameters change (load_state_dict, optimiser step)
        self.encoder_backbone_name = encoder_backbone_name
        self.net_code = encoder_backbone_name[:3]
        self.considered_tasks = list(considered_tasks)
        self.subtype_gland = subtype_gland
        self.subtype_nuclei = subtype_nuclei
        self.decoder_info_list = OrderedDict((k, OrderedDict(v)) for k, v in (decoder_kwargs or DEFAULT_DECODER_KWARGS).items())
        self._decoders = []  # (decoder_name, head_name, out_ch, output_key)
        for name, heads in self.decoder_info_list.items():
            if name not in self.considered_tasks:
                continue
            if name == "Patch-Class":  # the reference builds ONE Patch-Class branch whatever the dict holds: the last entry's width wins (net_desc.py:64-78)
                hname, och = list(heads.items())[-1]
                self._decoders.append((name, hname, int(och), name))
                continue
            # a decoder may carry several output heads over one trunk (models/net_desc.py:81-87, 196-198): one entry per head, output key
            # "<decoder without #suffix>-<head>"; the C side runs the trunk once (cerb_net_create)
            for hname, och in heads.items():
                key = name.split("#")[0] + "-" + hname
                if any(d[3] == key for d in self._decoders):
                    raise ValueError("two heads would both be returned as %r (the reference's output dict would keep the last one only)" % key)
                self._decoders.append((name, hname, int(och), key))
        self._schema = state_dict_schema(self.decoder_info_list, self.considered_tasks)
        if backbone_imagenet_pretrained:
            # the reference pulls torchvision's ImageNet ResNet34 here (models/backbone/__init__.py:67); no such file travels with this package
            raise NotImplementedError("backbone_imagenet_pretrained=True: load the backbone.* keys with load_state_dict(..., strict=False) instead")
        # the reference's initial state (weights_init_cnn, net_desc.py:89-103): kaiming-normal convs, identity BatchNorm.  The seeded
        # non-saturating TEST weights of cerberus_amd.weights.make_state_dict are never installed implicitly.
        self._sd = OrderedDict((k, torch.from_numpy(v)) for k, v in reference_init_state_dict(self.decoder_info_list, self.considered_tasks).items())
        self._handle = None
        self.training = False

    # ---- state dict (reference key names) -----------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        self._sync_state_dict()
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict=True):
        expected = OrderedDict((k, shp) for k, shp, _ in self._schema)
        missing = [k for k in expected if k not in state_dict]
        unexpected = [k for k in state_dict if k not in expected]
        errors = []
        if strict and missing:
            errors.append("Missing key(s) in state_dict: " + ", ".join('"%s"' % k for k in missing) + ".")
        if strict and unexpected:
            errors.append("Unexpected key(s) in state_dict: " + ", ".join('"%s"' % k for k in unexpected) + ".")
        self._sync_state_dict()
        new_sd = OrderedDict(self._sd)
        for k, shp in expected.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            v = v.detach().cpu() if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
            if tuple(v.shape) != tuple(shp):
                errors.append("size mismatch for %s: copying a param with shape %s from checkpoint, the shape in current model is %s."
                              % (k, tuple(v.shape), tuple(shp)))
                continue
            new_sd[k] = v.to(torch.int64) if k.endswith("num_batches_tracked") else v.to(torch.float32).contiguous()
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for NetDesc:\n\t" + "\n\t".join(errors))
        self._sd = new_sd
        self._param_version += 1
        for stale in ("_dev_params", "_dev_flat", "_dev_layout"):  # the optimiser's device copies (cerberus_amd.train) follow the state dict
            if hasattr(self, stale):
                delattr(self, stale)
        if self._handle is not None:  # same architecture, new values: re-pack in place (workspaces stay)
            try:
                _lib.check(_lib.lib().cerb_net_begin_reload(self._handle))
                self._load_and_finalize(self._handle)
            except Exception:
                self._release()
                raise
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- nn.Module conveniences ----------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            _lib.lib().cerb_net_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- library handle --------------------------------------------------------------------------------------
    def _ensure_handle(self):
        if self._handle is not None:
            return self._handle
        if not torch.cuda.is_available():
            raise _lib.CerberusHipError("cerberus_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        L = _lib.lib()
        n = len(self._decoders)
        names = (C.c_char_p * n)(*[d[0].encode() for d in self._decoders])
        heads = (C.c_char_p * n)(*[d[1].encode() for d in self._decoders])
        och = (C.c_int * n)(*[d[2] for d in self._decoders])
        h = C.c_void_p()
        _lib.check(L.cerb_net_create(names, heads, och, n, C.byref(h)))
        try:
            if getattr(self, "_train_packing", False):
                _lib.check(L.cerb_net_set_fold_bn(h, 0))
            self._load_and_finalize(h)
        except Exception:
            L.cerb_net_destroy(h)
            raise
        self._handle = h
        return h

    def _load_and_finalize(self, h):
        L = _lib.lib()
        self._sync_state_dict()
        for k, v in self._sd.items():
            if v.dtype != torch.float32:
                continue
            a = np.ascontiguousarray(v.numpy())
            shp = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(L.cerb_net_load_tensor(h, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
        _lib.check(L.cerb_net_finalize(h))
        if getattr(self, "_train_packing", False):
            self._apply_freeze(h)

    # ---- sub-typing fine-tune: frozen modules (models/net_desc.py:105-142) -------------------------------------------------------------
    def frozen_prefixes(self):
        """State-dict prefixes of the modules the reference's `_freeze_weight` freezes when subtype_gland / subtype_nuclei is set: backbone,
        conv_map, Patch-Class, every decoder + output head except the selected '#TYPE' one(s).  Empty when no sub-typing flag is set."""
        if not (self.subtype_gland or self.subtype_nuclei):
            return []
        keep = set()
        if self.subtype_gland:
            keep.add("Gland#TYPE")
        if self.subtype_nuclei:
            keep.add("Nuclei#TYPE")
        pre = ["backbone.", "conv_map."]
        for name, _, _, _ in self._decoders:
            if name in keep:
                continue
            pre.append("decoder_head.%s." % name)
            if name != "Patch-Class":
                pre.append("output_head.%s." % name)
        return pre

    def is_frozen(self, key):
        return any(key.startswith(p) for p in self.frozen_prefixes())

    def _apply_freeze(self, h):
        """BatchNorm layers of the frozen modules run in eval mode inside the train-mode device path (cerb_net_set_bn_eval): their running
        statistics, which no training step of this configuration ever changes, are handed over once per handle."""
        pre = self.frozen_prefixes()
        if not pre:
            return
        L = _lib.lib()
        for k, v in self._sd.items():
            if not k.endswith(".running_mean") or not any(k.startswith(p) for p in pre):
                continue
            bn = k[: -len(".running_mean")]
            if bn.startswith("backbone.fc"):
                continue
            mean = np.ascontiguousarray(v.numpy(), np.float32)
            var = np.ascontiguousarray(self._sd[bn + ".running_var"].numpy(), np.float32)
            _lib.check(L.cerb_net_set_bn_eval(h, bn.encode(), mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), int(mean.size)))

    def train(self, mode=True):
        """nn.Module.train(): a network whose handle has not been created yet is packed for training on first use (raw conv weights,
        BatchNorm with batch statistics; include/cerberus_hip.h cerb_net_set_fold_bn) and then serves forward_train only.  eval()
        on such a handle -- or train() on an inference handle -- needs a fresh NetDesc: the two packings are different device data."""
        if self._handle is not None and bool(mode) != bool(getattr(self, "_train_packing", False)):
            raise _lib.CerberusHipError("this network is already packed for %s; create another NetDesc for the other mode"
                                        % ("training" if getattr(self, "_train_packing", False) else "inference"))
        self._train_packing = bool(mode)
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def forward_train(self, tiles_u8, dropout_keep=None):
        """The reference's forward in model.train() mode (models/run_desc.py:79-86; BatchNorm with the batch's statistics).
        tiles_u8: CUDA uint8 [N, H, W, 3]; dropout_keep: CUDA bool / float [N, 512] keep mask of the Patch-Class dropout (p = 0.3), or
        None for no dropout.  -> OrderedDict head key -> logits, dense heads as [N, H, W, C] (channels last), Patch-Class [N, C]."""
        self.train(True)
        h = self._ensure_handle()
        assert tiles_u8.is_cuda and tiles_u8.dtype == torch.uint8 and tiles_u8.dim() == 4 and tiles_u8.shape[3] == 3
        tiles_u8 = tiles_u8.contiguous()
        n, hh, ww, _ = [int(v) for v in tiles_u8.shape]
        res, bufs = OrderedDict(), []
        for name, hname, och, key in self._decoders:
            t = torch.empty((n, och) if name == "Patch-Class" else (n, hh, ww, och), dtype=torch.float32, device=tiles_u8.device)
            res[key] = t
            bufs.append(t)
        io = _lib.TrainIO()
        io.tiles = tiles_u8.data_ptr()
        io.n, io.h, io.w = n, hh, ww
        scale = None
        if dropout_keep is not None:
            scale = (dropout_keep.to(tiles_u8.device).reshape(n, 512).float() / (1.0 - 0.3)).contiguous()
            io.dropout_scale = scale.data_ptr()
        arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        io.logits = arr
        stream = torch.cuda.current_stream(tiles_u8.device).cuda_stream
        with torch.cuda.device(tiles_u8.device):
            _lib.check(_lib.lib().cerb_net_forward_train(h, C.byref(io), C.c_void_p(stream)))
        return res

    def train_grads(self, tiles_u8, targets, has_target, loss_opts, dropout_keep=None, views=False, pixel_weights=None, logits_out=None, sync_losses=True):
        """One step of the reference's train_step up to all_loss.backward() (models/run_desc.py:79-170): train-mode forward, the head
        losses, the backward pass.  targets: head key -> CUDA float [N, H, W] class ids ([N] for Patch-Class); has_target: head key ->
        CUDA float [N]; loss_opts: the reference's loss_kwargs (cerberus_amd.losses.PARAMSET_LOSS).
        -> (losses: head key -> float as train_step reports them, grads: state-dict key -> CUDA float tensor shaped like the parameter;
        under the keys of the BatchNorm buffers (running_mean / running_var) it holds the step's batch mean / unbiased batch variance).
        pixel_weights: head key -> CUDA float [N, H, W], the head's '#WEIGHT-MAP' target (models/run_desc.py:111-117), optional.
        views=True returns tensors over the handle's own gradient memory instead of copies: valid until the next call on this network.
        logits_out: an (empty) dict that receives the train-mode logits of every head that has a target, channels last ([N, H, W, C]; Patch-Class
        [N, C]) -- what the reference's train_step turns into its `raw` visualisation payload.
        sync_losses=False: `losses` is (device tensor [n_heads], [head keys in its order]) instead of floats and the call returns without waiting for
        the device -- the caller (cerberus_amd.train.train_step) queues the optimiser, the running statistics and the weight re-pack underneath the
        backward pass that is still running and reads the losses last."""
        self.train(True)
        h = self._ensure_handle()
        L = _lib.lib()
        tiles_u8 = tiles_u8.contiguous()
        n, hh, ww, _ = [int(v) for v in tiles_u8.shape]
        nd = len(self._decoders)
        dev = tiles_u8.device
        keep = []
        tg, fl, cw, pwm = (C.c_void_p * nd)(), (C.c_void_p * nd)(), (C.c_void_p * nd)(), (C.c_void_p * nd)()
        ce, dc, hw = (C.c_float * nd)(), (C.c_float * nd)(), (C.c_float * nd)()
        for i, (name, hname, och, key) in enumerate(self._decoders):
            if key not in targets:
                continue
            t = targets[key].to(dev).float().contiguous()
            f = has_target[key].to(dev).float().contiguous()
            keep += [t, f]
            if pixel_weights and key in pixel_weights and key != "Patch-Class":
                pw_t = pixel_weights[key].to(dev).float().reshape(t.shape).contiguous()
                keep.append(pw_t)
                pwm[i] = pw_t.data_ptr()
            tg[i], fl[i] = t.data_ptr(), f.data_ptr()
            info = loss_opts["loss_info"][key]
            ce[i], dc[i], hw[i] = float(info["loss"].get("ce", 0)), float(info["loss"].get("dice", 0)), float(info["weight"])
            if key in loss_opts.get("class_weight", {}):
                w = torch.arange(och, dtype=torch.float32)
                for k, v in loss_opts["class_weight"][key].items():
                    w[int(k)] = float(v)
                w = w.to(dev)
                keep.append(w)
                cw[i] = w.data_ptr()
        loss = torch.zeros(nd, dtype=torch.float32, device=dev)
        io = _lib.TrainStepIO()
        io.tiles = tiles_u8.data_ptr()
        io.n, io.h, io.w = n, hh, ww
        if dropout_keep is not None:
            scale = (dropout_keep.to(dev).reshape(n, 512).float() / (1.0 - 0.3)).contiguous()
            keep.append(scale)
            io.dropout_scale = scale.data_ptr()
        io.target, io.has_target, io.class_weight, io.ce_w, io.dice_w, io.head_w = tg, fl, cw, ce, dc, hw
        if logits_out is not None:
            lg_arr = (C.c_void_p * nd)()
            for i, (name, hname, och, key) in enumerate(self._decoders):
                if key in targets:
                    logits_out[key] = torch.empty((n, och) if name == "Patch-Class" else (n, hh, ww, och), dtype=torch.float32, device=dev)
                    lg_arr[i] = logits_out[key].data_ptr()
            io.logits = lg_arr
            keep.append(lg_arr)
        io.loss_out = loss.data_ptr()
        # train_step's rule (models/run_desc.py:64-74): a decoder trains when its NAME is a substring of a target name that at least one
        # sample carries -- "Gland#TYPE" is not a substring of "Gland-TYPE", so the #TYPE decoders only ever train inside their blocks
        present = [k for k in targets if bool((has_target[k] > 0).any())]
        trained = (C.c_int * nd)(*[int(any(name in t for t in present)) for name, _, _, _ in self._decoders])
        io.decoder_trained = trained
        io.pixel_weight = pwm
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.cerb_net_train_grads(h, C.byref(io), C.c_void_p(stream)))
            grads = OrderedDict()
            frozen = self.frozen_prefixes()
            for k, v in self._sd.items():
                if v.dtype != torch.float32 or k.startswith("backbone.fc."):
                    continue
                if (k.endswith("running_mean") or k.endswith("running_var")) and frozen and any(k.startswith(p) for p in frozen):
                    continue  # eval-mode BatchNorm of a frozen module: no batch statistics, the running ones stay
                lk = k
                if k.endswith("running_mean"):  # the batch statistics behind the running-statistics update, under the buffer's own key
                    lk = k[: -len("running_mean")] + "batch_mean"
                elif k.endswith("running_var"):
                    lk = k[: -len("running_var")] + "batch_var"
                ptr, numel = C.c_void_p(), C.c_longlong()
                _lib.check(L.cerb_net_grad_lookup(h, lk.encode(), C.byref(ptr), C.byref(numel)))
                assert numel.value == v.numel(), (k, numel.value, v.numel())
                if views:  # zero-copy: a tensor over the handle's own gradient memory
                    g = torch.as_tensor(_DeviceArray(ptr.value, tuple(v.shape)), device=dev)
                else:
                    g = torch.empty(v.shape, dtype=torch.float32, device=dev)
                    _lib.check(L.cerb_copy_d2d(g.data_ptr(), ptr, 4 * v.numel(), C.c_void_p(stream)))
                grads[k] = g
        self._train_keepalive = keep  # (the call's device arguments must outlive the queued work when nobody synchronises here)
        if not sync_losses:
            return (loss, [(i, key) for i, (_, _, _, key) in enumerate(self._decoders) if key in targets]), grads
        torch.cuda.synchronize(dev)
        losses = OrderedDict((key, float(loss[i])) for i, (_, _, _, key) in enumerate(self._decoders) if key in targets)
        return losses, grads

    def load_updated_parameters(self, dev_params, flat=None, layout=None):
        """After an optimiser step: install the updated parameters (key -> CUDA tensor).  A live handle packed for training takes them
        device to device (cerb_net_update_params: copies into its raw tensors + the packing kernels) and the host state dict is brought
        up to date lazily, when somebody asks for it; otherwise they go through the host and the handle re-packs in place.
        flat / layout: dev_params as views of one buffer, layout = [(key, offset, numel, shape)] -- one transfer instead of one per key."""
        self._param_version += 1
        if self._handle is not None and getattr(self, "_train_packing", False):
            keys = [k for k in dev_params if dev_params[k].dtype == torch.float32]
            n = len(keys)
            ck, cp = (C.c_char_p * n)(*[k.encode() for k in keys]), (C.c_void_p * n)()
            for i, k in enumerate(keys):
                assert dev_params[k].is_contiguous() and dev_params[k].is_cuda, k
                cp[i] = dev_params[k].data_ptr()
            dev = dev_params[keys[0]].device
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().cerb_net_update_params(self._handle, n, ck, cp, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            self._sd_pending = (dev_params, flat, layout)  # the host copy follows on demand (state_dict, a new handle)
            return
        self._sd_pending = (dev_params, flat, layout)
        self._sync_state_dict()
        if self._handle is None:
            return
        try:  # keep the handle (workspaces); only the packed weights are rebuilt
            _lib.check(_lib.lib().cerb_net_begin_reload(self._handle))
            self._load_and_finalize(self._handle)
        except Exception:
            self._release()
            raise

    def _sync_state_dict(self):
        """Bring the host state dict up to date with the optimiser's device copies (see load_updated_parameters)."""
        pend = getattr(self, "_sd_pending", None)
        if pend is None:
            return
        self._sd_pending = None
        dev_params, flat, layout = pend
        if flat is not None:
            host = flat.detach().cpu()
            for k, o, n, shp in layout:
                self._sd[k] = host[o:o + n].view(shp)
        else:
            for k, v in dev_params.items():
                self._sd[k] = v.detach().cpu().clone()

    def handle_value(self):
        """Integer value of the finalized `cerb_net*` -- the `handle` argument of torch.ops.cerberus_amd.infer_tiles (cerberus_amd/ops.py)."""
        from . import ops

        return ops.register_net(self)

    def flops(self, n, h, w):
        return float(_lib.lib().cerb_net_flops(self._ensure_handle(), n, h, w))

    def twin(self):
        """A second NetDesc with the same parameters and switches and its OWN handle (weights packed again, its own workspace): cerberus_amd.wsi.WSIRunner
        alternates batches between the two on two streams, so that the launches of one batch that cannot fill the chip (the 16 x 16 encoder stage is
        256 work items for 256 CUs, every launch has a ramp and a tail) overlap the other batch's.  Measured: +3 % at batch 32, +1.6 % at batch 64."""
        t = NetDesc(**self._init_kwargs)
        t.load_state_dict(self.state_dict(), strict=True)
        auto = getattr(self, "_algo_is_auto", False)
        for name, value in getattr(self, "_switches", {}).items():
            getattr(t, name)(value)
        if getattr(self, "_precision_version", None) == self._param_version:  # same weights, same decision: the twin does not probe again
            t._algo_is_auto = auto
            t._precision_version = t._param_version
            if hasattr(self, "calibration_logit_absmax"):
                t.calibration_logit_absmax = self.calibration_logit_absmax
        return t

    def _remember(self, name, value):
        if not hasattr(self, "_switches"):
            self._switches = OrderedDict()
        self._switches[name] = value

    def set_head_algo(self, algo):
        """1 = all dense heads in one grouped launch, logits on 4x4x1 matrix instructions (default); 2 = round 3's grouped launch (zero-padded
        16-row instruction); 0 = one launch per head (include/cerberus_hip.h)."""
        self._remember("set_head_algo", algo)
        _lib.check(_lib.lib().cerb_net_set_head_algo(self._ensure_handle(), int(algo)))

    def set_conv_algo(self, algo):
        """Algorithm of the 3x3 stride-1 convolutions (include/cerberus_hip.h): 6 = Winograd F(4x4,3x3) for maps of 16 x 16 pixels and more,
        F(2x2,3x3) below (default); 5 / 7 = F(4x4) everywhere with conv_wino4 / conv_wino4b; 1 = F(2x2); 0 = direct implicit GEMM."""
        self._remember("set_conv_algo", algo)
        self._algo_is_auto = False  # the caller's choice is final (_auto_precision leaves it alone)
        _lib.check(_lib.lib().cerb_net_set_conv_algo(self._ensure_handle(), int(algo)))

    def set_planar(self, enable=True):
        """The two last decoder levels: 1 / True = tile-planar layout (conv_wino4p.hip), 0 / False = NHWC (conv_wino4.hip).  Bit-identical outputs."""
        self._remember("set_planar", enable)
        _lib.check(_lib.lib().cerb_net_set_planar(self._ensure_handle(), int(enable)))

    def set_packed_items(self, enable=True):
        """conv_wino4b.hip on maps that are not whole 16 x 16 blocks (28^2 / 56^2 of a 448-pixel patch): 1 / True = 16 consecutive tiles per work item
        (default), 0 / False = 16 x 16-pixel blocks with padding tiles.  Bit-identical outputs (include/cerberus_hip.h)."""
        self._remember("set_packed_items", enable)
        _lib.check(_lib.lib().cerb_net_set_packed_items(self._ensure_handle(), int(enable)))

    def set_crop_roi(self, enable=True):
        """Compute only what the centre crop keeps in the decoders / heads (default on; include/cerberus_hip.h)."""
        self._remember("set_crop_roi", enable)
        _lib.check(_lib.lib().cerb_net_set_crop_roi(self._ensure_handle(), int(bool(enable))))

    def profile(self, enable=True):
        _lib.check(_lib.lib().cerb_net_profile_enable(self._ensure_handle(), int(enable)))

    def profile_records(self):
        """[(layer name, kernel family, flops, ms)] of the last forward run with profiling enabled."""
        L, h = _lib.lib(), self._ensure_handle()
        out = []
        nm, kn = C.create_string_buffer(128), C.create_string_buffer(128)
        fl, ms = C.c_double(), C.c_float()
        for i in range(L.cerb_net_profile_count(h)):
            _lib.check(L.cerb_net_profile_get(h, i, nm, 128, kn, 128, C.byref(fl), C.byref(ms)))
            out.append((nm.value.decode(), kn.value.decode(), fl.value, ms.value))
        return out

    # (CERB_LOGIT_SATURATION overrides the bar: an operator who wants the guard tighter -- or a test that wants it to fire)
    LOGIT_SATURATION = float(__import__("os").environ.get("CERB_LOGIT_SATURATION", "100.0"))  # largest |logit| the F(4x4,3x3) default is held to the 1e-4 contract for (fixtures at 4 .. 17, 30 and 80: DESIGN.md par.5; the reference's default init: 650 .. 2200)

    def prepare(self, device=None):
        """Decide the 3x3 convolution algorithm from the WEIGHTS, once per parameter version, at LOAD time (not inside the first
        forward of a stream).  F(4x4,3x3) Winograd amplifies fp32 rounding ~3x more than a direct convolution: on every weight family whose dense
        logits stay below LOGIT_SATURATION that is 1e-7 .. 3e-5 on the probability maps (tests/test_net_gpu.py: calibration logits 4 .. 17, 30, 80;
        noise, stain-field, half-glass, white and black tiles), inside the 1e-4 contract measured from the reference's float64 evaluation; under the
        reference's DEFAULT initialisation (logits in the thousands, a step-function softmax) it was 5.8e-4 where the reference's own fp32 is 4.6e-4.
        So the network is probed with ONE fixed seeded tile on the device's DEFAULT stream -- a function of the weights alone: every rank, shard and
        handle of the same weights decides the same way -- and when a dense head's logits exceed LOGIT_SATURATION the handle runs F(2x2,3x3)
        (cerb_net_set_conv_algo(1): closer to float64 than the direct convolution there, ~1.3x slower).  An explicit set_conv_algo() is final;
        CERB_AUTO_PRECISION=0 switches the probe off.  What the probe cannot see -- real tiles that drive a trained model's logits past the bar --
        is watched per batch by the head kernels (watch_logits / cerb_forward_io.logit_absmax).
        Returns {"conv_algo", "calibration_logit_absmax", "auto", "probed"}; loaders call it and log it (run_infer_*.py), a handle nobody prepared
        calls it from its first forward."""
        import os

        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if getattr(self, "_precision_version", None) == self._param_version or getattr(self, "_train_packing", False):
            return self.precision_decision()
        self._precision_version = self._param_version
        auto = getattr(self, "_algo_is_auto", False)
        if ("set_conv_algo" in getattr(self, "_switches", {}) and not auto) or os.environ.get("CERB_AUTO_PRECISION", "1") == "0":
            return self.precision_decision()
        if auto:  # new weights: decide again from the default algorithm
            self._switches.pop("set_conv_algo", None)
            self._algo_is_auto = False
            _lib.check(_lib.lib().cerb_net_set_conv_algo(self._ensure_handle(), 6))
        tile = torch.from_numpy(np.random.RandomState(20240229).randint(0, 256, (1, 256, 256, 3)).astype(np.uint8))
        cur = torch.cuda.current_stream(device)
        dflt = torch.cuda.default_stream(device)
        with torch.cuda.stream(dflt):  # never on a caller's side stream; the host waits for the result below, so the handle is idle again on return
            dflt.wait_stream(cur)
            lg = self.forward(tile.to(device))
            amax = max([float(v.abs().max()) for k, v in lg.items() if k != "Patch-Class"] or [0.0])
        self.calibration_logit_absmax = amax
        import logging

        log = logging.getLogger("cerberus_amd")
        if amax > self.LOGIT_SATURATION:
            log.warning("calibration logits reach %.0f (saturated softmax): 3x3 convolutions run F(2x2,3x3) instead of F(4x4,3x3) on this handle", amax)
            self._remember("set_conv_algo", 1)
            self._algo_is_auto = True
            _lib.check(_lib.lib().cerb_net_set_conv_algo(self._ensure_handle(), 1))
        else:
            log.info("calibration logits reach %.1f: 3x3 convolutions run F(4x4,3x3)", amax)
        return self.precision_decision()

    def precision_decision(self):
        """What prepare() decided (or the caller set): the algorithm the next forward runs, the calibration measurement, who chose."""
        return {"conv_algo": int(getattr(self, "_switches", {}).get("set_conv_algo", 6)),
                "calibration_logit_absmax": getattr(self, "calibration_logit_absmax", None),
                "auto": bool(getattr(self, "_algo_is_auto", False)),
                "probed": getattr(self, "calibration_logit_absmax", None) is not None}

    _auto_precision = prepare  # (round 5's name)

    # ---- the data-aware half of the guard: the head kernels report the largest |logit| they produce ------------------------------------------
    def watch_logits(self, enable=True, device=None):
        """Every forward from now on raises a device word per dense head to the largest |logit| it produced (cerb_forward_io.logit_absmax: one
        wave-level maximum and at most one atomic per wave, nothing on the host) -- read with logit_absmax().  Callers that want the maximum
        per batch (cerberus_amd.wsi.WSIRunner) hand _run their own row instead."""
        if not enable:
            self._logit_watch = None
            return
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._logit_watch = torch.zeros(len(self._decoders), dtype=torch.int32, device=device)

    def logit_absmax(self, reset=True, words=None):
        """OrderedDict head key -> largest |logit| since the last reset (host sync).  words: another int32 [.., n_decoders] tensor of such words
        (a per-batch log): returns a float32 array of the same shape instead."""
        if words is not None:
            return words.detach().cpu().numpy().view(np.float32)
        w = getattr(self, "_logit_watch", None)
        if w is None:
            raise RuntimeError("logit_absmax: call watch_logits() first")
        vals = w.cpu().numpy().view(np.float32)
        if reset:
            w.zero_()
        return OrderedDict((d[3], float(v)) for d, v in zip(self._decoders, vals) if d[0] != "Patch-Class")

    def _run(self, tiles_u8, out_h, out_w, outs, logits, tile_off=None, tile_stride=0, row_stride=0, type_is_u8=False, feats=None, logit_absmax=None):
        # uint8 tiles (what infer_step receives), or float32 NHWC pixel values for forward() on inputs that are not whole numbers in 0..255
        assert tiles_u8.is_cuda and tiles_u8.dtype in (torch.uint8, torch.float32) and tiles_u8.dim() == 4 and tiles_u8.shape[3] == 3
        if getattr(self, "_precision_version", None) != self._param_version:
            self.prepare(tiles_u8.device)
        tiles_u8 = tiles_u8.contiguous()
        n, h, w, _ = tiles_u8.shape
        nd = len(self._decoders)
        io = _lib.ForwardIO()
        if tiles_u8.dtype == torch.uint8:
            io.tiles = tiles_u8.data_ptr()
        else:
            io.tiles, io.tiles_f32 = None, tiles_u8.data_ptr()
        io.n, io.h, io.w, io.out_h, io.out_w = n, h, w, out_h, out_w
        out_arr = (C.c_void_p * nd)(*[(t.data_ptr() if t is not None else None) for t in outs])
        io.out = out_arr
        if logits is not None:
            lg_arr = (C.c_void_p * nd)(*[(t.data_ptr() if t is not None else None) for t in logits])
            io.logits = lg_arr
        if tile_off is not None:
            assert tile_off.is_cuda and tile_off.dtype == torch.int64 and tile_off.numel() == n
            io.tile_off = tile_off.data_ptr()
        io.tile_stride, io.row_stride, io.type_is_u8 = int(tile_stride), int(row_stride), int(bool(type_is_u8))
        if feats is not None:
            f_arr = (C.c_void_p * 6)(*[(t.data_ptr() if t is not None else None) for t in feats])
            io.feats = f_arr
        if logit_absmax is None:
            logit_absmax = getattr(self, "_logit_watch", None)
        if logit_absmax is not None:
            assert logit_absmax.is_cuda and logit_absmax.dtype == torch.int32 and logit_absmax.numel() == nd and logit_absmax.is_contiguous()
            io.logit_absmax = logit_absmax.data_ptr()
        stream = torch.cuda.current_stream(tiles_u8.device).cuda_stream
        with torch.cuda.device(tiles_u8.device):
            _lib.check(_lib.lib().cerb_net_forward(self._ensure_handle(), C.byref(io), C.c_void_p(stream)))

    # ---- reference-compatible forward: logits ----------------------------------------------------------------
    def forward(self, imgs, train_decoder_list=[]):
        """imgs: NCHW float tensor (any values; the reference divides by 255 whatever they are, models/net_desc.py:144-147), or uint8 NHWC
        (what infer_step passes on after `.float()`, run_desc.py:440-449).  Returns OrderedDict key -> NCHW fp32 logits on the GPU
        (reference net_desc.py:144-200).  Whole numbers in 0..255 take the uint8 stem (a quarter of the input bytes), anything else the
        float-input instantiation of the same kernel (`cerb_forward_io.tiles_f32`): same `x / 255.0f` in fp32, same arithmetic after it."""
        self._ensure_handle()
        if imgs.dtype == torch.uint8 and imgs.shape[-1] == 3:
            tiles = imgs
        else:
            imgs = imgs.float()
            q = imgs.round()
            if torch.equal(q, imgs) and float(imgs.min()) >= 0 and float(imgs.max()) <= 255:
                tiles = imgs.permute(0, 2, 3, 1).to(torch.uint8)
            else:
                tiles = imgs.permute(0, 2, 3, 1)
        tiles = tiles.cuda().contiguous()
        n, h, w, _ = tiles.shape
        lg = []
        for name, hname, och, key in self._decoders:
            if name == "Patch-Class":
                lg.append(torch.empty((n, och), dtype=torch.float32, device=tiles.device))
            else:
                lg.append(torch.empty((n, h, w, och), dtype=torch.float32, device=tiles.device))
        self._run(tiles, h, w, [None] * len(self._decoders), lg)
        out = OrderedDict()
        for (name, hname, och, key), t in zip(self._decoders, lg):
            out[key] = t.view(n, och, 1, 1) if name == "Patch-Class" else t.permute(0, 3, 1, 2)
        return out

    # ---- device-resident output wrapper (softmax / crop / argmax fused into the head kernels) ---------------
    def infer_tiles(self, tiles_u8, output_shape, head_name_list=None, type_dtype=torch.int64):
        """F7 of SURVEY.md par.8a on device: returns OrderedDict head-key -> CUDA tensor
        ('*-INST' (N,oh,ow,2) float32; '*-TYPE' (N,oh,ow) int64|uint8; 'Patch-Class' (N,oh,ow) float32)."""
        if not isinstance(output_shape, (list, tuple)):
            output_shape = [output_shape, output_shape]
        oh, ow = int(output_shape[0]), int(output_shape[1])
        self._ensure_handle()  # raises CerberusHipError when there is no GPU / no built library
        tiles_u8 = tiles_u8.cuda()
        n = tiles_u8.shape[0]
        wanted = None if head_name_list is None else set(HEAD_NAME_MAP[h] for h in head_name_list)
        outs, res = [], OrderedDict()
        for name, hname, och, key in self._decoders:
            if wanted is not None and key not in wanted:
                outs.append(None)
                continue
            if hname == "INST":
                t = torch.empty((n, oh, ow, 2), dtype=torch.float32, device=tiles_u8.device)
            elif hname == "TYPE":
                t = torch.empty((n, oh, ow), dtype=type_dtype, device=tiles_u8.device)
            else:
                t = torch.empty((n, oh, ow), dtype=torch.float32, device=tiles_u8.device)
            outs.append(t)
            res[key] = t
        self._run(tiles_u8, oh, ow, outs, None, type_is_u8=(type_dtype == torch.uint8))
        if head_name_list is not None:  # reference orders the result by head_name_list (run_desc.py:475-492)
            res = OrderedDict((HEAD_NAME_MAP[h], res[HEAD_NAME_MAP[h]]) for h in head_name_list)
        return res

    def encoder_features(self, tiles_u8):
        """Test hook: NHWC dumps of x0, x1, x2, x3, conv_map(x4), x4."""
        self._ensure_handle()
        tiles_u8 = tiles_u8.cuda().contiguous()
        n, h, w, _ = tiles_u8.shape
        shp = [(h, w, 64), (h // 2, w // 2, 64), (h // 4, w // 4, 128), (h // 8, w // 8, 256), (h // 16, w // 16, 256), (h // 16, w // 16, 512)]
        feats = [torch.empty((n,) + s, dtype=torch.float32, device=tiles_u8.device) for s in shp]
        self._run(tiles_u8, h, w, [None] * len(self._decoders), None, feats=feats)
        return feats


def create_model(**kwargs):
    return NetDesc(**kwargs)
