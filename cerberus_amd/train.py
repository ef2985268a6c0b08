"""Mirror of the reference's train_step (models/run_desc.py:25-230) on the GPU: train-mode forward, the six head losses, the backward
pass, Adam (models/opt.py:47-58) and the BatchNorm running statistics -- BASELINE configs[4].  Every piece is checked against
the reference's own train_step (tests/test_train_loss_gpu.py).  After the optimiser the updated parameters go back into the state dict
in one transfer and the handle re-packs its conv weights on the device (cerb_net_begin_reload).

Multi-GPU: one process per GPU, `allreduce_grads` averages the gradients over ranks in buckets (backend "nccl" = RCCL over xGMI,
113.5 MB per step) between the backward pass and the optimiser -- the DistributedDataParallel arithmetic of a reference that only
has DataParallel (infer/base.py:46; it ships no training launcher)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib


class Adam(object):
    """torch.optim.Adam's state and arithmetic (no weight decay, no amsgrad) over a state dict of CUDA tensors."""

    def __init__(self, lr=1.0e-3, betas=(0.9, 0.999), eps=1.0e-8):
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.state = {}

    def step(self, params, grads):
        """params / grads: key -> CUDA float tensor (contiguous); params are updated in place.  One launch for all tensors."""
        L = _lib.lib()
        self.step_count += 1
        keys = list(grads.keys())
        if not keys:
            return
        dev = params[keys[0]].device
        fresh = [k for k in keys if k not in self.state]
        if fresh:  # moments of the new tensors as views of two flat buffers (one allocation, one memset each)
            sizes = [(params[k].numel() + 63) // 64 * 64 for k in fresh]
            fm, fv = torch.zeros(sum(sizes), dtype=torch.float32, device=dev), torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
            o = 0
            for k, sz in zip(fresh, sizes):
                n = params[k].numel()
                self.state[k] = (fm[o:o + n].view(params[k].shape), fv[o:o + n].view(params[k].shape))
                o += sz
        n = len(keys)
        gs = [grads[k].contiguous() for k in keys]
        pp, gg, mm, vv = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        nn = (C.c_longlong * n)()
        for i, k in enumerate(keys):
            p = params[k]
            assert p.is_contiguous() and p.dtype == torch.float32 and gs[i].numel() == p.numel(), k
            pp[i], gg[i], mm[i], vv[i], nn[i] = p.data_ptr(), gs[i].data_ptr(), self.state[k][0].data_ptr(), self.state[k][1].data_ptr(), p.numel()
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.cerb_adam_step_multi(n, pp, gg, mm, vv, nn, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, C.c_void_p(st)))


class StepLR(object):
    """torch.optim.lr_scheduler.StepLR over cerberus_amd.train.Adam (models/opt.py:55-58: StepLR(opt, 75000), gamma 0.1):
    after `step()` number e the rate is base_lr * gamma ** (e // step_size)."""

    def __init__(self, optimizer, step_size, gamma=0.1):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lr, self.last_epoch = optimizer.lr, 0

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)

    def get_last_lr(self):
        return [self.optimizer.lr]


def allreduce_grads(grads, dist, world_size, bucket_bytes=32 << 20):
    """Average gradients over ranks in flat buckets (a ring all-reduce over xGMI is bound per link: few large messages beat 300 small
    ones).  grads: key -> tensor (any device); in place.  dist: torch.distributed or None."""
    if dist is None:  # (a communicator of ONE rank still reduces: bench.py --force-dist / the world-1 RCCL tests run the N > 1 code path on one device)
        return
    keys, bucket, size = list(grads.keys()), [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([grads[k].reshape(-1) for k in bucket])
        if flat.is_cuda and dist.get_backend() == "gloo":  # plumbing tests only: gloo reduces host memory
            host = flat.cpu()
            dist.all_reduce(host)
            flat.copy_(host)
        else:
            dist.all_reduce(flat)
        flat /= world_size
        o = 0
        for k in bucket:
            n = grads[k].numel()
            grads[k].copy_(flat[o:o + n].view_as(grads[k]))
            o += n

    for k in keys:
        bucket.append(k)
        size += grads[k].numel() * 4
        if size >= bucket_bytes:
            flush()
            bucket, size = [], 0
    flush()


def train_step(batch_data, run_info, dist=None, world_size=1, dropout_keep=None, timings=None):
    """batch_data: {'img': uint8 [N, H, W, 3], 'dummy_target': object array [N, B] of head names / None, '<head>': [N, H, W, 1] class
    ids, ...}; run_info: ({'net': {'desc': NetDesc, 'optimizer': cerberus_amd.train.Adam, 'extra_info': {'loss': loss_kwargs}}}, state)
    -- the reference's protocol (models/run_desc.py:25-60).  Returns {'EMA': {'<head>_loss': ..., 'overall_loss': ...},
    'raw': {'img', 'true', 'pred'}} (two random samples for the visualisation callbacks, models/run_desc.py:172-230).
    timings: optional dict that receives the DEVICE time (ms, stream events) of the step's phases outside cerb_net_train_grads' own per-launch
    records -- 'batch_to_device', 'train_grads' (the whole call), 'allreduce', 'adam_and_running_stats', 'param_update_and_repack', 'raw_payload'."""
    marks = []

    def mark(name):
        if timings is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))

    run_info, _ = run_info
    model, opt = run_info["net"]["desc"], run_info["net"]["optimizer"]
    # Sub-typing fine-tune (subtype_gland / subtype_nuclei; models/run_desc.py:83-84 -> net_desc.py:105-142): the backbone, conv_map, Patch-Class,
    # every INST decoder / head and the unselected TYPE decoder / head are frozen -- their BatchNorm layers normalise with the running statistics
    # (cerb_net_set_bn_eval, registered when the handle is packed) and neither their parameters nor their statistics move; see `frozen` below.
    frozen = model.frozen_prefixes() if hasattr(model, "frozen_prefixes") else []
    loss_opts = run_info["net"]["extra_info"]["loss"]
    batch = dict(batch_data)
    img = batch.pop("img")
    has = batch.pop("dummy_target")
    dev = torch.device("cuda", torch.cuda.current_device())
    mark("start")
    targets, flags = OrderedDict(), OrderedDict()
    wmaps = OrderedDict()
    for k, v in batch.items():
        if k.endswith("#WEIGHT-MAP"):  # models/run_desc.py:111-117: "<head>#WEIGHT-MAP" multiplies that head's per-pixel cross-entropy
            t = torch.as_tensor(v).float()
            wmaps[k[: -len("#WEIGHT-MAP")]] = t.reshape(t.shape[0], t.shape[1], t.shape[2]).to(dev)
            continue
        t = torch.as_tensor(v).float()
        targets[k] = (t.reshape(t.shape[0]) if k == "Patch-Class" else t.reshape(t.shape[0], t.shape[1], t.shape[2])).to(dev)
        flags[k] = torch.from_numpy(np.any(np.asarray(has) == k, axis=-1).astype(np.float32)).to(dev)
    if dropout_keep is None and "Patch-Class" in targets:  # nn.Dropout(p=0.3) of the Patch-Class branch (models/net_desc.py:70)
        dropout_keep = torch.rand((img.shape[0], 512), device=dev) >= 0.3
    logits = {}
    img_dev = torch.as_tensor(img).to(dev)
    mark("batch_to_device")
    # (sync_losses=False: the call returns while the backward pass is still running; everything below is QUEUED behind it -- the host's share of
    # the optimiser, the running statistics and the re-pack, ~4 ms of Python and table building, no longer sits between device phases)
    (loss_dev, loss_keys), grads = model.train_grads(img_dev, targets, flags, loss_opts, dropout_keep, views=True, pixel_weights=wmaps, logits_out=logits,
                                                     sync_losses=False)
    mark("train_grads")
    buf_keys = [k for k in grads if k.endswith("running_mean") or k.endswith("running_var")]
    stats = OrderedDict((k, grads.pop(k)) for k in buf_keys)
    if frozen:  # requires_grad = False there: the optimiser never sees these tensors (their moments stay unborn, as in torch.optim.Adam)
        for k in [k for k in grads if any(k.startswith(p) for p in frozen)]:
            del grads[k]
        for k in [k for k in stats if any(k.startswith(p) for p in frozen)]:
            del stats[k]
    allreduce_grads(grads, dist, world_size)
    mark("allreduce")
    # parameters live in the model's state dict (host); the optimiser works on device copies that persist across steps
    if not hasattr(model, "_dev_params"):
        model._sync_state_dict()
    if not hasattr(model, "_dev_params"):  # views into ONE flat device buffer: the copy back to the state dict is a single transfer
        layout, off = [], 0
        for k, v in model._sd.items():
            if v.dtype == torch.float32:
                layout.append((k, off, v.numel(), tuple(v.shape)))
                off += (v.numel() + 63) // 64 * 64
        model._dev_flat = torch.zeros(off, dtype=torch.float32, device=dev)
        model._dev_layout = layout
        model._dev_params = OrderedDict((k, model._dev_flat[o:o + n].view(shp)) for k, o, n, shp in layout)
        for k, t in model._dev_params.items():
            t.copy_(model._sd[k])
    opt.step(model._dev_params, grads)
    if stats:  # running = 0.9 running + 0.1 batch (torch BatchNorm momentum 0.1; the variance is the unbiased one), all buffers per launch
        run = [model._dev_params[k] for k in stats]
        torch._foreach_mul_(run, 0.9)
        torch._foreach_add_(run, [s.reshape(r.shape) for s, r in zip(stats.values(), run)], alpha=0.1)
        for k in stats:  # BatchNorm2d.num_batches_tracked += 1 per training forward (the checkpoint's int64 buffers)
            if k.endswith("running_mean"):
                nk = k[: -len("running_mean")] + "num_batches_tracked"
                if nk in model._sd:
                    model._sd[nk] = model._sd[nk] + 1
    mark("adam_and_running_stats")
    model.load_updated_parameters(model._dev_params, model._dev_flat, model._dev_layout)
    mark("param_update_and_repack")
    loss_host = loss_dev.cpu()  # (the step's one wait for the device)
    losses = OrderedDict((key, float(loss_host[i])) for i, key in loss_keys)
    ema = OrderedDict(("%s_loss" % k, v) for k, v in losses.items())
    ema["overall_loss"] = float(sum(losses.values()))
    out = {"EMA": ema, "raw": _raw_payload(torch.as_tensor(img), targets, logits, has)}
    mark("raw_payload")
    if timings is not None:
        torch.cuda.synchronize(dev)
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            timings[name] = a.elapsed_time(b)
    return out


def _raw_payload(img, targets, logits, has):
    """train_step's visualisation payload (models/run_desc.py:172-230): two randomly drawn samples of the batch -- the uint8 images, the targets
    and the train-mode predictions read out per head ('*-INST': softmax channels 1:, '*-TYPE': argmax, 'Patch-Class': argmax spread over the
    tile), every array torch.squeeze'd as the reference does.  When a sample carries a Patch-Class target the reference pushes every head
    through F.interpolate(nearest, size = tile): the identity for the dense heads, the broadcast over the tile for Patch-Class."""
    import torch.nn.functional as F

    n, h, w = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
    idx = torch.randint(0, n, (2,))
    names = np.asarray(has)
    pc_in_targets = bool(np.any(names == "Patch-Class"))
    true, pred = OrderedDict(), OrderedDict()
    for key, lg in logits.items():
        lg = lg[idx.to(lg.device)]
        if key == "Patch-Class":
            p = torch.argmax(torch.softmax(lg, -1), dim=-1, keepdim=True).reshape(2, 1, 1, 1)  # NHWC [2, 1, 1, 1], read as NCHW by interpolate
            p = F.interpolate(p.float(), size=(h, w), mode="nearest").permute(0, 2, 3, 1)
        else:
            sm = torch.softmax(lg, -1)
            p = sm if key.endswith("TYPE") else sm[..., 1:]
            if pc_in_targets:
                p = F.interpolate(p.permute(0, 3, 1, 2).float(), size=(h, w), mode="nearest").permute(0, 2, 3, 1)
        p = torch.squeeze(p)
        if "TYPE" in key:
            p = torch.argmax(p, dim=-1, keepdim=False)
        pred[key] = p.detach()
        t = targets[key][idx.to(targets[key].device)]
        t = t.reshape(2, 1, 1, 1) if key == "Patch-Class" else t.reshape(2, h, w, 1)
        if key == "Patch-Class" or pc_in_targets:
            t = F.interpolate(t.permute(0, 3, 1, 2).float(), size=(h, w), mode="nearest").permute(0, 2, 3, 1)
        true[key] = torch.squeeze(t).detach()
    out = {"img": img[idx.to(img.device)].to(torch.uint8), "true": true, "pred": pred}
    # 13 arrays, ~30 MB at batch 16 x 448^2: every device tensor goes to a fresh PINNED host tensor without a wait in between and the host waits once (thirteen
    # `.cpu()` calls were thirteen waits and pageable copies: ~2 ms of a 98 ms step with the device idle).  The numpy arrays keep their pinned tensors alive.
    pend = []
    for d in (out, true, pred):
        for k, v in list(d.items()):
            if torch.is_tensor(v) and v.is_cuda:
                hbuf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                hbuf.copy_(v.contiguous(), non_blocking=True)
                pend.append((d, k, hbuf, v.device))
            elif torch.is_tensor(v):
                d[k] = v.contiguous().numpy()
    if pend:
        for dev in set(e[3] for e in pend):
            torch.cuda.current_stream(dev).synchronize()
        for d, k, hbuf, _ in pend:
            d[k] = hbuf.numpy()
    return out


def _eval_twin(model):
    """valid_step runs the network in eval mode (running statistics folded into the convolutions): a handle packed for training
    cannot serve that, so a model in training mode gets an inference twin that is refreshed whenever the parameters have changed."""
    if not getattr(model, "_train_packing", False):
        return model
    from .net_desc import NetDesc  # noqa: F401  (type of the twin)

    version = getattr(model, "_param_version", 0)
    twin = getattr(model, "_valid_twin", None)
    if twin is None or model._valid_twin_version != version:
        if twin is None:
            twin = model.__class__(**model._init_kwargs)
        twin.load_state_dict(model.state_dict(), strict=True)
        model._valid_twin, model._valid_twin_version = twin, version
    return twin


def valid_step(batch_data, run_info):
    """The reference's valid_step (models/run_desc.py:332-436): eval-mode forward of the whole batch at full size and the per-head
    read-outs -- '*-INST' softmax channels 1: , '*-TYPE' argmax, 'Patch-Class' argmax spread over the tile -- returned on the host as
    {'raw': {'img', 'true', 'pred', 'dummy', 'channel_info'}}.  Read-outs come from the inference kernels (cerb_net_forward with the
    output window = the tile).  Quirks kept: torch.squeeze on every array, and when a sample carries a Patch-Class target the dense
    heads' 'true' maps pass through F.interpolate in NHWC order, which turns [N, H, W, 1] into [N, H, H, W] (:418-421)."""
    run_info, _ = run_info
    model = run_info["net"]["desc"]
    batch = dict(batch_data)
    img = torch.as_tensor(batch.pop("img"))
    has = np.asarray(batch.pop("dummy_target"))
    tgt_names = list(np.unique(has[has != None]))  # noqa: E711  (object array, as in the reference)
    dev = torch.device("cuda", torch.cuda.current_device())
    n, h, w = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
    tiles = img.to(dev).float().to(torch.uint8).contiguous()  # the reference computes on float32 of the input and hands back .byte()
    pred_dev = _eval_twin(model).infer_tiles(tiles, [h, w])
    pc_in_targets = "Patch-Class" in tgt_names
    pred, true = {}, {}
    for key, p in pred_dev.items():
        pred[key] = np.squeeze(p.cpu().numpy())
        t = torch.as_tensor(batch[key]).float().numpy()  # NHWC, one channel
        if key == "Patch-Class":
            t = np.broadcast_to(t.reshape(n, 1, 1), (n, h, w))
        elif pc_in_targets:  # F.interpolate(nearest, size = tile) over NHWC read as NCHW: [N, H, W, 1] -> [N, H, H, W], last axis repeated
            t = np.broadcast_to(t[:, :, :, 0][:, :, :, None], (n, t.shape[1], t.shape[2], w))
            if t.shape[2] != h:
                t = t[:, :, (np.arange(h) * t.shape[2] // h), :]
        true[key] = np.squeeze(np.array(t, dtype=np.float32))
    info = OrderedDict((name, {hname: och}) for name, hname, och, _ in model._decoders)
    return {"raw": {"img": tiles.cpu().numpy(), "true": true, "pred": pred, "dummy": has, "channel_info": info}}
