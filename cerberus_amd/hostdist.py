"""torch.distributed facade for backends that move host memory only ("gloo"): the multi-GPU drivers (wsi.gather_to_root,
shard_postproc.run_distributed) hand CUDA tensors to gather / all_gather / batched send-recv, which "nccl" (RCCL over xGMI) moves
device to device; under gloo the same calls are staged through the host here.  Selected with CERB_DIST_BACKEND=gloo -- clusters
without RCCL, and the two-ranks-on-one-GPU runs of the test suite (RCCL refuses two ranks on one device)."""
import torch


def _host(t):
    return t.detach().cpu() if t.is_cuda else t


class _Reqs(object):
    def __init__(self, reqs, back):
        self._reqs, self._back = reqs, back

    def wait(self):
        for r in self._reqs:
            r.wait()
        for dst, src in self._back:
            dst.copy_(src)
        self._back = []


class HostStagedDist(object):
    def __init__(self, dist):
        self._d = dist

    def __getattr__(self, name):  # barrier, get_rank, isend / irecv (used as P2POp tags), ReduceOp, ...
        return getattr(self._d, name)

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def batch_isend_irecv(self, ops):
        real, back = [], []
        for op, t, peer in ops:
            if op is self._d.isend:
                real.append(self._d.P2POp(op, _host(t).contiguous(), peer))
            else:
                h = torch.empty(t.shape, dtype=t.dtype)
                real.append(self._d.P2POp(op, h, peer))
                back.append((t, h))
        return [_Reqs(self._d.batch_isend_irecv(real), back)]

    def all_gather(self, tensor_list, tensor):
        hl = [torch.empty(x.shape, dtype=x.dtype) for x in tensor_list]
        self._d.all_gather(hl, _host(tensor).contiguous())
        for dst, src in zip(tensor_list, hl):
            dst.copy_(src)

    def gather(self, tensor, gather_list=None, dst=0):
        hl = None if gather_list is None else [torch.empty(x.shape, dtype=x.dtype) for x in gather_list]
        self._d.gather(_host(tensor).contiguous(), hl, dst=dst)
        if gather_list is not None:
            for a, b in zip(gather_list, hl):
                a.copy_(b)

    def broadcast(self, tensor, src=0):
        h = _host(tensor).contiguous()
        self._d.broadcast(h, src=src)
        tensor.copy_(h)

    def all_reduce(self, tensor, op=None):
        h = _host(tensor).contiguous()
        self._d.all_reduce(h) if op is None else self._d.all_reduce(h, op=op)
        tensor.copy_(h)
