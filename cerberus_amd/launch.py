"""One process per GPU without an external launcher.

The reference drives every visible GPU from ONE process (`torch.nn.DataParallel`, infer/base.py:46-47): `python run_infer_tile.py --gpu 0,1`
needs no launcher.  Here a rank is a process (one HIP context, one RCCL communicator per GPU), so the entry points accept both ways in:

* under `python -m torch.distributed.run --nproc-per-node N ...` (WORLD_SIZE / RANK / LOCAL_RANK in the environment) they are a rank;
* started plainly with `--gpus N` (bench.py) or several ids in `--gpu` (run_infer_*.py) they re-execute themselves N times through
  `ensure_world`, one child per device, rendezvous on 127.0.0.1 -- and NEVER fall back to one GPU silently: a request for N ranks that
  cannot be met (fewer devices, WORLD_SIZE disagreeing with --gpus) ends with a non-zero exit code and a sentence saying why.

`init_dist` opens the process group with a finite timeout, `rank_identity` all-gathers (rank, device index, device UUID, pid) so that a
result line can show which devices the communicator really spanned, and `PhaseWatch` names the phase a rank is stuck in when a
collective does not return (RCCL's own watchdog only reports a sequence number)."""
import os
import socket
import subprocess
import sys
import threading
import time


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices():
    import torch

    return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0


def ensure_world(gpus, backend="nccl", argv=None, oversubscribe=False):
    """Call first thing in main().  Returns normally when this process should go on as a rank (or as the only process); re-executes the
    script `gpus` times and exits with the children's worst return code otherwise.

    gpus: ranks asked for on the command line.  backend "nccl" (= RCCL) needs one device per rank; "gloo" (host-staged collectives,
    cerberus_amd/hostdist.py) may time-share devices when `oversubscribe` is set (the plumbing tests)."""
    gpus = int(gpus)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if gpus > 1 and int(env_world) != gpus:
            sys.stderr.write("error: --gpus %d but the launcher set WORLD_SIZE=%s: refusing to run a job whose size is ambiguous\n" % (gpus, env_world))
            sys.exit(2)
        return
    if gpus <= 1:
        return
    n_dev = visible_devices()
    if n_dev < gpus and not (oversubscribe and backend != "nccl" and n_dev >= 1):
        sys.stderr.write("error: --gpus %d requested but %d GPU(s) are visible: not running on fewer devices than asked "
                         "(RCCL needs one device per rank)\n" % (gpus, n_dev))
        sys.exit(2)
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for r in range(gpus):
        env = dict(os.environ)
        env.update(WORLD_SIZE=str(gpus), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CERB_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer buffers (see the environment notes of this build)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    rc = 0
    alive = set(range(gpus))
    while alive:
        for r in list(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0:
                rc = rc or c
                sys.stderr.write("rank %d exited with code %d: stopping the other ranks\n" % (r, c))
                for o in alive:
                    procs[o].terminate()
        if alive:
            time.sleep(0.05)
    sys.exit(rc if rc >= 0 else 1)


def init_dist(backend, local_rank, timeout_s=None):
    """Open the default process group for this rank and return the `dist` object the drivers use ("nccl": torch.distributed itself,
    device-to-device over RCCL / xGMI; "gloo": the host-staging facade).  timeout_s: collective timeout (default CERB_DIST_TIMEOUT_S or 600)."""
    import datetime

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if timeout_s is None:
        timeout_s = float(os.environ.get("CERB_DIST_TIMEOUT_S", "600"))
    to = datetime.timedelta(seconds=float(timeout_s))
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=to)
        return dist
    from .hostdist import HostStagedDist

    dist.init_process_group(backend, timeout=to)
    return HostStagedDist(dist)


def rank_identity(dist, dev, backend):
    """-> {"world", "backend", "ranks": [{"rank", "device", "uuid", "name", "pid"} ...]} gathered over the communicator itself (so the list
    has as many entries as ranks the backend really connected); without a process group: the one local device."""
    import torch

    me = {"rank": 0, "device": -1, "uuid": None, "name": None, "pid": os.getpid()}
    if dev is not None and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(dev)
        me.update(device=int(torch.device(dev).index or 0), uuid=str(getattr(pr, "uuid", "")), name=pr.name)
    if dist is None:
        return {"world": 1, "backend": None, "ranks": [me], "distinct_devices": 1}
    me["rank"] = int(dist.get_rank())
    world = int(dist.get_world_size())
    # fixed-size byte tensor through the backend's own all_gather (device tensors under RCCL): what comes back IS what the communicator spans
    import json

    raw = json.dumps(me).encode()[:255]
    buf = torch.zeros(256, dtype=torch.uint8)
    buf[: len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    buf[255] = len(raw)
    on = torch.device(dev) if (backend == "nccl" or (dev is not None and torch.cuda.is_available())) else torch.device("cpu")
    buf = buf.to(on)
    lst = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(lst, buf)
    ranks = []
    for t in lst:
        b = bytes(t.cpu().tolist())
        ranks.append(json.loads(b[: b[255]].decode()))
    ranks.sort(key=lambda r: r["rank"])
    return {"world": world, "backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend + " (host-staged)", "ranks": ranks,
            "distinct_devices": len(set((r["uuid"] or r["device"]) for r in ranks))}


class PhaseWatch(object):
    """`with watch.phase("halo exchange"): ...` around blocking collectives.  A daemon thread ends the process with a message that NAMES the
    phase when one lasts longer than `timeout_s` -- a stuck rank says where it is instead of hanging until the launcher is killed."""

    def __init__(self, rank=0, timeout_s=None, on_timeout=None):
        self.rank = int(rank)
        self.timeout_s = float(os.environ.get("CERB_PHASE_TIMEOUT_S", "900") if timeout_s is None else timeout_s)
        self._cur = None
        self._lock = threading.Lock()
        self._on_timeout = on_timeout
        self._stop = False
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()

    def _loop(self):
        while not self._stop:
            time.sleep(min(1.0, max(0.01, self.timeout_s / 20)))
            with self._lock:
                cur = self._cur
            if cur is not None and time.monotonic() - cur[1] > self.timeout_s:
                msg = "rank %d: phase '%s' has not finished after %.0f s (collective timeout): giving up\n" % (self.rank, cur[0], self.timeout_s)
                if self._on_timeout is not None:
                    self._on_timeout(msg)
                    with self._lock:
                        self._cur = None
                    continue
                sys.stderr.write(msg)
                sys.stderr.flush()
                os._exit(3)

    def phase(self, name):
        return _Phase(self, name)

    def close(self):
        self._stop = True


class _Phase(object):
    def __init__(self, watch, name):
        self.w, self.name = watch, name

    def __enter__(self):
        with self.w._lock:
            self.w._cur = (self.name, time.monotonic())
        return self

    def __exit__(self, et, ev, tb):
        with self.w._lock:
            self.w._cur = None
        if et is not None and et is not SystemExit and et is not KeyboardInterrupt:
            sys.stderr.write("rank %d: failed in phase '%s': %s\n" % (self.w.rank, self.name, ev))
        return False


_NULL = None


def null_watch():
    """A PhaseWatch that never fires (single-process runs)."""
    global _NULL
    if _NULL is None:
        _NULL = PhaseWatch(0, timeout_s=1e18)
    return _NULL
