"""Launcher plumbing without a GPU (cerberus_amd/launch.py): `--gpus N` never degrades to one GPU silently, the rank list is gathered over
the communicator itself, a collective that does not return names its phase, and the paired halo rounds move the right strips."""
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 8, reason="box really has 8 GPUs")
def test_bench_gpus_n_without_launcher_refuses_instead_of_running_one_gpu():
    """`python bench.py --gpus 8` with fewer than 8 devices and no launcher: exit code 2 and a sentence, never a JSON line with n_gpus 1
    (round 3's bench parsed --gpus and ignored it)."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--mode", "batch", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_clean_env())
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 8" in r.stderr and "visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus_disagreeing_with_world_size_refuses():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--mode", "batch"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
from cerberus_amd import launch
launch.ensure_world(int(sys.argv[1]), "gloo", oversubscribe=True)
if "WORLD_SIZE" not in os.environ:
    sys.exit(7)
import torch
dist = launch.init_dist("gloo", int(os.environ["LOCAL_RANK"]), timeout_s=60)
idn = launch.rank_identity(dist, None, "gloo")
from cerberus_amd.shard_postproc import halo_exchange
r, w = dist.get_rank(), dist.get_world_size()
up, down = torch.full((4,), 10.0 * r + 1), torch.full((4,), 10.0 * r + 2)
above = torch.zeros(4) if r > 0 else None
below = torch.zeros(4) if r < w - 1 else None
halo_exchange(dist, r, w, up, down, above, below)
ok = (above is None or float(above[0]) == 10.0 * (r - 1) + 2) and (below is None or float(below[0]) == 10.0 * (r + 1) + 1)
if len(sys.argv) > 2 and sys.argv[2] == "fail" and r == 1:
    sys.exit(5)
dist.barrier()
if r == 0:
    print(json.dumps({"identity": idn, "ok": bool(ok)}))
sys.exit(0 if ok else 9)
"""


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only plumbing test (the GPU suite has its own self-spawn test)")
@pytest.mark.parametrize("world", [2, 5])
def test_self_spawn_rendezvous_identity_and_halo_rounds(tmp_path, world):
    import json

    script = tmp_path / "child.py"
    script.write_text(_CHILD % ROOT)
    # no devices here: oversubscribe needs >= 1 device, so pretend through the launcher's own probe
    env = _clean_env(CERB_TEST_FAKE_DEVICES="1")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.argv = [%r, %r]; import cerberus_amd.launch as l; l.visible_devices = lambda: 1; "
                        "exec(open(%r).read())" % (str(script), str(world), str(script))], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    # the parent's argv[0] is the script: children are `python child.py <world>` and take the WORLD_SIZE branch
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["ok"] and out["identity"]["world"] == world and [x["rank"] for x in out["identity"]["ranks"]] == list(range(world))
    assert len(set(x["pid"] for x in out["identity"]["ranks"])) == world


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only plumbing test")
def test_self_spawn_propagates_a_rank_failure(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(_CHILD % ROOT)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.argv = [%r, '2', 'fail']; import cerberus_amd.launch as l; l.visible_devices = lambda: 1; "
                        "exec(open(%r).read())" % (str(script), str(script))], capture_output=True, text=True, timeout=300, cwd=ROOT, env=_clean_env())
    assert r.returncode == 5 and "rank 1 exited with code 5" in r.stderr


def test_phase_watch_names_the_stuck_phase():
    from cerberus_amd.launch import PhaseWatch

    msgs = []
    w = PhaseWatch(rank=3, timeout_s=0.2, on_timeout=msgs.append)
    with w.phase("quick"):
        pass
    time.sleep(0.4)
    assert not msgs
    with w.phase("halo exchange (Gland)"):
        time.sleep(0.6)
    w.close()
    assert len(msgs) == 1 and "rank 3" in msgs[0] and "halo exchange (Gland)" in msgs[0]


def test_phase_watch_ends_a_hung_process():
    code = ("import sys, time; sys.path.insert(0, %r); from cerberus_amd.launch import PhaseWatch; w = PhaseWatch(1, timeout_s=0.3)\n"
            "with w.phase('label-band gather to rank 0 (Nuclei)'):\n    time.sleep(30)\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and time.time() - t0 < 20
    assert "label-band gather to rank 0 (Nuclei)" in r.stderr and "rank 1" in r.stderr
