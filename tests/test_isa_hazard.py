"""ISA-level guard for the gfx950 store hazard documented in cerberus_amd/csrc/conv_wino.hip (`buf_store`): a
`buffer_store_dwordx4` whose soffset is an SGPR, followed directly by a VALU write of its data VGPRs, stores corrupted data and
hipcc (ROCm 7.2) does not pad it.  The kernels pin `s_nop 1` behind every such store; this test compiles the translation units
to gfx950 assembly (hipcc cross-compiles without a GPU) and checks that NO later compiler, flag or source change reopens the
window: between each SGPR-soffset store and the first instruction that rewrites one of its data registers there must be at
least two wait states."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cerberus_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
UNITS = ["conv_wino.hip", "conv_igemm.hip", "conv_wino4.hip", "conv_wino4b.hip"]

STORE = re.compile(r"^\s*buffer_store_dwordx4\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*(v_\w+|ds_read\w*|ds_load\w*|buffer_load\w*|global_load\w*|flat_load\w*|scratch_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def _asm(unit, tmp):
    out = os.path.join(tmp, unit.replace(".hip", ".s"))
    from cerberus_amd.build import EXTRA_FLAGS  # the per-unit flags of the real build
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out] + EXTRA_FLAGS.get(unit, [])
                          + [os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
    return [l for l in open(out).read().splitlines() if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]


def _writes(line, lo, hi):
    m = VDST.match(line)
    if not m:
        return False
    if m.group(3) is not None:
        a, b = int(m.group(3)), int(m.group(4))
    else:
        a = b = int(m.group(5))
    if m.group(1).startswith("v_cmp") or m.group(1).startswith("v_readlane") or m.group(1).startswith("v_readfirstlane"):
        return False  # destination is not a VGPR
    return not (b < lo or a > hi)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("unit", UNITS)
def test_sgpr_soffset_stores_are_padded(unit, tmp_path):
    lines = _asm(unit, str(tmp_path))
    n_sgpr_stores = 0
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        lo, hi, soff = int(m.group(1)), int(m.group(2)), m.group(4)
        if not re.match(r"^s\d+$", soff):  # immediate / `off` soffset: the compiler handles that form itself
            continue
        n_sgpr_stores += 1
        wait = 0
        for nxt in lines[i + 1: i + 40]:
            if _writes(nxt, lo, hi):
                break
            sn = re.match(r"^\s*s_nop\s+(\d+)", nxt)
            wait += int(sn.group(1)) + 1 if sn else 1
            if wait >= 2:
                break
        assert wait >= 2, "%s: data registers v[%d:%d] of `%s` are rewritten after %d wait state(s)" % (unit, lo, hi, l.strip(), wait)
    assert n_sgpr_stores >= 8, "expected the output stage's SGPR-soffset stores in %s (found %d): has the kernel changed shape?" % (unit, n_sgpr_stores)


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("unit", ["conv_wino4.hip", "conv_wino4b.hip"])
def test_f4x4_kernels_keep_their_arrays_in_registers(unit, tmp_path):
    """The F(4x4) kernels hold 144 / 288 accumulators, the raw patch and the weight ring in (Acc)VGPRs.  Twice during round 2 hipcc put
    them in scratch without a word: below the full unroll of the 36-step chunk (the pragma-unroll budget of cerberus_amd/build.py) and when
    it stopped inlining a lambda that takes the patch by reference (3.4x slower, same results).  Guard: no scratch, no out-of-line
    lambda, no call, and the fully unrolled chunk (>= 1152 matrix instructions per translation unit)."""
    from cerberus_amd.build import EXTRA_FLAGS

    out = os.path.join(str(tmp_path), unit.replace(".hip", ".s"))
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out, "-Rpass-analysis=kernel-resource-usage"]
                         + EXTRA_FLAGS.get(unit, []) + [os.path.join(CSRC, unit)], stderr=subprocess.PIPE, universal_newlines=True, check=True)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res.stderr)]
    assert scratch and max(scratch) == 0, scratch
    asm = open(out).read()
    assert not re.search(r"^_ZZ", asm, re.M), "a lambda was compiled out of line"
    assert "s_swappc_b64" not in asm and "scratch_" not in asm
    assert asm.count("v_mfma_f32_16x16x4") >= 1152
