"""Build recipe for libcerberus_hip.so (gfx950 only, in-tree so the .so travels with gpurun snapshots) and libcerberus_host.so (the reader's
host-side byte codecs: plain C, gcc, no HIP -- include/cerberus_host.h).

    python -m cerberus_amd.build            # incremental
    python -m cerberus_amd.build --force

hipcc cross-compiles without a GPU.  One object per .hip translation unit, linked into one shared library.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcerberus_hip.so")
LIB_DEV = os.path.join(HERE, "libcerberus_hip_dev.so")
LIB_HOST = os.path.join(HERE, "libcerberus_host.so")
HOST_SOURCES = ["host_codecs.c"]
SOURCES = ["conv_igemm.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wino4b.hip", "conv_wino4p.hip", "net_kernels.hip", "postproc.hip", "slide_kernels.hip", "train_kernels.hip", "head_train.hip", "conv_wgrad.hip", "conv_wgrad_wino.hip", "pack_kernels.hip", "cerb_api.hip", "cerb_train.hip"]
# The translation units that read developer A/B switches (cerb_common.h: cerb_dev_getenv).  The product library compiles them WITHOUT the switches
# (every one folds to its default); the same units compiled with -DCERB_DEV_SWITCHES, linked with the other units' objects, make
# libcerberus_hip_dev.so -- loaded only by the A/B tests' child processes (CERB_DEV_LIB=1, cerberus_amd/_lib.py).
DEV_SOURCES = ["cerb_api.hip", "cerb_train.hip", "postproc.hip", "conv_wino4b.hip", "train_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# conv_wino4.hip: the 36-step chunk (288 matrix instructions) must be fully unrolled for its 288 accumulators to be registers (the default
# pragma-unroll budget is 16 k instructions); the matrix instructions start in VGPR form and the register allocator moves the ones that do
# not fit to the AccVGPR half of the 512-register file (the AGPR-only form spills 32 accumulators to scratch).
EXTRA_FLAGS = {"conv_wino4.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               # 144 accumulators fit the AccVGPR half: the default AGPR form (the VGPR-form rewrite pass of ROCm 7.2 crashes on this kernel)
               "conv_wino4b.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
               # accumulators pinned by hand (inline-asm matrix instructions with "a" / "v" constraints): no allocator flag needed
               "conv_wino4p.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_host(force=False, verbose=True):
    """libcerberus_host.so: the TIFF LZW / PackBits / predictor codecs of cerberus_amd/reader.py (torch-free decode workers load it too)."""
    srcs = [os.path.join(CSRC, f) for f in HOST_SOURCES]
    if force or _stale(LIB_HOST, srcs + [os.path.join(os.path.dirname(HERE), "include", "cerberus_host.h")]):
        cmd = [os.environ.get("CC", "gcc"), "-O3", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-pthread", "-o", LIB_HOST] + srcs + ["-lz"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_HOST


def build(force=False, verbose=True):
    build_host(force, verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cerberus_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    dev_objs = list(objs)
    for src in DEV_SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, "dev_" + src.replace(".hip", ".o"))
        dev_objs[dev_objs.index(os.path.join(CSRC, src.replace(".hip", ".o")))] = obj
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc] + FLAGS + ["-DCERB_DEV_SWITCHES"] + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src + " (dev)", subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    for lib, lobjs in ((LIB, objs), (LIB_DEV, dev_objs)):
        if force or procs or _stale(lib, lobjs):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + lobjs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
