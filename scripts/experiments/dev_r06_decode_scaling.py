"""Developer microbenchmark (host only): why does the reader's JPEG tile decode stop scaling at 2 threads (270 Mpx/s on the 256-core GPU box)?
Stages of reader.TiffReader._decode on one atlas tile, alone and on thread pools; a process pool for comparison."""
import io
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from PIL import Image  # noqa: E402

import dev_r06_giant_slide as g  # noqa: E402
from cerberus_amd import reader as rd  # noqa: E402

PATH = "/tmp/decode_scaling.tif"
N = 4096


def stream_of(r, i):
    p = r.levels[0]
    data = os.pread(r.fh.fileno(), p.counts[i], p.offsets[i])
    data = rd._strip_jfif_app0(data)
    return data[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00" + data[2:]


def open_and_load(data):
    img = Image.open(io.BytesIO(data))
    return np.asarray(img)


def proc_job(args):
    lo, hi = args
    r = rd.WSIReader.open(input_img=PATH)
    p = r.levels[0]
    n = 0
    for i in range(lo, hi):
        n += r._decode(p, i, 256, 256).shape[0]
    return n


def main():
    g.write_atlas_tiff(PATH, 16384, 16384)
    r = rd.WSIReader.open(input_img=PATH)
    p = r.levels[0]
    datas = [stream_of(r, i) for i in range(N)]
    t = time.perf_counter()
    for i in range(512):
        stream_of(r, i)
    print("pread + marker splice: %.1f us / tile" % ((time.perf_counter() - t) / 512 * 1e6))
    t = time.perf_counter()
    for d in datas[:512]:
        Image.open(io.BytesIO(d))
    print("Image.open (marker parse, no decode): %.1f us / tile" % ((time.perf_counter() - t) / 512 * 1e6))
    t = time.perf_counter()
    for d in datas[:512]:
        open_and_load(d)
    one = (time.perf_counter() - t) / 512
    print("open + decode + asarray: %.1f us / tile = %.0f Mpx/s on one thread" % (one * 1e6, 65536 / one / 1e6))
    for nt in (1, 2, 4, 8, 16, 32):
        with ThreadPoolExecutor(nt) as ex:
            t = time.perf_counter()
            list(ex.map(open_and_load, datas))
            dt = time.perf_counter() - t
        print("threads %2d: open + decode: %.0f Mpx/s" % (nt, N * 65536 / dt / 1e6))
    for nt in (8, 32):
        with ThreadPoolExecutor(nt) as ex:
            t = time.perf_counter()
            list(ex.map(lambda ds: [open_and_load(d) for d in ds], [datas[i:i + 32] for i in range(0, N, 32)]))
            dt = time.perf_counter() - t
        print("threads %2d, 32 tiles per task: %.0f Mpx/s" % (nt, N * 65536 / dt / 1e6))
    out = np.empty((256 * 8, 256 * 64, 3), np.uint8)
    for nt in (8, 32):
        os.environ["CERB_DECODE_THREADS"] = str(nt)
        t = time.perf_counter()
        for k in range(0, 16384, 2048):
            r._read_level(0, 0, k, 16384, k + 2048)
        dt = time.perf_counter() - t
        print("reader._read_level, CERB_DECODE_THREADS=%d: %.0f Mpx/s" % (nt, 16384 * 16384 / dt / 1e6))
    for nproc in (4, 8, 16, 32):
        with ProcessPoolExecutor(nproc) as ex:
            list(ex.map(proc_job, [(0, 8)] * nproc))  # start-up
            t = time.perf_counter()
            per = N // nproc
            list(ex.map(proc_job, [(i * per, (i + 1) * per) for i in range(nproc)]))
            dt = time.perf_counter() - t
        print("processes %2d (decode only, nothing returned): %.0f Mpx/s" % (nproc, N * 65536 / dt / 1e6))


if __name__ == "__main__":
    main()
