"""Slide readers behind the interface the reference drives (tiatoolbox's `WSIReader`, infer/wsi.py:521-531):

    reader = WSIReader.open(input_img=path)
    w, h   = reader.slide_dimensions(resolution=0.5, units="mpp")      # XY, as tiatoolbox returns it
    mpp    = reader.info.mpp                                            # scan resolution, microns per pixel (x, y) or None
    rgb    = reader.read_bounds((x0, y0, x1, y1), resolution, units)   # uint8 [h, w, 3]; bounds in the REQUESTED resolution's pixels
    rows   = reader.rows(resolution, units)                            # lazy (H, W, 3) row source for wsi.SlabUploader

tiatoolbox / OpenSlide are not in this image, so the back ends are this package's own:
  * ArrayReader      -- `.npy` (memory-mapped), PNG / JPG (PIL) and in-memory arrays: one level, no resolution metadata unless given;
  * SyntheticReader  -- `.txt` holding `synthetic:<H>x<W>:<seed>`: pixels are generated on the device (wsi.synth_slide);
  * TiffReader       -- baseline / BigTIFF, striped or TILED, pyramid pages, compression none / deflate (+ horizontal predictor) /
                        JPEG tiles (decoded by PIL, JPEGTables spliced in), resolution from XResolution / ResolutionUnit or an
                        Aperio `MPP = ...` description: generic tiled TIFFs and `.svs` files whose tiles are JPEG or JPEG 2000
                        (Aperio 33003 / 33005, TIFF 34712: OpenJPEG behind PIL; 33003 with 4:2:2 / 4:2:0 chroma on even tile sizes, anything else subsampled refused by name); LZW and PackBits tiles are decoded by libcerberus_host.so (csrc/host_codecs.c, include/cerberus_host.h).
Resampling: the pyramid level with the largest downsample not above the request is read and reduced by a box (area) filter --
exact pixel means for integer factors, PIL's BOX filter otherwise (tiatoolbox uses cv2 INTER_AREA there; unpinned, both libraries
are absent).  Everything here is host I/O; pixels reach the GPU through wsi.SlabUploader chunk by chunk under the inference.
"""
import io
import os
import struct
import zlib

import numpy as np


_POOL = {"n": None, "pool": None}


def decode_threads():
    """Tile-decode threads of the TIFF / .svs reader (the Python pool of the JPEG tiles and the native pthreads of cerb_host_tiff_read_tiles):
    CERB_DECODE_THREADS, default min(32, host cores / ranks on this host) -- `bench.py --mode ingest` reports the count that saturates one GPU's
    inference (a JPEG tile decodes at ~160 Mpx/s per core, an LZW / deflate tile at ~100, the network eats ~150 Mpx/s: four to eight threads)."""
    v = os.environ.get("CERB_DECODE_THREADS")
    if v:
        return max(1, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    except ValueError:
        ranks = 1
    return max(1, min(32, n // ranks))


def decode_pool():
    """The shared thread pool (None for one thread); re-made when CERB_DECODE_THREADS changes (bench.py's sweep)."""
    n = decode_threads()
    if n <= 1:
        return None
    if _POOL["n"] != n:
        from concurrent.futures import ThreadPoolExecutor

        if _POOL["pool"] is not None:
            _POOL["pool"].shutdown(wait=True)
        _POOL["pool"], _POOL["n"] = ThreadPoolExecutor(max_workers=n, thread_name_prefix="cerb-decode"), n
    return _POOL["pool"]


# ---- decode PROCESSES ------------------------------------------------------------------------------------------------------------
# Threads stop paying at two: PIL parses the JPEG markers, hands pixels over and closes the image under the interpreter lock, and only
# libjpeg's inner loop runs without it -- 236 Mpx/s on one thread, ~350 on any number (256-core host; scripts/experiments/dev_r06_decode_scaling.py),
# enough for a slide stored at the processing resolution (150 Mpx/s) and a quarter of what a 40x scan read at 0.5 mpp needs (4 source pixels per
# pixel).  Worker processes scale linearly (8: 1.9 Gpx/s, 32: 4.8): the reference's own answer (12 DataLoader workers, infer/wsi.py:936-950).
# A read of many tiles is cut into groups; every worker opens the file itself, decodes its group and writes the pixels into a shared-memory
# window; nothing but a few integers crosses the pipes.  Workers are fresh interpreters (cerberus_amd/decode_worker.py: no torch, no GPU context,
# the parent's __main__ is not re-imported).
_PROCS = {"n": None, "pool": None}
JP2K_CODECS = (33003, 33005, 34712)  # Aperio JPEG 2000 (YCbCr components / RGB components) and the TIFF JPEG 2000 tag of generic writers
PROC_CODECS = (7,) + JP2K_CODECS     # tiles decoded through PIL (libjpeg / OpenJPEG): the compressions whose large reads go to worker processes
_PROCS_LOCK = __import__("threading").Lock()
PROC_MIN_TILES = 96  # below this many tiles a read stays on the thread pool (thumbnails, edge strips, tests)


def decode_procs():
    """Worker processes of large tile reads: CERB_DECODE_PROCS, default min(16, host cores / (4 x ranks on this host)); 0 = threads only."""
    if os.environ.get("CERB_DECODE_WORKER") == "1":
        return 0
    v = os.environ.get("CERB_DECODE_PROCS")
    if v is not None and v != "":
        return max(0, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    except ValueError:
        ranks = 1
    return max(0, min(16, n // (4 * ranks)))


class _WorkerPool(object):
    """n decode_worker processes behind pipes.  run(tasks): every task goes to the next idle worker (n feeder threads: the pipe I/O and the wait for
    the reply run without the interpreter lock); raises with the worker's traceback when one fails."""

    def __init__(self, n):
        import queue
        import subprocess
        import sys
        from concurrent.futures import ThreadPoolExecutor

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = "import sys; sys.path.insert(0, %r); from cerberus_amd import decode_worker; decode_worker.main()" % root
        env = dict(os.environ, CERB_DECODE_WORKER="1", CERB_DECODE_THREADS="1", OMP_NUM_THREADS="1")
        self.procs = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env) for _ in range(n)]
        self.idle = queue.Queue()
        for p in self.procs:
            self.idle.put(p)
        self.feeders = ThreadPoolExecutor(max_workers=n, thread_name_prefix="cerb-decode-feed")
        self.n = n

    def _one(self, task):
        import pickle

        p = self.idle.get()
        try:
            body = pickle.dumps(task)
            p.stdin.write(struct.pack("<I", len(body)) + body)
            p.stdin.flush()
            head = p.stdout.read(4)
            if len(head) < 4:
                raise RuntimeError("a tile-decode worker died (exit code %s)" % p.poll())
            n = struct.unpack("<I", head)[0]
            buf = b""
            while len(buf) < n:
                part = p.stdout.read(n - len(buf))
                if not part:
                    raise RuntimeError("a tile-decode worker died mid-reply (exit code %s)" % p.poll())
                buf += part
            kind, val = pickle.loads(buf)
            if kind != "ok":
                raise RuntimeError("tile-decode worker: " + str(val))
            return val
        finally:
            self.idle.put(p)

    def run(self, tasks):
        return list(self.feeders.map(self._one, tasks))

    def shutdown(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:  # noqa: BLE001
                p.kill()
        self.feeders.shutdown(wait=False)


def _proc_pool():
    n = decode_procs()
    if n <= 0:
        return None
    with _PROCS_LOCK:
        if _PROCS["n"] != n:
            import atexit

            if _PROCS["pool"] is not None:
                _PROCS["pool"].shutdown()
            else:
                atexit.register(_shutdown_procs)
            _PROCS["pool"], _PROCS["n"] = _WorkerPool(n), n
        return _PROCS["pool"]


def warm_decode_workers():
    """Start the worker processes now, on a thread (0.2 - 0.3 s of interpreter + numpy + PIL start-up each, in parallel): a command line that knows it
    will read tiled slides calls this before it loads the model, so that the first chunk of the first slide does not wait for them."""
    import threading

    if decode_procs() > 0 and _PROCS["pool"] is None:
        threading.Thread(target=_proc_pool, name="cerb-decode-warm", daemon=True).start()


def _shm_has_room(nbytes):
    """Worker processes hand pixels over through a POSIX shared-memory block: a container with the default 64 MB /dev/shm has no room for a chunk
    of a slide, and writing past a tmpfs' size is a SIGBUS, not an exception -- such hosts decode on threads."""
    try:
        st = os.statvfs("/dev/shm")
        return st.f_bavail * st.f_frsize > int(nbytes * 1.25) + (64 << 20)
    except OSError:
        return False


def _shutdown_procs():
    if _PROCS["pool"] is not None:
        _PROCS["pool"].shutdown()
        _PROCS["pool"], _PROCS["n"] = None, None


_SHM_LIVE = {}  # shared blocks of live readers: unlinked at exit whatever became of their readers (/dev/shm outlives the process otherwise)


def _unlink_live_shm():
    for sh in list(_SHM_LIVE.values()):
        try:
            sh.close()
            sh.unlink()
        except Exception:  # noqa: BLE001
            pass
    _SHM_LIVE.clear()


__import__("atexit").register(_unlink_live_shm)
_WORKER = {"readers": {}, "shm": {}}


def _worker_decode(path, mpp, level, tiles, window, shm_name, shape):
    """In a worker: decode `tiles` [(ty, tx)] of a level and write their parts inside `window` = (x0, y0, x1, y1) into the shared block."""
    from multiprocessing import shared_memory

    r = _WORKER["readers"].get(path)
    if r is None:
        if len(_WORKER["readers"]) > 4:
            _WORKER["readers"].clear()
        r = _WORKER["readers"][path] = TiffReader(path, mpp=mpp)
    sh = _WORKER["shm"].get(shm_name)
    if sh is None:
        for k in list(_WORKER["shm"]):
            _WORKER["shm"].pop(k).close()
        sh = _WORKER["shm"][shm_name] = shared_memory.SharedMemory(name=shm_name)
        try:  # (Python < 3.13 registers every ATTACHMENT with the resource tracker, which then unlinks the parent's block when this worker exits)
            from multiprocessing import resource_tracker

            resource_tracker.unregister(sh._name, "shared_memory")
        except Exception:  # noqa: BLE001
            pass
    out = np.ndarray(shape, np.uint8, buffer=sh.buf)
    p = r.levels[level]
    for tt in tiles:
        r._place_tile(p, tt, window, out)
    return len(tiles)


class SlideInfo(object):
    def __init__(self, path, dims_wh, mpp=None, level_dimensions=None, level_downsamples=None):
        self.file_path = path
        self.slide_dimensions = (int(dims_wh[0]), int(dims_wh[1]))  # XY at baseline
        self.mpp = None if mpp is None else np.array([float(mpp[0]), float(mpp[1])], np.float64)
        self.level_dimensions = list(level_dimensions or [self.slide_dimensions])
        self.level_downsamples = list(level_downsamples or [1.0])
        self.level_count = len(self.level_dimensions)


def _strip_jfif_app0(data):
    """Remove APP0 ("JFIF" / "JFXX") segments that precede the first SOF / SOS of a JPEG stream; everything else is kept byte for byte."""
    out, i, n = bytearray(data[:2]), 2, len(data)
    while i + 4 <= n and data[i] == 0xFF:
        m = data[i + 1]
        if m in (0xC0, 0xC1, 0xC2, 0xDA) or m == 0xD9:  # frame header / scan / EOI: the rest is copied as it is
            break
        seg = (data[i + 2] << 8) | data[i + 3]
        if m != 0xE0:
            out += data[i:i + 2 + seg]
        i += 2 + seg
    return bytes(out) + data[i:]


def area_tables(o0, o1, rel, s0, n, size):
    """The numbers of _resample_axis' area-mean branch (rel >= 1) for output pixels o0 .. o1-1, as tables a device kernel can apply
    (cerb_resample_area: same float32 products, sums and quotient in the same order -> the same bytes): idx int32 [n_out, T] (positions in the
    source window s0 .. s0 + n - 1, clipped), w float32 [n_out, T] (0 where a tap lies outside), ws float32 [n_out] (the divisor), rep int32 [n_out]
    (>= 0: the output pixel lies wholly past the level's end and repeats that source position)."""
    o = np.arange(o0, o1, dtype=np.float64)
    lo, hi = o * rel, np.minimum((o + 1) * rel, float(size))
    first = np.floor(lo).astype(np.int64)
    T = int(np.ceil(rel)) + 1
    idx = np.zeros((len(o), T), np.int32)
    w = np.zeros((len(o), T), np.float32)
    wsum = np.zeros(len(o), np.float64)
    for t in range(T):
        i = first + t
        wgt = np.clip(np.minimum(hi, i + 1.0) - np.maximum(lo, i.astype(np.float64)), 0.0, 1.0)
        ok = (i - s0 >= 0) & (i - s0 < n)
        wgt = np.where(ok, wgt, 0.0)
        idx[:, t] = np.clip(i - s0, 0, n - 1)
        w[:, t] = wgt.astype(np.float32)
        wsum += wgt
    ws = np.maximum(wsum, 1e-12).astype(np.float32)
    rep = np.where(wsum <= 0.0, np.clip(size - 1 - s0, 0, n - 1), -1).astype(np.int32)
    return idx, w, ws, rep


def _resample_axis(src, axis, o0, o1, rel, s0, size):
    """Output pixels o0 .. o1-1 of a global resampling grid along `axis`: src holds the source pixels s0 .. s0 + n - 1 of a level that is
    `size` pixels long.  rel >= 1: area mean over [o * rel, (o + 1) * rel) clipped to the level (fractional end pixels weighted by their overlap);
    rel < 1: linear interpolation at (o + 0.5) * rel - 0.5 with edge replication."""
    src = np.moveaxis(src, axis, 0)
    n = src.shape[0]
    o = np.arange(o0, o1, dtype=np.float64)
    out = np.zeros((len(o),) + src.shape[1:], np.float32)
    if rel >= 1:
        idx, w, ws, rep = area_tables(o0, o1, rel, s0, n, size)
        tail = (-1,) + (1,) * (src.ndim - 1)
        for t in range(idx.shape[1]):
            out += src[idx[:, t]] * w[:, t].reshape(tail)
        out /= ws.reshape(tail)
        beyond = rep >= 0  # output pixels wholly past the level's end: replicate the last source pixel like the integer-factor path
        if beyond.any():
            out[beyond] = src[rep[beyond]]
    else:
        c = np.clip((o + 0.5) * rel - 0.5, 0.0, size - 1.0)
        i0 = np.floor(c).astype(np.int64)
        i1 = np.minimum(i0 + 1, size - 1)
        f = (c - i0).astype(np.float32).reshape((-1,) + (1,) * (src.ndim - 1))
        out = src[np.clip(i0 - s0, 0, n - 1)] * (1.0 - f) + src[np.clip(i1 - s0, 0, n - 1)] * f
    return np.moveaxis(out, 0, axis)


class _Rows(object):
    """(H, W, 3) row source: rows[a:b] -> uint8 [b - a, W, 3] at the reader's requested resolution."""

    def __init__(self, reader, resolution, units):
        self.reader, self.resolution, self.units = reader, resolution, units
        w, h = reader.slide_dimensions(resolution, units)
        self.shape = (int(h), int(w), 3)
        self.dtype = np.dtype(np.uint8)
        # rows per storage tile when this resolution IS a stored level: a caller that reads whole multiples of it (wsi.SlabUploader's chunks)
        # decodes every tile once instead of once per chunk that touches it
        self.row_align = 1
        lv = getattr(reader, "levels", None)
        if lv is not None and abs(reader._scale(resolution, units) - 1.0) < 1e-9 and getattr(lv[0], "tiled", False):
            self.row_align = int(lv[0].th)

    def device_plan(self):
        """None, or how a caller with a GPU gets these rows WITHOUT the host-side reduction (wsi.SlabUploader): a 40x scan (0.25 mpp, pyramid levels
        x1 / x4 / x16) read at the 0.5 mpp the network runs on is a x2 reduction of level 0 -- 4 source pixels per output pixel through numpy on ONE
        thread ran a slide at 9 Mpx/s against 150 from a file at the processing resolution.  The plan names the stored level, the remaining
        reduction `rel` >= 1 (`k`: the integer factor, exact box means; else area means on the global grid), and hands out source rows (decoded on
        the reader's pool) and the tables of the reduction; the device applies them (cerb_resample_box / cerb_resample_area) to the same bytes
        read_bounds returns."""
        r = self.reader
        if not hasattr(r, "levels"):
            return None
        s = r._scale(self.resolution, self.units)
        if abs(s - 1.0) < 1e-9:
            return None
        want = 1.0 / s
        lvl = 0
        for i, d in enumerate(r.info.level_downsamples):
            if d <= want * (1 + 1e-6):
                lvl = i
        rel = want / r.info.level_downsamples[lvl]
        if rel < 1.0 - 1e-9:  # enlarging (bilinear on the host): rare, and no reduction to save
            return None
        return _DevicePlan(r, lvl, rel, self.shape)

    def __getitem__(self, key):
        rows = key[0] if isinstance(key, tuple) else key
        if not isinstance(rows, slice):
            raise TypeError("row slices only")
        a, b, step = rows.indices(self.shape[0])
        assert step == 1
        out = self.reader.read_bounds((0, a, self.shape[1], b), self.resolution, self.units)
        return out if not isinstance(key, tuple) else out[(slice(None),) + tuple(key[1:])]


class _DevicePlan(object):
    """See _Rows.device_plan.  Source windows follow read_bounds exactly (same first / last source row for an output row range)."""

    def __init__(self, reader, lvl, rel, out_shape):
        self.reader, self.lvl, self.rel = reader, int(lvl), float(rel)
        self.lw, self.lh = [int(v) for v in reader.info.level_dimensions[lvl]]
        self.out_h, self.out_w = int(out_shape[0]), int(out_shape[1])
        k = int(round(rel))
        self.k = k if abs(rel - k) < 1e-9 else None
        p = reader.levels[lvl]
        self.tile_rows = int(p.th) if getattr(p, "tiled", False) else 1

    def source_rows(self, a, b):
        """[sy0, sy1) of the level for output rows [a, b)"""
        if self.k is not None:
            return a * self.k, min(b * self.k, self.lh)
        return max(int(np.floor(a * self.rel)), 0), min(int(np.ceil(b * self.rel)), self.lh)

    def out_rows_for_source_tiles(self, a, tiles):
        """the last output row (exclusive) whose source rows end within `tiles` storage tile rows of the tile row output row a starts in"""
        sy0 = self.source_rows(a, a + 1)[0]
        end = (sy0 // self.tile_rows + tiles) * self.tile_rows
        b = int(np.floor(end / self.rel))
        return max(a + 1, min(b, self.out_h))

    def read(self, sy0, sy1, out=None):
        """uint8 [sy1 - sy0, lw, 3] of the level (tiles decoded on the reader's pool)"""
        return self.reader._read_level(self.lvl, 0, sy0, self.lw, sy1, out=out)

    def row_tables(self, a, b, sy0, n):
        return area_tables(a, b, self.rel, sy0, n, self.lh)

    def col_tables(self):
        return area_tables(0, self.out_w, self.rel, 0, self.lw, self.lw)


class WSIReader(object):
    info = None

    @staticmethod
    def open(input_img, mpp=None, power=None):
        if isinstance(input_img, np.ndarray):
            return ArrayReader(input_img, mpp=mpp)
        path = str(input_img)
        ext = os.path.splitext(path)[1].lower()
        if ext == ".npy":
            return ArrayReader(np.load(path, mmap_mode="r"), path=path, mpp=mpp)
        if ext == ".txt":
            return SyntheticReader(path, mpp=mpp)
        if ext in (".tif", ".tiff", ".svs"):
            return TiffReader(path, mpp=mpp)
        from PIL import Image

        return ArrayReader(np.array(Image.open(path).convert("RGB")), path=path, mpp=mpp)

    # ---- resolution arithmetic (tiatoolbox semantics: "mpp", "baseline" = scale w.r.t. level 0, "level") ---------------------
    def _scale(self, resolution, units):
        """baseline pixels per requested pixel's inverse: requested = baseline * scale"""
        if units == "baseline":
            return float(np.atleast_1d(resolution)[0])
        if units == "level":
            return 1.0 / float(self.info.level_downsamples[int(resolution)])
        if units == "mpp":
            if self.info.mpp is None:
                return 1.0  # no scan resolution recorded (arrays, synthetic slides): the pixels ARE the processing resolution
            res = np.atleast_1d(np.asarray(resolution, np.float64))
            return float(self.info.mpp[0] / res[0])
        raise ValueError("units must be 'mpp', 'baseline' or 'level' (objective power needs metadata these readers do not carry), got %r" % (units,))

    def slide_dimensions(self, resolution, units):
        s = self._scale(resolution, units)
        w, h = self.info.slide_dimensions
        return np.array([int(round(w * s)), int(round(h * s))], np.int64)

    def rows(self, resolution=1.0, units="baseline"):
        return _Rows(self, resolution, units)

    def read_bounds(self, bounds, resolution=1.0, units="baseline"):
        """bounds = (x0, y0, x1, y1) in pixels of the REQUESTED resolution; returns uint8 [y1 - y0, x1 - x0, 3]."""
        x0, y0, x1, y1 = [int(v) for v in bounds]
        s = self._scale(resolution, units)
        if abs(s - 1.0) < 1e-9:
            return self._read_level(0, x0, y0, x1, y1)
        # level whose downsample is the largest one not above 1 / s
        want = 1.0 / s
        lvl = 0
        for i, d in enumerate(self.info.level_downsamples):
            if d <= want * (1 + 1e-6):
                lvl = i
        d = self.info.level_downsamples[lvl]
        rel = want / d  # remaining reduction from that level (>= 1), or < 1 when upsampling from level 0
        lw, lh = self.info.level_dimensions[lvl]
        k = int(round(rel))
        if rel >= 1 and abs(rel - k) < 1e-9:  # integer factor: exact box means
            sx0, sy0 = x0 * k, y0 * k
            sx1, sy1 = min(x1 * k, lw), min(y1 * k, lh)
            src = self._read_level(lvl, sx0, sy0, sx1, sy1).astype(np.float32)
            hh, ww = y1 - y0, x1 - x0
            pad = np.zeros((hh * k, ww * k, 3), np.float32)
            pad[: src.shape[0], : src.shape[1]] = src
            if src.shape[0] < hh * k:  # the slide ends inside the last output row / column: replicate the edge
                pad[src.shape[0]:, : src.shape[1]] = src[-1:]
            if src.shape[1] < ww * k:
                pad[:, src.shape[1]:] = pad[:, src.shape[1] - 1: src.shape[1]]
            return np.clip(np.rint(pad.reshape(hh, k, ww, k, 3).mean(axis=(1, 3))), 0, 255).astype(np.uint8)
        # non-integer factor: resample on ONE global grid -- output pixel o covers the source interval [o * rel, (o + 1) * rel) of the level (exact
        # area means; bilinear at (o + 0.5) * rel - 0.5 when enlarging) -- so that the pixels do not depend on how a caller cuts the slide into row
        # chunks or rank bands (a per-window resize would shift scale and phase with every window)
        pad = 1 if rel < 1 else 0
        sx0, sy0 = max(int(np.floor(x0 * rel)) - pad, 0), max(int(np.floor(y0 * rel)) - pad, 0)
        sx1, sy1 = min(int(np.ceil(x1 * rel)) + pad, lw), min(int(np.ceil(y1 * rel)) + pad, lh)
        src = self._read_level(lvl, sx0, sy0, sx1, sy1).astype(np.float32)
        tmp = _resample_axis(src, 0, y0, y1, rel, sy0, lh)
        out = _resample_axis(tmp, 1, x0, x1, rel, sx0, lw)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)

    def _read_level(self, level, x0, y0, x1, y1):
        raise NotImplementedError


class ArrayReader(WSIReader):
    def __init__(self, arr, path=None, mpp=None):
        assert arr.ndim == 3 and arr.shape[2] >= 3 and arr.dtype == np.uint8, "slides are uint8 [H, W, 3] arrays"
        self.arr = arr
        self.info = SlideInfo(path, (arr.shape[1], arr.shape[0]), mpp)

    def _read_level(self, level, x0, y0, x1, y1):
        return np.ascontiguousarray(self.arr[y0:y1, x0:x1, :3])

    def rows(self, resolution=1.0, units="baseline"):
        if abs(self._scale(resolution, units) - 1.0) < 1e-9 and self.arr.shape[2] == 3:
            return self.arr  # the array itself (np.memmap for .npy files: SlabUploader reads it chunk by chunk)
        return _Rows(self, resolution, units)


class SyntheticReader(WSIReader):
    """`synthetic:<H>x<W>:<seed>`: there are no host pixels; run_infer_wsi.py generates the band on the device."""

    def __init__(self, path, mpp=None):
        _, dims, seed = open(path).read().strip().split(":")
        h, w = [int(v) for v in dims.split("x")]
        self.seed = int(seed)
        self.info = SlideInfo(path, (w, h), mpp)

    def _read_level(self, level, x0, y0, x1, y1):
        from .wsi import synth_slide

        return synth_slide(y1 - y0, self.info.slide_dimensions[0], y0=y0, seed=self.seed)[:, x0:x1].cpu().numpy()


# ---- TIFF ------------------------------------------------------------------------------------------------------------------
_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4), 10: ("ii", 8),
          11: ("f", 4), 12: ("d", 8), 13: ("I", 4), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


class _Page(object):
    pass


class TiffReader(WSIReader):
    def __init__(self, path, mpp=None):
        self.path = path
        self.fh = open(path, "rb")
        head = self.fh.read(16)
        self.bo = "<" if head[:2] == b"II" else ">"
        magic = struct.unpack(self.bo + "H", head[2:4])[0]
        if magic == 42:
            self.big, off = False, struct.unpack(self.bo + "I", head[4:8])[0]
        elif magic == 43:
            self.big, off = True, struct.unpack(self.bo + "Q", head[8:16])[0]
        else:
            raise ValueError("%s: not a TIFF file" % path)
        pages = []
        while off:
            page, off = self._read_ifd(off)
            pages.append(page)
        # pyramid = the full-resolution page plus every REDUCED page of the same aspect (labels / macros of .svs files drop out)
        base = pages[0]
        levels = [base]
        for p in pages[1:]:
            ds_x, ds_y = base.w / p.w, base.h / p.h
            if p.w < base.w and abs(ds_x - ds_y) / ds_x < 0.02 and p.samples >= 3 and (p.subfile & 1 or p.tiled):
                levels.append(p)
        levels.sort(key=lambda p: -p.w)
        self.levels = levels
        file_mpp = base.mpp
        self.info = SlideInfo(path, (base.w, base.h), mpp if mpp is not None else file_mpp, [(p.w, p.h) for p in levels],
                              [base.w / p.w for p in levels])

    def _read_ifd(self, off):
        bo, fh = self.bo, self.fh
        fh.seek(off)
        n = struct.unpack(bo + ("Q" if self.big else "H"), fh.read(8 if self.big else 2))[0]
        esz = 20 if self.big else 12
        raw = fh.read(n * esz + (8 if self.big else 4))
        tags = {}
        for i in range(n):
            e = raw[i * esz:(i + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            cnt = struct.unpack(bo + ("Q" if self.big else "I"), e[4:12] if self.big else e[4:8])[0]
            val = e[12:20] if self.big else e[8:12]
            if typ not in _TYPES:
                continue
            fmt, size = _TYPES[typ]
            nbytes = cnt * size
            if nbytes > len(val):
                pos = struct.unpack(bo + ("Q" if self.big else "I"), val)[0]
                fh.seek(pos)
                data = fh.read(nbytes)
            else:
                data = val[:nbytes]
            if typ == 2:
                tags[tag] = data.split(b"\x00")[0].decode("latin1")
            elif typ == 7:
                tags[tag] = data
            elif typ in (5, 10):
                v = struct.unpack(bo + fmt[0] * (2 * cnt), data)
                tags[tag] = [v[2 * k] / v[2 * k + 1] if v[2 * k + 1] else 0.0 for k in range(cnt)]
            else:
                tags[tag] = list(struct.unpack(bo + fmt * cnt, data))
        nxt = struct.unpack(bo + ("Q" if self.big else "I"), raw[n * esz:])[0]
        p = _Page()
        p.w, p.h = tags[256][0], tags[257][0]
        p.samples = tags.get(277, [1])[0]
        p.bits = tags.get(258, [8])[0]
        p.compression = tags.get(259, [1])[0]
        p.photometric = tags.get(262, [2])[0]
        p.planar = tags.get(284, [1])[0]
        p.predictor = tags.get(317, [1])[0]
        p.subfile = tags.get(254, [0])[0]
        p.jpeg_tables = tags.get(347)
        p.tiled = 322 in tags
        if p.tiled:
            p.tw, p.th = tags[322][0], tags[323][0]
            p.offsets, p.counts = tags[324], tags[325]
        else:
            p.tw, p.th = p.w, tags.get(278, [p.h])[0]
            p.offsets, p.counts = tags[273], tags[279]
        p.mpp = None
        desc = tags.get(270, "")
        if isinstance(desc, str) and "MPP" in desc:  # Aperio: "...|MPP = 0.2520|..."
            try:
                v = float(desc.split("MPP")[1].split("=")[1].split("|")[0])
                p.mpp = (v, v)
            except (IndexError, ValueError):
                pass
        if p.mpp is None and 282 in tags and 283 in tags and tags.get(296, [2])[0] in (2, 3) and tags[282][0] > 0 and tags[283][0] > 0:
            per_um = 25400.0 if tags.get(296, [2])[0] == 2 else 10000.0  # pixels per inch / per centimetre
            if tags[282][0] > 100:  # 72 / 96 dpi are display defaults, not scan resolutions
                p.mpp = (per_um / tags[282][0], per_um / tags[283][0])
        if p.bits != 8 or p.planar != 1:
            raise NotImplementedError("%s: only 8-bit chunky RGB(A) pages are supported" % self.path)
        return p, nxt

    def _decode(self, p, idx, rows, cols):
        data = os.pread(self.fh.fileno(), p.counts[idx], p.offsets[idx])  # positional read: decode threads share the descriptor
        c = p.compression
        if c == 1:
            buf = np.frombuffer(data, np.uint8)
        elif c in (8, 32946):
            buf = np.frombuffer(zlib.decompress(data), np.uint8)
        elif c == 7:
            from PIL import Image

            if p.jpeg_tables:  # abbreviated streams: tables (minus EOI) + tile (minus SOI)
                data = p.jpeg_tables[:-2] + data[2:]
            if p.photometric == 2:
                # PhotometricInterpretation = RGB (most Aperio .svs): the components ARE R, G, B, but the stream carries neither a JFIF nor
                # an Adobe marker, and libjpeg then guesses YCbCr for component ids 1, 2, 3 and converts.  An Adobe APP14 segment with
                # transform = 0 right behind SOI states "no colour transform" (libjpeg honours it before any guess).
                # A JFIF APP0 segment takes precedence over the Adobe marker in libjpeg (JFIF means YCbCr for three components), so any APP0
                # in front of the frame header is dropped first.
                data = _strip_jfif_app0(data)
                data = data[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00" + data[2:]
            img = Image.open(io.BytesIO(data))
            if img.mode != "RGB":
                img = img.convert("RGB")
            return np.asarray(img)[:rows, :cols]
        elif c in JP2K_CODECS:
            return _decode_jp2k_tile(data, c, self.path, idx)[:rows, :cols]
        elif c == 5:  # libcerberus_host.so (csrc/host_codecs.c) through ctypes: the interpreter lock is released, decode threads run side by side
            from . import _hostlib

            buf = _hostlib.lzw_decode(data, rows * cols * p.samples)
        elif c == 32773:
            from . import _hostlib

            buf = _hostlib.packbits_decode(data, rows * cols * p.samples)
        else:
            raise NotImplementedError("%s: TIFF compression %d is not supported" % (self.path, c))
        if buf.size < rows * cols * p.samples:
            raise ValueError("%s: strip / tile %d holds %d bytes, %d x %d x %d pixels need %d" % (self.path, idx, buf.size, rows, cols, p.samples, rows * cols * p.samples))
        arr = buf[: rows * cols * p.samples].reshape(rows, cols, p.samples)
        if p.predictor == 2:
            if arr.flags.writeable and arr.flags.c_contiguous:
                from . import _hostlib

                arr = _hostlib.unpredict_u8(arr)  # in place on the decoder's own buffer
            else:  # (read-only views of raw / zlib bytes)
                arr = np.cumsum(arr, axis=1, dtype=np.uint8)
        return arr[:, :, :3]

    def _place_tile(self, p, tt, window, out):
        """decode tile (ty, tx) of page p and write the part of it inside window = (x0, y0, x1, y1) into out (the window's pixels)"""
        ty, tx = tt
        x0, y0, x1, y1 = window
        across = -(-p.w // p.tw)
        # tiles are stored whole (padded); strips are cropped to the image on the last rows
        rows = p.th if p.tiled else min(p.th, p.h - ty * p.th)
        cols = p.tw if p.tiled else p.w
        tile = self._decode(p, ty * across + tx, rows, cols)
        gy0, gx0 = ty * p.th, tx * p.tw
        a0, a1 = max(y0, gy0), min(y1, gy0 + tile.shape[0])
        b0, b1 = max(x0, gx0), min(x1, gx0 + tile.shape[1])
        if a1 > a0 and b1 > b0:  # tiles do not overlap: every thread / process writes its own window of `out`
            out[a0 - y0:a1 - y0, b0 - x0:b1 - x0] = tile[a0 - gy0:a1 - gy0, b0 - gx0:b1 - gx0]

    def _shared_block(self, nbytes):
        """this reader's shared-memory block (grown on demand, unlinked when the reader goes)"""
        from multiprocessing import shared_memory

        sh = getattr(self, "_shm", None)
        if sh is None or sh.size < nbytes:
            if sh is not None:
                _SHM_LIVE.pop(sh.name, None)
                sh.close()
                sh.unlink()
            self._shm = sh = shared_memory.SharedMemory(create=True, size=int(nbytes * 1.25) + 4096)
            _SHM_LIVE[sh.name] = sh
        return sh

    def __del__(self):
        sh = getattr(self, "_shm", None)
        if sh is not None:
            try:
                _SHM_LIVE.pop(sh.name, None)
                sh.close()
                sh.unlink()
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def _read_level(self, level, x0, y0, x1, y1, out=None):
        """out: a uint8 array at least as large as the window -- the pixels go into its first rows / columns and that view is returned
        (wsi.SlabUploader hands its pinned staging buffer in: one copy less per chunk)"""
        p = self.levels[level]
        x0, y0, x1, y1 = max(0, x0), max(0, y0), min(p.w, x1), min(p.h, y1)
        shape = (max(0, y1 - y0), max(0, x1 - x0), 3)
        dest = None if out is None else out[: shape[0], : shape[1]]
        window = (x0, y0, x1, y1)
        tiles = [(ty, tx) for ty in range(y0 // p.th, -(-y1 // p.th)) for tx in range(x0 // p.tw, -(-x1 // p.tw))]
        nbytes = shape[0] * shape[1] * 3
        procs = _proc_pool() if (len(tiles) >= PROC_MIN_TILES and p.compression in PROC_CODECS and not _PROCS.get("off") and _shm_has_room(nbytes)) else None
        if procs is not None:
            import threading

            lock = self.__dict__.setdefault("_shm_lock", threading.Lock())
            with lock:
                sh = self._shared_block(nbytes)
                n = _PROCS["n"]
                per = max(8, -(-len(tiles) // (4 * n)))
                mpp = None if self.info.mpp is None else tuple(float(v) for v in self.info.mpp)
                try:
                    procs.run([(self.path, mpp, level, tiles[i:i + per], window, sh.name, shape) for i in range(0, len(tiles), per)])
                    got = np.ndarray(shape, np.uint8, buffer=sh.buf)
                    if dest is None:
                        return got.copy()
                    np.copyto(dest, got)
                    return dest
                except Exception as e:  # noqa: BLE001
                    # a worker is gone or failed (a container whose /dev/shm filled up kills it with SIGBUS): no more worker processes in this run, and
                    # this read is done again on the threads -- which raise the real error themselves if the file is the problem
                    import logging

                    _shutdown_procs()
                    _PROCS["off"] = True
                    logging.getLogger("cerberus_amd.reader").warning("tile-decode worker processes switched off (%s): decoding on threads", str(e).splitlines()[-1][:200])
        out = np.zeros(shape, np.uint8) if dest is None else dest
        if dest is not None:
            dest[...] = 0
        from . import _hostlib

        if p.compression in _hostlib.NATIVE_CODECS and tiles and out.strides[1:] == (3, 1):
            # raw / deflate / LZW / PackBits: the whole window in ONE native call (csrc/host_codecs.c: pread + decode + predictor + placement on its own
            # pthreads, the interpreter lock released throughout) -- per-tile calls from Python threads stop scaling at two threads
            across = -(-p.w // p.tw)
            idx = np.array([ty * across + tx for ty, tx in tiles], np.int64)
            tys, txs = np.array([t[0] for t in tiles], np.int32), np.array([t[1] for t in tiles], np.int32)
            rows_t = np.full(len(tiles), p.th, np.int32) if p.tiled else np.minimum(p.th, p.h - tys * p.th).astype(np.int32)
            try:
                _hostlib.read_tiles(self.fh.fileno(), p.compression, p.predictor, p.samples, p.tw if p.tiled else p.w, np.asarray(p.offsets, np.int64)[idx],
                                    np.asarray(p.counts, np.int64)[idx], rows_t, txs * p.tw, tys * p.th, window, out, decode_threads())
            except _hostlib.HostCodecError as e:
                raise ValueError("%s: %s" % (self.path, e)) from None
            return out
        pool = decode_pool()
        if pool is None or len(tiles) < 2:
            for tt in tiles:
                self._place_tile(p, tt, window, out)
        else:  # libjpeg / zlib release the interpreter lock while they decode
            list(pool.map(lambda tt: self._place_tile(p, tt, window, out), tiles))
        return out


def _jp2k_siz(data):
    """(width, height, [(XRsiz, YRsiz)] per component) out of a JPEG 2000 codestream's SIZ marker segment (ISO 15444-1 A.5.1; a JP2 file's boxes
    are skipped to its codestream) -- None when there is no SIZ where one belongs."""
    k = data.find(b"\xff\x4f\xff\x51")
    if k < 0 or len(data) < k + 42:
        return None
    xsiz, ysiz, xo, yo = struct.unpack(">IIII", data[k + 8:k + 24])
    csiz = struct.unpack(">H", data[k + 40:k + 42])[0]
    body = data[k + 42:k + 42 + 3 * csiz]
    if len(body) < 3 * csiz:
        return None
    return xsiz - xo, ysiz - yo, [(body[3 * i + 1], body[3 * i + 2]) for i in range(csiz)]


def _jp2k_component_sampling(data):
    siz = _jp2k_siz(data)
    return None if siz is None else siz[2]


def _decode_jp2k_tile(data, compression, path, idx):
    """One JPEG 2000 tile (a raw codestream, as Aperio .svs files and generic TIFF writers store them) -> uint8 [rows, cols, 3], through OpenJPEG
    behind PIL.  33005 / 34712: the components are R, G, B.  33003: the components are Y, Cb, Cr (full range, JFIF matrix -- what OpenSlide's
    Aperio back end converts with).  Chroma SUBSAMPLED inside the codestream (most scanner-written 33003 files: 4:2:2): OpenJPEG + PIL return
    the chroma samples replicated (sample x / XRsiz, y / YRsiz) and ALREADY converted to RGB (PIL takes a subsampled three-component codestream for
    sYCC) -- pinned by tests/golden/jp2k_subsampled.npz (codestreams written by the bundled OpenJPEG itself with per-component dx / dy; the
    generator is committed beside the other golden-vector scripts) for 4:2:2 and 4:2:0 on sizes that are multiples of the factors, the only case accepted: on an odd width the same encode -> decode round
    trip does NOT return the stored planes (the fixture's third stream; encoder or decoder, not established), so anything else is refused by name.  Parity with OpenSlide's own arithmetic is unpinned (it is
    not in the image); the rule -- replication + JFIF -- is the one its Aperio back end documents."""
    from PIL import Image, features

    if not features.check_codec("jpg_2000"):
        raise NotImplementedError("%s: JPEG 2000 tiles (TIFF compression %d) need a PIL built with OpenJPEG" % (path, compression))
    siz = _jp2k_siz(data)
    subsampled = siz is not None and any(s_ != (1, 1) for s_ in siz[2][:3])
    if subsampled:
        w, h, samp = siz
        ok = (compression == 33003 and len(samp) == 3 and samp[0] == (1, 1) and samp[1] == samp[2] and samp[1] in ((2, 1), (2, 2))
              and w % samp[1][0] == 0 and h % samp[1][1] == 0)
        if not ok:
            raise NotImplementedError("%s: tile %d is a %d x %d JPEG 2000 codestream with component sampling %s under TIFF compression %d: only YCbCr (33003) 4:2:2 / 4:2:0 "
                                      "on sizes that are multiples of the factors is supported (what the decoders in this image return correctly)" % (path, idx, w, h, samp, compression))
    try:
        img = Image.open(io.BytesIO(data))
        img.load()
    except Exception as e:  # noqa: BLE001
        raise ValueError("%s: tile %d: JPEG 2000 codestream not decodable (%s)" % (path, idx, e)) from None
    if subsampled:
        if img.mode != "RGB":
            raise NotImplementedError("%s: tile %d: PIL returned mode %s for a subsampled YCbCr codestream (expected its sYCC -> RGB conversion)" % (path, idx, img.mode))
        return np.asarray(img)
    if img.mode not in ("RGB", "RGBA", "YCbCr") and compression != 33003:
        img = img.convert("RGB")
    bands = img.split()
    if len(bands) < 3:
        raise NotImplementedError("%s: tile %d: %d-component JPEG 2000 tiles are not supported" % (path, idx, len(bands)))
    if compression == 33003:
        return np.asarray(Image.merge("YCbCr", bands[:3]).convert("RGB"))
    return np.asarray(Image.merge("RGB", bands[:3]))


def tiff_lzw_encode(data):
    """TIFF 6.0 section 13 encoder (MSB-first, early change, ClearCode first, a ClearCode when the table holds 4094 entries, EndOfInformation last):
    write_tiled_tiff(compress="lzw") -- tiles of a TILED file, which PIL cannot write; plain Python (a writer for tests, benchmarks and array
    conversion: ~0.3 s per 256 x 256 tile), the DEcoder is native (csrc/host_codecs.c)."""
    out = bytearray()
    bitbuf, bitcnt = 0, 0
    nbits = 9

    def put(code):
        nonlocal bitbuf, bitcnt
        bitbuf = (bitbuf << nbits) | code
        bitcnt += nbits
        while bitcnt >= 8:
            out.append((bitbuf >> (bitcnt - 8)) & 0xFF)
            bitcnt -= 8
        bitbuf &= (1 << bitcnt) - 1

    table = {bytes((i,)): i for i in range(256)}
    nxt = 258
    put(256)
    w = b""
    for byte in bytes(data):
        wc = w + bytes((byte,))
        if wc in table:
            w = wc
            continue
        put(table[w])
        table[wc] = nxt
        nxt += 1
        # the DECODER's table is one entry behind the encoder's, and it widens its codes when ITS table holds 511 / 1023 / 2047 entries
        # (one code early): seen from here that is nxt = 512 / 1024 / 2048
        if nxt == 4094:
            put(256)
            table = {bytes((i,)): i for i in range(256)}
            nxt, nbits = 258, 9
        else:
            nbits = 12 if nxt >= 2048 else 11 if nxt >= 1024 else 10 if nxt >= 512 else 9
        w = bytes((byte,))
    if w:
        put(table[w])
        nxt += 1  # (the decoder adds an entry for this code too, and widens on it)
        if nxt != 4094:
            nbits = 12 if nxt >= 2048 else 11 if nxt >= 1024 else 10 if nxt >= 512 else 9
    put(257)
    if bitcnt:
        out.append((bitbuf << (8 - bitcnt)) & 0xFF)
    return bytes(out)


def tiff_packbits_encode(data):
    """TIFF 6.0 section 9 encoder: runs of 3 and more as replicate packets, everything else as literal packets of up to 128 bytes."""
    data = bytes(data)
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 127:
            j += 1
        if j - i >= 2:
            out += bytes((257 - (j - i + 1), data[i]))
            i = j + 1
            continue
        k = i
        while k < n and k - i < 128 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
            k += 1
        out += bytes((k - i - 1,)) + data[i:k]
        i = k
    return bytes(out)


def write_tiled_tiff(path, levels, tile=256, mpp=None, compress=True, description=None, encode=None, predictor=1):
    """Minimal pyramidal tiled TIFF writer (deflate, raw, LZW or PackBits tiles: `compress`) -- for tests and for converting arrays; levels[0] is full
    resolution, the others are reduced pages (NewSubfileType 1).  encode = (function tile [t, t, 3] uint8 -> bytes, TIFF compression
    code) replaces the built-in tile encoders (the tests write Aperio-style JPEG tiles and LZW tiles through it); predictor=2 stores every row as
    differences to the pixel on its left (TIFF 6.0 section 14: tag 317) before the tile is encoded."""
    bo = "<"
    # compress: True / "deflate" (zlib level 6), False (raw), "lzw", "packbits" -- (tile encoder, TIFF Compression tag)
    builtin = {True: (lambda t: zlib.compress(t.tobytes(), 6), 8), "deflate": (lambda t: zlib.compress(t.tobytes(), 6), 8), False: (lambda t: t.tobytes(), 1),
               None: (lambda t: t.tobytes(), 1), "lzw": (lambda t: tiff_lzw_encode(t.tobytes()), 5), "packbits": (lambda t: tiff_packbits_encode(t.tobytes()), 32773)}[compress]
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack(bo + "HI", 42, 0))
        prev_next_pos = 4
        for li, img in enumerate(levels):
            img = np.ascontiguousarray(img[:, :, :3], np.uint8)
            h, w = img.shape[:2]
            offs, cnts = [], []
            for ty in range(-(-h // tile)):
                for tx in range(-(-w // tile)):
                    t = np.zeros((tile, tile, 3), np.uint8)
                    blk = img[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
                    t[: blk.shape[0], : blk.shape[1]] = blk
                    if predictor == 2:
                        t[:, 1:] = t[:, 1:] - t[:, :-1]  # (uint8 arithmetic wraps: modulo 256; numpy evaluates the right-hand side first)
                    data = encode[0](t) if encode else builtin[0](t)
                    offs.append(fh.tell())
                    cnts.append(len(data))
                    fh.write(data)
                    if fh.tell() & 1:
                        fh.write(b"\0")
            entries = []

            def put(tag, typ, vals):
                entries.append((tag, typ, vals))

            put(254, 4, [1 if li else 0])
            put(256, 4, [w])
            put(257, 4, [h])
            put(258, 3, [8, 8, 8])
            put(259, 3, [encode[1] if encode else builtin[1]])
            put(262, 3, [2])
            if description and li == 0:
                put(270, 2, description.encode("latin1") + b"\0")
            put(277, 3, [3])
            if mpp is not None:
                scale = levels[0].shape[1] / w
                put(282, 5, [(int(round(1e4 / (mpp * scale) * 1000)), 1000)])
                put(283, 5, [(int(round(1e4 / (mpp * scale) * 1000)), 1000)])
            put(284, 3, [1])
            if mpp is not None:
                put(296, 3, [3])
            if predictor == 2:
                put(317, 3, [2])
            put(322, 4, [tile])
            put(323, 4, [tile])
            put(324, 4, offs)
            put(325, 4, cnts)
            entries.sort(key=lambda e: e[0])
            # out-of-line values first
            blobs = {}
            for tag, typ, vals in entries:
                if typ == 2:
                    raw = vals
                elif typ == 5:
                    raw = b"".join(struct.pack(bo + "II", a, b) for a, b in vals)
                else:
                    raw = struct.pack(bo + {3: "H", 4: "I"}[typ] * len(vals), *vals)
                if len(raw) > 4:
                    if fh.tell() & 1:
                        fh.write(b"\0")
                    blobs[tag] = fh.tell()
                    fh.write(raw)
            if fh.tell() & 1:
                fh.write(b"\0")
            ifd_pos = fh.tell()
            fh.write(struct.pack(bo + "H", len(entries)))
            for tag, typ, vals in entries:
                if typ == 2:
                    raw, cnt = vals, len(vals)
                elif typ == 5:
                    raw, cnt = b"".join(struct.pack(bo + "II", a, b) for a, b in vals), len(vals)
                else:
                    raw, cnt = struct.pack(bo + {3: "H", 4: "I"}[typ] * len(vals), *vals), len(vals)
                fh.write(struct.pack(bo + "HHI", tag, typ, cnt))
                fh.write(struct.pack(bo + "I", blobs[tag]) if len(raw) > 4 else raw.ljust(4, b"\0"))
            next_pos = fh.tell()
            fh.write(struct.pack(bo + "I", 0))
            end = fh.tell()
            fh.seek(prev_next_pos)
            fh.write(struct.pack(bo + "I", ifd_pos))
            fh.seek(end)
            prev_next_pos = next_pos
