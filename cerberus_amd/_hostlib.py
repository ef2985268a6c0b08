"""ctypes binding of libcerberus_host.so (include/cerberus_host.h): the slide reader's host-side byte codecs.  torch-free -- the decode worker
processes load it too.  There is no Python fallback: a missing library is an error that names the build command."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcerberus_host.so")
EXPORTS = ["cerb_host_version", "cerb_host_lzw_decode", "cerb_host_packbits_decode", "cerb_host_unpredict_u8", "cerb_host_tiff_read_tiles"]
NATIVE_CODECS = (1, 5, 8, 32946, 32773)  # TIFF Compression tags cerb_host_tiff_read_tiles decodes (7 = JPEG stays with libjpeg behind the reader)
_LIB = None


class HostCodecError(ValueError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -m cerberus_amd.build` (gcc; no GPU needed)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)  # CDLL calls release the interpreter lock: the reader's decode threads run these side by side
        u8p, i64 = ctypes.c_void_p, ctypes.c_int64
        L.cerb_host_version.restype = ctypes.c_int
        L.cerb_host_version.argtypes = []
        for name in ("cerb_host_lzw_decode", "cerb_host_packbits_decode"):
            f = getattr(L, name)
            f.restype, f.argtypes = i64, [ctypes.c_char_p, i64, u8p, i64]
        L.cerb_host_unpredict_u8.restype = None
        L.cerb_host_unpredict_u8.argtypes = [u8p, i64, i64, ctypes.c_int]
        ci = ctypes.c_int
        L.cerb_host_tiff_read_tiles.restype = ci
        L.cerb_host_tiff_read_tiles.argtypes = [ci, ci, ci, ci, ci, ci, u8p, u8p, u8p, u8p, u8p, ci, ci, ci, ci, u8p, i64, ci, ctypes.POINTER(ctypes.c_int32)]
        _LIB = L
    return _LIB


def _decode(name, what, data, expected):
    out = np.empty(int(expected), np.uint8)
    n = getattr(lib(), name)(data, len(data), out.ctypes.data, int(expected))
    if n == -2:
        raise HostCodecError("%s: pre-TIFF-6.0 'old-style' (LSB-first) stream is not supported" % what)
    if n < 0:
        raise HostCodecError("corrupt %s stream in a TIFF strip / tile" % what)
    if n < expected:  # a stream that ends early: the rest of the tile is zero (what the pure-Python decoders' short result meant after reshape would not)
        out[n:] = 0
    return out


def lzw_decode(data, expected):
    """TIFF LZW strip / tile -> uint8 array of `expected` bytes (zero-filled past an early end of the stream)."""
    return _decode("cerb_host_lzw_decode", "LZW", data, expected)


def packbits_decode(data, expected):
    """TIFF PackBits strip / tile -> uint8 array of `expected` bytes."""
    return _decode("cerb_host_packbits_decode", "PackBits", data, expected)


def unpredict_u8(arr):
    """Undo TIFF Predictor 2 in place on a C-contiguous writable uint8 [rows, cols, samples] array; returns it."""
    assert arr.dtype == np.uint8 and arr.ndim == 3 and arr.flags.c_contiguous and arr.flags.writeable
    lib().cerb_host_unpredict_u8(arr.ctypes.data, arr.shape[0], arr.shape[1], arr.shape[2])
    return arr


_READ_ERRORS = {-1: "corrupt LZW / PackBits stream", -2: "pre-TIFF-6.0 'old-style' (LSB-first) LZW stream is not supported", -3: "the file ends inside the tile's bytes",
                -4: "fewer decoded bytes than the tile's pixels need", -5: "corrupt deflate stream (zlib)", -6: "out of memory", -7: "unsupported compression",
                -8: "bad arguments"}


def read_tiles(fd, codec, predictor, samples, tile_cols, offsets, counts, rows, gx0, gy0, window, out, n_threads):
    """cerb_host_tiff_read_tiles: every tile / strip of the lists read (pread on fd), decoded, un-predicted and placed into `out` -- a writable uint8
    [>= y1 - y0, >= x1 - x0, 3] array whose rows are contiguous (a view of a wider / taller buffer is fine) -- on n_threads native threads, the
    interpreter lock released for the whole call.  Raises HostCodecError naming the failing tile."""
    x0, y0, x1, y1 = (int(v) for v in window)
    assert out.dtype == np.uint8 and out.ndim == 3 and out.shape[2] == 3 and out.strides[2] == 1 and out.strides[1] == 3 and out.flags.writeable
    assert out.shape[0] >= y1 - y0 and out.shape[1] >= x1 - x0
    offs, cnts = np.ascontiguousarray(offsets, np.int64), np.ascontiguousarray(counts, np.int64)
    rws, gx, gy = (np.ascontiguousarray(v, np.int32) for v in (rows, gx0, gy0))
    n = int(offs.shape[0])
    assert cnts.shape[0] == n and rws.shape[0] == n and gx.shape[0] == n and gy.shape[0] == n
    bad = ctypes.c_int32(-1)
    rc = lib().cerb_host_tiff_read_tiles(int(fd), int(codec), int(predictor), int(samples), int(tile_cols), n, offs.ctypes.data, cnts.ctypes.data, rws.ctypes.data,
                                         gx.ctypes.data, gy.ctypes.data, x0, y0, x1, y1, out.ctypes.data, int(out.strides[0]), max(1, int(n_threads)), ctypes.byref(bad))
    if rc != 0:
        raise HostCodecError("TIFF strip / tile %d (at level pixel %s): %s" % (bad.value, (int(gx[bad.value]), int(gy[bad.value])) if 0 <= bad.value < n else "?",
                                                                               _READ_ERRORS.get(rc, "error %d" % rc)))
    return out
