"""TEST INFRASTRUCTURE: writes tests/golden/jp2k_subsampled.npz -- three small JPEG 2000 codestreams whose chroma components are SUBSAMPLED inside the
codestream (4:2:2 on a 128 x 128 tile, 4:2:0 on 64 x 64, 4:2:2 with an odd width -- whose round trip through this library does NOT return the stored planes, so the reader
refuses such layouts), as Aperio's compression 33003 stores its tiles, together with the component planes they were
encoded from (lossless 5-3 wavelet).  PIL's encoder cannot subsample, so the codestreams come from the OpenJPEG 2.5 library PIL bundles, driven
through ctypes (opj_image_create with per-component dx / dy).  Run in the build container: python oracle/gen_golden_jp2k.py
The reference reads such slides through OpenSlide (infer/wsi.py:521-531), which is not in this image: the fixture pins what the reader does with
subsampled chroma (nearest replication + JFIF YCbCr -> RGB, what OpenJPEG + PIL return), not OpenSlide's arithmetic."""
import ctypes as C
import glob
import os
import tempfile

import numpy as np
import PIL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cmpt(C.Structure):  # opj_image_cmptparm_t
    _fields_ = [(n, C.c_uint32) for n in ("dx", "dy", "w", "h", "x0", "y0", "prec", "bpp", "sgnd")]


class Comp(C.Structure):  # opj_image_comp_t
    _fields_ = [(n, C.c_uint32) for n in ("dx", "dy", "w", "h", "x0", "y0", "prec", "bpp", "sgnd", "resno", "factor")] + [("data", C.POINTER(C.c_int32)), ("alpha", C.c_uint16)]


class Img(C.Structure):  # opj_image_t
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32), ("numcomps", C.c_uint32), ("cs", C.c_int),
                ("comps", C.POINTER(Comp)), ("icc", C.c_void_p), ("icclen", C.c_uint32)]


def _lib():
    path = glob.glob(os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs", "libopenjp2-*.so*"))[0]
    L = C.CDLL(path)
    L.opj_version.restype = C.c_char_p
    assert L.opj_version().startswith(b"2.5"), L.opj_version()
    L.opj_image_create.restype, L.opj_image_create.argtypes = C.POINTER(Img), [C.c_uint32, C.POINTER(Cmpt), C.c_int]
    L.opj_create_compress.restype, L.opj_create_compress.argtypes = C.c_void_p, [C.c_int]
    L.opj_setup_encoder.restype, L.opj_setup_encoder.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Img)]
    L.opj_stream_create_default_file_stream.restype, L.opj_stream_create_default_file_stream.argtypes = C.c_void_p, [C.c_char_p, C.c_int]
    L.opj_start_compress.restype, L.opj_start_compress.argtypes = C.c_int, [C.c_void_p, C.POINTER(Img), C.c_void_p]
    for f in ("opj_encode", "opj_end_compress"):
        getattr(L, f).restype, getattr(L, f).argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    L.opj_stream_destroy.argtypes = [C.c_void_p]
    L.opj_destroy_codec.argtypes = [C.c_void_p]
    L.opj_image_destroy.argtypes = [C.POINTER(Img)]
    L.opj_set_default_encoder_parameters.argtypes = [C.c_void_p]
    return L


def encode(planes, subs):
    """planes: 2-D uint8 arrays at their own (subsampled) sizes; subs: [(dx, dy)] -> the raw codestream (lossless, OpenJPEG's defaults)"""
    L = _lib()
    H, W = planes[0].shape
    cm = (Cmpt * len(planes))()
    for i, (p, (dx, dy)) in enumerate(zip(planes, subs)):
        cm[i].dx, cm[i].dy, cm[i].w, cm[i].h, cm[i].prec, cm[i].bpp = dx, dy, p.shape[1], p.shape[0], 8, 8
    img = L.opj_image_create(len(planes), cm, 3)  # OPJ_CLRSPC_SYCC (a raw codestream does not record it)
    im = img.contents
    im.x0, im.y0, im.x1, im.y1 = 0, 0, W, H
    for i, p in enumerate(planes):
        c = im.comps[i]
        assert (c.w, c.h, c.dx, c.dy) == (p.shape[1], p.shape[0], subs[i][0], subs[i][1])
        flat = np.ascontiguousarray(p, np.int32).ravel()
        C.memmove(c.data, flat.ctypes.data, flat.nbytes)
    params = C.create_string_buffer(1 << 16)
    L.opj_set_default_encoder_parameters(params)
    codec = L.opj_create_compress(0)  # OPJ_CODEC_J2K
    assert L.opj_setup_encoder(codec, params, img)
    fd, path = tempfile.mkstemp(suffix=".j2k")
    os.close(fd)
    st = L.opj_stream_create_default_file_stream(path.encode(), 0)
    ok = L.opj_start_compress(codec, img, st) and L.opj_encode(codec, st) and L.opj_end_compress(codec, st)
    L.opj_stream_destroy(st)
    L.opj_destroy_codec(codec)
    L.opj_image_destroy(img)
    assert ok
    raw = open(path, "rb").read()
    os.remove(path)
    return raw


def main():
    rng = np.random.RandomState(11)
    out = {}
    for name, (H, W, dx, dy) in {"s422": (128, 128, 2, 1), "s420": (64, 64, 2, 2), "s422_odd": (40, 67, 2, 1)}.items():
        ch, cw = -(-H // dy), -(-W // dx)
        yy, xx = np.mgrid[:H, :W]
        Y = np.clip((xx * 2 + yy * 3) % 256 + rng.randint(-6, 7, (H, W)), 0, 255).astype(np.uint8)
        cy, cx = np.mgrid[:ch, :cw]
        Cb = np.clip((cx * 5 + 60) % 256 + rng.randint(-4, 5, (ch, cw)), 0, 255).astype(np.uint8)
        Cr = np.clip((cy * 4 + 90) % 256 + rng.randint(-4, 5, (ch, cw)), 0, 255).astype(np.uint8)
        raw = encode([Y, Cb, Cr], [(1, 1), (dx, dy), (dx, dy)])
        out[name + "/stream"] = np.frombuffer(raw, np.uint8)
        out[name + "/Y"], out[name + "/Cb"], out[name + "/Cr"] = Y, Cb, Cr
        out[name + "/sub"] = np.array([dx, dy])
    dst = os.path.join(ROOT, "tests", "golden", "jp2k_subsampled.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
