"""Slide readers (cerberus_amd/reader.py: the WSIReader-shaped interface of infer/wsi.py:521-531 over this package's own back ends).
The tiled-TIFF reader and the small writer are checked against an independent implementation (PIL's libtiff) both ways."""
import os

import numpy as np
import pytest

from cerberus_amd.reader import ArrayReader, TiffReader, WSIReader, write_tiled_tiff


def _pyramid(h=700, w=900, seed=3):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // w), (yy * 255 // h), rs.randint(0, 256, (h, w))], -1).astype(np.uint8)

    def half(a):
        hh, ww = a.shape[0] // 2, a.shape[1] // 2
        return np.clip(np.rint(a[: hh * 2, : ww * 2].astype(np.float32).reshape(hh, 2, ww, 2, 3).mean(axis=(1, 3))), 0, 255).astype(np.uint8)

    l1 = half(base)
    return [base, l1, half(l1)]


def test_tiled_pyramid_tiff_roundtrip_and_resolutions(tmp_path):
    levels = _pyramid()
    path = str(tmp_path / "slide.tif")
    write_tiled_tiff(path, levels, tile=256, mpp=0.25)
    r = WSIReader.open(input_img=path)
    assert isinstance(r, TiffReader) and r.info.level_count == 3
    assert np.allclose(r.info.mpp, [0.25, 0.25], rtol=1e-3) and r.info.slide_dimensions == (900, 700)
    assert r.info.level_dimensions == [(900, 700), (450, 350), (225, 175)]
    # XY order, rounded (reference: wsi_proc_shape = slide_dimensions(**resolution)[::-1])
    assert r.slide_dimensions(0.25, "mpp").tolist() == [900, 700] and r.slide_dimensions(0.5, "mpp").tolist() == [450, 350]
    assert r.slide_dimensions(r.info.mpp, "mpp").tolist() == [900, 700]
    assert np.array_equal(r.read_bounds((0, 0, 900, 700), 0.25, "mpp"), levels[0])
    assert np.array_equal(r.read_bounds((100, 37, 415, 300), 0.25, "mpp"), levels[0][37:300, 100:415])  # crosses tile borders
    # a resolution that is a pyramid level is read from that level, pixel for pixel
    assert np.array_equal(r.read_bounds((5, 9, 440, 333), 0.5, "mpp"), levels[1][9:333, 5:440])
    assert np.array_equal(r.read_bounds((0, 0, 225, 175), 1.0, "mpp"), levels[2])
    rows = r.rows(0.5, "mpp")
    assert rows.shape == (350, 450, 3) and np.array_equal(rows[40:90], levels[1][40:90])
    # independent decoder: PIL (libtiff) reads the file this package wrote
    from PIL import Image

    im = Image.open(path)
    assert np.array_equal(np.array(im), levels[0])
    im.seek(1)
    assert np.array_equal(np.array(im), levels[1])


@pytest.mark.parametrize("compression", [None, "tiff_adobe_deflate", "jpeg", "tiff_lzw", "packbits"])
def test_reads_tiffs_written_by_pil(tmp_path, compression):
    """independent encoder: files written by PIL's libtiff (strips; raw, deflate, JPEG, LZW, PackBits) through this package's reader"""
    from PIL import Image

    base = _pyramid(300, 420, 5)[0]
    base[..., 2] = base[..., 0] // 2  # smooth planes so that JPEG stays close
    path = str(tmp_path / "p.tif")
    kw = {} if compression is None else {"compression": compression}
    Image.fromarray(base).save(path, dpi=(101600.0, 101600.0), **kw)  # 101600 dpi = 0.25 um / px
    r = WSIReader.open(path)
    got = r.read_bounds((0, 0, 420, 300), 1.0, "baseline")
    if compression == "jpeg":
        assert np.abs(got.astype(int) - base.astype(int)).mean() < 6.0
        assert np.array_equal(got, np.array(Image.open(path).convert("RGB")))
    else:
        assert np.array_equal(got, base)
    assert np.allclose(r.info.mpp, [0.25, 0.25], rtol=1e-3)
    # x2 reduction from a file without a pyramid: exact box means
    half = r.read_bounds((0, 0, 210, 150), 0.5, "mpp")
    if compression != "jpeg":
        want = np.clip(np.rint(base.astype(np.float32).reshape(150, 2, 210, 2, 3).mean(axis=(1, 3))), 0, 255).astype(np.uint8)
        assert np.array_equal(half, want)


def test_array_and_synthetic_specs(tmp_path):
    arr = _pyramid(120, 90, 7)[0]
    np.save(str(tmp_path / "a.npy"), arr)
    r = WSIReader.open(str(tmp_path / "a.npy"))
    assert isinstance(r, ArrayReader) and r.info.mpp is None
    assert r.slide_dimensions(0.5, "mpp").tolist() == [90, 120]  # no scan resolution recorded: pixels are the processing resolution
    rows = r.rows(0.5, "mpp")
    assert isinstance(rows, np.memmap) and np.array_equal(rows[10:20], arr[10:20])
    r2 = WSIReader.open(arr, mpp=(0.25, 0.25))
    assert r2.slide_dimensions(0.5, "mpp").tolist() == [45, 60] and r2.read_bounds((0, 0, 45, 60), 0.5, "mpp").shape == (60, 45, 3)
    (tmp_path / "s.txt").write_text("synthetic:1500x1100:9")
    r3 = WSIReader.open(str(tmp_path / "s.txt"))
    assert r3.info.slide_dimensions == (1100, 1500) and r3.seed == 9
    with pytest.raises(ValueError):
        r.slide_dimensions(20, "power")


def test_jpeg_tiles_with_rgb_photometric_and_plain_component_ids(tmp_path):
    """Aperio-style JPEG tiles: PhotometricInterpretation = RGB, component ids 1, 2, 3, no JFIF / Adobe marker -- the components are R, G, B
    and must come out unconverted (libjpeg alone would guess YCbCr and convert; libtiff-written files hide this because libtiff names
    its components 'R', 'G', 'B')."""
    import io

    from PIL import Image

    yy, xx = np.mgrid[0:200, 0:330]
    base = np.stack([xx * 255 // 330, yy * 255 // 200, (xx + yy) * 255 // 530], -1).astype(np.uint8)

    def enc(t):  # encode the R, G, B planes as they are: PIL writes a 'YCbCr' image's components untouched (ids 1, 2, 3); drop its JFIF APP0
        buf = io.BytesIO()
        Image.merge("YCbCr", [Image.fromarray(t[..., i]) for i in range(3)]).save(buf, format="JPEG", quality=95, subsampling=0)
        b = buf.getvalue()
        assert b[2:4] == b"\xff\xe0"
        n = (b[4] << 8) | b[5]
        b = b[:2] + b[4 + n:]
        assert b"JFIF" not in b[:64] and b"Adobe" not in b[:64]
        return b

    path = str(tmp_path / "aperio_like.tif")
    write_tiled_tiff(path, [base], tile=128, mpp=0.25, encode=(enc, 7))
    r = WSIReader.open(path)
    assert r.levels[0].photometric == 2 and r.levels[0].compression == 7
    got = r.read_bounds((0, 0, 330, 200), 1.0, "baseline")
    assert np.abs(got.astype(int) - base.astype(int)).mean() < 2.0, np.abs(got.astype(int) - base.astype(int)).mean()


def test_non_integer_reduction_does_not_depend_on_the_row_chunks(tmp_path):
    """mpp 0.252 -> 0.5 (a factor of 1.984 from level 0: Aperio's scan resolution against the reference's default processing resolution):
    the rows a SlabUploader reads chunk by chunk, or two ranks read as bands, are the rows of ONE read of the whole slide -- the resampling
    grid is global -- and they are area means of the level's pixels."""
    levels = _pyramid(523, 389, 11)[:1]
    path = str(tmp_path / "s.tif")
    write_tiled_tiff(path, levels, tile=128, mpp=0.252)
    r = WSIReader.open(path)
    w, h = [int(v) for v in r.slide_dimensions(0.5, "mpp")]
    whole = r.read_bounds((0, 0, w, h), 0.5, "mpp")
    assert whole.shape == (h, w, 3)
    rows = r.rows(0.5, "mpp")
    for a, b in ((0, 37), (37, 100), (100, 101), (101, h)):
        assert np.array_equal(rows[a:b], whole[a:b]), (a, b)
    assert np.array_equal(r.read_bounds((13, 50, 140, 77), 0.5, "mpp"), whole[50:77, 13:140])
    # a constant image stays constant, a linear ramp keeps its slope: area means on the right grid
    rel = 0.5 / 0.252
    ramp = levels[0][..., 0].astype(np.float64)  # x * 255 // w
    exp = np.array([ramp[:, int(np.floor(o * rel)):int(np.ceil((o + 1) * rel))].mean() for o in range(5, 60)])
    assert np.abs(whole[:, 5:60, 0].mean(axis=0) - exp).max() < 1.5
    # enlarging (0.252 -> 0.2) goes through the same global grid
    w2, h2 = [int(v) for v in r.slide_dimensions(0.2, "mpp")]
    big = r.read_bounds((0, 0, w2, h2), 0.2, "mpp")
    assert np.array_equal(r.read_bounds((0, 211, w2, 305), 0.2, "mpp"), big[211:305])


def test_rgb_photometric_jpeg_tile_with_a_jfif_header_and_resampling_past_the_edge():
    """(i) a JPEG tile of a PhotometricInterpretation = RGB page whose stream ALSO carries a JFIF APP0 segment -- libjpeg lets JFIF win over
    the injected Adobe transform-0 marker and converts as YCbCr -- has its APP0 dropped before the Adobe marker goes in; (ii) a non-integer
    down-sampling factor whose last output pixels lie wholly past the level's end replicates the edge pixel (it returned black)."""
    import io

    from PIL import Image

    from cerberus_amd.reader import _resample_axis, _strip_jfif_app0

    rgb = np.random.RandomState(3).randint(0, 256, (48, 64, 3)).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=95, subsampling=0)
    data = buf.getvalue()
    assert b"JFIF" in data[:32]
    plain = np.array(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
    stripped = _strip_jfif_app0(data)
    assert b"JFIF" not in stripped[:32] and stripped[:2] == data[:2]
    assert np.array_equal(np.array(Image.open(io.BytesIO(stripped)).convert("RGB")).astype(int), plain)  # nothing but the APP0 segment changed
    adobe = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    with_jfif = np.array(Image.open(io.BytesIO(data[:2] + adobe + data[2:])).convert("RGB")).astype(int)
    without = np.array(Image.open(io.BytesIO(stripped[:2] + adobe + stripped[2:])).convert("RGB")).astype(int)
    assert np.abs(without - plain).max() > 20   # transform 0 honoured: the components are taken as R, G, B (no YCbCr conversion)
    assert np.abs(with_jfif - plain).max() <= 1  # ... which the JFIF segment would have prevented
    src = np.arange(10, dtype=np.float32).reshape(10, 1)
    out = _resample_axis(src, 0, 0, 5, 2.5, 0, 10).ravel()
    assert abs(out[3] - 8.2) < 1e-5 and out[4] == 9.0


def test_large_reads_on_decode_processes_equal_the_thread_pool(tmp_path, monkeypatch):
    """Reads of many JPEG tiles go to worker PROCESSES (cerberus_amd/decode_worker.py; threads stop scaling at two because PIL parses and hands over
    pixels under the interpreter lock): fresh interpreters behind pipes, tiles decoded into a shared-memory window.  Same bytes as the thread pool,
    for a window that cuts tiles on all four sides, into a caller's buffer too; a worker's failure switches the processes off and the read is repeated on the threads."""
    import io

    from PIL import Image

    from cerberus_amd import reader as rd

    rs = np.random.RandomState(5)
    small = rs.randint(0, 256, (17, 23, 3)).astype(np.uint8)
    base = np.kron(small, np.ones((64, 64, 1), np.uint8))[:1000, :1400]  # smooth blocks: JPEG keeps them recognisable

    def enc(t):  # Aperio-style: the R, G, B planes are the stream's components (the writer tags the page PhotometricInterpretation = RGB)
        buf = io.BytesIO()
        Image.merge("YCbCr", [Image.fromarray(np.ascontiguousarray(t[..., i])) for i in range(3)]).save(buf, format="JPEG", quality=90, subsampling=0)
        return buf.getvalue()

    path = str(tmp_path / "many_tiles.tif")
    write_tiled_tiff(path, [base], tile=64, mpp=0.5, encode=(enc, 7))  # 16 x 22 = 352 tiles
    monkeypatch.setenv("CERB_DECODE_PROCS", "0")
    r = WSIReader.open(path)
    ref = r._read_level(0, 0, 0, 1400, 1000)
    cut = r._read_level(0, 37, 21, 1333, 977)
    assert np.abs(ref.astype(int) - base.astype(int)).mean() < 4.0
    monkeypatch.setenv("CERB_DECODE_PROCS", "3")
    try:
        r2 = WSIReader.open(path)
        assert rd._proc_pool() is not None and rd._proc_pool().n == 3
        assert np.array_equal(r2._read_level(0, 0, 0, 1400, 1000), ref)
        assert np.array_equal(r2._read_level(0, 37, 21, 1333, 977), cut)
        buf = np.full((1100, 1500, 3), 7, np.uint8)
        got = r2._read_level(0, 0, 0, 1400, 1000, out=buf)
        assert got.base is buf or got is buf or np.shares_memory(got, buf)
        assert np.array_equal(got, ref) and (buf[1000:] == 7).all() and (buf[:, 1400:] == 7).all()
        assert np.array_equal(r2.rows(0.5, "mpp")[100:900], ref[100:900])
        # a small read stays on the threads
        assert np.array_equal(r2._read_level(0, 0, 0, 300, 200), ref[:200, :300])
        # a failing worker: it is told to open a file that is not there
        r3 = WSIReader.open(path)
        r3.path = str(tmp_path / "nowhere.tif")
        # ... the pool is switched off for the rest of the run and the read is done on the threads, which read r3's own open file
        assert np.array_equal(r3._read_level(0, 0, 0, 1400, 1000), ref)
        assert rd._PROCS.get("off") and rd._PROCS["pool"] is None
        assert np.array_equal(r2._read_level(0, 0, 0, 1400, 1000), ref)  # (threads from now on)
    finally:
        rd._shutdown_procs()
        rd._PROCS.pop("off", None)


# ---- libcerberus_host.so: TIFF LZW / PackBits / predictor in C (csrc/host_codecs.c) against the plain-Python restatement and against libtiff ------------

def _codec_cases():
    rng = np.random.RandomState(1)
    return {"noise": rng.randint(0, 256, 40000).astype(np.uint8).tobytes(),   # fills the 4094-entry table many times over
            "smooth": (np.arange(50000) // 7 % 256).astype(np.uint8).tobytes(),
            "flat": bytes(30000), "empty": b"", "one": b"\x07", "two": b"\x07\x07",
            "low": rng.randint(0, 4, 60000).astype(np.uint8).tobytes(),
            "kwkwk": b"ab" * 5 + b"a" * 2400}


def test_c_codecs_equal_the_python_restatement_on_encoded_streams():
    from cerberus_amd import _hostlib as H
    from oracle import tiff_codecs_ref as R

    assert H.lib().cerb_host_version() >= 1
    for name, d in _codec_cases().items():
        e = R.lzw_encode(d)
        assert R.lzw_decode(e, len(d)) == d and H.lzw_decode(e, len(d)).tobytes() == d, name
        pe = R.packbits_encode(d)
        assert R.packbits_decode(pe, len(d)) == d and H.packbits_decode(pe, len(d)).tobytes() == d, name
        if len(d) > 10:  # a destination smaller than the stream: the prefix, never a byte more
            assert H.lzw_decode(e, len(d) - 7).tobytes() == d[:-7] and H.packbits_decode(pe, len(d) - 7).tobytes() == d[:-7], name
            # a stream cut short: what there is, zero-filled (the Python decoders return the short prefix)
            cut = e[: len(e) // 2]
            ref = R.lzw_decode(cut, len(d))
            got = H.lzw_decode(cut, len(d)).tobytes()
            assert got[: len(ref)] == ref and not any(got[len(ref):]), name
    # corrupt: a code beyond the table right after the first literal; old-style (LSB-first) streams are named, not decoded into noise
    with pytest.raises(ValueError, match="corrupt LZW"):
        H.lzw_decode(bytes([0x80, 0x00, 0xFF, 0xFF, 0xFF]), 64)
    with pytest.raises(ValueError, match="old-style"):
        H.lzw_decode(bytes([0x00, 0x01, 0x02, 0x03]), 64)
    with pytest.raises(ValueError, match="corrupt PackBits"):
        H.packbits_decode(bytes([5, 1, 2]), 64)
    # predictor 2: the running sum per sample, modulo 256, in place
    a = np.random.RandomState(2).randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(H.unpredict_u8(a.copy()), np.cumsum(a, axis=1, dtype=np.uint8))
    a4 = np.random.RandomState(3).randint(0, 256, (5, 9, 4)).astype(np.uint8)
    assert np.array_equal(H.unpredict_u8(a4.copy()), np.cumsum(a4, axis=1, dtype=np.uint8))


def test_c_lzw_decoder_on_libtiff_streams(tmp_path):
    """independent ENCODER: the strips of files PIL's libtiff wrote (with and without the horizontal predictor), pulled out with this package's
    TIFF parser, through the C decoder and the Python restatement -- and the whole file through the reader against PIL's own decode."""
    from PIL import Image

    from cerberus_amd import _hostlib as H
    from oracle import tiff_codecs_ref as R

    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[:120, :200]
    imgs = {"noise": rng.randint(0, 256, (120, 200, 3)).astype(np.uint8),
            "smooth": np.stack([(xx // 3) % 256, (yy // 2) % 256, ((xx + yy) // 5) % 256], -1).astype(np.uint8),
            "flat": np.full((120, 200, 3), 200, np.uint8)}
    for name, a in imgs.items():
        for pred in (False, True):
            path = str(tmp_path / ("%s%d.tif" % (name, pred)))
            kw = {"tiffinfo": {317: 2}} if pred else {}
            Image.fromarray(a).save(path, compression="tiff_lzw", **kw)
            r = WSIReader.open(path)
            p = r.levels[0]
            assert p.compression == 5 and p.predictor == (2 if pred else 1), (name, pred, p.predictor)
            for i in range(len(p.offsets)):
                s = os.pread(r.fh.fileno(), p.counts[i], p.offsets[i])
                exp = min(p.th, p.h - i * p.th) * p.w * p.samples
                assert H.lzw_decode(s, exp).tobytes() == R.lzw_decode(s, exp), (name, pred, i)
            assert np.array_equal(r.read_bounds((0, 0, 200, 120), 1.0, "baseline"), a), (name, pred)
            assert np.array_equal(np.array(Image.open(path)), a)


@pytest.mark.parametrize("predictor", [1, 2])
def test_tiled_lzw_pyramid_through_the_reader_and_through_libtiff(tmp_path, predictor):
    """A TILED pyramidal LZW TIFF (what bioformats / libvips exports look like; PIL cannot write tiles): written by write_tiled_tiff with the
    restatement's encoder, read by the product on its decode threads (C decoder, lock released) and by PIL's libtiff -- an independent DECODER for
    the encoder, an independent container reader for the file.  Every level, windows across tile borders."""
    from PIL import Image

    from oracle import tiff_codecs_ref as R

    levels = _pyramid(520, 610, 11)
    path = str(tmp_path / "lzw.tif")
    write_tiled_tiff(path, levels, tile=128, mpp=0.5, encode=(lambda t: R.lzw_encode(t.tobytes()), 5), predictor=predictor)
    r = WSIReader.open(path)
    assert r.levels[0].compression == 5 and r.levels[0].predictor == predictor and r.info.level_count == 3
    assert np.array_equal(r.read_bounds((0, 0, 610, 520), 0.5, "mpp"), levels[0])
    assert np.array_equal(r.read_bounds((100, 37, 415, 300), 0.5, "mpp"), levels[0][37:300, 100:415])
    assert np.array_equal(r.read_bounds((3, 5, 300, 255), 1.0, "mpp"), levels[1][5:255, 3:300])
    im = Image.open(path)
    assert np.array_equal(np.array(im), levels[0])
    im.seek(1)
    assert np.array_equal(np.array(im), levels[1])
    # PackBits tiles the same way (without the predictor: libtiff implements tag 317 inside its LZW / deflate codecs only)
    path2 = str(tmp_path / "pb.tif")
    write_tiled_tiff(path2, levels[:1], tile=128, encode=(lambda t: R.packbits_encode(t.tobytes()), 32773))
    assert np.array_equal(WSIReader.open(path2).read_bounds((0, 0, 610, 520), 1.0, "baseline"), levels[0])
    assert np.array_equal(np.array(Image.open(path2)), levels[0])


@pytest.mark.parametrize("codec", ["raw", "deflate", "lzw", "packbits"])
def test_whole_window_native_read_equals_the_per_tile_path(tmp_path, codec):
    """cerb_host_tiff_read_tiles (one native call per window: pread + decode + predictor + placement on pthreads) against the reader's per-tile
    Python path (TiffReader._place_tile, what the JPEG tiles and the worker processes use) on ragged windows, a destination that is a view of a wider
    buffer, 1 / 3 / 8 threads, and a level whose last tiles are padding."""
    from oracle import tiff_codecs_ref as R

    levels = _pyramid(333, 471, 5)[:2]
    path = str(tmp_path / "w.tif")
    kw = {"raw": dict(compress=False), "deflate": dict(compress=True), "lzw": dict(encode=(lambda t: R.lzw_encode(t.tobytes()), 5), predictor=2),
          "packbits": dict(encode=(lambda t: R.packbits_encode(t.tobytes()), 32773))}[codec]
    write_tiled_tiff(path, levels, tile=64, mpp=0.5, **kw)
    r = WSIReader.open(path)
    for lvl, img in enumerate(levels):
        p = r.levels[lvl]
        h, w = img.shape[:2]
        for (x0, y0, x1, y1) in ((0, 0, w, h), (1, 2, 3, 5), (63, 63, 65, 66), (100, 37, w - 3, h - 1), (w - 10, h - 7, w, h)):
            want = np.zeros((y1 - y0, x1 - x0, 3), np.uint8)
            for ty in range(y0 // p.th, -(-y1 // p.th)):
                for tx in range(x0 // p.tw, -(-x1 // p.tw)):
                    r._place_tile(p, (ty, tx), (x0, y0, x1, y1), want)
            assert np.array_equal(want, img[y0:y1, x0:x1])
            for th in ("1", "3", "8"):
                os.environ["CERB_DECODE_THREADS"] = th
                try:
                    assert np.array_equal(r._read_level(lvl, x0, y0, x1, y1), want), (codec, lvl, th)
                    wide = np.full((y1 - y0 + 5, x1 - x0 + 9, 3), 7, np.uint8)  # a pinned staging buffer larger than the window
                    got = r._read_level(lvl, x0, y0, x1, y1, out=wide)
                    assert np.array_equal(got, want) and np.shares_memory(got, wide) and bool((wide[y1 - y0:] == 7).all()) and bool((wide[:, x1 - x0:] == 7).all())
                finally:
                    os.environ.pop("CERB_DECODE_THREADS", None)


def test_native_read_names_the_tile_it_could_not_decode(tmp_path):
    """A damaged tile ends the read with a ValueError that names the file, the tile and the reason -- for a corrupt deflate stream, a corrupt LZW
    stream and a file that ends inside a tile -- never with pixels."""
    from oracle import tiff_codecs_ref as R

    img = _pyramid(200, 260, 9)[0]
    for codec, kw, what in (("deflate", dict(compress=True), "deflate"), ("lzw", dict(encode=(lambda t: R.lzw_encode(t.tobytes()), 5)), "LZW")):
        path = str(tmp_path / (codec + ".tif"))
        write_tiled_tiff(path, [img], tile=64, **kw)
        r = WSIReader.open(path)
        p = r.levels[0]
        k = 6  # damage tile 6 = (ty 1, tx 1) of the 4 x 5 grid: bytes in the middle of its stream set to 0xFF
        raw = bytearray(open(path, "rb").read())
        a = p.offsets[k] + 2
        raw[a:a + 24] = b"\xff" * 24
        bad = str(tmp_path / (codec + "_bad.tif"))
        open(bad, "wb").write(bytes(raw))
        rb = WSIReader.open(bad)
        assert np.array_equal(rb.read_bounds((0, 0, 64, 64), 1.0, "baseline"), img[:64, :64])  # the other tiles still read
        with pytest.raises(ValueError) as ei:
            rb.read_bounds((0, 0, 260, 200), 1.0, "baseline")
        assert "tile %d" % k in str(ei.value) and what in str(ei.value) and "_bad.tif" in str(ei.value), str(ei.value)
    # a file cut inside its last tile (tiles are written before the directory: cut the bytes, keep the directory's offsets by padding zeros elsewhere)
    path = str(tmp_path / "raw.tif")
    write_tiled_tiff(path, [img], tile=64, compress=False)
    r = WSIReader.open(path)
    p = r.levels[0]
    p.counts[3] = p.counts[3] + (1 << 30)  # a count that runs past the end of the file
    with pytest.raises(ValueError, match="ends inside"):
        r.read_bounds((0, 0, 260, 64), 1.0, "baseline")


def test_jpeg2000_tiles_of_aperio_and_generic_tiffs(tmp_path):
    """TIFF compressions 33005 (Aperio, RGB components), 34712 (generic) and 33003 (Aperio, YCbCr components) through OpenJPEG behind PIL.  The only
    pin this image offers is OpenJPEG's own encoder: lossless codestreams written tile by tile come back bit for bit (33003: as PIL's JFIF
    YCbCr -> RGB of the stored planes); a codestream whose chroma is subsampled and a damaged one are refused by name, never decoded into something."""
    import io

    from PIL import Image, features

    if not features.check_codec("jpg_2000"):
        pytest.skip("PIL without OpenJPEG")
    levels = _pyramid(300, 410, 4)[:2]

    def enc(t, **kw):
        b = io.BytesIO()
        Image.fromarray(t).save(b, format="JPEG2000", no_jp2=True, irreversible=False, **kw)
        return b.getvalue()

    for code in (33005, 34712):
        path = str(tmp_path / ("c%d.tif" % code))
        write_tiled_tiff(path, levels, tile=128, mpp=0.5, encode=(enc, code))
        r = WSIReader.open(path)
        assert r.levels[0].compression == code
        assert np.array_equal(r.read_bounds((0, 0, 410, 300), 0.5, "mpp"), levels[0])
        assert np.array_equal(r.read_bounds((33, 60, 400, 290), 0.5, "mpp"), levels[0][60:290, 33:400])
        assert np.array_equal(r.read_bounds((0, 0, 205, 150), 1.0, "mpp"), levels[1])
    ycc = np.asarray(Image.fromarray(levels[0]).convert("YCbCr"))
    path = str(tmp_path / "c33003.tif")
    write_tiled_tiff(path, [ycc], tile=128, mpp=0.5, encode=(lambda t: enc(t, mct=0), 33003))
    got = WSIReader.open(path).read_bounds((0, 0, 410, 300), 0.5, "mpp")
    assert np.array_equal(got, np.asarray(Image.fromarray(ycc, "YCbCr").convert("RGB")))
    assert np.abs(got.astype(int) - levels[0].astype(int)).max() <= 4  # (the 8-bit YCbCr round trip of the source)
    # chroma subsampled INSIDE the codestream (what scanners write under 33003): fixtures out of the bundled OpenJPEG itself (oracle/gen_golden_jp2k.py) --
    # the reader returns the stored planes' nearest-replicated chroma through the JFIF matrix, for 4:2:2 and 4:2:0 tiles; an odd-sized codestream (whose round trip through
    # this image's OpenJPEG does not return the stored planes) and subsampled RGB (33005) are refused by name
    from cerberus_amd.reader import _decode_jp2k_tile

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jp2k_subsampled.npz"))

    def expected(name):
        Y, Cb, Cr = g[name + "/Y"], g[name + "/Cb"], g[name + "/Cr"]
        dx, dy = (int(v) for v in g[name + "/sub"])
        rep = [np.repeat(np.repeat(c, dy, axis=0), dx, axis=1)[: Y.shape[0], : Y.shape[1]] for c in (Cb, Cr)]
        return np.asarray(Image.merge("YCbCr", [Image.fromarray(Y), Image.fromarray(rep[0]), Image.fromarray(rep[1])]).convert("RGB"))

    for name in ("s422", "s420"):
        want = expected(name)
        side = want.shape[0]
        stream = g[name + "/stream"].tobytes()
        sub = str(tmp_path / (name + ".tif"))
        write_tiled_tiff(sub, [np.zeros((side, 2 * side, 3), np.uint8)], tile=side, mpp=0.5, encode=(lambda t: stream, 33003))  # two tiles, the same codestream
        got = WSIReader.open(sub).read_bounds((0, 0, 2 * side, side), 0.5, "mpp")
        assert np.array_equal(got[:, :side], want) and np.array_equal(got[:, side:], want), name
    with pytest.raises(NotImplementedError, match="67 x 40"):
        _decode_jp2k_tile(g["s422_odd/stream"].tobytes(), 33003, "odd", 0)
    with pytest.raises(NotImplementedError, match="component sampling"):
        _decode_jp2k_tile(g["s422/stream"].tobytes(), 33005, "rgb", 0)
    r = WSIReader.open(path)
    p = r.levels[0]
    # a damaged codestream: an error that names file and tile
    raw = bytearray(open(path, "rb").read())
    a = p.offsets[1] + 120
    raw[a:a + 400] = b"\x00" * 400
    raw[p.offsets[1] + 2:p.offsets[1] + 4] = b"\xff\x00"  # no SIZ behind SOC
    bad2 = str(tmp_path / "bad.tif")
    open(bad2, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="tile 1"):
        WSIReader.open(bad2).read_bounds((128, 0, 256, 128), 0.5, "mpp")


def test_native_reader_against_libtiff_on_random_files(tmp_path):
    """30 seeded files out of PIL's libtiff -- RGB and RGBA (the reader drops the alpha samples), raw / deflate / LZW / PackBits strips, with and without
    the horizontal predictor, sizes down to one pixel -- and a random window of each through cerb_host_tiff_read_tiles on 1 .. 5 threads: the
    pixels libtiff itself decodes."""
    from PIL import Image

    rng = np.random.RandomState(5)
    for trial in range(30):
        h, w = int(rng.randint(1, 260)), int(rng.randint(1, 330))
        mode = ["RGB", "RGBA"][rng.randint(2)]
        a = (rng.randint(0, 25, (h, w, len(mode))) + np.arange(w)[None, :, None] // 2).astype(np.uint8)
        comp = [None, "tiff_adobe_deflate", "tiff_lzw", "packbits"][rng.randint(4)]
        kw = {} if comp is None else {"compression": comp}
        if rng.randint(2) and comp in ("tiff_adobe_deflate", "tiff_lzw"):
            kw["tiffinfo"] = {317: 2}
        path = str(tmp_path / ("f%d.tif" % trial))
        Image.fromarray(a, mode).save(path, **kw)
        r = WSIReader.open(path)
        x0, x1 = sorted(int(v) for v in rng.randint(0, w + 1, 2))
        y0, y1 = sorted(int(v) for v in rng.randint(0, h + 1, 2))
        if x1 == x0:
            x1 = min(w, x0 + 1)
            x0 = x1 - 1
        if y1 == y0:
            y1 = min(h, y0 + 1)
            y0 = y1 - 1
        os.environ["CERB_DECODE_THREADS"] = str(rng.randint(1, 6))
        try:
            got = r._read_level(0, x0, y0, x1, y1)
        finally:
            os.environ.pop("CERB_DECODE_THREADS", None)
        want = np.asarray(Image.open(path))[y0:y1, x0:x1, :3]
        assert np.array_equal(got, want), (trial, mode, kw, (h, w), (x0, y0, x1, y1), r.levels[0].th)
