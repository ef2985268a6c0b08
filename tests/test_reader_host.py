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
