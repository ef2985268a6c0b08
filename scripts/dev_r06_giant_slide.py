"""Developer run (GPU box): slides with MORE THAN 2^31 pixels through run_infer_wsi.py -- the range a 0.5-mpp scan of a large section reaches
(60000 x 50000) and nothing in tests/ or bench.py touches (40000^2 = 1.6e9 < 2^31): every pixel index past int32, canvases of 100 - 300 GB, and, for the
second slide, `stream_bands.plan_slide` deciding against the REAL free HBM (no CERB_HBM_BUDGET_* override) that the band must be walked in sub-bands.

The slide file is a JPEG-tiled pyramidal TIFF whose TileOffsets point into an atlas of 64 encoded stain-field tiles (a few MB on disk for
10 Gpx; the reader does not care that offsets repeat) plus a x16 level for the tissue thumbnail.

    python scripts/dev_r06_giant_slide.py <H> <W> [out.json] [share of glass blocks] [ranks] [mask]
"""
import io
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TILE = 256


def _jpeg_rgb_components(t):
    """Aperio-style JPEG tile: R, G, B are the stream's components (4:4:4, no JFIF marker) -- what bench.py's ingest leg writes."""
    from PIL import Image

    b = io.BytesIO()
    Image.merge("YCbCr", [Image.fromarray(np.ascontiguousarray(t[..., i])) for i in range(3)]).save(b, format="JPEG", quality=80, subsampling=0)
    raw = b.getvalue()
    n = (raw[4] << 8) | raw[5]
    return raw[:2] + raw[4 + n:] if raw[2:4] == b"\xff\xe0" else raw


def write_atlas_tiff(path, H, W, mpp=0.5, seed=17, glass=0.0):
    """Level 0: tile (ty, tx) = atlas[pick[ty, tx]] (the image's last tile row / column may be partial: TIFF tiles are whole, the image size crops them);
    level 1 (x16): assembled from the atlas tiles' x16 reductions.  glass: share of 16 x 16-tile blocks that are white (a slide is 30 - 70 % glass:
    the tissue mask then drops patches, which an all-tissue slide never exercises)."""
    from cerberus_amd.synth_tiles import stain_field

    rs = np.random.RandomState(seed)
    atlas = [np.clip(stain_field(TILE, 100 + i).astype(np.int16) + rs.randint(-10, 11, (TILE, TILE, 3)), 0, 255).astype(np.uint8) for i in range(64)]
    atlas.append(np.full((TILE, TILE, 3), 255, np.uint8))
    ny, nx = -(-H // TILE), -(-W // TILE)
    pick = rs.randint(0, 64, (ny, nx))
    if glass > 0:
        blocks = rs.rand(-(-ny // 16), -(-nx // 16)) < glass
        pick[np.kron(blocks, np.ones((16, 16), bool))[:ny, :nx]] = 64
    small = np.stack([a.reshape(16, 16, 16, 16, 3).mean(axis=(1, 3)).astype(np.uint8) for a in atlas])  # [65, 16, 16, 3]
    h1, w1 = -(-H // 16), -(-W // 16)
    l1 = np.full((-(-h1 // TILE) * TILE, -(-w1 // TILE) * TILE, 3), 255, np.uint8)
    l1[: ny * 16, : nx * 16] = small[pick].transpose(0, 2, 1, 3, 4).reshape(ny * 16, nx * 16, 3)
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, 0))
        link = 4
        for li in range(2):
            offs, cnts = [], []
            if li == 0:
                where = []
                for a in atlas:
                    data = _jpeg_rgb_components(a)
                    where.append((fh.tell(), len(data)))
                    fh.write(data + (b"\0" if len(data) & 1 else b""))
                for i in pick.reshape(-1):
                    offs.append(where[i][0])
                    cnts.append(where[i][1])
                h, w = H, W
            else:
                h, w = h1, w1
                for ty in range(l1.shape[0] // TILE):
                    for tx in range(l1.shape[1] // TILE):
                        data = _jpeg_rgb_components(l1[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE])
                        offs.append(fh.tell())
                        cnts.append(len(data))
                        fh.write(data + (b"\0" if len(data) & 1 else b""))
            res = (int(round(1e4 / (mpp * (16 if li else 1)) * 1000)), 1000)
            entries = [(254, 4, [li]), (256, 4, [w]), (257, 4, [h]), (258, 3, [8, 8, 8]), (259, 3, [7]), (262, 3, [2]), (277, 3, [3]), (282, 5, [res]), (283, 5, [res]),
                       (284, 3, [1]), (296, 3, [3]), (322, 4, [TILE]), (323, 4, [TILE]), (324, 4, offs), (325, 4, cnts)]
            packed = []
            for tag, typ, vals in entries:
                raw = b"".join(struct.pack("<II", a, b) for a, b in vals) if typ == 5 else struct.pack("<" + {3: "H", 4: "I"}[typ] * len(vals), *vals)
                pos = None
                if len(raw) > 4:
                    if fh.tell() & 1:
                        fh.write(b"\0")
                    pos = fh.tell()
                    fh.write(raw)
                packed.append((tag, typ, len(vals), raw, pos))
            if fh.tell() & 1:
                fh.write(b"\0")
            ifd = fh.tell()
            fh.write(struct.pack("<H", len(packed)))
            for tag, typ, cnt, raw, pos in packed:
                fh.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<I", pos) if pos is not None else raw.ljust(4, b"\0")))
            fh.write(struct.pack("<I", 0))
            end = fh.tell()
            fh.seek(link)
            fh.write(struct.pack("<I", ifd))
            fh.seek(end)
            link = ifd + 2 + 12 * len(packed)
    return os.path.getsize(path), pick


def write_model_dir(path, q=0.02):
    """settings.yml + weights.tar with ~q of a stain-field tile's pixels foreground (tests/tools/model_dir.py): the plain seeded weights call half
    the slide one nucleus, and a flood of 10^9 pixels is no test of anything."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from model_dir import stain_atlas, write_sparse_model_dir

    if os.environ.get("GIANT_Q"):  # e.g. GIANT_Q='{"Gland": 0.2, "Lumen": 0.08, "default": 0.02}': glands / lumina large enough to survive their size filters
        q = json.loads(os.environ["GIANT_Q"])
    return write_sparse_model_dir(path, np.stack(stain_atlas(4)), q)


def main():
    H, W = int(sys.argv[1]), int(sys.argv[2])
    out_json = sys.argv[3] if len(sys.argv) > 3 else None
    glass = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    ranks = int(sys.argv[5]) if len(sys.argv) > 5 else 1  # > 1: that many ranks on THIS box's one GPU (torch.distributed.run, collectives staged through gloo)
    with_mask = len(sys.argv) > 6 and sys.argv[6] == "mask"  # --msk_dir: the tissue mask = the blocks that are not glass (4 mask pixels per tile side)
    td = tempfile.mkdtemp(prefix="giant_")
    os.makedirs(os.path.join(td, "in"))
    path = os.path.join(td, "in", "giant.tif")
    t0 = time.perf_counter()
    base_mpp = float(os.environ.get("GIANT_BASE_MPP", "0.5"))  # 0.25: a 40x scan read at the 0.5 mpp the network runs on (H, W are the FILE's pixels)
    size, pick = write_atlas_tiff(path, H, W, mpp=base_mpp, glass=glass)
    build_s = time.perf_counter() - t0
    from cerberus_amd import reader as rd

    r = rd.WSIReader.open(input_img=path)
    rows = r.rows(0.5, "mpp")
    fH, fW = H, W
    H, W = int(rows.shape[0]), int(rows.shape[1])  # the processing resolution
    assert rows[H - 300:H - 290].shape == (10, W, 3)
    del rows, r
    res = {"file_pixels": [fH, fW], "base_mpp": base_mpp, "slide": [H, W], "pixels": H * W, "over_int32": H * W > 2 ** 31, "glass_share_of_blocks": glass, "file_MB": round(size / 1e6, 1), "build_s": round(build_s, 1)}
    for c in range(1, int(os.environ.get("GIANT_COPIES", "1"))):  # a directory of slides: the same file under more names
        os.symlink(path, os.path.join(td, "in", "giant%d.tif" % (c + 1)))
    res["background_bias_shifts"] = write_model_dir(os.path.join(td, "model"))
    cmd = [sys.executable, os.path.join(ROOT, "run_infer_wsi.py"), "--model=%s" % os.path.join(td, "model"), "--gpu=0", "--input_dir=%s" % os.path.join(td, "in"), "--wsi_file_ext=.tif",
           "--output_dir=%s" % os.path.join(td, "out"), "--logging_dir=%s" % os.path.join(td, "log"), "--batch_size=64", "--patch_input_shape=256",
           "--patch_output_shape=256"]
    cmd += [a for a in os.environ.get("GIANT_EXTRA_FLAGS", "").split() if a]  # e.g. --reference_tiling
    res["extra_flags"] = os.environ.get("GIANT_EXTRA_FLAGS")
    if with_mask:
        from PIL import Image

        os.makedirs(os.path.join(td, "msk"))
        m = np.kron((pick != 64).astype(np.uint8) * 255, np.ones((4, 4), np.uint8))
        Image.fromarray(np.stack([m] * 3, -1)).save(os.path.join(td, "msk", "giant.png"))
        cmd.append("--msk_dir=%s" % os.path.join(td, "msk"))
        res["mask"] = {"shape": list(m.shape), "tissue_share": round(float((m > 0).mean()), 4)}
    env = dict(os.environ)
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", "29611"] + cmd[1:]
        env.update(MASTER_ADDR="127.0.0.1", CERB_DIST_BACKEND="gloo")
    res["ranks"] = ranks
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env)
    wall = time.perf_counter() - t0
    res["rc"] = p.returncode
    res["wall_s"] = round(wall, 1)
    res["whole_command_Mpx_s"] = round(H * W / wall / 1e6, 1)
    logs = []
    ld = os.path.join(td, "log")
    if os.path.isdir(ld):
        for f in sorted(os.listdir(ld)):
            logs += open(os.path.join(ld, f)).read().splitlines()
    res["log"] = logs[-60:]
    res["memory_plans"] = [l.split("Memory plan: ")[1] for l in logs if "Memory plan: " in l]
    res["overall_times"] = [float(l.split("Overall Time: ")[1]) for l in logs if "Overall Time: " in l]
    res["stderr_tail"] = p.stderr.splitlines()[-25:]
    if p.returncode != 0 and out_json:
        open(out_json + ".stderr.txt", "w").write(p.stderr[-200000:])
    res["CERB_HBM_BUDGET_GB"] = os.environ.get("CERB_HBM_BUDGET_GB")
    res["stdout_tail"] = p.stdout.splitlines()[-25:]
    dat = os.path.join(td, "out", "dat", "giant.dat")
    if os.path.exists(dat):
        res["dat_MB"] = round(os.path.getsize(dat) / 1e6, 1)
        try:
            import joblib

            d = joblib.load(dat)
            res["proc_dimensions"] = [int(v) for v in d["proc_dimensions"]]
            cnt = {}
            far = {}
            for t in ("Nuclei", "Gland", "Lumen"):
                if t in d:
                    cnt[t] = len(d[t])
                    # instances whose centroid lies past pixel index 2^31 in raster order: they exist only if nothing wrapped
                    far[t] = int(sum(1 for v in d[t].values() if int(v["centroid"][1]) * W + int(v["centroid"][0]) > 2 ** 31))
            res["entries"] = cnt
            res["entries_past_int32_raster_index"] = far
        except Exception as e:  # noqa: BLE001
            res["dat_error"] = repr(e)
    s = json.dumps(res, indent=1)
    print(s)
    if out_json:
        os.makedirs(os.path.dirname(out_json) or ".", exist_ok=True)
        open(out_json, "w").write(s + "\n")
    subprocess.run(["rm", "-rf", td])


if __name__ == "__main__":
    main()
