"""Host side of the instance dictionary (loader/postproc.py:12-98, infer/wsi.py:805-853) WITHOUT torch: numpy only.

Lives apart from postproc.py / wsi.py so that the writer process of a slide's `.dat` file (`python -m cerberus_amd.inst_info <parts.npz> <out.dat>`,
started by cerberus_amd.wsi.DatWriter.from_arrays) imports in a fraction of a second and shares nothing with the process that drives the GPU:
it receives the per-instance TABLES (cerb_inst_table rows, contour point lists) as arrays, builds the ~1e6 per-instance dictionaries, draws
the uuid keys and writes the pickle -- all of which used to sit between two slides of the parent."""
import os
import sys
from collections import OrderedDict

import numpy as np


def info_from_table(tab, cnts, pts, offs, has_type, ds_factor=1.0, flat_box=False):
    """Host side of get_inst_info_dict: the per-instance dictionaries from cerb_inst_table's rows and the compact contour list.
    Everything numeric is computed for all instances at once; the per-instance values are row views of those arrays (a slide has
    ~1e6 nuclei: array construction per instance would dominate the whole slide).  Rules kept from loader/postproc.py:34-35,55-75,
    78-96 (flat_box: the slide dictionary's [x1, y1, x2, y2] form of infer/wsi.py:817-825 instead of [[y1, x1], [y2, x2]]):
    drop contours with < 3 points; majority type, skipping background when a second class exists; type_prob =
    votes / (area + 1e-6); with ds_factor != 1 box / centroid / contour are np.round(x / ds_factor).astype(int)."""
    from collections import OrderedDict

    n = tab.shape[0]
    info = OrderedDict()
    if n == 0:
        return info
    area = tab[:, 0]
    keep = np.nonzero((area > 0) & (cnts >= 3))[0]
    if keep.size == 0:
        return info
    a = np.maximum(area, 1).astype(np.float64)
    box = np.stack([tab[:, [3, 5]], tab[:, [4, 6]]], axis=1)  # [[y1, x1], [y2, x2]]
    cen = np.stack([tab[:, 1] / a, tab[:, 2] / a], axis=1)
    if ds_factor != 1.0:
        box = np.round(box / ds_factor).astype("int")
        cen = np.round(cen / ds_factor).astype("int")
        pts = np.round(pts / ds_factor).astype("int")
    if flat_box:
        box = box.reshape(n, 4)[:, [1, 0, 3, 2]]
    if has_type:
        votes = tab[:, 8:16]
        # dominant class first, ties towards the smaller class id (np.unique order + stable sort of the reference)
        order = np.argsort(-votes, axis=1, kind="stable")
        t0, t1 = order[:, 0], order[:, 1]
        second = np.take_along_axis(votes, t1[:, None], 1)[:, 0] > 0
        typ = np.where((t0 == 0) & second, t1, t0)
        prob = np.take_along_axis(votes, typ[:, None], 1)[:, 0] / (area + 1.0e-6)
        typ_l, prob_l = typ.tolist(), prob.tolist()
    starts, ends = offs.tolist(), (offs + cnts).tolist()
    for i in keep.tolist():
        d = {"box": box[i], "centroid": cen[i], "contour": pts[starts[i]:ends[i]]}
        if has_type:
            d["type"] = typ_l[i]
            d["type_prob"] = prob_l[i]
        info[i + 1] = d
    return info


def _uuid4_hex(n):
    """n random uuid4().hex strings from one os.urandom call (a slide has ~1e6 instances; uuid.uuid4() costs a syscall each)."""
    import os

    if n == 0:
        return []
    raw = np.frombuffer(os.urandom(16 * n), np.uint8).reshape(n, 16).copy()
    raw[:, 6] = (raw[:, 6] & 0x0F) | 0x40  # version 4
    raw[:, 8] = (raw[:, 8] & 0x3F) | 0x80  # RFC 4122 variant
    hx = raw.tobytes().hex()
    return [hx[i:i + 32] for i in range(0, 32 * n, 32)]


def write_dat(obj, path):
    """dat/<slide>.dat (infer/wsi.py:853 `joblib.dump(wsi_inst_info, ...)`).  Written as a plain protocol-4 pickle: `joblib.load`
    -- what consumers of the reference's files call -- reads it back to the same objects, and for a dictionary of ~1e6 instances
    with three small arrays each it is an order of magnitude faster to write (and twice as fast to load) than joblib's per-array
    wrapper stream."""
    import pickle

    with open(path, "wb") as fh:
        pickle.dump(obj, fh, protocol=4)



def build_from_parts(parts, meta):
    """parts: [(tissue, tab, cnts, pts, offs, has_type, ds_factor)]; meta: the resolution entries.  -> the dictionary of infer/wsi.py:805-853."""
    out = OrderedDict()
    for tissue, tab, cnts, pts, offs, has_type, ds_factor in parts:
        info = info_from_table(tab, cnts, pts, offs, bool(has_type), float(ds_factor), flat_box=True)
        out[tissue] = OrderedDict(zip(_uuid4_hex(len(info)), info.values()))
    out.update(meta)
    return out


def save_parts(path, parts, meta):
    """One uncompressed .npz holding every array the writer process needs."""
    arrs = {"tissues": np.array([p[0] for p in parts]), "has_type": np.array([int(bool(p[5])) for p in parts]), "ds_factor": np.array([float(p[6]) for p in parts])}
    for i, p in enumerate(parts):
        arrs["tab%d" % i], arrs["cnts%d" % i], arrs["pts%d" % i], arrs["offs%d" % i] = p[1], p[2], p[3], p[4]
    for k, v in meta.items():
        arrs["meta/" + k] = np.asarray(v["resolution"] if isinstance(v, dict) else v)
        if isinstance(v, dict):
            arrs["metaunits/" + k] = np.array(v["units"])
    with open(path, "wb") as fh:
        np.savez(fh, **arrs)


def load_parts(path):
    z = np.load(path, allow_pickle=False)
    parts = [(str(t), z["tab%d" % i], z["cnts%d" % i], z["pts%d" % i], z["offs%d" % i], bool(z["has_type"][i]), float(z["ds_factor"][i]))
             for i, t in enumerate(z["tissues"])]
    meta = OrderedDict()
    for k in z.files:
        if k.startswith("meta/"):
            name = k[5:]
            if ("metaunits/" + name) in z.files:
                meta[name] = {"resolution": float(z[k]), "units": str(z["metaunits/" + name])}
            else:
                meta[name] = z[k]
    return parts, meta


def main(argv=None):
    """python -m cerberus_amd.inst_info <parts.npz> <out.dat> [<extra.pkl>]: build the dictionary, merge pre-built entries, write + rename, clean up."""
    import pickle
    import time

    argv = list(sys.argv[1:] if argv is None else argv)
    src, dst = argv[0], argv[1]
    t0 = time.perf_counter()
    parts, meta = load_parts(src)
    obj = build_from_parts(parts, OrderedDict())
    if len(argv) > 2 and argv[2]:
        with open(argv[2], "rb") as fh:
            extra = pickle.load(fh)
        for k, v in extra.items():
            obj[k] = v
        os.remove(argv[2])
    obj.update(meta)
    t1 = time.perf_counter()
    write_dat(obj, dst + ".part")
    os.replace(dst + ".part", dst)
    os.remove(src)
    if os.environ.get("CERB_DAT_WRITER_TIMING"):
        with open(dst + ".time", "w") as fh:
            fh.write("%.6f %.6f" % (t1 - t0, time.perf_counter() - t1))


if __name__ == "__main__":
    main()
