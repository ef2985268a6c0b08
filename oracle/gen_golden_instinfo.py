"""Pin the instance-dictionary oracle (oracle/postproc_ref.py::inst_info_ref) and the HIP instance table against the REFERENCE's own
get_inst_info_dict (/root/reference/loader/postproc.py:12-98), run in this container.

Run:  /opt/conda/bin/python3.9 oracle/gen_golden_instinfo.py      (real scipy / scikit-image; cv2 = oracle/cv2_standin.py)

What this pins: the box arithmetic (get_bounding_box, misc/utils.py:82-91), the `< 3 contour points -> skip` rule (:34-41), the
majority vote over the instance's class pixels with its 'background loses to a second class' rule and the stable-sort tie order
(:55-75), type_prob = votes / (area + 1e-6), the key set and key order, and the ds_factor rounding (:78-96).  What it cannot pin:
cv2.moments / cv2.findContours themselves (OpenCV is not installed anywhere here; the stand-in restates them).

Label maps: the reference-generated maps of tests/golden/pp_cases.npz plus three hand-made ones (ties in the vote, background
majority, one- and two-pixel instances, an instance in two pieces).  Type maps: seeded blocky class maps with background holes.
Stored per case: label map, type map, ds, and the reference's dictionary flattened into arrays.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import oracle.cv2_standin as cv2_standin  # noqa: E402

sys.modules["cv2"] = cv2_standin
from loader.postproc import get_inst_info_dict as ref_info  # noqa: E402  (reference)

from oracle import postproc_ref as pr  # noqa: E402


def type_map_for(shape, seed, n_cls=7, cell=9, p_bg=0.35):
    """Blocky class map: cells of `cell` pixels carry one class, a share of the cells is background (0), then 3 % pixel noise."""
    rs = np.random.RandomState(seed)
    h, w = shape
    gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
    g = rs.randint(1, n_cls, (gh, gw))
    g[rs.rand(gh, gw) < p_bg] = 0
    t = np.kron(g, np.ones((cell, cell), np.int64))[:h, :w]
    noise = rs.rand(h, w) < 0.03
    t[noise] = rs.randint(0, n_cls, int(noise.sum()))
    return t.astype(np.uint8)


def handmade():
    out = []
    # exact ties in the vote, background majorities with and without a second class, tiny instances
    L = np.zeros((40, 48), np.int32)
    T = np.zeros((40, 48), np.uint8)
    L[2:8, 2:10] = 1; T[2:8, 2:6] = 3; T[2:8, 6:10] = 5          # 24 / 24 tie between classes 3 and 5 -> 3 (stable sort after np.unique)
    L[10:16, 2:10] = 2; T[10:16, 2:4] = 4                          # background majority, second class exists -> 4
    L[18:24, 2:10] = 3                                             # only background -> 0
    L[26:30, 2:6] = 4; T[26:28, 2:6] = 2; T[28:30, 2:4] = 6       # 8 / 4 / 4(bg)
    L[2, 20] = 5; T[2, 20] = 1                                     # one pixel: contour of 1 point -> skipped
    L[5, 20:22] = 6; T[5, 20:22] = 1                               # two pixels: 2 points -> skipped
    L[8:10, 20:22] = 7; T[8:10, 20:22] = 2                         # 2 x 2: 4 points
    L[12:15, 20] = 8; T[12:15, 20] = 3                             # vertical line of 3: 2 points -> skipped
    L[20:26, 20:30] = 9; L[22:24, 23:27] = 0; T[20:26, 20:30] = 6  # ring
    L[30:34, 20:24] = 10; L[36:39, 30:34] = 10; T[30:34, 20:24] = 1; T[36:39, 30:34] = 2  # one id in two pieces: 16 vs 12 votes
    L[0:3, 40:48] = 11; T[0:3, 40:48] = 5                          # touches the top and right border
    L[37:40, 0:3] = 12; T[37:40, 0:3] = 0                          # bottom-left corner, background only
    out.append(("hand_votes", L, T))
    # ids with gaps (np.unique skips them) and an id far from 1
    L2 = np.zeros((32, 32), np.int32)
    L2[1:6, 1:6] = 2; L2[10:20, 8:15] = 5; L2[22:30, 20:31] = 40
    T2 = type_map_for((32, 32), 7, cell=4)
    out.append(("hand_gaps", L2, T2))
    return out


def flatten(info):
    ids = np.array(list(info.keys()), dtype=np.int64)
    n = len(ids)
    box = np.zeros((n, 2, 2), np.int64)
    cen = np.zeros((n, 2), np.float64)
    typ = np.full(n, -1, np.int64)
    prob = np.full(n, -1.0, np.float64)
    counts = np.zeros(n, np.int64)
    pts = []
    for i, k in enumerate(ids):
        d = info[k]
        box[i] = d["box"]
        cen[i] = d["centroid"]
        c = np.asarray(d["contour"]).reshape(-1, 2)
        counts[i] = c.shape[0]
        pts.append(c.astype(np.int64))
        if "type" in d:
            typ[i] = d["type"]
            prob[i] = d["type_prob"]
    pts = np.concatenate(pts, 0) if pts else np.zeros((0, 2), np.int64)
    return {"ids": ids, "box": box, "centroid": cen, "type": typ, "type_prob": prob, "ncont": counts, "contour": pts,
            "centroid_is_int": np.array(n > 0 and np.issubdtype(np.asarray(info[ids[0]]["centroid"]).dtype, np.integer))}


def same(a, b):
    if list(a.keys()) != list(b.keys()):
        return False
    for k in a:
        if sorted(a[k].keys()) != sorted(b[k].keys()):
            return False
        for f in a[k]:
            x, y = np.asarray(a[k][f]), np.asarray(b[k][f])
            if x.shape != y.shape or x.dtype.kind != y.dtype.kind:
                return False
            if f == "type_prob":
                if abs(float(x) - float(y)) > 1e-12:
                    return False
            elif x.dtype.kind == "f":
                if not np.allclose(x, y, rtol=0, atol=1e-9):
                    return False
            elif not np.array_equal(x, y):
                return False
    return True


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "pp_cases.npz"))
    cases = []
    for name, seed in (("nuc_generic", 1), ("nuc_touching", 2), ("nuc_small", 3), ("nuc_border", 4), ("gland_generic", 5), ("lumen_generic", 6),
                       ("gland_ds05", 7), ("lumen_ds05", 8), ("gland_holes", 9), ("nuc_empty", 10)):
        lab = g["out/" + name].astype(np.int32)
        ds = float(g["ds/" + name])
        cases.append((name, lab, type_map_for(lab.shape, 1000 + seed, cell=5 if name.startswith("nuc") else 17), ds))
    for name, L, T in handmade():
        cases.append((name, L, T, 1.0))
        cases.append((name + "_ds05", L, T, 0.5))
    cases.append(("nuc_generic_ds05", g["out/nuc_generic"].astype(np.int32), type_map_for(g["out/nuc_generic"].shape, 1011, cell=5), 0.5))
    store, names = {}, []
    for name, lab, typ, ds in cases:
        for with_type in (True, False):
            tag = name + ("" if with_type else "_notype")
            if not with_type and not (name.startswith("hand") or name in ("nuc_generic", "gland_ds05")):
                continue
            ref = ref_info(lab.copy(), typ.copy() if with_type else None, ds_factor=ds)
            orc = pr.inst_info_ref(lab, typ if with_type else None, ds_factor=ds)
            ok = same(ref, orc)
            print("%-24s ds=%.1f type=%d  instances in map %4d, in dict %4d   oracle == reference: %s" %
                  (tag, ds, with_type, len(np.unique(lab)) - 1, len(ref), ok))
            assert ok, tag
            names.append(tag)
            store["lab/" + tag] = lab.astype(np.int32)
            store["typ/" + tag] = typ
            store["ds/" + tag] = np.float32(ds)
            store["with_type/" + tag] = np.array(with_type)
            for k, v in flatten(ref).items():
                store["ref/%s/%s" % (tag, k)] = v
    store["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "inst_info.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
