"""TEST INFRASTRUCTURE: reader / comparator for tests/golden/inst_info.npz -- the dictionaries the REFERENCE's get_inst_info_dict
(/root/reference/loader/postproc.py:12-98) returned for the golden label maps (generator: oracle/gen_golden_instinfo.py)."""
import os

import numpy as np


def cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "inst_info.npz"))
    for tag in [str(n) for n in g["names"]]:
        ref = {k: g["ref/%s/%s" % (tag, k)] for k in ("ids", "box", "centroid", "type", "type_prob", "ncont", "contour", "centroid_is_int")}
        yield tag, g["lab/" + tag], g["typ/" + tag], float(g["ds/" + tag]), bool(g["with_type/" + tag]), ref


def check(info, ref, with_type, tag=""):
    """info: a get_inst_info_dict result; ref: the flattened reference dictionary.  Key set AND order, box, contour exact; centroid to
    1e-9 (exact integers after ds rounding); type exact; type_prob to 1e-12."""
    ids = [int(k) for k in info.keys()]
    assert ids == [int(i) for i in ref["ids"]], (tag, ids[:8], ref["ids"][:8])
    off = 0
    for i, k in enumerate(info.keys()):
        d = info[k]
        assert set(d.keys()) == ({"box", "centroid", "contour", "type", "type_prob"} if with_type else {"box", "centroid", "contour"}), (tag, k)
        assert np.array_equal(np.asarray(d["box"]), ref["box"][i]), (tag, k, d["box"], ref["box"][i])
        c = np.asarray(d["centroid"])
        if bool(ref["centroid_is_int"]):
            assert np.issubdtype(c.dtype, np.integer) and np.array_equal(c, ref["centroid"][i].astype(np.int64)), (tag, k)
        else:
            assert np.allclose(c, ref["centroid"][i], rtol=0, atol=1e-9), (tag, k, c, ref["centroid"][i])
        n = int(ref["ncont"][i])
        pts = np.asarray(d["contour"])
        assert pts.shape == (n, 2) and np.array_equal(pts, ref["contour"][off:off + n]), (tag, k)
        off += n
        if with_type:
            assert int(d["type"]) == int(ref["type"][i]), (tag, k, d["type"], ref["type"][i])
            assert abs(float(d["type_prob"]) - float(ref["type_prob"][i])) < 1e-12, (tag, k)
    assert off == ref["contour"].shape[0], tag
