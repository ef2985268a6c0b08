"""Host side of the instance dictionary (loader/postproc.py:12-98, infer/wsi.py:805-853) WITHOUT torch: numpy only.

Lives apart from postproc.py / wsi.py so that the writer process of a slide's `.dat` file (`python -m cerberus_amd.inst_info <parts.npz> <out.dat>`,
started by cerberus_amd.wsi.DatWriter.from_arrays) imports in a fraction of a second and shares nothing with the process that drives the GPU:
it receives the per-instance TABLES (cerb_inst_table rows, contour point lists) as arrays, builds the ~1e6 per-instance dictionaries, draws
the uuid keys and writes the pickle -- all of which used to sit between two slides of the parent."""
import os
import sys
from collections import OrderedDict

import numpy as np


def info_from_table(tab, cnts, pts, offs, has_type, ds_factor=1.0, flat_box=False, origin=None):
    """Host side of get_inst_info_dict: the per-instance dictionaries from cerb_inst_table's rows and the compact contour list.
    Everything numeric is computed for all instances at once; the per-instance values are row views of those arrays (a slide has
    ~1e6 nuclei: array construction per instance would dominate the whole slide).  Rules kept from loader/postproc.py:34-35,55-75,
    78-96 (flat_box: the slide dictionary's [x1, y1, x2, y2] form of infer/wsi.py:817-825 instead of [[y1, x1], [y2, x2]]):
    drop contours with < 3 points; majority type, skipping background when a second class exists; type_prob =
    votes / (area + 1e-6); with ds_factor != 1 box / centroid / contour are np.round(x / ds_factor).astype(int).
    origin: optional int [n, 2] (x, y) added to every coordinate of instance i AFTER all of the above -- instances measured inside a tile
    (cerberus_amd/ref_tiling.py) placed in the slide, with the arithmetic of `value + offset` on the finished values."""
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
    if origin is not None:
        origin = np.asarray(origin, dtype=np.int64).reshape(n, 2)
        box = box + (np.concatenate([origin, origin], axis=1) if flat_box else origin[:, None, ::-1])
        cen = cen + origin
        pts = pts + np.repeat(origin, np.asarray(cnts, dtype=np.int64), axis=0)[: len(pts)] if len(pts) == int(np.sum(cnts)) else _shift_points(pts, cnts, offs, origin)
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


def _shift_points(pts, cnts, offs, origin):
    """pts + origin[i] for the points of instance i when the point list is not simply the concatenation of the instances' runs."""
    out = np.array(pts, copy=True)
    for i in np.nonzero(np.asarray(cnts) > 0)[0].tolist():
        out[offs[i]:offs[i] + cnts[i]] += origin[i]
    return out


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



# ---- the same file, written without building a Python object per instance -----------------------------------------------------------
# pickle.dump walks ~1e6 dictionaries and ~3e6 small arrays through Python-level reduce calls: 8 s for a 40000^2 slide, as long as the slide's
# inference.  The stream it produces is regular, though: per instance the same opcodes around a uuid, three array payloads, an int and a
# float.  write_dat_fast assembles those records as numpy byte matrices (one row per instance) and writes them out: a protocol-4 pickle that
# pickle.load / joblib.load turn into the same dictionary (tests/test_host_logic.py compares it entry for entry with write_dat's).
#   * contour arrays have a variable length, everything else is fixed size: the contours are emitted FIRST, grouped by point count (fixed-size
#     rows per group), each stored in the unpickler's memo (LONG_BINPUT) and popped; the per-instance records then refer to them (LONG_BINGET);
#   * arrays are rebuilt through `numpy.core.multiarray._reconstruct`, the module path numpy 1.x (the reference's) AND 2.x resolve -- numpy 2's
#     own pickles name `numpy._core`, which numpy 1.x cannot load;
#   * no MEMOIZE / FRAME opcodes: every memo index is explicit.
def _u8(b):
    return np.frombuffer(b, np.uint8)


def _le32(v):
    return np.ascontiguousarray(np.asarray(v, dtype="<u4")).view(np.uint8).reshape(-1, 4)


def _sbu(text):  # SHORT_BINUNICODE
    b = text.encode("utf-8")
    assert len(b) < 256
    return b"\x8c" + bytes([len(b)]) + b


def _emit_small(obj, out):
    """Generic emitter for the handful of non-instance entries (resolution records, dimension arrays, pre-built dictionaries): no memo use."""
    import struct

    if obj is None:
        out.append(b"N")
    elif obj is True or obj is False:
        out.append(b"\x88" if obj else b"\x89")
    elif isinstance(obj, (int, np.integer)):
        v = int(obj)
        out.append(b"J" + struct.pack("<i", v) if -2 ** 31 <= v < 2 ** 31 else b"\x8a\x08" + struct.pack("<q", v))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"G" + struct.pack(">d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(_sbu(obj) if len(b) < 256 else b"X" + struct.pack("<I", len(b)) + b)
    elif isinstance(obj, bytes):
        out.append(b"B" + struct.pack("<I", len(obj)) + obj)
    elif isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        if a.dtype.hasobject or a.dtype.names:
            raise TypeError("write_dat_fast: object / structured arrays are not handled")
        out.append(b"cnumpy.core.multiarray\n_reconstruct\ncnumpy\nndarray\nK\x00\x85C\x01b\x87R(K\x01")
        _emit_small(tuple(int(d) for d in a.shape), out)
        out.append(_dtype_ops(a.dtype) + b"\x89")
        _emit_small(a.tobytes(), out)
        out.append(b"tb")
    elif isinstance(obj, OrderedDict):
        out.append(b"ccollections\nOrderedDict\n)R(")
        for k, v in obj.items():
            _emit_small(k, out)
            _emit_small(v, out)
        out.append(b"u")
    elif isinstance(obj, dict):
        out.append(b"}(")
        for k, v in obj.items():
            _emit_small(k, out)
            _emit_small(v, out)
        out.append(b"u")
    elif isinstance(obj, tuple):
        out.append(b"(")
        for v in obj:
            _emit_small(v, out)
        out.append(b"t")
    elif isinstance(obj, list):
        out.append(b"](")
        for v in obj:
            _emit_small(v, out)
        out.append(b"e")
    else:
        raise TypeError("write_dat_fast: cannot emit %r" % type(obj))


def _dtype_ops(dt):
    """numpy.dtype(<kind+size>, False, True) + its state, for plain numeric dtypes (what ndarray.__reduce__ writes)."""
    dt = np.dtype(dt)
    order = b"|" if dt.itemsize == 1 else (b"<" if dt.str[0] in "<=|" else b">")
    return b"cnumpy\ndtype\n" + _sbu(dt.str[1:]) + b"\x89\x88\x87R(K\x03" + b"\x8c\x01" + order + b"NNNJ\xff\xff\xff\xffJ\xff\xff\xff\xffK\x00tb"


class _Memo(object):
    def __init__(self):
        self.n = 0

    def put(self):  # LONG_BINPUT <next index>, POP
        i = self.n
        self.n += 1
        return i, b"r" + int(i).to_bytes(4, "little") + b"0"


def _get(i):  # BINGET / LONG_BINGET
    return b"h" + bytes([i]) if i < 256 else b"j" + int(i).to_bytes(4, "little")


def _array_row_template(g_rec, g_nd, shape, dt_idx, nbytes):
    """-> (bytes of the construct-and-BUILD sequence of one array with a placeholder payload, offset of the payload inside it)."""
    import struct

    def small_int(v):
        return b"K" + bytes([v]) if v < 256 else b"J" + struct.pack("<i", v)

    head = _get(g_rec) + _get(g_nd) + b"K\x00\x85C\x01b\x87R(K\x01"
    head += b"".join(small_int(d) for d in shape) + (b"\x85" if len(shape) == 1 else b"\x86")
    head += _get(dt_idx) + b"\x89" + (b"C" + bytes([nbytes]) if nbytes < 256 else b"B" + struct.pack("<I", nbytes))
    return head + b"\x00" * nbytes + b"tb", len(head)


def write_dat_fast(parts, meta, path, extra=None):
    """The file write_dat(build_from_parts(parts, {}) + extra + meta) writes -- same keys in the same order, equal values of equal dtypes, fresh
    uuid keys -- assembled as byte matrices.  parts: as build_from_parts; extra: {key: small ready object} placed after the tissues."""
    memo = _Memo()
    with open(path, "wb") as fh:
        fh.write(b"\x80\x04")
        g_rec, ops = memo.put()
        fh.write(b"cnumpy.core.multiarray\n_reconstruct\n" + ops)
        g_nd, ops = memo.put()
        fh.write(b"cnumpy\nndarray\n" + ops)
        dts = {}

        def dtype_index(dt):
            key = np.dtype(dt).str
            if key not in dts:
                i, ops_ = memo.put()
                fh.write(_dtype_ops(dt) + ops_)
                dts[key] = i
            return dts[key]

        prepared = []
        for part in parts:
            tissue, tab, cnts, pts, offs, has_type, ds_factor = part[:7]
            origin = np.asarray(part[7], dtype=np.int64).reshape(-1, 2) if len(part) > 7 and part[7] is not None else None
            n = tab.shape[0]
            area = tab[:, 0] if n else np.zeros(0, np.int64)
            keep = np.nonzero((area > 0) & (cnts >= 3))[0] if n else np.zeros(0, np.int64)
            if keep.size == 0:
                prepared.append((tissue, None))
                continue
            a = np.maximum(area, 1).astype(np.float64)
            box = np.stack([tab[:, [3, 5]], tab[:, [4, 6]]], axis=1)
            cen = np.stack([tab[:, 1] / a, tab[:, 2] / a], axis=1)
            if float(ds_factor) != 1.0:
                box = np.round(box / ds_factor).astype("int")
                cen = np.round(cen / ds_factor).astype("int")
                pts = np.round(pts / ds_factor).astype("int")
            box = box.reshape(n, 4)[:, [1, 0, 3, 2]]
            if origin is not None:
                box = box + np.concatenate([origin, origin], axis=1)
                cen = cen + origin
            box = np.ascontiguousarray(box[keep])
            cen = np.ascontiguousarray(cen[keep])
            pts = np.ascontiguousarray(pts.astype(np.int64) if origin is not None else pts)  # (points + int64 origin: int64, as numpy promotes in the object path)
            kc, ko = cnts[keep].astype(np.int64), offs[keep].astype(np.int64)
            korg = origin[keep] if origin is not None else None
            dt_box, dt_cen, dt_pts = dtype_index(box.dtype), dtype_index(cen.dtype), dtype_index(pts.dtype)
            # ---- the contours, grouped by point count: fixed-size rows, each array memoised and popped
            cidx = np.zeros(keep.size, np.int64)
            isz = pts.dtype.itemsize * 2
            for k in np.unique(kc).tolist():
                rows = np.nonzero(kc == k)[0]
                tmpl, pay = _array_row_template(g_rec, g_nd, (k, 2), dt_pts, k * isz)
                rec = np.empty((rows.size, len(tmpl) + 6), np.uint8)
                rec[:, :len(tmpl)] = _u8(tmpl)
                gathered = pts[ko[rows][:, None] + np.arange(k)[None, :]]  # [rows, k, 2]
                if korg is not None:
                    gathered = gathered + korg[rows][:, None, :].astype(gathered.dtype)
                rec[:, pay:pay + k * isz] = gathered.reshape(rows.size, -1).view(np.uint8).reshape(rows.size, k * isz)
                base = memo.n
                memo.n += rows.size
                cidx[rows] = base + np.arange(rows.size)
                rec[:, len(tmpl)] = ord("r")
                rec[:, len(tmpl) + 1:len(tmpl) + 5] = _le32(cidx[rows])
                rec[:, len(tmpl) + 5] = ord("0")
                fh.write(rec.tobytes())
            if has_type:
                votes = tab[:, 8:16]
                order = np.argsort(-votes, axis=1, kind="stable")
                t0, t1 = order[:, 0], order[:, 1]
                second = np.take_along_axis(votes, t1[:, None], 1)[:, 0] > 0
                typ = np.where((t0 == 0) & second, t1, t0)
                prob = np.take_along_axis(votes, typ[:, None], 1)[:, 0] / (area + 1.0e-6)
                typ, prob = typ[keep], prob[keep]
            else:
                typ = prob = None
            prepared.append((tissue, (box, cen, cidx, typ, prob, dt_box, dt_cen)))
        # ---- the dictionary itself
        fh.write(b"ccollections\nOrderedDict\n)R")
        for tissue, pr in prepared:
            fh.write(_sbu(tissue) + b"ccollections\nOrderedDict\n)R")
            if pr is not None:
                box, cen, cidx, typ, prob, dt_box, dt_cen = pr
                n = box.shape[0]
                tb, pb = _array_row_template(g_rec, g_nd, (4,), dt_box, box.dtype.itemsize * 4)
                tc, pc = _array_row_template(g_rec, g_nd, (2,), dt_cen, cen.dtype.itemsize * 2)
                pieces, at, pos = [], {}, 0

                def add(name, b):
                    nonlocal pos
                    at[name] = pos
                    pieces.append(b)
                    pos += len(b)

                add("key", b"\x8c\x20" + b"0" * 32)
                add("open", b"}(" + _sbu("box"))
                add("box", tb)
                add("k2", _sbu("centroid"))
                add("cen", tc)
                add("k3", _sbu("contour") + b"j")
                add("cidx", b"\x00" * 4)
                if typ is not None:
                    assert int(typ.min()) >= 0 and int(typ.max()) < 256
                    add("k4", _sbu("type") + b"K")
                    add("typ", b"\x00")
                    add("k5", _sbu("type_prob") + b"G")
                    add("prob", b"\x00" * 8)
                add("close", b"u")
                tmpl = b"".join(pieces)
                hexkeys = _u8("".join(_uuid4_hex(n)).encode("ascii")).reshape(n, 32)
                chunk = 1 << 16
                fh.write(b"(")
                for c0 in range(0, n, chunk):  # bounded working set: 64 k instances (~15 MB) at a time
                    c1 = min(n, c0 + chunk)
                    rec = np.empty((c1 - c0, len(tmpl)), np.uint8)
                    rec[:] = _u8(tmpl)
                    rec[:, at["key"] + 2:at["key"] + 34] = hexkeys[c0:c1]
                    rec[:, at["box"] + pb:at["box"] + pb + box.dtype.itemsize * 4] = box[c0:c1].view(np.uint8).reshape(c1 - c0, -1)
                    rec[:, at["cen"] + pc:at["cen"] + pc + cen.dtype.itemsize * 2] = cen[c0:c1].view(np.uint8).reshape(c1 - c0, -1)
                    rec[:, at["cidx"]:at["cidx"] + 4] = _le32(cidx[c0:c1])
                    if typ is not None:
                        rec[:, at["typ"]] = typ[c0:c1].astype(np.uint8)
                        rec[:, at["prob"]:at["prob"] + 8] = np.ascontiguousarray(prob[c0:c1].astype(">f8")).view(np.uint8).reshape(c1 - c0, 8)
                    fh.write(rec.tobytes())
                fh.write(b"u")
            fh.write(b"s")
        tail = []
        for src in (extra or {}, meta or {}):
            for k, v in src.items():
                _emit_small(k, tail)
                _emit_small(v, tail)
                tail.append(b"s")
        fh.write(b"".join(tail) + b".")


def build_from_parts(parts, meta):
    """parts: [(tissue, tab, cnts, pts, offs, has_type, ds_factor[, origin])]; meta: the resolution entries.  -> the dictionary of infer/wsi.py:805-853."""
    out = OrderedDict()
    for part in parts:
        tissue, tab, cnts, pts, offs, has_type, ds_factor = part[:7]
        info = info_from_table(tab, cnts, pts, offs, bool(has_type), float(ds_factor), flat_box=True, origin=part[7] if len(part) > 7 else None)
        out[tissue] = OrderedDict(zip(_uuid4_hex(len(info)), info.values()))
    out.update(meta)
    return out


def save_parts(path, parts, meta):
    """One uncompressed .npz holding every array the writer process needs."""
    arrs = {"tissues": np.array([p[0] for p in parts]), "has_type": np.array([int(bool(p[5])) for p in parts]), "ds_factor": np.array([float(p[6]) for p in parts])}
    for i, p in enumerate(parts):
        arrs["tab%d" % i], arrs["cnts%d" % i], arrs["pts%d" % i], arrs["offs%d" % i] = p[1], p[2], p[3], p[4]
        if len(p) > 7 and p[7] is not None:
            arrs["origin%d" % i] = np.asarray(p[7], dtype=np.int64)
    for k, v in meta.items():
        arrs["meta/" + k] = np.asarray(v["resolution"] if isinstance(v, dict) else v)
        if isinstance(v, dict):
            arrs["metaunits/" + k] = np.array(v["units"])
    with open(path, "wb") as fh:
        np.savez(fh, **arrs)


def load_parts(path):
    z = np.load(path, allow_pickle=False)
    parts = [(str(t), z["tab%d" % i], z["cnts%d" % i], z["pts%d" % i], z["offs%d" % i], bool(z["has_type"][i]), float(z["ds_factor"][i]))
             + ((z["origin%d" % i],) if ("origin%d" % i) in z.files else ())
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
    extra = OrderedDict()
    if len(argv) > 2 and argv[2]:
        with open(argv[2], "rb") as fh:
            extra = pickle.load(fh)
        os.remove(argv[2])
    big_extra = any(isinstance(v, dict) and len(v) > 20000 for v in extra.values())  # e.g. the --reference_tiling nuclei, built as objects already
    clash = any(k in extra for k in (p[0] for p in parts))
    if os.environ.get("CERB_DAT_WRITER") != "pickle" and not big_extra and not clash:
        t1 = time.perf_counter()
        write_dat_fast(parts, meta, dst + ".part", extra)  # no per-instance Python objects at all
    else:
        obj = build_from_parts(parts, OrderedDict())
        for k, v in extra.items():
            obj[k] = v
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
