"""ORACLE (test infrastructure only): ctypes wrapper of oracle/postproc_ref.c, same call protocol as the reference's
PostProcInstErodedContourMap.post_process (loader/postproc.py:383-407).  Never imported by cerberus_amd/."""
import copy
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpostproc_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "postproc_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_SO)
        for f in ("ref_proc_gland", "ref_proc_lumen"):
            getattr(_lib, f).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        _lib.ref_proc_nuclei.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.ref_label4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.ref_watershed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.ref_fill_holes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib.ref_dilate_ellipse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.ref_erode_cross3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.ref_remove_small_labels.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def label4(mask):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros(m.shape, np.int32)
    n = lib().ref_label4(_p(m), m.shape[0], m.shape[1], _p(out))
    return out, n


def watershed(image_f32, markers_i32, mask):
    img = np.ascontiguousarray(image_f32, dtype=np.float32)
    mk = np.ascontiguousarray(markers_i32, dtype=np.int32)
    ms = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros(img.shape, np.int32)
    lib().ref_watershed(_p(img), _p(mk), _p(ms), img.shape[0], img.shape[1], _p(out))
    return out


def fill_holes(mask):
    m = np.ascontiguousarray(mask, dtype=np.uint8).copy()
    lib().ref_fill_holes(_p(m), m.shape[0], m.shape[1])
    return m


def dilate_ellipse(mask, k):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros_like(m)
    lib().ref_dilate_ellipse(_p(m), m.shape[0], m.shape[1], int(k), _p(out))
    return out


def erode_cross3(mask):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros_like(m)
    lib().ref_erode_cross3(_p(m), m.shape[0], m.shape[1], _p(out))
    return out


def proc(inst_fg, tissue, ds_factor=1.0):
    """inst_fg (H,W,2) float32 -> instance map with the reference's dtype (int32 for the nuclei watershed branch,
    float64 otherwise: postproc.py:290,331,380)."""
    a = np.ascontiguousarray(inst_fg, dtype=np.float32)
    H, W = a.shape[:2]
    out = np.zeros((H, W), np.int32)
    t = tissue.upper()
    if t == "NUCLEI":
        ran = lib().ref_proc_nuclei(_p(a), H, W, _p(out))
        return out if ran else out.astype(np.float64)
    if t == "GLAND":
        lib().ref_proc_gland(_p(a), H, W, C.c_float(ds_factor), _p(out))
    elif t == "LUMEN":
        lib().ref_proc_lumen(_p(a), H, W, C.c_float(ds_factor), _p(out))
    else:
        raise AssertionError(tissue)
    return out.astype(np.float64)


class PostProcInstErodedContourMap(object):
    @classmethod
    def post_process(cls, raw_map, idx_dict, tissue_mode, ds_factor=1.0):
        assert tissue_mode.upper() in ("LUMEN", "GLAND", "NUCLEI")
        tissue_ch = "%s-INST" % tissue_mode
        idx_dict = copy.deepcopy(idx_dict)
        assert tissue_ch in list(idx_dict.keys())
        inst_fg = raw_map[..., idx_dict[tissue_ch][0]: idx_dict[tissue_ch][1]]
        inst_map = proc(inst_fg, tissue_mode, ds_factor)
        type_ch = tissue_mode + "-" + "TYPE"
        if type_ch in list(idx_dict.keys()):
            type_map = np.squeeze(raw_map[..., idx_dict[type_ch][0]: idx_dict[type_ch][1]])
        else:
            type_map = None
        return inst_map, type_map


def inst_info_ref(inst_map, type_map=None, ds_factor=1.0):
    """ORACLE restatement of get_inst_info_dict (loader/postproc.py:12-98; pinned against the reference's own function on the golden label
    maps by oracle/gen_golden_instinfo.py -> tests/golden/inst_info.npz, OpenCV's two calls through the stand-in): box from get_bounding_box (misc/utils.py:82-91),
    centroid = cv2.moments m10/m00, m01/m00 of the cropped binary mask (= mean x, mean y), contour =
    findContours(crop, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0] shifted by the box origin with the `< 3 points -> skip` filter
    (postproc.py:26-41; OpenCV is absent here, so findContours is oracle/cv2_standin.py's Suzuki-Abe restatement -- parity
    unpinned for it), majority type vote incl. the 'skip background if a second class exists' rule."""
    from collections import OrderedDict

    from . import cv2_standin as cv2

    info = OrderedDict()
    for inst_id in np.unique(inst_map)[1:]:
        m = inst_map == inst_id
        rows, cols = np.any(m, axis=1), np.any(m, axis=0)
        rmin, rmax = np.where(rows)[0][[0, -1]]
        cmin, cmax = np.where(cols)[0][[0, -1]]
        rmax += 1
        cmax += 1
        ys, xs = np.nonzero(m)
        crop = m[rmin:rmax, cmin:cmax].astype(np.uint8)
        contour = np.squeeze(cv2.findContours(crop, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0].astype("int32"))
        if contour.shape[0] < 3 or len(contour.shape) != 2:
            continue
        contour[:, 0] += cmin
        contour[:, 1] += rmin
        d = {"box": np.array([[rmin, cmin], [rmax, cmax]]), "centroid": np.array([xs.mean(), ys.mean()]), "contour": contour}
        if type_map is not None:
            t = type_map[m]
            tl, tc = np.unique(t, return_counts=True)
            lst = sorted(zip(tl, tc), key=lambda x: x[1], reverse=True)
            it = lst[0][0]
            if it == 0 and len(lst) > 1:
                it = lst[1][0]
            d["type"] = int(it)
            d["type_prob"] = float(dict(lst)[it] / (m.sum() + 1.0e-6))
        info[int(inst_id)] = d
    if ds_factor != 1.0:  # loader/postproc.py:78-96: back to the resolution the slide is annotated at
        for k in list(info.keys()):
            for f in ("box", "centroid", "contour"):
                info[k][f] = np.round(info[k][f] / ds_factor).astype("int")
    return info
