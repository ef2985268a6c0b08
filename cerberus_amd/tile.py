"""Tile-mode inference manager: mirror of the reference's infer/base.py + infer/tile.py on the HIP path.

Kept from the reference: patch geometry (`_prepare_patching`, infer/tile.py:43-106), the channel layout of the
stitched canvas (infer/tile.py:116-134), the post-processing dispatch and the lumen-inside-gland rule (:168-191), the
x2 nearest up-scaling of the saved maps (:196-199), the `.mat` outputs and the resume-by-skip rule (:225-238, :261-287).

Different by design (MI355X-first):
  * every patch is inferred ONCE.  With patch_output_overlap == 0 the reference appends every patch twice and
    averages (infer/tile.py:90-103,160): (x + x) / (2 + 1e-8 -> 2.0f) == x exactly in float32, so the stitched maps
    are identical.
  * stitching is not a host loop: the head kernels write each patch's centre crop straight into device-resident
    per-head canvases (tile_off / row_stride of cerb_net_forward), and post-processing reads those canvases in place;
    only the final integer label maps leave the GPU.
  * post-processing runs in the main process (GPU handles are not picklable) -- the reference's
    nr_post_proc_workers=0 path.
Not implemented (SURVEY.md par.8f "next" rows): contour tracing (cv2.findContours) and the overlay jpg.
"""
import math
import os
import pathlib
from collections import OrderedDict

import numpy as np
import torch

from .net_desc import create_model
from .postproc import get_inst_info_dict, mask_lumen_by_gland, postproc_device
from .run_desc import infer_step

POSTPROC_CODES = ("IP-ERODED-CONTOUR-3", "IP-ERODED-CONTOUR-11")  # -> PostProcInstErodedContourMap (infer/tile.py:35-40)


def _pad_reflect_numpy1(img, pads):
    """np.pad(img, pads, "reflect") AS NUMPY < 2.0 COMPUTES IT (the reference pins numpy 1.21.5, environment.yml:11).  When a pad
    is wider than the image side minus one, numpy 1.x fills it iteratively from whatever is already valid on BOTH sides, which
    is not the periodic mirror numpy >= 2.0 produces; an image narrower than a quarter of the window shows the difference.
    Restated from numpy/lib/arraypad.py (_set_reflect_both, 1.21); pinned by tests/golden/tile_patching.npz cases 8, 12, 15, 17, 27."""
    img = np.asarray(img)
    out_shape = tuple(img.shape[a] + pads[a][0] + pads[a][1] for a in range(img.ndim))
    out = np.zeros(out_shape, img.dtype)
    out[tuple(slice(pads[a][0], pads[a][0] + img.shape[a]) for a in range(img.ndim))] = img
    for axis in range(img.ndim):
        left, right = int(pads[axis][0]), int(pads[axis][1])
        if left == 0 and right == 0:
            continue
        # view of everything already valid along the other axes (numpy pads axis after axis on the growing region of interest)
        roi = tuple(slice(None) if a < axis else (slice(None) if a == axis else slice(pads[a][0], pads[a][0] + img.shape[a])) for a in range(img.ndim))
        v = out[roi]
        v = np.moveaxis(v, axis, 0)  # a view: writes go through to `out`
        n = v.shape[0]
        if img.shape[axis] == 1:  # numpy: a length-1 axis is padded with its edge value
            v[:left] = v[left]
            v[n - right:] = v[n - right - 1]
            continue
        while left > 0 or right > 0:
            old_length = n - right - left - 1
            if left > 0:
                chunk = min(old_length, left)
                stop = left
                start = stop + chunk
                v[left - chunk:left] = v[start:stop:-1]
                left -= chunk
            if right > 0:
                chunk = min(old_length, right)
                start = n - right - 2
                stop = start - chunk
                v[n - right:n - right + chunk] = v[start:stop:-1] if stop >= 0 else v[start::-1][:chunk]
                right -= chunk
    return out


def _prepare_patching(img, input_size, output_size, output_overlap_size):
    """Mirror padding + patch placement; same return values as the reference (infer/tile.py:43-106):
    padded_img, info_list int32 [P, 2(in/out), 2(tl/br), 2(y/x)], [padt, padl]."""
    win, step = int(input_size), int(output_size)
    im_h, im_w = img.shape[:2]

    def last_step(length):
        nr = math.ceil((length - step) / step)
        return int((nr + 1) * step)

    last_h, last_w = last_step(im_h), last_step(im_w)
    diff = win - step
    padt = padl = diff // 2
    padb, padr = last_h + win - im_h, last_w + win - im_w
    padded = _pad_reflect_numpy1(img, ((padt, padb), (padl, padr), (0, 0)))
    ys = np.arange(0, last_h, step, dtype=np.int32)
    xs = np.arange(0, last_w, step, dtype=np.int32)
    # the reference's np.meshgrid(y, x) (xy indexing) flattens with x slow, y fast
    in_tl = np.stack([np.tile(ys, len(xs)), np.repeat(xs, len(ys))], axis=-1).astype(np.int32)
    out_tl = in_tl + diff // 2

    def boxes(i_tl, o_tl):
        i_br, o_br = i_tl + win, o_tl + step
        keep = ~np.any(i_br > np.array(padded.shape[:2]), axis=-1)
        return np.stack([np.stack([i_tl[keep], i_br[keep]], axis=1), np.stack([o_tl[keep], o_br[keep]], axis=1)], axis=1)

    info = boxes(in_tl, out_tl)
    if output_overlap_size == 0:
        # reference quirk kept for protocol parity: the patch list is appended to itself (infer/tile.py:90-103)
        o2 = out_tl + output_overlap_size
        info = np.concatenate([info, boxes(o2 - diff // 2, o2)], axis=0)
    return padded, info, [padt, padl]


def channel_layout(decoder_kwargs):
    """idx_dict of the stitched canvas (infer/tile.py:119-134): INST -> nr_chans-1 channels, TYPE / other -> 1."""
    idx, n = OrderedDict(), 0
    for tissue_name, info in decoder_kwargs.items():
        for chann_type, nr in info.items():
            s = n
            if chann_type == "INST":
                n += nr - 1
                idx[tissue_name + "-INST"] = [s, n]
            elif chann_type == "TYPE":
                n += 1
                idx[tissue_name.split("#")[0] + "-TYPE"] = [s, n]
            else:
                n += 1
                idx[tissue_name] = [s, n]
    return idx, n


def inst_info_table(inst_map, type_map=None):
    """Per-instance box / centroid / majority type (the non-contour part of get_inst_info_dict,
    loader/postproc.py:12-75).  Host numpy over the final integer maps; contour tracing is a 'next' row."""
    inst_map = np.asarray(inst_map)
    ids = np.unique(inst_map)
    ids = ids[ids != 0]
    info = OrderedDict()
    if ids.size == 0:
        return info
    lab = inst_map.astype(np.int64)
    H, W = lab.shape
    yy, xx = np.divmod(np.arange(H * W), W)
    flat = lab.ravel()
    sel = flat > 0
    f, y, x = flat[sel], yy[sel], xx[sel]
    order = np.argsort(f, kind="stable")
    f, y, x = f[order], y[order], x[order]
    starts = np.flatnonzero(np.r_[True, f[1:] != f[:-1]])
    ends = np.r_[starts[1:], f.size]
    t = type_map.ravel()[sel][order] if type_map is not None else None
    for s, e in zip(starts, ends):
        iid = int(f[s])
        rmin, rmax, cmin, cmax = y[s:e].min(), y[s:e].max() + 1, x[s:e].min(), x[s:e].max() + 1
        d = {
            "box": np.array([[rmin, cmin], [rmax, cmax]]),
            "centroid": np.array([x[s:e].mean(), y[s:e].mean()]),  # cv2.moments m10/m00, m01/m00 of the binary mask
            "contour": None,
        }
        if t is not None:
            tl, tc = np.unique(t[s:e], return_counts=True)
            o = np.argsort(-tc, kind="stable")
            tl, tc = tl[o], tc[o]
            it = tl[0]
            if it == 0 and len(tl) > 1:  # pick the 2nd most dominant if exist (postproc.py:69-71)
                it = tl[1]
            d["type"] = int(it)
            d["type_prob"] = float(dict(zip(tl, tc))[it] / ((e - s) + 1.0e-6))
        info[iid] = d
    return info


class InferManager(object):
    """Tile inference manager (reference infer/base.py:9-54 + infer/tile.py:215-429)."""

    def __init__(self, **kwargs):
        self.run_step = None
        for k, v in kwargs.items():
            setattr(self, k, v)
        self._load_model()

    def _load_model(self):
        net = create_model(**self.model_args)
        ckpt = getattr(self, "checkpoint_path", None)
        if ckpt is None:
            # no checkpoint: the seeded, non-saturating TEST weights (cerberus_amd.weights.make_state_dict) -- what the benchmarks and
            # tests run on; the command lines only get here with an explicit --synthetic (run_infer_*.py)
            from .weights import make_state_dict

            net.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(int(getattr(self, "synthetic_seed", 0)), net.decoder_info_list,
                                                                                   net.considered_tasks).items()}, strict=True)
        else:
            saved = torch.load(ckpt, map_location="cpu")["desc"]
            if all(k.split(".")[0] == "module" for k in saved.keys()):  # data-parallel checkpoint (infer/base.py:30-44)
                saved = {".".join(k.split(".")[1:]): v for k, v in saved.items()}
            net.load_state_dict(saved, strict=True)
        self.net = net
        # the precision decision belongs to LOADING: one calibration tile on the default stream, the algorithm of the 3x3 convolutions
        # fixed and logged before the first batch (NetDesc.prepare; without a GPU the first compute call raises CerberusHipError as before)
        self.precision = None
        if torch.cuda.is_available():
            self.precision = net.prepare()
            import logging

            logging.getLogger("cerberus_amd").info("precision decision at load: %s", self.precision)
        self.run_step = lambda input_batch, output_shape: infer_step(input_batch, net, output_shape, self.model_args["considered_tasks"])

    # ---- one image, everything on the GPU ---------------------------------------------------------------------
    def infer_image(self, img, patch_input_shape, patch_output_shape, batch_size=32, postproc_list=("gland", "lumen", "nuclei", "patch-class")):
        """img: HxWx3 uint8 RGB (numpy).  Returns dict with device tensors:
        'raw': per-head stitched canvases cropped to the source image, 'inst': {Tissue: int32 label map},
        'type': {Tissue: uint8 map or None}, 'pclass': float32 map or None, 'info': per-tissue postproc info."""
        return self.infer_images([img], patch_input_shape, patch_output_shape, batch_size, postproc_list)[0]

    def infer_images(self, imgs, patch_input_shape, patch_output_shape, batch_size=32, postproc_list=("gland", "lumen", "nuclei", "patch-class")):
        """Several images through SHARED batches (the reference caches a group of files and runs one DataLoader over all their
        patches, infer/tile.py:300-420; a folder of 256 x 256 tiles would otherwise run at batch 1).  The canvases of the group are
        row blocks of one buffer per head (width = the widest canvas), so that a batch mixing patches of different images still
        scatters with one row stride; every image is then post-processed on its own window of that buffer.  A single image gives
        exactly what it gave alone (same offsets, same kernels)."""
        net = self.net
        dev = torch.device("cuda", torch.cuda.current_device())
        win, osz = int(patch_input_shape), int(patch_output_shape)
        preps, row0, wmax = [], [0], 0
        for img in imgs:
            padded, info, src_pos = _prepare_patching(img, patch_input_shape, patch_output_shape, 0)
            uniq = info[: info.shape[0] // 2]  # second half duplicates the first (see module docstring)
            hw = np.max(info[:, 1, 1], axis=0).tolist()
            preps.append((padded, uniq, src_pos, int(hw[0]), int(hw[1])))
            row0.append(row0[-1] + int(hw[0]))
            wmax = max(wmax, int(hw[1]))
        canv = OrderedDict()
        for name, hname, och, key in net._decoders:
            if hname == "INST":
                canv[key] = torch.zeros((row0[-1], wmax, 2), dtype=torch.float32, device=dev)
            elif hname == "TYPE":
                canv[key] = torch.zeros((row0[-1], wmax), dtype=torch.uint8, device=dev)
            else:
                canv[key] = torch.zeros((row0[-1], wmax), dtype=torch.float32, device=dev)
        outs = [canv[d[3]] for d in net._decoders]
        tiles_all, off_all = [], []
        for k, (padded, uniq, _, _, _) in enumerate(preps):
            pad_dev = torch.from_numpy(np.ascontiguousarray(padded)).to(dev)
            tiles_all += [pad_dev[int(i[0, 0, 0]):int(i[0, 0, 0]) + win, int(i[0, 0, 1]):int(i[0, 0, 1]) + win] for i in uniq]
            off_all += [(row0[k] + int(i[1, 0, 0])) * wmax + int(i[1, 0, 1]) for i in uniq]
        off_dev = torch.tensor(off_all, dtype=torch.int64, device=dev)
        for b0 in range(0, len(tiles_all), batch_size):
            tiles = torch.stack(tiles_all[b0:b0 + batch_size])
            net._run(tiles, osz, osz, outs, None, tile_off=off_dev[b0:b0 + batch_size], row_stride=wmax, type_is_u8=True)
        results = []
        for k, img in enumerate(imgs):
            y0, x0 = preps[k][2]
            sh, sw = img.shape[:2]
            raw = OrderedDict((key, v[row0[k] + y0:row0[k] + y0 + sh, x0:x0 + sw]) for key, v in canv.items())
            inst, types, pp_info = OrderedDict(), OrderedDict(), OrderedDict()
            pclass = None
            for tissue in postproc_list:
                tissue = tissue.capitalize()
                code = self.decoder_dict.get(tissue + "-INST") if getattr(self, "decoder_dict", None) else "IP-ERODED-CONTOUR-3"
                if tissue + "-INST" in raw:
                    if code not in POSTPROC_CODES:
                        raise NotImplementedError("post-proc code %r: only IP-ERODED-CONTOUR-* (PostProcInstErodedContourMap) is on the HIP path" % code)
                    inst[tissue], pp_info[tissue] = postproc_device(raw[tissue + "-INST"], tissue)
                    types[tissue] = raw.get(tissue + "-TYPE")
                elif tissue == "Patch-class":
                    pclass = raw.get("Patch-Class")
            if "Lumen" in inst and "Gland" in inst:
                mask_lumen_by_gland(inst["Lumen"], inst["Gland"])
            results.append({"raw": raw, "inst": inst, "type": types, "pclass": pclass, "info": pp_info})
        return results

    # ---- reference CLI behaviour --------------------------------------------------------------------------------
    def process_file_list(self, run_args):
        """Process every *.png / *.jpg of input_dir (tiles < 5000x5000) and write <tissue>_mat/<name>.mat + pclass_mat."""
        import scipy.io as sio
        from PIL import Image

        for k, v in run_args.items():
            setattr(self, k, v)
        files = []
        for root, _, fs in os.walk(self.input_dir):
            files += [os.path.join(root, f) for f in fs if f.lower().endswith((".png", ".jpg"))]
        todo = []
        for fp in sorted(files):
            base = os.path.basename(fp).split(".")[0]
            if any(not os.path.exists("%s/%s_mat/%s.mat" % (self.output_dir, t, base)) for t in self.postproc_list):
                todo.append(fp)
        assert len(todo) > 0, "Not Detected Any Files From Path"
        # several GPUs (run_infer_tile.py --gpu=0,1,...): the reference splits every batch over the devices (DataParallel, infer/base.py:46);
        # files are independent, so here rank r of w takes files r, r + w, ... of the sorted list -- no collective, no shared canvas; a tile's
        # values do not depend on what it is batched with, so the outputs are those of the one-GPU run
        rank, world = int(getattr(self, "rank", 0)), int(getattr(self, "world_size", 1))
        todo = todo[rank::world]
        if not todo:
            return
        # groups of files share batches (infer/tile.py:300-420 caches several files per DataLoader pass): a group is closed once it
        # holds 8 batches' worth of patches or 64 Mpx of padded pixels
        win, osz = int(self.patch_input_shape), int(self.patch_output_shape)

        # Host work around the GPU (decoding PNG / JPG files, .mat files, overlays) runs on small thread pools -- the reference gives both sides to worker
        # processes (nr_inference_workers / nr_post_proc_workers, infer/tile.py:300-420): 96 tiles of 1000^2 spent 1.7 s on the GPU and 8 s in the main
        # thread's decodes, savemat calls and overlay drawing, one after the other.  zlib, libjpeg, file I/O and numpy's copies release the interpreter
        # lock.  CERB_TILE_IO_THREADS (default 4; 0 = everything in the main thread, the old order).
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        n_io = max(0, int(os.environ.get("CERB_TILE_IO_THREADS", "4")))
        readers = ThreadPoolExecutor(max_workers=n_io, thread_name_prefix="cerb-tile-read") if n_io else None
        writers = ThreadPoolExecutor(max_workers=n_io, thread_name_prefix="cerb-tile-write") if n_io else None

        def load(fp):
            return np.array(Image.open(fp).convert("RGB"))

        def decoded():  # (path, pixels) in order, up to 4 x n_io files decoded ahead
            if readers is None:
                for fp in todo:
                    yield fp, load(fp)
                return
            ahead, it = deque(), iter(todo)
            for fp in it:
                ahead.append((fp, readers.submit(load, fp)))
                if len(ahead) >= 4 * n_io:
                    break
            while ahead:
                fp, fut = ahead.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    ahead.append((nxt, readers.submit(load, nxt)))
                yield fp, fut.result()

        def grouped_results():  # lazily: one group of decoded files and its canvases alive at a time
            cur, n_patch, n_px = [], 0, 0
            for idx, (fp, img) in enumerate(decoded()):
                cur.append((fp, img))
                n_patch += int(math.ceil(img.shape[0] / osz)) * int(math.ceil(img.shape[1] / osz))
                n_px += (img.shape[0] + win) * (img.shape[1] + win)
                if n_patch >= 8 * int(self.batch_size) or n_px >= (64 << 20) or idx == len(todo) - 1:
                    res_list = self.infer_images([im for _, im in cur], self.patch_input_shape, self.patch_output_shape, self.batch_size,
                                                 self.postproc_list)
                    for (f, im), r in zip(cur, res_list):
                        yield f, im, r
                    cur, n_patch, n_px = [], 0, 0

        def write_outputs(base, img, mats, info_all, pclass_np):
            for tissue, mat in mats:
                os.makedirs("%s/%s_mat/" % (self.output_dir, tissue.lower()), exist_ok=True)
                sio.savemat("%s/%s_mat/%s.mat" % (self.output_dir, tissue.lower(), base), mat)
            # overlay of the x2 nearest-upscaled source with every instance contour (infer/tile.py:251-257)
            from .viz import up2_nearest, visualize_instances_dict_orig

            os.makedirs("%s/overlay/" % self.output_dir, exist_ok=True)
            Image.fromarray(visualize_instances_dict_orig(up2_nearest(img), info_all)).save("%s/overlay/%s.jpg" % (self.output_dir, base))
            if pclass_np is not None:
                os.makedirs("%s/pclass_mat/" % self.output_dir, exist_ok=True)
                sio.savemat("%s/pclass_mat/%s.mat" % (self.output_dir, base), {"pclass": pclass_np})
            print("Done Assembling %s" % base)

        pending = deque()
        for fp, img, res in grouped_results():
            base = pathlib.Path(fp).stem
            prev_type = None

            def up2(t):  # cv2.resize(fx=2, fy=2, INTER_NEAREST) of an integer map, on the device
                return t.repeat_interleave(2, dim=0).repeat_interleave(2, dim=1).contiguous()

            info_all, mats = {}, []
            for tissue, lab in res["inst"].items():
                lab_np = lab.cpu().numpy()
                tmap = res["type"].get(tissue)
                tmap_np = tmap.cpu().numpy() if tmap is not None else None
                if tissue != "Lumen" and tmap is not None:
                    prev_type = up2(tmap)
                # instance table on the GPU; the reference re-uses the previous tissue's type map for Lumen (infer/tile.py:196-202)
                info = get_inst_info_dict(up2(lab), prev_type)
                info_all[tissue] = info
                mat = {"inst_map": lab_np.astype(np.float64) if tissue != "Nuclei" else lab_np,
                       "type": [d.get("type", -1) for d in info.values()], "id": list(info.keys())}
                if tmap_np is not None:
                    mat["type_map"] = tmap_np.astype(np.float32)
                mats.append((tissue, mat))
            pclass_np = res["pclass"].cpu().numpy() if res["pclass"] is not None else None
            if writers is None:
                write_outputs(base, img, mats, info_all, pclass_np)
            else:
                pending.append(writers.submit(write_outputs, base, img, mats, info_all, pclass_np))
                while len(pending) > 4 * n_io:  # bounded: a tile's maps stay in RAM until its files are written
                    pending.popleft().result()
        while pending:
            pending.popleft().result()
        for pool in (readers, writers):
            if pool is not None:
                pool.shutdown(wait=True)
