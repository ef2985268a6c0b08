"""Seeded structured probability maps (blobs with inner / contour channels, the shape of a trained Cerberus head's output): synthetic
INPUT for the post-processing benchmarks and tests -- `bench.py`'s postproc leg, scripts/, and (re-exported by the test
infrastructure's `synth` module) the golden-vector generators.  numpy only; no checker lives here."""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.clip(x, -80.0, 80.0)))  # (clipped: exp(88.7) overflows float32 with a RuntimeWarning; sigmoid(+-80) is 0 / 1 in fp32 anyway)


def blob_maps(H, W, seed, n_blobs, rmin, rmax, sharp=1.5, noise=0.0, holes=0.0, rim=2.0, border_bias=False):
    """Returns (H,W,2) float32: ch0 inner prob, ch1 contour prob."""
    rs = np.random.RandomState(seed)
    d = np.full((H, W), -1e9, np.float32)  # signed distance-ish field: max over blobs of (r - dist)
    for i in range(n_blobs):
        r = rs.uniform(rmin, rmax)
        if border_bias and i % 3 == 0:
            cy, cx = (rs.choice([0.0, H - 1.0]), rs.uniform(0, W)) if rs.rand() < 0.5 else (rs.uniform(0, H), rs.choice([0.0, W - 1.0]))
        else:
            cy, cx = rs.uniform(0, H), rs.uniform(0, W)
        ax = rs.uniform(0.7, 1.3)
        th = rs.uniform(0, np.pi)
        hole = holes > 0 and rs.rand() < holes
        # each blob only touches a window around itself (cost O(blob area), not O(H*W))
        ext = int(np.ceil(r * 1.45 + 14.0))
        y0, y1 = max(0, int(cy) - ext), min(H, int(cy) + ext + 1)
        x0, x1 = max(0, int(cx) - ext), min(W, int(cx) + ext + 1)
        if y0 >= y1 or x0 >= x1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        dy, dx = yy - np.float32(cy), xx - np.float32(cx)
        u = (np.cos(th) * dx + np.sin(th) * dy) * ax
        v = (-np.sin(th) * dx + np.cos(th) * dy) / ax
        dist = np.sqrt(u * u + v * v)
        f = r - dist
        if hole:
            f = np.minimum(f, dist - 0.35 * r)  # annulus: a hole in the middle
        d[y0:y1, x0:x1] = np.maximum(d[y0:y1, x0:x1], f.astype(np.float32))
    d = np.maximum(d, np.float32(-40.0))
    inner = _sigmoid(sharp * (d - rim))
    cnt = _sigmoid(sharp * (rim - np.abs(d - 0.5 * rim))) * 0.95
    if noise > 0:
        inner = inner + rs.normal(0, noise, inner.shape)
        cnt = cnt + rs.normal(0, noise, cnt.shape)
    out = np.stack([np.clip(inner, 0, 1), np.clip(cnt, 0, 1)], axis=-1)
    return out.astype(np.float32)


def nuclei_maps(H, W, seed, density_per_mpx=600.0, **kw):
    n = max(1, int(round(density_per_mpx * H * W / 1e6)))
    return blob_maps(H, W, seed, n, 4.0, 9.0, **kw)


def gland_maps(H, W, seed, n=None, **kw):
    n = n if n is not None else max(1, int(round(12.0 * H * W / 1e6)))
    kw.setdefault("rim", 4.0)
    kw.setdefault("sharp", 1.0)
    return blob_maps(H, W, seed, n, 25.0, 150.0, **kw)


def softmax_nuclei_maps(H, W, seed, density_per_mpx=600.0, gain=4.0, logit_noise=0.5, rim=2.0, rmin=4.0, rmax=9.0):
    """(H,W,2) float32 inner / contour maps computed the way a trained head produces them: a float32 softmax over three logits
    (background, inner, contour).  With gain >= ~3 the nucleus cores saturate to EXACTLY 1.0f (logit gap > 17) while rim pixels
    keep generic, pairwise distinct floats -- the tie pattern of a confident network (SURVEY par.7 "Hard parts")."""
    n = max(1, int(round(density_per_mpx * H * W / 1e6)))
    rs = np.random.RandomState(seed)
    d = np.full((H, W), -40.0, np.float32)
    for _ in range(n):
        r = rs.uniform(rmin, rmax)
        cy, cx = rs.uniform(0, H), rs.uniform(0, W)
        ax, th = rs.uniform(0.7, 1.3), rs.uniform(0, np.pi)
        ext = int(np.ceil(r * 1.45 + 14.0))
        y0, y1 = max(0, int(cy) - ext), min(H, int(cy) + ext + 1)
        x0, x1 = max(0, int(cx) - ext), min(W, int(cx) + ext + 1)
        if y0 >= y1 or x0 >= x1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        dy, dx = yy - np.float32(cy), xx - np.float32(cx)
        u = (np.cos(th) * dx + np.sin(th) * dy) * ax
        v = (-np.sin(th) * dx + np.cos(th) * dy) / ax
        f = (r - np.sqrt(u * u + v * v)).astype(np.float32)
        # touching nuclei keep a contour ridge between them: union of blobs by max, ridge where the two largest are close
        d[y0:y1, x0:x1] = np.maximum(d[y0:y1, x0:x1], f)
    g = np.float32(gain)
    z_in = g * (d - np.float32(rim))
    z_ct = g * (np.float32(rim) - np.abs(d - np.float32(0.5 * rim))) - np.float32(1.0)
    z_bg = -g * d
    if logit_noise > 0:
        z_in = z_in + rs.normal(0, logit_noise, d.shape).astype(np.float32)
        z_ct = z_ct + rs.normal(0, logit_noise, d.shape).astype(np.float32)
    z = np.stack([z_bg, z_in, z_ct], axis=-1).astype(np.float32)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z, dtype=np.float32)
    p = e / e.sum(axis=-1, keepdims=True, dtype=np.float32)
    return np.ascontiguousarray(p[..., 1:3].astype(np.float32))
