"""Dev experiment (CPU, imports the oracle): how far do the probability maps move when every 3x3 stride-1 convolution is evaluated as
Winograd F(4x4,3x3) / F(2x2,3x3) in fp32, against an fp64 direct evaluation of the same network?  Decides whether F(4x4) fits the 1e-4 bar."""
import sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, "/root/repo")
from oracle import net_ref  # noqa: E402
from cerberus_amd import weights, net_desc  # noqa: E402


def mats(kind, pts=None):
    if kind == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
        return BT, G, AT
    # general Toom-Cook construction for F(4,3) from 5 finite points + infinity
    p = np.array(pts, np.float64)
    n = 6
    # A^T: 4 x 6 rows i: p^i, last col = [0,0,0,1]
    AT = np.zeros((4, n))
    for i in range(4):
        AT[i, :5] = p ** i
    AT[3, 5] = 1
    # G: 6 x 3: rows: [1, p, p^2] / prod_{k != j}(p_j - p_k); last row [0,0,1]
    G = np.zeros((n, 3))
    for j in range(5):
        d = np.prod([p[j] - p[k] for k in range(5) if k != j])
        G[j] = np.array([1, p[j], p[j] ** 2]) / d
    G[5] = [0, 0, 1]
    # B^T from the polynomial identity: solve so that Y = A^T[(G g) * (B^T d)] equals correlation for all g, d
    # build linear system numerically: unknown B^T (6x6).  For basis g_e, d_f the correct output y(i) = sum_k g[k] d[i+k].
    # A^T diag(G g) B^T d = T(g) d where T(g)[i, i+k] = g[k].  For each basis g (3): A^T diag(G[:,e]) B^T = T_e  -> stack and solve least squares
    rows, rhs = [], []
    for e in range(3):
        M = AT * G[:, e][None, :]  # 4 x 6
        T = np.zeros((4, 6))
        for i in range(4):
            T[i, i + e] = 1
        rows.append(M)
        rhs.append(T)
    Mst = np.concatenate(rows, 0)  # 12 x 6
    Tst = np.concatenate(rhs, 0)  # 12 x 6
    BT, res, rk, _ = np.linalg.lstsq(Mst, Tst, rcond=None)
    assert np.abs(Mst @ BT - Tst).max() < 1e-9, np.abs(Mst @ BT - Tst).max()
    return BT, G, AT


def wino_conv(x, w, b, kind, pts, acc_dtype=torch.float32):
    BT, G, AT = mats(kind, pts)
    m = 4 if kind == 4 else 2
    a = m + 2
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.from_numpy(np.einsum("ij,kcjl,ml->imkc", G, w.double().numpy(), G)).to(acc_dtype)  # [a, a, K, C]
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = F.pad(x, (1, Wp - W + 1, 1, Hp - H + 1))
    pt = xp.unfold(2, a, m).unfold(3, a, m)  # N, C, th, tw, a, a
    BTt = torch.from_numpy(BT).to(acc_dtype)
    ATt = torch.from_numpy(AT).to(acc_dtype)
    # sequential fp32 evaluation of the 1-D passes (as a kernel would: a chain of adds / fmas)
    V = torch.einsum("ij,ncyxjl->ncyxil", BTt, pt.to(acc_dtype))
    V = torch.einsum("ncyxil,ml->ncyxim", V, BTt)
    Mm = torch.einsum("imkc,ncyxim->nkyxim", U, V)
    Y = torch.einsum("ij,nkyxjl->nkyxil", ATt, Mm)
    Y = torch.einsum("nkyxil,ml->nkyxim", Y, ATt)
    th, tw = Y.shape[2], Y.shape[3]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, th * m, tw * m)[:, :, :H, :W]
    if b is not None:
        Y = Y + b.view(1, -1, 1, 1)
    return Y.to(x.dtype)


def run(mode, sd, x, dk, tasks, pts=None):
    orig = F.conv2d

    def conv(inp, w, b=None, stride=1, padding=0, *a, **k):
        if mode in (2, 4) and w.shape[2] == 3 and stride == 1 and padding == 1 and inp.shape[1] >= 32:
            return wino_conv(inp, w, b, mode, pts)
        return orig(inp, w, b, stride, padding, *a, **k)

    net_ref.F.conv2d = conv
    try:
        return net_ref.net_forward(sd, x, dk, tasks)
    finally:
        net_ref.F.conv2d = orig


def main():
    torch.manual_seed(0)
    torch.set_num_threads(32)
    dk = weights.DEFAULT_DECODER_KWARGS
    tasks = list(dk.keys())
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sd32 = weights.make_state_dict(seed=seed) if (len(sys.argv) <= 3 or sys.argv[3] == "make") else weights.reference_init_state_dict(generator=torch.Generator().manual_seed(seed))
    sd32 = {k: torch.from_numpy(np.asarray(v)) if not isinstance(v, torch.Tensor) else v for k, v in sd32.items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.integers(0, 256, (2, 3, S, S)).astype(np.float32))
    ref = run(0, sd64, x.double(), dk, tasks)
    outs = {"direct32": run(0, sd32, x, dk, tasks), "F2x2": run(2, sd32, x, dk, tasks)}
    for name, pts in (("F4x4 (0,1,-1,2,-2)", (0, 1, -1, 2, -2)), ("F4x4 (0,1,-1,.5,-.5)", (0, 1, -1, .5, -.5)),
                      ("F4x4 (0,.5,-.5,1.5,-1.5)", (0, .5, -.5, 1.5, -1.5)), ("F4x4 (0,1,-1,.5,-2)", (0, 1, -1, .5, -2))):
        outs[name] = run(4, sd32, x, dk, tasks, pts)
    for name, o in outs.items():
        worst_logit, worst_prob = 0.0, 0.0
        for k in ref:
            r = ref[k]
            d = (o[k].double() - r).abs().max().item()
            worst_logit = max(worst_logit, d / max(r.abs().max().item(), 1e-30))
            if r.dim() == 4 and r.shape[1] > 1:
                pr = torch.softmax(r, 1)
                po = torch.softmax(o[k].double(), 1)
                worst_prob = max(worst_prob, (pr - po).abs().max().item())
        print("%-28s max |dlogit|/max|logit| = %.3e   max |dprob| = %.3e" % (name, worst_logit, worst_prob))


if __name__ == "__main__":
    main()
