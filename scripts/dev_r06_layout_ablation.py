"""Round 6, VERDICT r5 items 6 + 7: what the tile-planar kernel (conv_wino4p) would buy where the NHWC Winograd kernels still run, measured on the launches
themselves (per-launch HIP events, cerb_net_profile_*), before any plumbing is written.
  A. inference, batch 32 x 256^2 (the headline's inner loop): the encoder's 128^2 stage (conv_wino4, 6 launches) against the decoder's 128^2 level
     (conv_wino4p<0>: the same 64 -> 64 shape per group) -> best case of moving layer1 to planar items; the 64^2 / 32^2 decoder levels (conv_wino4b).
  B. the training geometry (batch 16 x 448^2) through the inference forward with the last two decoder levels planar (default) and NHWC
     (set_planar(0)): the forward time the training step's 448^2 / 224^2 levels would save on conv_wino4p, and by the same ratio its data gradients."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cerberus_amd.net_desc import create_model  # noqa: E402
from cerberus_amd.weights import default_model_kwargs, make_state_dict  # noqa: E402


def records(m, tiles, out, reps=3):
    for _ in range(2):
        m.infer_tiles(tiles, out)
    torch.cuda.synchronize()
    acc = None
    for _ in range(reps):
        m.profile(True)
        m.infer_tiles(tiles, out)
        torch.cuda.synchronize()
        r = m.profile_records()
        m.profile(False)
        acc = [(a, b, c, d) for a, b, c, d in r] if acc is None else [(a, b, c, d0 + d) for (a, b, c, d0), (_, _, _, d) in zip(acc, r)]
    return [(a, b, c, d / reps) for a, b, c, d in acc]


def main():
    m = create_model(**default_model_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(0).items()}, strict=True)
    m.prepare()
    t = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
    rec = records(m, t, 256)
    tot = sum(r[3] for r in rec)
    print("# A. inference batch 32 x 256^2: %.3f ms per step (sum of launches)" % tot)
    enc = [r for r in rec if r[0].startswith("backbone.layer1") and r[1].startswith("conv_wino4<")]
    dec128 = [r for r in rec if r[1].startswith("conv_wino4p") and "half-res" in r[1]]
    w4b_dec = [r for r in rec if r[0].startswith("dec.") and r[1].startswith("conv_wino4b")]
    for r in enc + dec128 + w4b_dec:
        print("%-30s %-44s %8.3f ms %7.1f TFLOP/s algorithmic" % (r[0], r[1], r[3], r[2] / r[3] / 1e9))
    e_ms, e_fl = sum(r[3] for r in enc), sum(r[2] for r in enc)
    d_ms, d_fl = sum(r[3] for r in dec128), sum(r[2] for r in dec128)
    best = e_fl / (d_fl / d_ms)
    print("encoder 128^2 stage: %d launches %.3f ms at %.1f TFLOP/s; the decoder's 128^2 level on conv_wino4p runs %.1f TFLOP/s -> best case %.3f ms (%.3f ms saved = step %.3f ms)"
          % (len(enc), e_ms, e_fl / e_ms / 1e9, d_fl / d_ms / 1e9, best, e_ms - best, tot - (e_ms - best)))
    b_ms, b_fl = sum(r[3] for r in w4b_dec), sum(r[2] for r in w4b_dec)
    bestb = b_fl / (d_fl / d_ms)
    print("decoder 64^2 / 32^2 levels on conv_wino4b: %d launches %.3f ms at %.1f TFLOP/s -> at the planar kernel's rate %.3f ms (%.3f saved)" % (len(w4b_dec), b_ms, b_fl / b_ms / 1e9, bestb, b_ms - bestb))
    print("gate (VERDICT r5 item 7): step <= 13.7 ms; best case with BOTH moved and no conversion / dual-store cost: %.3f ms" % (tot - (e_ms - best) - max(0.0, b_ms - bestb)))
    # B. training geometry
    del t
    t = torch.randint(0, 256, (16, 448, 448, 3), dtype=torch.uint8, device="cuda")
    res = {}
    for planar in (1, 0):
        m.set_planar(planar)
        rec = records(m, t, 448)
        lv = [r for r in rec if r[0].startswith(("dec.3.", "dec.2.")) and r[1].startswith("conv_wino4")]
        res[planar] = lv
        print("# B. batch 16 x 448^2, planar=%d: %.3f ms per forward" % (planar, sum(r[3] for r in rec)))
        for r in lv:
            print("%-30s %-44s %8.3f ms %7.1f TFLOP/s algorithmic" % (r[0], r[1], r[3], r[2] / r[3] / 1e9))
    m.set_planar(1)
    p_ms, n_ms = sum(r[3] for r in res[1]), sum(r[3] for r in res[0])
    print("448^2 + 224^2 decoder levels, 4 launches: NHWC conv_wino4 %.3f ms, planar conv_wino4p %.3f ms -> %.3f ms per forward; the training step's forward AND data gradient "
          "run these shapes (STATS instantiations): about %.1f ms of a 97 ms step if EVERY tape consumer of the two levels read the planar layout" % (n_ms, p_ms, n_ms - p_ms, 2.0 * (n_ms - p_ms)))


if __name__ == "__main__":
    main()
