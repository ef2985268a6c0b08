// head_train.hip -- the output heads of the TRAINING step (BASELINE.json configs[4]; models/utils/net_layers.py:31-38 in train mode,
// models/run_desc.py:79-170): per head  prev[rows][64] -> 1x1 64->96 (+bias) -> BatchNorm(batch statistics) -> ReLU -> 1x1 96->out (+bias).
//
// Round 4 ran this chain as separate passes (pointwise, BN apply, pointwise; backward: small pointwise backward, two BN-backward passes, pointwise
// weight gradient, pointwise data gradient): 18.5 GB of HBM traffic per head and step at batch 16 x 448^2 -- 26 ms of a 163 ms step for 3.5 % of the FLOPs.
// Here the 96-channel hidden map is the ONLY large tensor that is stored (once, by the first pointwise layer, which also leaves the BatchNorm
// statistics partials), and it is read three times:
//   forward   cerb_launch_head_fwd2   hid -> BN -> ReLU -> 96->out logits                                   (reads hid)
//   backward  cerb_launch_head_bwd1   hid, dlogits -> dW2, db2 and the BatchNorm-backward sums (sum dz, sum dz xhat)   (reads hid)
//             cerb_launch_head_bwd2   hid, dlogits, prev -> dhid in registers / LDS -> dprev = dhid W1, dW1 += dhid^T prev, db1   (reads hid + prev, writes dprev)
// 7.4 GB per head.  The normalised / rectified hidden map and both of its gradients never exist in memory; every kernel recomputes
// z = bn_out(hid) with the ONE expression the other BatchNorm kernels use (identical ReLU masks by construction).
// HBM-bound by design: the matrix work (dprev, dW1: 12 v_mfma_f32_16x16x4_f32 per row) is sized to hide under the hidden map's stream.
#include "cerb_common.h"
#include <algorithm>

namespace {
constexpr int HC = 96;    // hidden channels (get_classification_head: ConvBlock(64, [96]))
constexpr int HQ = 24;    // channel quads of a hidden row
constexpr int PC = 64;    // decoder channels

__device__ __forceinline__ float bn_out(float y, float m, float sc, float be) { return __fmaf_rn(y - m, sc, be); }  // sc = rstd * gamma (train_kernels.hip: bn_out)

// ---------------------------------------------------------------------------------------------------------------------------------
// forward 1:  hid[r][c] = b1[c] + sum_ci prev[r][ci] W1[c][ci]  (+ the BatchNorm statistics partials of hid, as cerb_launch_pw_mfma leaves them).
// One wave = 16 rows per step: A = the rows (lane = (row l & 15, k-slot l >> 4) reads prev[row][16 S + 4 kq .. + 3]: 64-byte segments, four 16-byte loads
// per lane), B = W1 held in registers (96 values per lane), 96 v_mfma_f32_16x16x4_f32 per 16 rows; the 16 x 96 result tile -- 6 KiB CONTIGUOUS in hid --
// goes through LDS and leaves as six 1-KiB stores.  (Round 4's pw_mfma_kernel<64, 96>: row-per-lane 16-byte loads 256 bytes apart and 128-byte store
// pieces, 3.5 TB/s on the largest stream of the heads.)
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int F1S = 100;  // LDS row stride of the output tile (floats)
// in_mean != nullptr: `prev` is the RAW output of the decoder's last convolution and its BatchNorm + ReLU are applied here, on the load
// (z = relu(bn_out(y)) with the one expression every BatchNorm kernel uses): the normalised copy of the largest decoder tensor is never written.
__global__ __launch_bounds__(256) void head_fwd1_kernel(const float* __restrict__ prev, const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ hid,
                                                        long long rows, double* __restrict__ bn_part, const float* __restrict__ in_mean, const float* __restrict__ in_rstd,
                                                        const float* __restrict__ in_gamma, const float* __restrict__ in_beta) {
    __shared__ __attribute__((aligned(16))) float tile[4][16 * F1S];
    __shared__ __attribute__((aligned(16))) float wl[6 * 4 * 64 * 4];  // [nt][S][lane][e] = W1[16 nt + (lane & 15)][16 S + 4 (lane >> 4) + e]
    __shared__ double red[4][HC][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    for (int i = threadIdx.x; i < 6 * 4 * 64 * 4; i += 256) {
        const int e = i & 3, l = (i >> 2) & 63, S = (i >> 8) & 3, nt = i >> 10;
        wl[i] = w1[(16 * nt + (l & 15)) * PC + 16 * S + 4 * (l >> 4) + e];
    }
    const bool in_bn = in_mean != nullptr;
    f32x4 pm[4], ps[4], pb[4];
#pragma unroll
    for (int S = 0; S < 4; ++S) {
        const int c = 16 * S + 4 * kq;
        pm[S] = ps[S] = pb[S] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (in_bn) {
            pm[S] = *reinterpret_cast<const f32x4*>(in_mean + c);
            ps[S] = *reinterpret_cast<const f32x4*>(in_rstd + c) * *reinterpret_cast<const f32x4*>(in_gamma + c);
            pb[S] = *reinterpret_cast<const f32x4*>(in_beta + c);
        }
    }
    float bias[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) bias[nt] = b1[16 * nt + r];
    double bs[6], bq[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) bs[nt] = bq[nt] = 0.0;
    float* tl = tile[wave];
    __syncthreads();
    const long long ntiles = rows >> 4, stride = (long long)gridDim.x * 4;
    long long t = (long long)blockIdx.x * 4 + wave;
    f32x4 v[4], nx[4];
    if (t < ntiles) {
#pragma unroll
        for (int S = 0; S < 4; ++S) v[S] = *reinterpret_cast<const f32x4*>(prev + (t * 16 + r) * PC + 16 * S + 4 * kq);
    }
    for (; t < ntiles; t += stride) {
        const bool more = t + stride < ntiles;
        if (more) {
#pragma unroll
            for (int S = 0; S < 4; ++S) nx[S] = *reinterpret_cast<const f32x4*>(prev + ((t + stride) * 16 + r) * PC + 16 * S + 4 * kq);
        }
        if (in_bn) {
#pragma unroll
            for (int S = 0; S < 4; ++S)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[S][e] = fmaxf(bn_out(v[S][e], pm[S][e], ps[S][e], pb[S][e]), 0.f);
        }
        asm volatile("" ::: "memory");  // (keeps the 24 W1 reads below inside the loop: hoisted, they are 96 registers -- the reason W1 sits in LDS)
        f32x4 acc[6];
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + ((nt * 4 + S) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[S][e], b[e], acc[nt], 0, 0, 0);
            }
        // D[row 4 kq + e][channel 16 nt + r] -> LDS tile; statistics of what is stored
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float val = acc[nt][e] + bias[nt];
                tl[(4 * kq + e) * F1S + 16 * nt + r] = val;
                ts += val;
                tq = fmaf(val, val, tq);
            }
            bs[nt] += (double)ts;
            bq[nt] += (double)tq;
        }
        // (a wave's LDS accesses complete in order: no barrier between its own writes and reads)
        float* dst = hid + t * 16 * HC;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int f = lane + 64 * k, row = f / HQ, c4 = f % HQ;
            *reinterpret_cast<f32x4*>(dst + 4 * f) = *reinterpret_cast<const f32x4*>(tl + row * F1S + 4 * c4);
        }
        if (more) {
#pragma unroll
            for (int S = 0; S < 4; ++S) v[S] = nx[S];
        }
    }
    if (bn_part) {  // one partial row per workgroup: [gridDim.x][96][(sum, sum of squares)]
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            double s = bs[nt], q = bq[nt];
            s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (kq == 0) {
                red[wave][16 * nt + r][0] = s;
                red[wave][16 * nt + r][1] = q;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < HC; c += 256) {
            double* o = bn_part + ((long long)blockIdx.x * HC + c) * 2;
            o[0] = ((red[0][c][0] + red[1][c][0]) + red[2][c][0]) + red[3][c][0];
            o[1] = ((red[0][c][1] + red[1][c][1]) + red[2][c][1]) + red[3][c][1];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward 2:  logits[r][o] = b2[o] + sum_c relu(bn(hid[r][c])) W2[o][c].   One wave = 16 rows per step on v_mfma_f32_16x16x4_f32:
// A[m = row][k] = the lane's own rectified values (lane = (row l & 15, k-slot kq = l >> 4) reads hid[row][16 S + 4 kq .. + 3]: 64-byte segments),
// B[k][n = output] = W2 (zero for n >= out), 24 instructions per 16 rows.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd2_kernel(const float* __restrict__ hid, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ logits, long long rows, int out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    f32x4 pm[6], ps[6], pb[6], wb[6];
#pragma unroll
    for (int S = 0; S < 6; ++S) {
        const int c = 16 * S + 4 * kq;
        pm[S] = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c), ga = *reinterpret_cast<const f32x4*>(gamma + c);
        ps[S] = rs * ga;
        pb[S] = *reinterpret_cast<const f32x4*>(beta + c);
        wb[S] = r < out ? *reinterpret_cast<const f32x4*>(w2 + (long long)r * HC + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bias = r < out ? b2[r] : 0.f;
    const long long ntiles = rows >> 4, stride = (long long)gridDim.x * 4;
    long long t = (long long)blockIdx.x * 4 + wave;
    f32x4 v[6], nx[6];
    if (t < ntiles) {
#pragma unroll
        for (int S = 0; S < 6; ++S) v[S] = *reinterpret_cast<const f32x4*>(hid + (t * 16 + r) * HC + 16 * S + 4 * kq);
    }
    for (; t < ntiles; t += stride) {
        const bool more = t + stride < ntiles;
        if (more) {
#pragma unroll
            for (int S = 0; S < 6; ++S) nx[S] = *reinterpret_cast<const f32x4*>(hid + ((t + stride) * 16 + r) * HC + 16 * S + 4 * kq);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < 6; ++S)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaxf(bn_out(v[S][e], pm[S][e], ps[S][e], pb[S][e]), 0.f);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z, wb[S][e], acc, 0, 0, 0);
            }
        if (r < out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) logits[(t * 16 + 4 * kq + e) * out + r] = acc[e] + bias;  // D[row 4 kq + e][output l & 15]
        }
        if (more) {
#pragma unroll
            for (int S = 0; S < 6; ++S) v[S] = nx[S];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward 1:  per channel c (thread = one channel quad, fixed; 8 rows per pass of a workgroup: wave w, half-wave s -> row 2 w + s, lanes 24..31 idle)
//   hz = relu(bn(hid)), dhz[c] = sum_o dl[o] W2[o][c], dz = dhz where hz > 0
//   dW2[o][c] += dl[o] hz[c];  sum dz;  sum dz xhat  (= dbeta, dgamma);  db2[o] += dl[o]
// Per-workgroup partials (a slab of `rows_per_block` rows), added in slab order by slab_sum / bn_bwd_finalize: reproducible.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(256) void head_bwd1_kernel(const float* __restrict__ hid, const float* __restrict__ dlog, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ w2, float* __restrict__ part_w, float* __restrict__ part_b,
                                                        double* __restrict__ part_bn, long long rows, long long rows_per_block) {
    __shared__ float red[8][HC][OUT + 2];
    __shared__ float redb[8][OUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 5, q = lane & 31, ro = 2 * wave + slot;
    const bool live = q < HQ;
    const int c0 = live ? 4 * q : 0;
    const f32x4 pm = *reinterpret_cast<const f32x4*>(mean + c0), prs = *reinterpret_cast<const f32x4*>(rstd + c0);
    const f32x4 ps = prs * *reinterpret_cast<const f32x4*>(gamma + c0), pb = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 wc[OUT], aw[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        wc[o] = *reinterpret_cast<const f32x4*>(w2 + o * HC + c0);
        aw[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    float ab[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) ab[o] = 0.f;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    auto one = [&](const f32x4& y, const float* dl) {
        f32x4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OUT; ++o) dh += dl[o] * wc[o];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = bn_out(y[e], pm[e], ps[e], pb[e]);
            const bool on = z > 0.f;
            const float hz = on ? z : 0.f, dz = on ? dh[e] : 0.f;
            s1[e] += dz;
            s2[e] = fmaf(dz, (y[e] - pm[e]) * prs[e], s2[e]);
#pragma unroll
            for (int o = 0; o < OUT; ++o) aw[o][e] = fmaf(dl[o], hz, aw[o][e]);
        }
        if (q == 0) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) ab[o] += dl[o];
        }
    };
    if (live) {
        long long r = r0 + ro;
        for (; r + 24 < r1; r += 32) {  // four rows in flight per thread
            f32x4 y[4];
            float dl[4][OUT];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = *reinterpret_cast<const f32x4*>(hid + (r + 8 * k) * HC + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int o = 0; o < OUT; ++o) dl[k][o] = dlog[(r + 8 * k) * OUT + o];
#pragma unroll
            for (int k = 0; k < 4; ++k) one(y[k], dl[k]);
        }
        for (; r < r1; r += 8) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(hid + r * HC + c0);
            float dl[OUT];
#pragma unroll
            for (int o = 0; o < OUT; ++o) dl[o] = dlog[r * OUT + o];
            one(y, dl);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) red[ro][c0 + e][o] = aw[o][e];
            red[ro][c0 + e][OUT] = s1[e];
            red[ro][c0 + e][OUT + 1] = s2[e];
        }
        if (q == 0) {
#pragma unroll
            for (int o = 0; o < OUT; ++o) redb[ro][o] = ab[o];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HC * (OUT + 2); i += 256) {  // the eight row slots in slot order
        const int c = i / (OUT + 2), o = i % (OUT + 2);
        if (o < OUT) {
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += red[k][c][o];
            part_w[((long long)blockIdx.x * OUT + o) * HC + c] = s;
        } else {
            double s = 0.0;
            for (int k = 0; k < 8; ++k) s += (double)red[k][c][o];
            part_bn[((long long)blockIdx.x * HC + c) * 2 + (o - OUT)] = s;  // bn_bwd_finalize_kernel's layout: [block][C][(sum dz, sum dz xhat)]
        }
    }
    if (threadIdx.x < OUT) {
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += redb[k][threadIdx.x];
        part_b[(long long)blockIdx.x * OUT + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward 2:  a workgroup (4 waves) walks 64-row tiles:
//   stage   dlogits (64 x OUT) and prev (64 x 64) in LDS, coalesced
//   E       thread = fixed channel quad (as in backward 1), 8 rows per pass: dhid = gamma rstd (dz - dbeta / M - xhat dgamma / M) -> LDS tile D[64][96];  db1 += dhid
//   M       wave w:  dprev[rows 16 w ..][64] = D W1          (A = D rows from LDS as b128, B = W1 from LDS, 4 x 24 instructions)
//                    dW1[c][16 w ..] += D^T prev           (A = D columns, B = prev columns from LDS, 6 x 16 instructions, 24 accumulators kept over the tiles)
// LDS 68 KB per workgroup -> two workgroups per CU: one's element-wise phase overlaps the other's matrix phase.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DS = 100, PS_ = 68;  // row strides (floats) of the LDS tiles: 100 = 25 16-byte chunks (odd: b128 row reads conflict-free), 68 likewise
struct Bwd2Params {
    const float *hid, *dlog, *prev, *mean, *rstd, *gamma, *dgamma, *dbeta, *beta, *w1, *w2;
    const float *in_mean, *in_rstd, *in_gamma, *in_beta;  // optional: `prev` is the raw input of the BatchNorm + ReLU in front of the head (applied on the LDS reads)
    double* in_part;  // optional, with in_*: [gridDim.x][64][2] -- that BatchNorm's backward sums (sum dz, sum dz xhat over this workgroup's rows; dz = dprev masked by its ReLU)
    float *dprev, *part_w1, *part_b1;
    long long rows;
    float inv_m;   // 1 / rows; 0 for an eval-mode BatchNorm (no batch-statistics terms)
    int assign;    // dprev holds nothing yet: assign instead of accumulate
};
template <int OUT>
__global__ __launch_bounds__(256, 2) void head_bwd2_kernel(Bwd2Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Dl = lds;                  // [64][DS]
    float* Pl = Dl + 64 * DS;         // [64][PS_]
    float* Wl = Pl + 64 * PS_;        // [S 6][nt 4][lane 64][e 4] = W1[16 S + 4 (lane >> 4) + e][16 nt + (lane & 15)]
    float* DLl = Wl + HC * PC;        // [64][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = lane >> 5, q = lane & 31, ro = 2 * wave + slot;
    const int mr = lane & 15, kq = lane >> 4;
    const bool live = q < HQ;
    const int c0 = live ? 4 * q : 0;
    for (int i = tid; i < HC * PC; i += 256) {
        const int e = i & 3, l = (i >> 2) & 63, nt = (i >> 8) & 3, S = i >> 10;
        Wl[i] = p.w1[(16 * S + 4 * (l >> 4) + e) * PC + 16 * nt + (l & 15)];
    }
    const f32x4 pm = *reinterpret_cast<const f32x4*>(p.mean + c0), prs = *reinterpret_cast<const f32x4*>(p.rstd + c0);
    const f32x4 pga = *reinterpret_cast<const f32x4*>(p.gamma + c0);
    const f32x4 ps = prs * pga, pb = *reinterpret_cast<const f32x4*>(p.beta + c0);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(p.dbeta + c0) * p.inv_m, k2 = *reinterpret_cast<const f32x4*>(p.dgamma + c0) * p.inv_m;
    f32x4 wc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) wc[o] = *reinterpret_cast<const f32x4*>(p.w2 + o * HC + c0);
    f32x4 sb = {0.f, 0.f, 0.f, 0.f};
    const bool in_bn = p.in_mean != nullptr;
    // the channels this lane meets `prev` at: 16 wave + mr as the B operand of the weight gradient, 16 nt + mr in the data gradient's output tile
    float wm = 0.f, wsc = 0.f, wb_ = 0.f;
    float em[4], es[4], eb[4], er[4], s1[4], s2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) em[nt] = es[nt] = eb[nt] = er[nt] = s1[nt] = s2[nt] = 0.f;
    if (in_bn) {
        const int cw = 16 * wave + mr;
        wm = p.in_mean[cw]; wsc = p.in_rstd[cw] * p.in_gamma[cw]; wb_ = p.in_beta[cw];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int ce = 16 * nt + mr;
            em[nt] = p.in_mean[ce]; er[nt] = p.in_rstd[ce]; es[nt] = er[nt] * p.in_gamma[ce]; eb[nt] = p.in_beta[ce];
        }
    }
    f32x4 accw[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) accw[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long ntiles = p.rows >> 6;
    f32x4 hy[8], pv[4];
    float dlv[2];
    auto fetch = [&](long long t) {
        const long long rb = t * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) hy[k] = live ? *reinterpret_cast<const f32x4*>(p.hid + (rb + 8 * k + ro) * HC + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const f32x4*>(p.prev + rb * PC + 4 * (tid + 256 * k));
#pragma unroll
        for (int k = 0; k < 2; ++k) dlv[k] = (tid + 256 * k < 64 * OUT) ? p.dlog[rb * OUT + tid + 256 * k] : 0.f;
    };
    long long t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads();  // the previous tile's matrix phase has read D / P (first pass: W1 is staged)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k, row = i >> 4, c4 = i & 15;
            *reinterpret_cast<f32x4*>(Pl + row * PS_ + 4 * c4) = pv[k];  // (raw: with in_bn the readers below normalise, and the epilogue needs the raw values)
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            if (i < 64 * OUT) DLl[(i / OUT) * 8 + (i % OUT)] = dlv[k];
        }
        __syncthreads();
        // ---- E: dhid of this thread's channel quad for its 8 rows ------------------------------------------------------------------------
        if (live) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 8 * k + ro;
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(DLl + row * 8), d1 = *reinterpret_cast<const f32x4*>(DLl + row * 8 + 4);
                const float dl[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
                f32x4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < OUT; ++o) dh += dl[o] * wc[o];
                f32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = hy[k][e];
                    const float z = bn_out(y, pm[e], ps[e], pb[e]);
                    const float dz = z > 0.f ? dh[e] : 0.f;
                    const float xh = (y - pm[e]) * prs[e];
                    o4[e] = ps[e] * (dz - k1[e] - xh * k2[e]);
                }
                sb += o4;
                *reinterpret_cast<f32x4*>(Dl + row * DS + c0) = o4;
            }
        }
        const long long tn = t + gridDim.x;
        const long long rb = t * 64;
        if (tn < ntiles) fetch(tn);  // the next tile's rows travel underneath the matrix phase
        __syncthreads();
        // ---- M1: dprev rows 16 wave .. + 15 ------------------------------------------------------------------------------------------------
        {
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < 6; ++S) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Dl + (16 * wave + mr) * DS + 16 * S + 4 * kq);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(Wl + ((S * 4 + nt) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[nt], 0, 0, 0);
                }
            }
            // D[row 4 kq + e][ci 16 nt + mr]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* o = p.dprev + (rb + 16 * wave + 4 * kq + e) * PC + 16 * nt + mr;
                    *o = p.assign ? acc[nt][e] : *o + acc[nt][e];
                    if (in_bn && p.in_part) {  // the BatchNorm in front: its backward sums over what this head contributes (assign mode: dprev IS this head's dz)
                        const float y = Pl[(16 * wave + 4 * kq + e) * PS_ + 16 * nt + mr];
                        const float dz = bn_out(y, em[nt], es[nt], eb[nt]) > 0.f ? acc[nt][e] : 0.f;
                        s1[nt] += dz;
                        s2[nt] = fmaf(dz, (y - em[nt]) * er[nt], s2[nt]);
                    }
                }
        }
        // ---- M2: dW1[c = 16 mt + mr][ci = 16 wave + ..] += sum over the tile's rows ---------------------------------------------------
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            float b = Pl[(4 * s + kq) * PS_ + 16 * wave + mr];
            if (in_bn) b = fmaxf(bn_out(b, wm, wsc, wb_), 0.f);
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const float a = Dl[(4 * s + kq) * DS + 16 * mt + mr];
                accw[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accw[mt], 0, 0, 0);
            }
        }
    }
    // ---- this workgroup's partials: dW1 [96][64] (wave w holds the columns 16 w .. + 15), db1 [96] -----------------------------------------
    float* ow = p.part_w1 + (long long)blockIdx.x * HC * PC;
#pragma unroll
    for (int mt = 0; mt < 6; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) ow[(16 * mt + 4 * kq + e) * PC + 16 * wave + mr] = accw[mt][e];  // D[c 4 kq + e][ci mr]
    __syncthreads();
    if (in_bn && p.in_part) {  // lanes kq = 0 .. 3 and the four waves hold different rows of the same 64 channels
        float* r2 = Pl;  // [wave][64][2]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float a = s1[nt], b = s2[nt];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (kq == 0) {
                r2[(wave * 64 + 16 * nt + mr) * 2] = a;
                r2[(wave * 64 + 16 * nt + mr) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, which = tid & 1;
            p.in_part[((long long)blockIdx.x * 64 + c) * 2 + which] =
                (double)(((r2[(0 * 64 + c) * 2 + which] + r2[(1 * 64 + c) * 2 + which]) + r2[(2 * 64 + c) * 2 + which]) + r2[(3 * 64 + c) * 2 + which]);
        }
        __syncthreads();
    }
    float* redb = Dl;  // [8 row slots][96]
    if (live)
#pragma unroll
        for (int e = 0; e < 4; ++e) redb[ro * HC + c0 + e] = sb[e];
    __syncthreads();
    if (tid < HC) {
        float s = 0.f;
        for (int k = 0; k < 8; ++k) s += redb[k * HC + tid];
        p.part_b1[(long long)blockIdx.x * HC + tid] = s;
    }
}
}  // namespace

hipError_t cerb_launch_slab_sum(const float* part, float* out, int n, int blocks, int groups, hipStream_t st);

bool cerb_head_train_supported(long long rows, int cin, int chid, int out) { return cin == PC && chid == HC && (out == 3 || out == 7) && rows > 0 && rows % 64 == 0; }

// hid [rows][96] = prev [rows][64] W1^T + b1; bn_part: [*bn_blocks][96][2] doubles for cerb_launch_bn_finalize (rows of one group)
// in_bn (optional): {mean, rstd, gamma, beta} [64] of the BatchNorm in FRONT of the head -- `prev` is then its raw input (see the kernel)
hipError_t cerb_launch_head_fwd1(const float* prev, const float* w1, const float* b1, float* hid, long long rows, double* bn_part, int* bn_blocks, hipStream_t st,
                                 const float* const* in_bn) {
    if (rows % 16) return hipErrorInvalidValue;
    const long long nt = rows / 16;
    const unsigned blocks = (unsigned)std::min<long long>((nt + 3) / 4, 2048);
    if (bn_blocks) *bn_blocks = bn_part ? (int)blocks : 0;
    hipLaunchKernelGGL(head_fwd1_kernel, dim3(blocks), dim3(256), 0, st, prev, w1, b1, hid, rows, bn_part, in_bn ? in_bn[0] : (const float*)nullptr,
                       in_bn ? in_bn[1] : (const float*)nullptr, in_bn ? in_bn[2] : (const float*)nullptr, in_bn ? in_bn[3] : (const float*)nullptr);
    return hipGetLastError();
}
hipError_t cerb_launch_head_fwd2(const float* hid, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2, const float* b2,
                                 float* logits, long long rows, int out, hipStream_t st) {
    if (rows % 16 || out > 16) return hipErrorInvalidValue;
    const long long nt = rows / 16;
    const unsigned blocks = (unsigned)std::min<long long>((nt + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(head_fwd2_kernel, dim3(blocks), dim3(256), 0, st, hid, mean, rstd, gamma, beta, w2, b2, logits, rows, out);
    return hipGetLastError();
}

constexpr long long B1_ROWS = 2048;
constexpr int B2_BLOCKS = 512;
int cerb_head_bwd2_blocks() { return B2_BLOCKS; }
// workspace of the two backward launches (floats): bwd1 partials [blocks1][out][96] + [blocks1][out] + BN partials (doubles) [blocks1][96][2];
// bwd2 partials [512][96][64] + [512][96]
size_t cerb_head_bwd_workspace_bytes(long long rows, int out) {
    const size_t b1 = (size_t)((rows + B1_ROWS - 1) / B1_ROWS);
    return b1 * ((size_t)out * HC + out) * 4 + 256 + b1 * HC * 2 * 8 + (size_t)B2_BLOCKS * (HC * PC + HC) * 4 + 256;
}
struct HeadBwdWs {
    float *pw, *pb, *pw1, *pb1;
    double* pbn;
    int blocks1;
};
static HeadBwdWs head_ws(void* ws, long long rows, int out) {
    HeadBwdWs w;
    w.blocks1 = (int)((rows + B1_ROWS - 1) / B1_ROWS);
    char* b = (char*)ws;
    w.pbn = (double*)b;  b += (size_t)w.blocks1 * HC * 2 * 8;
    w.pw = (float*)b;    b += (size_t)w.blocks1 * out * HC * 4;
    w.pb = (float*)b;    b += ((size_t)w.blocks1 * out * 4 + 255) / 256 * 256;
    w.pw1 = (float*)b;   b += (size_t)B2_BLOCKS * HC * PC * 4;
    w.pb1 = (float*)b;
    return w;
}
void cerb_bn_bwd_finalize_launch(const double* partial, int C, int blocks, float* dgamma, float* dbeta, hipStream_t st);

// dW2 [out][96], db2 [out], dgamma / dbeta [96] of the head's BatchNorm
hipError_t cerb_launch_head_bwd1(const float* hid, const float* dlog, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2,
                                 float* dw2, float* db2, float* dgamma, float* dbeta, long long rows, int out, void* ws, hipStream_t st) {
    const HeadBwdWs w = head_ws(ws, rows, out);
    if (out == 3) hipLaunchKernelGGL((head_bwd1_kernel<3>), dim3(w.blocks1), dim3(256), 0, st, hid, dlog, mean, rstd, gamma, beta, w2, w.pw, w.pb, w.pbn, rows, B1_ROWS);
    else if (out == 7) hipLaunchKernelGGL((head_bwd1_kernel<7>), dim3(w.blocks1), dim3(256), 0, st, hid, dlog, mean, rstd, gamma, beta, w2, w.pw, w.pb, w.pbn, rows, B1_ROWS);
    else return hipErrorInvalidValue;
    (void)cerb_launch_slab_sum(w.pw, dw2, out * HC, w.blocks1, 1, st);
    (void)cerb_launch_slab_sum(w.pb, db2, out, w.blocks1, 1, st);
    cerb_bn_bwd_finalize_launch(w.pbn, HC, w.blocks1, dgamma, dbeta, st);
    return hipGetLastError();
}
// dprev [rows][64] (assigned or accumulated), dW1 [96][64], db1 [96]; dgamma / dbeta: what cerb_launch_head_bwd1 left
hipError_t cerb_launch_head_bwd2(const float* hid, const float* dlog, const float* prev, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 const float* dgamma, const float* dbeta, const float* w1, const float* w2, float* dprev, float* dw1, float* db1, long long rows, int out,
                                 int eval_mode, int assign, void* ws, hipStream_t st, const float* const* in_bn, double* in_part) {
    if (rows % 64 || (in_part && !assign)) return hipErrorInvalidValue;  // (the sums are over dprev as THIS launch leaves it: first writer only)
    const HeadBwdWs w = head_ws(ws, rows, out);
    Bwd2Params p;
    p.hid = hid; p.dlog = dlog; p.prev = prev; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.dgamma = dgamma; p.dbeta = dbeta; p.w1 = w1; p.w2 = w2;
    p.in_mean = in_bn ? in_bn[0] : nullptr; p.in_rstd = in_bn ? in_bn[1] : nullptr; p.in_gamma = in_bn ? in_bn[2] : nullptr; p.in_beta = in_bn ? in_bn[3] : nullptr;
    p.in_part = in_bn ? in_part : nullptr;
    p.dprev = dprev; p.part_w1 = w.pw1; p.part_b1 = w.pb1; p.rows = rows; p.inv_m = eval_mode ? 0.f : 1.f / (float)rows; p.assign = assign;
    const int blocks = (int)std::min<long long>(rows / 64, B2_BLOCKS);
    constexpr size_t LDS_BYTES = (size_t)(64 * DS + 64 * PS_ + HC * PC + 64 * 8) * 4;  // 69.6 KB: two workgroups per CU
    static bool attr3[64], attr7[64];
    if (out == 3) {
        if (cerb_attr_needed(attr3)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_bwd2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((head_bwd2_kernel<3>), dim3(blocks), dim3(256), LDS_BYTES, st, p);
    } else if (out == 7) {
        if (cerb_attr_needed(attr7)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_bwd2_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((head_bwd2_kernel<7>), dim3(blocks), dim3(256), LDS_BYTES, st, p);
    } else return hipErrorInvalidValue;
    (void)cerb_launch_slab_sum(w.pw1, dw1, HC * PC, blocks, 1, st);
    (void)cerb_launch_slab_sum(w.pb1, db1, HC, blocks, 1, st);
    return hipGetLastError();
}
