// train_kernels.hip -- first pieces of the training step (BASELINE.json configs[4]; models/run_desc.py:25-230 of the reference).
// NOT a training step yet: what is here is the per-head loss of train_step with its gradient on the logits, HBM-bound
// elementwise / reduction work (one read of the logits per pass):
//   cross entropy (models/utils/loss_utils.py:6-21) x pixel weight map, mean over the pixels of a sample, samples without the
//   target masked out:   sum_n flag_n * mean_hw(ce * w) / (sum_n flag_n + 1e-8)                  (models/run_desc.py:147-156)
//   TYPE heads: w = class weight of the target class, 0 on background (:118-124), plus the Dice term over the positive classes with
//   mask = target > 0, smooth 1e-3, summed over classes and NOT flag-masked (:137-146, loss_utils.py:60-75)
//   Patch-Class: the reference's `[N] * [N,1,1]` broadcast makes every sample's loss the mean over all samples (:126-128,152-154)
// Three launches: per-block partial sums (fixed order -> bitwise reproducible), one-block finalise (double accumulation; loss value
// and the coefficients of the gradient), gradient.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <stdint.h>

#include <string>

#include "../../include/cerberus_hip.h"
#include "cerb_dev.h"

int cerb_set_error(const std::string& m);

hipError_t cerb_launch_pw_mfma(const float* a, const float* w, int w_trans, const float* bias, float* out, long long rows, int K, int NC, int accumulate, hipStream_t st,
                               double* bn_part = nullptr, int* bn_blocks = nullptr);
hipError_t cerb_launch_slab_sum(const float* part, float* out, int n, int blocks, int groups, hipStream_t st);
namespace {
constexpr int MAXC = 16;   // classes per head (reference: 3, 7, 9)
constexpr int PIX_PER_BLOCK = 1024;

struct LossParams {
    const float* logits;
    long long sn, sc, sy, sx;  // element strides of logits / dlogits
    const float* target;       // [N][H][W] class ids as float
    const float* has_target;   // [N]
    const float* class_weight; // [C] or nullptr
    const float* pixel_weight; // [N][H][W] or nullptr (the '#WEIGHT-MAP' target of the head; ignored when class_weight is given)
    int N, H, W, C;
    float ce_w, dice_w, head_w;
    int pc_mode;
    float* dlogits;
    float* partial;            // [blocks][1 + 3 * MAXC]: sum(ce * w), then I_c, L_c, R_c
    double* coef;              // [N] ce factor per sample, then [MAXC] dLoss/dI_c, [MAXC] dLoss/dL_c, [1] loss
    float* loss_out;
    int blocks_per_sample;
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;  // valid in thread 0
}

// softmax statistics of one pixel; returns the pixel's weight
__device__ __forceinline__ float pixel_softmax(const LossParams& p, int n, int y, int x, float* prob, int* t_out, float* ce_out) {
    const float* l = p.logits + n * p.sn + y * p.sy + x * p.sx;
    float v[MAXC], m = -3.402823466e38f;
    for (int c = 0; c < p.C; ++c) {
        v[c] = l[c * p.sc];
        m = fmaxf(m, v[c]);
    }
    float se = 0.f;
    for (int c = 0; c < p.C; ++c) {
        prob[c] = expf(v[c] - m);
        se += prob[c];
    }
    const float inv = 1.f / se;
    for (int c = 0; c < p.C; ++c) prob[c] *= inv;
    const int t = (int)p.target[((long long)n * p.H + y) * p.W + x];
    *t_out = t;
    *ce_out = (m + logf(se)) - v[t];
    if (!p.class_weight) return p.pixel_weight ? p.pixel_weight[((long long)n * p.H + y) * p.W + x] : 1.f;
    return t > 0 ? p.class_weight[t] : 0.f;  // get_class_wmap: listed classes get their weight, background keeps its own value 0
}

__global__ __launch_bounds__(256) void loss_partial_kernel(LossParams p) {
    __shared__ float sh[4];
    const int n = blockIdx.x / p.blocks_per_sample, b = blockIdx.x % p.blocks_per_sample;
    const int hw = p.H * p.W;
    float ce_sum = 0.f, I[MAXC], L[MAXC], R[MAXC];
    for (int c = 0; c < MAXC; ++c) I[c] = L[c] = R[c] = 0.f;
    for (int i = b * PIX_PER_BLOCK + threadIdx.x; i < min(hw, (b + 1) * PIX_PER_BLOCK); i += blockDim.x) {
        float prob[MAXC], ce;
        int t;
        const float w = pixel_softmax(p, n, i / p.W, i % p.W, prob, &t, &ce);
        ce_sum += ce * w;
        if (p.dice_w != 0.f && t > 0) {  // mask = target > 0
            for (int c = 1; c < p.C; ++c) {
                L[c] += prob[c];
                if (c == t) {
                    I[c] += prob[c];
                    R[c] += 1.f;
                }
            }
        }
    }
    float* out = p.partial + (long long)blockIdx.x * (1 + 3 * MAXC);
    float r = block_sum(ce_sum, sh);
    if (threadIdx.x == 0) out[0] = r;
    if (p.dice_w != 0.f)
        for (int c = 1; c < p.C; ++c) {
            const float a = block_sum(I[c], sh), bb = block_sum(L[c], sh), cc = block_sum(R[c], sh);
            if (threadIdx.x == 0) {
                out[1 + c] = a;
                out[1 + MAXC + c] = bb;
                out[1 + 2 * MAXC + c] = cc;
            }
        }
}

// one wave: every lane adds a strided share of the partials in double, lane 0 adds the 64 lane sums in lane order (fixed order: reproducible)
__device__ __forceinline__ double lane_ordered_sum(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += sh[k];
        sh[64] = t;
    }
    __syncthreads();
    const double t = sh[64];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(64) void loss_finalize_kernel(LossParams p) {
    __shared__ double sh[65];
    const int lane = threadIdx.x;
    const int stride = 1 + 3 * MAXC;
    const double hw = (double)p.H * p.W;
    double flag_sum = 0.0;
    for (int n = 0; n < p.N; ++n) flag_sum += p.has_target[n];
    const double denom = flag_sum + 1.0e-8;
    double ce_term = 0.0;
    if (!p.pc_mode) {
        for (int n = 0; n < p.N; ++n) {
            double s = 0.0;
            for (int b = lane; b < p.blocks_per_sample; b += 64) s += p.partial[(long long)(n * p.blocks_per_sample + b) * stride];
            s = lane_ordered_sum(s, sh);
            ce_term += p.has_target[n] * (s / hw);
            if (lane == 0) p.coef[n] = (double)p.ce_w * p.head_w * p.has_target[n] / (denom * hw);
        }
        ce_term /= denom;
    } else {  // every sample's loss is the mean over all samples; H = W = 1
        double s = 0.0;
        for (int n = 0; n < p.N; ++n) s += p.partial[(long long)n * stride];
        ce_term = (s / p.N) * flag_sum / denom;
        if (lane == 0)
            for (int n = 0; n < p.N; ++n) p.coef[n] = (double)p.ce_w * p.head_w * flag_sum / (denom * p.N);
    }
    double dice = 0.0;
    if (p.dice_w != 0.f)
        for (int c = 1; c < p.C; ++c) {
            double I = 0.0, L = 0.0, R = 0.0;
            for (int b = lane; b < p.N * p.blocks_per_sample; b += 64) {
                I += p.partial[(long long)b * stride + 1 + c];
                L += p.partial[(long long)b * stride + 1 + MAXC + c];
                R += p.partial[(long long)b * stride + 1 + 2 * MAXC + c];
            }
            I = lane_ordered_sum(I, sh);
            L = lane_ordered_sum(L, sh);
            R = lane_ordered_sum(R, sh);
            const double D = L + R + 1.0e-3;
            dice += 1.0 - (2.0 * I + 1.0e-3) / D;
            if (lane == 0) {
                p.coef[p.N + c] = (double)p.dice_w * p.head_w * (-2.0 / D);                         // d/dI_c
                p.coef[p.N + MAXC + c] = (double)p.dice_w * p.head_w * ((2.0 * I + 1.0e-3) / (D * D));  // d/dL_c
            }
        }
    const double loss = ((double)p.ce_w * ce_term + (double)p.dice_w * dice) * p.head_w;
    if (lane == 0) {
        p.coef[p.N + 2 * MAXC] = loss;
        if (p.loss_out) *p.loss_out = (float)loss;
    }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(LossParams p) {
    const long long total = (long long)p.N * p.H * p.W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.W);
        const long long r = i / p.W;
        const int y = (int)(r % p.H), n = (int)(r / p.H);
        float prob[MAXC], ce;
        int t;
        const float w = pixel_softmax(p, n, y, x, prob, &t, &ce);
        const float f = (float)p.coef[n] * w;
        float g[MAXC];
        for (int c = 0; c < p.C; ++c) g[c] = f * (prob[c] - (c == t ? 1.f : 0.f));
        if (p.dice_w != 0.f && t > 0) {  // q_c = dLoss/dp_c on masked pixels, chained through the softmax Jacobian
            float q[MAXC], dot = 0.f;
            q[0] = 0.f;
            for (int c = 1; c < p.C; ++c) {
                q[c] = (float)p.coef[p.N + MAXC + c] + (c == t ? (float)p.coef[p.N + c] : 0.f);
                dot += q[c] * prob[c];
            }
            for (int c = 0; c < p.C; ++c) g[c] += prob[c] * (q[c] - dot);
        }
        float* d = p.dlogits + n * p.sn + y * p.sy + x * p.sx;
        for (int c = 0; c < p.C; ++c) d[c * p.sc] = g[c];
    }
}
}  // namespace

extern "C" size_t cerb_head_loss_workspace_bytes(int n, int h, int w) {
    const long long bps = ((long long)h * w + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK;
    return (size_t)(n * bps) * (1 + 3 * MAXC) * 4 + (size_t)(n + 2 * MAXC + 1) * 8 + 512;
}

extern "C" int cerb_head_loss_wmap(const float* logits, long long stride_n, long long stride_c, long long stride_y, long long stride_x, const float* target,
                                   const float* has_target, int N, int H, int W, int C, const float* class_weight, const float* pixel_weight, float ce_weight,
                                   float dice_weight, float head_weight, int patch_class_mode, float* loss_out, float* dlogits, void* ws, size_t ws_bytes,
                                   void* hip_stream);
extern "C" int cerb_head_loss(const float* logits, long long stride_n, long long stride_c, long long stride_y, long long stride_x, const float* target,
                              const float* has_target, int N, int H, int W, int C, const float* class_weight, float ce_weight, float dice_weight,
                              float head_weight, int patch_class_mode, float* loss_out, float* dlogits, void* ws, size_t ws_bytes, void* hip_stream) {
    return cerb_head_loss_wmap(logits, stride_n, stride_c, stride_y, stride_x, target, has_target, N, H, W, C, class_weight, nullptr, ce_weight, dice_weight,
                               head_weight, patch_class_mode, loss_out, dlogits, ws, ws_bytes, hip_stream);
}
extern "C" int cerb_head_loss_wmap(const float* logits, long long stride_n, long long stride_c, long long stride_y, long long stride_x, const float* target,
                                   const float* has_target, int N, int H, int W, int C, const float* class_weight, const float* pixel_weight, float ce_weight,
                                   float dice_weight, float head_weight, int patch_class_mode, float* loss_out, float* dlogits, void* ws, size_t ws_bytes,
                                   void* hip_stream) {
    if (!logits || !target || !has_target || !ws || N <= 0 || H <= 0 || W <= 0 || C < 2 || C > MAXC) return cerb_set_error("cerb_head_loss: bad arguments");
    if (patch_class_mode && (H != 1 || W != 1)) return cerb_set_error("cerb_head_loss: Patch-Class logits are [N][C][1][1]");
    if (ws_bytes < cerb_head_loss_workspace_bytes(N, H, W)) return cerb_set_error("cerb_head_loss: workspace too small");
    hipStream_t st = (hipStream_t)hip_stream;
    LossParams p;
    p.logits = logits; p.sn = stride_n; p.sc = stride_c; p.sy = stride_y; p.sx = stride_x;
    p.target = target; p.has_target = has_target; p.class_weight = class_weight; p.pixel_weight = patch_class_mode ? nullptr : pixel_weight;
    p.N = N; p.H = H; p.W = W; p.C = C;
    p.ce_w = ce_weight; p.dice_w = dice_weight; p.head_w = head_weight; p.pc_mode = patch_class_mode;
    p.dlogits = dlogits; p.loss_out = loss_out;
    p.blocks_per_sample = (int)(((long long)H * W + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK);
    const size_t part_bytes = (size_t)N * p.blocks_per_sample * (1 + 3 * MAXC) * 4;
    p.partial = (float*)ws;
    p.coef = (double*)((char*)ws + ((part_bytes + 255) & ~(size_t)255));
    hipLaunchKernelGGL(loss_partial_kernel, dim3(N * p.blocks_per_sample), dim3(256), 0, st, p);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, p);
    if (dlogits) {
        long long blocks = ((long long)N * H * W + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cerb_set_error(std::string("cerb_head_loss: ") + hipGetErrorString(e));
    return 0;
}

// =================================================================================================================
// Train-mode forward pieces (models/run_desc.py:79-86 `model.train()` forward): BatchNorm with BATCH statistics cannot be folded
// into the convolutions, so every conv runs with its raw weights (the inference kernels, packed without the fold) and is followed by
//   cerb_launch_bn_stats : per (group, channel) mean and biased variance over the N*H*W rows of an NHWC tensor -- per-block partial
//                          sums in double, one finalising block per group (fixed order: reproducible)
//   cerb_launch_bn_apply : y = (x - mean) * gamma / sqrt(var + eps) + beta (+ residual) (ReLU), in place or from `src` into x, float4
// plus the small dense pieces that the fused inference head / Patch-Class kernels cannot serve in train mode:
//   cerb_launch_pointwise: out[r][co] = bias[co] + sum_ci in[r][ci] * W[co][ci]   (1x1 convs 64->96, 96->C, 512->256, 256->9)
//   cerb_launch_crop_gap : centre crop (Python-slice semantics of cropping_center) + global average pool of the bottom features
// First version: correctness and reproducibility; these are HBM-bound passes that a later round fuses into the conv epilogues.
// =================================================================================================================
namespace {
constexpr int BN_ROWS_PER_BLOCK = 2048;
// Rows one workgroup of the statistics passes reduces.  2048 suits the decoder levels (3.2 M rows x 5 groups = 7840 workgroups); the encoder's small maps
// got 7 .. 49 workgroups for 256 CUs that way and ran latency-bound (round 5's per-layer table: 0.3 ms per layer4 BatchNorm at 0.6 TB/s, ~8 ms of a
// 123 ms step on layers whose bytes need 1.5 ms): aim at ~2048 workgroups, between 32 and 2048 rows each.
static inline int bn_rpb(long long rows, int groups) {
    long long r = (rows * (long long)(groups > 0 ? groups : 1) + 2047) / 2048;
    r = (r + 31) / 32 * 32;
    if (r < 32) r = 32;
    if (r > BN_ROWS_PER_BLOCK) r = BN_ROWS_PER_BLOCK;
    return (int)r;
}

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, long long group_stride, long long rows, int C, int blocks_per_group,
                                                         double* __restrict__ partial, int rpb) {
    extern __shared__ double shd[];  // [256][2] per float4 lane component handled below
    const int g = blockIdx.x / blocks_per_group, b = blockIdx.x % blocks_per_group;
    const int c4n = C >> 2, tid = threadIdx.x;
    const int c4 = tid % c4n, rl = tid / c4n, nrl = 256 / c4n;  // threads with rl >= nrl idle (C / 4 need not divide 256)
    const long long r0 = (long long)b * rpb, r1 = min(rows, r0 + rpb);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rl < nrl) {
        // four rows in flight per thread (one dependent load per iteration left the pass latency-bound at 0.45 of the HBM peak); the
        // sums are still added in row order, so the statistics are bit-identical to the one-row loop
        auto add = [&](const float4& v) {
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        };
        const float* xb = x + g * group_stride + 4 * c4;
        long long r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            const float4 v0 = *reinterpret_cast<const float4*>(xb + r * C), v1 = *reinterpret_cast<const float4*>(xb + (r + nrl) * C),
                         v2 = *reinterpret_cast<const float4*>(xb + (r + 2 * nrl) * C), v3 = *reinterpret_cast<const float4*>(xb + (r + 3 * nrl) * C);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < r1; r += nrl) add(*reinterpret_cast<const float4*>(xb + r * C));
    }
    // reduce over the row lanes of each channel quad through LDS, fixed order
    for (int e = 0; e < 4; ++e) {
        __syncthreads();
        shd[tid * 2] = s[e];
        shd[tid * 2 + 1] = q[e];
        __syncthreads();
        if (rl == 0) {
            double ss = 0, qq = 0;
            for (int k = 0; k < nrl; ++k) {
                ss += shd[(k * c4n + c4) * 2];
                qq += shd[(k * c4n + c4) * 2 + 1];
            }
            double* o = partial + ((long long)blockIdx.x * C + 4 * c4 + e) * 2;
            o[0] = ss;
            o[1] = qq;
        }
    }
}
// (sum, sum of squares) partials of one channel added over the row blocks in a fixed order: 64 channels x 16 block chunks per workgroup of 1024
__device__ __forceinline__ bool bn_sum_partials(const double* __restrict__ partial, int C, int blocks_per_group, int* c_out, double* s_out, double* q_out) {
    __shared__ double sh[2][16][64];
    const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, g = blockIdx.y;
    const int per = (blocks_per_group + 15) / 16, b0 = ch * per, b1 = min(blocks_per_group, b0 + per);
    double s = 0, q = 0;
    if (c < C)
        // sixteen loads in flight, then the additions in the SAME order as before (identical bits): one load per iteration made a finalize launch a chain
        // of up to 16 dependent round trips -- 30 us for a kernel that moves 260 KB, 51 + 51 of them per training step
        for (int bb = b0; bb < b1; bb += 16) {
            double2 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int b = bb + k < b1 ? bb + k : b1 - 1;
                v[k] = *reinterpret_cast<const double2*>(partial + ((long long)(g * blocks_per_group + b) * C + c) * 2);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (bb + k < b1) {
                    s += v[k].x;
                    q += v[k].y;
                }
        }
    sh[0][ch][lane] = s;
    sh[1][ch][lane] = q;
    __syncthreads();
    if (ch != 0 || c >= C) return false;
    s = q = 0;
    for (int k = 0; k < 16; ++k) {
        s += sh[0][k][lane];
        q += sh[1][k][lane];
    }
    *c_out = c; *s_out = s; *q_out = q;
    return true;
}
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ partial, long long rows, int C, int blocks_per_group, float eps,
                                                            float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ var_unbiased) {
    int c;
    double s, q;
    if (!bn_sum_partials(partial, C, blocks_per_group, &c, &s, &q)) return;
    const int g = blockIdx.y;
    const double m = s / (double)rows;
    double v = q / (double)rows - m * m;
    if (v < 0) v = 0;
    mean[g * C + c] = (float)m;
    rstd[g * C + c] = (float)(1.0 / sqrt(v + (double)eps));
    if (var_unbiased) var_unbiased[g * C + c] = (float)(rows > 1 ? v * (double)rows / (double)(rows - 1) : v);
}
// z = gamma * (y - mean) * rstd + beta, ONE expression shared by the forward (bn_apply_kernel) and by the backward kernels, which
// recompute z's sign from y instead of reading z back (the ReLU mask costs 4 bytes per element otherwise): identical bits by construction
__device__ __forceinline__ float bn_out(float y, float m, float rs, float ga, float be) { return __fmaf_rn(y - m, rs * ga, be); }

// Thread = (channel quad c4, row lane rl) like the statistics pass; a block walks rows rl + nrl * (blockIdx.x + k * gridDim.x): the channel quad
// of a thread never changes, so the four per-channel vectors are loaded ONCE and an element costs one 64-bit multiply-add of address
// arithmetic (round 3 decoded a flat index with two 64-bit divisions per element and re-read 16 parameter values for it: the pass ran at
// 0.5 of the HBM peak on instruction issue).  Same expression per element (bn_out): identical bits.  blockIdx.y = group.
__global__ __launch_bounds__(256) void bn_apply_kernel(float* x, const float* src, const float* __restrict__ resid, long long group_stride, long long rows, int C,
                                                       int groups, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int relu) {
    const int c4n = C >> 2, tid = threadIdx.x, g = blockIdx.y;
    const int c4 = tid % c4n, rl = tid / c4n, nrl = 256 / c4n;
    if (rl >= nrl) return;
    const int gc = g * C + 4 * c4;
    const float4 m = *reinterpret_cast<const float4*>(mean + gc), rs = *reinterpret_cast<const float4*>(rstd + gc),
                 ga = *reinterpret_cast<const float4*>(gamma + gc), be = *reinterpret_cast<const float4*>(beta + gc);
    const long long base = g * group_stride + 4 * c4;
    const long long step = (long long)gridDim.x * nrl;
    // FOUR rows per iteration with every load issued before the first store: the pass is in place (no __restrict__ on x / src), so the compiler
    // cannot move a later row's loads above an earlier row's store by itself, and one row in flight per thread left the pass latency-bound
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&](long long r, float4 v, const float4& r4) {
        v.x = bn_out(v.x, m.x, rs.x, ga.x, be.x);
        v.y = bn_out(v.y, m.y, rs.y, ga.y, be.y);
        v.z = bn_out(v.z, m.z, rs.z, ga.z, be.z);
        v.w = bn_out(v.w, m.w, rs.w, ga.w, be.w);
        if (resid) {
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(x + base + r * C) = v;
    };
    long long r = (long long)blockIdx.x * nrl + rl;
    for (; r + 3 * step < rows; r += 4 * step) {
        float4 v[4], q4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(src + base + (r + k * step) * C);  // src == x: in place
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = resid ? *reinterpret_cast<const float4*>(resid + base + (r + k * step) * C) : zero4;
#pragma unroll
        for (int k = 0; k < 4; ++k) finish(r + k * step, v[k], q4[k]);
    }
    for (; r < rows; r += step)
        finish(r, *reinterpret_cast<const float4*>(src + base + r * C), resid ? *reinterpret_cast<const float4*>(resid + base + r * C) : zero4);
}
// out[r][co] = bias[co] + sum_ci in[r][ci] * W[co][ci]; thread = (row, 4 couts); weights read through the caches
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, long long rows, int cin, int cout, const float* __restrict__ in_scale) {
    const int cq = (cout + 3) >> 2;
    const long long total = rows * cq;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cq;
        const int co = 4 * (int)(i % cq);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* x = in + r * cin;
        for (int ci = 0; ci < cin; ci += 4) {
            float4 v = *reinterpret_cast<const float4*>(x + ci);
            if (in_scale) {  // dropout mask * 1 / (1 - p), per (row, channel)
                const float4 s = *reinterpret_cast<const float4*>(in_scale + r * cin + ci);
                v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            }
            for (int e = 0; e < 4; ++e)
                if (co + e < cout) {
                    const float4 ww = *reinterpret_cast<const float4*>(w + (long long)(co + e) * cin + ci);
                    acc[e] = fmaf(v.x, ww.x, acc[e]);
                    acc[e] = fmaf(v.y, ww.y, acc[e]);
                    acc[e] = fmaf(v.z, ww.z, acc[e]);
                    acc[e] = fmaf(v.w, ww.w, acc[e]);
                }
        }
        for (int e = 0; e < 4; ++e)
            if (co + e < cout) out[r * cout + co + e] = acc[e] + (bias ? bias[co + e] : 0.f);
    }
}
__global__ void crop_gap_kernel(const float* __restrict__ x, int N, int H, int W, int C, int y0, int ch, int x0, int cw, float* __restrict__ out) {
    const int total = N * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / C, c = i % C;
        float s = 0.f;  // torch's adaptive_avg_pool2d sums the window in order and divides once
        for (int y = y0; y < y0 + ch; ++y)
            for (int xx = x0; xx < x0 + cw; ++xx) s += x[(((long long)n * H + y) * W + xx) * C + c];
        out[i] = s / (float)(ch * cw);
    }
}
}  // namespace

// [partials: groups x bpg x C x (sum, sum of squares) doubles][fold area: groups x 256 x C x 2 doubles] -- more than 256 partial rows per group are folded
// to <= 256 by a parallel pass before the finalising kernel walks them (its 16 threads per channel took 30 us over 1500 rows: 1.4 ms per step)
static inline size_t bn_partial_bytes(int groups, long long bpg, int C) { return ((size_t)groups * bpg * C * 2 * 8 + 255) & ~(size_t)255; }
size_t cerb_bn_workspace_bytes(int groups, long long rows, int C) {
    const int rpb = bn_rpb(rows, groups);
    const long long bpg = (rows + rpb - 1) / rpb;
    return bn_partial_bytes(groups, bpg, C) + (size_t)groups * 256 * C * 16 + 256;
}
hipError_t cerb_launch_bn_finalize(const double* partial, int blocks, long long rows, int C, float eps, float* mean, float* rstd, float* var_unbiased, hipStream_t st,
                                   int groups, void* fold_ws);
hipError_t cerb_launch_bn_stats(const float* x, long long group_stride, long long rows, int C, int groups, float eps, float* mean, float* rstd,
                                float* var_unbiased, void* ws, hipStream_t st) {
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;  // a block covers all channel quads of a row; spare threads idle (C = 96: 24 quads x 10 row lanes)
    const int rpb = bn_rpb(rows, groups);
    const int bpg = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(groups * bpg), dim3(256), 256 * 2 * sizeof(double), st, x, group_stride, rows, C, bpg, (double*)ws, rpb);
    return cerb_launch_bn_finalize((const double*)ws, bpg, rows, C, eps, mean, rstd, var_unbiased, st, groups, (char*)ws + bn_partial_bytes(groups, bpg, C));
}
// statistics from partials some producer already wrote ([groups][blocks][C][2] doubles): mean, 1 / sqrt(var + eps), unbiased variance
// [G][B][C][2] -> [G][B2][C][2]: output row b2 = the sum of input rows b2 * per .. (b2 + 1) * per - 1, in that order (thread = one output value pair)
__global__ __launch_bounds__(256) void bn_partial_fold_kernel(const double* __restrict__ in, double* __restrict__ out, int B, int B2, int per, int C, int G) {
    const long long total = (long long)G * B2 * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), b2 = (int)((i / C) % B2), g = (int)(i / ((long long)C * B2));
        const int b0 = b2 * per, b1 = min(B, b0 + per);
        double s = 0, q = 0;
        for (int bb = b0; bb < b1; bb += 8) {  // eight loads in flight, additions in the original order (see bn_sum_partials)
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int b = bb + k < b1 ? bb + k : b1 - 1;
                v[k] = *reinterpret_cast<const double2*>(in + (((long long)g * B + b) * C + c) * 2);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (bb + k < b1) {
                    s += v[k].x;
                    q += v[k].y;
                }
        }
        double* o = out + i * 2;
        o[0] = s;
        o[1] = q;
    }
}
size_t cerb_bn_fold_workspace_bytes(int groups, int C) { return (size_t)groups * 256 * C * 16; }
// blocks > 1024: the partial rows are first folded to 256 per group into `fold_ws` (cerb_bn_fold_workspace_bytes; the finalising kernel walks its
// rows with 16 threads per channel -- 12,544 rows of a 448^2 decoder level took it 0.3 ms)
hipError_t cerb_launch_bn_finalize(const double* partial, int blocks, long long rows, int C, float eps, float* mean, float* rstd, float* var_unbiased, hipStream_t st,
                                   int groups, void* fold_ws) {
    if (blocks > 256 && fold_ws) {
        const int per = (blocks + 255) / 256, b2 = (blocks + per - 1) / per;
        const long long total = (long long)groups * b2 * C;
        hipLaunchKernelGGL(bn_partial_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, (double*)fold_ws, blocks, b2, per, C, groups);
        partial = (const double*)fold_ws;
        blocks = b2;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, partial, rows, C, blocks, eps, mean, rstd, var_unbiased);
    return hipGetLastError();
}
hipError_t cerb_launch_bn_apply(float* x, const float* src, const float* resid, long long group_stride, long long rows, int C, int groups, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, int relu, hipStream_t st) {
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;
    const int nrl = 256 / (C / 4);
    long long blocks = (rows + nrl - 1) / nrl;
    const long long cap = std::max(1ll, (256ll * 32) / std::max(1, groups));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, st, x, src ? src : x, resid, group_stride, rows, C, groups, mean, rstd, gamma,
                       beta, relu);
    return hipGetLastError();
}
// bn_part / bn_blocks (optional): BatchNorm statistics partials of the output, produced inside the layer when the MFMA path serves it (*bn_blocks > 0
// then: rows of [cout][2] doubles for cerb_launch_bn_finalize; 0 = not produced, run cerb_launch_bn_stats)
hipError_t cerb_launch_pointwise(const float* in, const float* w, const float* bias, float* out, long long rows, int cin, int cout, const float* in_scale,
                                 hipStream_t st, double* bn_part, int* bn_blocks) {
    if (bn_blocks) *bn_blocks = 0;
    if (cin % 4) return hipErrorInvalidValue;
    if (!in_scale && rows >= 4096 && cerb_launch_pw_mfma(in, w, 1, bias, out, rows, cin, cout, 0, st, bn_part, bn_blocks) == hipSuccess) return hipSuccess;
    if (bn_blocks) *bn_blocks = 0;
    long long blocks = (rows * ((cout + 3) / 4) + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pointwise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, w, bias, out, rows, cin, cout, in_scale);
    return hipGetLastError();
}
hipError_t cerb_launch_crop_gap(const float* x, int N, int H, int W, int C, int y0, int ch, int x0, int cw, float* out, hipStream_t st) {
    hipLaunchKernelGGL(crop_gap_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, x, N, H, W, C, y0, ch, x0, cw, out);
    return hipGetLastError();
}

// =================================================================================================================
// Backward pieces, FIRST VERSION (correctness against the reference's gradients; plain gather kernels, no MFMA yet -- the data
// gradient of the 3x3 stride-1 convs is the forward Winograd kernel on rotated weights and the weight gradient a split-K MFMA GEMM in
// the design).  Every kernel is a deterministic gather (no float atomics): bitwise reproducible.
// Layouts: activations NHWC [G][N][H][W][C]; weights as in the state dict [G][Cout][Cin][ks][ks].
// =================================================================================================================
namespace {
// ---- BatchNorm backward: dz (+ReLU mask from z) -> per-channel sum(dz), sum(dz * xhat); dy = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ y,
                                                             long long group_stride, long long rows, int C, int blocks_per_group,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ partial, int rpb) {
    extern __shared__ double shd[];
    const int g = blockIdx.x / blocks_per_group, b = blockIdx.x % blocks_per_group;
    const int c4n = C >> 2, tid = threadIdx.x;
    const int c4 = tid % c4n, rl = tid / c4n, nrl = 256 / c4n;  // thread = (channel quad, row lane); spare threads idle
    const long long r0 = (long long)b * rpb, r1 = min(rows, r0 + rpb);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rl < nrl) {
        const float4 m = *reinterpret_cast<const float4*>(mean + g * C + 4 * c4), rs = *reinterpret_cast<const float4*>(rstd + g * C + 4 * c4);
        float4 ga = {0.f, 0.f, 0.f, 0.f}, be = ga;
        if (relu == 2) {  // no residual: the mask is the sign of bn_out(y), recomputed instead of read
            ga = *reinterpret_cast<const float4*>(gamma + g * C + 4 * c4);
            be = *reinterpret_cast<const float4*>(beta + g * C + 4 * c4);
        }
        auto add = [&](float4 d, const float4& yy, const float4& zr) {
            if (relu) {
                float4 zz = zr;
                if (relu == 2) {
                    zz.x = bn_out(yy.x, m.x, rs.x, ga.x, be.x);
                    zz.y = bn_out(yy.y, m.y, rs.y, ga.y, be.y);
                    zz.z = bn_out(yy.z, m.z, rs.z, ga.z, be.z);
                    zz.w = bn_out(yy.w, m.w, rs.w, ga.w, be.w);
                }
                if (!(zz.x > 0.f)) d.x = 0.f;
                if (!(zz.y > 0.f)) d.y = 0.f;
                if (!(zz.z > 0.f)) d.z = 0.f;
                if (!(zz.w > 0.f)) d.w = 0.f;
            }
            s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
            q[0] += (double)d.x * ((yy.x - m.x) * rs.x);
            q[1] += (double)d.y * ((yy.y - m.y) * rs.y);
            q[2] += (double)d.z * ((yy.z - m.z) * rs.z);
            q[3] += (double)d.w * ((yy.w - m.w) * rs.w);
        };
        const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const long long base = g * group_stride + 4 * c4;
        long long r = r0 + rl;
        // four rows (eight to twelve 16-byte loads) in flight per thread, then two; sums still in row order: bit-identical to the one-row loop
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            const long long i0 = base + r * C, i1 = base + (r + nrl) * C, i2 = base + (r + 2 * nrl) * C, i3 = base + (r + 3 * nrl) * C;
            const float4 d0 = *reinterpret_cast<const float4*>(dz + i0), d1 = *reinterpret_cast<const float4*>(dz + i1),
                         d2 = *reinterpret_cast<const float4*>(dz + i2), d3 = *reinterpret_cast<const float4*>(dz + i3);
            const float4 y0 = *reinterpret_cast<const float4*>(y + i0), y1 = *reinterpret_cast<const float4*>(y + i1),
                         y2 = *reinterpret_cast<const float4*>(y + i2), y3 = *reinterpret_cast<const float4*>(y + i3);
            float4 z0 = zero4, z1 = zero4, z2 = zero4, z3 = zero4;
            if (relu == 1) {
                z0 = *reinterpret_cast<const float4*>(z + i0);
                z1 = *reinterpret_cast<const float4*>(z + i1);
                z2 = *reinterpret_cast<const float4*>(z + i2);
                z3 = *reinterpret_cast<const float4*>(z + i3);
            }
            add(d0, y0, z0);
            add(d1, y1, z1);
            add(d2, y2, z2);
            add(d3, y3, z3);
        }
        for (; r + nrl < r1; r += 2 * nrl) {
            const long long i0 = base + r * C, i1 = base + (r + nrl) * C;
            const float4 d0 = *reinterpret_cast<const float4*>(dz + i0), d1 = *reinterpret_cast<const float4*>(dz + i1);
            const float4 y0 = *reinterpret_cast<const float4*>(y + i0), y1 = *reinterpret_cast<const float4*>(y + i1);
            float4 z0 = zero4, z1 = zero4;
            if (relu == 1) {
                z0 = *reinterpret_cast<const float4*>(z + i0);
                z1 = *reinterpret_cast<const float4*>(z + i1);
            }
            add(d0, y0, z0);
            add(d1, y1, z1);
        }
        for (; r < r1; r += nrl) {
            const long long i = base + r * C;
            add(*reinterpret_cast<const float4*>(dz + i), *reinterpret_cast<const float4*>(y + i), relu == 1 ? *reinterpret_cast<const float4*>(z + i) : zero4);
        }
    }
    for (int e = 0; e < 4; ++e) {
        __syncthreads();
        shd[tid * 2] = s[e];
        shd[tid * 2 + 1] = q[e];
        __syncthreads();
        if (rl == 0) {
            double ss = 0, qq = 0;
            for (int k = 0; k < nrl; ++k) {
                ss += shd[(k * c4n + c4) * 2];
                qq += shd[(k * c4n + c4) * 2 + 1];
            }
            double* o = partial + ((long long)blockIdx.x * C + 4 * c4 + e) * 2;
            o[0] = ss;
            o[1] = qq;
        }
    }
}
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int C, int blocks_per_group, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    int c;
    double s, q;
    if (!bn_sum_partials(partial, C, blocks_per_group, &c, &s, &q)) return;
    dbeta[blockIdx.y * C + c] = (float)s;
    dgamma[blockIdx.y * C + c] = (float)q;
}
// dy (+)= gamma * rstd * (dzm - dbeta / M - xhat * dgamma / M); dresid += dzm (dzm = ReLU-masked dz).  Thread = four channels of a row;
// ASSIGN: dy holds nothing yet (this BatchNorm is its first and only writer): no read of it, and no zero fill before the launch
template <bool ASSIGN>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ y,
                                                           float* __restrict__ dy, float* __restrict__ dresid, long long group_stride, long long rows, int C,
                                                           int groups, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu,
                                                           unsigned long long eval_mask, int resid_assign) {
    // thread = (channel quad, row lane), rows strided by the grid, blockIdx.y = group (see bn_apply_kernel): per-channel vectors loaded once
    const int c4n = C >> 2, tid = threadIdx.x, g = blockIdx.y;
    const int c4 = tid % c4n, rl = tid / c4n, nrl = 256 / c4n;
    if (rl >= nrl) return;
    // a group in EVAL mode (cerb_net_set_bn_eval) normalises with constants: dy = dz * gamma * rstd, no mean / xhat-projection terms
    // (they are the derivative of the BATCH statistics, which an eval-mode BatchNorm does not use)
    const float invM = ((eval_mask >> (g & 63)) & 1ull) ? 0.f : 1.f / (float)rows;
    const int gc = g * C + 4 * c4;
    const float4 mu = *reinterpret_cast<const float4*>(mean + gc), rs = *reinterpret_cast<const float4*>(rstd + gc),
                 ga = *reinterpret_cast<const float4*>(gamma + gc), dg = *reinterpret_cast<const float4*>(dgamma + gc), db = *reinterpret_cast<const float4*>(dbeta + gc);
    float4 be = {0.f, 0.f, 0.f, 0.f};
    if (relu == 2) be = *reinterpret_cast<const float4*>(beta + gc);
    const long long base = g * group_stride + 4 * c4;
    const long long step = (long long)gridDim.x * nrl;
    // four rows per iteration, every load before the first store (see bn_apply_kernel)
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&](long long r, float4 d, const float4& yv, const float4& zm, float4 r4, const float4& p) {
        const long long a = base + r * C;
        if (relu) {
            float4 m = zm;
            if (relu == 2) {
                m.x = bn_out(yv.x, mu.x, rs.x, ga.x, be.x);
                m.y = bn_out(yv.y, mu.y, rs.y, ga.y, be.y);
                m.z = bn_out(yv.z, mu.z, rs.z, ga.z, be.z);
                m.w = bn_out(yv.w, mu.w, rs.w, ga.w, be.w);
            }
            if (!(m.x > 0.f)) d.x = 0.f;
            if (!(m.y > 0.f)) d.y = 0.f;
            if (!(m.z > 0.f)) d.z = 0.f;
            if (!(m.w > 0.f)) d.w = 0.f;
        }
        if (dresid) {
            r4.x += d.x; r4.y += d.y; r4.z += d.z; r4.w += d.w;
            *reinterpret_cast<float4*>(dresid + a) = r4;
        }
        float4 o;
        o.x = ga.x * rs.x * (d.x - db.x * invM - ((yv.x - mu.x) * rs.x) * dg.x * invM);
        o.y = ga.y * rs.y * (d.y - db.y * invM - ((yv.y - mu.y) * rs.y) * dg.y * invM);
        o.z = ga.z * rs.z * (d.z - db.z * invM - ((yv.z - mu.z) * rs.z) * dg.z * invM);
        o.w = ga.w * rs.w * (d.w - db.w * invM - ((yv.w - mu.w) * rs.w) * dg.w * invM);
        if (!ASSIGN) {
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *reinterpret_cast<float4*>(dy + a) = o;
    };
    auto ld = [&](const float* t, long long r) { return *reinterpret_cast<const float4*>(t + base + r * C); };
    long long r = (long long)blockIdx.x * nrl + rl;
    for (; r + 3 * step < rows; r += 4 * step) {
        float4 d[4], yv[4], zm[4], r4[4], p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = ld(dz, r + k * step);
            yv[k] = ld(y, r + k * step);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            zm[k] = relu == 1 ? ld(z, r + k * step) : zero4;
            r4[k] = (dresid && !resid_assign) ? ld(dresid, r + k * step) : zero4;
            p[k] = ASSIGN ? zero4 : ld(dy, r + k * step);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) finish(r + k * step, d[k], yv[k], zm[k], r4[k], p[k]);
    }
    for (; r < rows; r += step)
        finish(r, ld(dz, r), ld(y, r), relu == 1 ? ld(z, r) : zero4, (dresid && !resid_assign) ? ld(dresid, r) : zero4, ASSIGN ? zero4 : ld(dy, r));
}
// ---- convolution backward, gather form (any ks / stride / pad = ks / 2; groups = independent convs stacked along G) ------------------
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int G, int N, int H,
                                                         int W, int Cin, int Ho, int Wo, int Cout, int ks, int stride, long long x_gs) {
    const int pad = ks / 2;
    const long long per_g = (long long)N * H * W * Cin, total = per_g * G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i / per_g);
        long long r = i - (long long)g * per_g;
        const int ci = (int)(r % Cin);
        r /= Cin;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H), n = (int)(r / H);
        float acc = 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const float* d = dy + (((long long)(g * N + n) * Ho + oy) * Wo + ox) * Cout;
                const float* ww = w + ((long long)g * Cout * Cin + ci) * ks * ks + ky * ks + kx;
                for (int co = 0; co < Cout; ++co) acc = fmaf(d[co], ww[(long long)co * Cin * ks * ks], acc);
            }
        }
        dx[g * x_gs + (i - (long long)g * per_g)] += acc;
    }
}
// dW[g][co][ci][ky][kx] = sum over pixels; one block per (g, co, ci), threads = taps x pixel lanes, fixed-order reduction
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N, int H, int W,
                                                         int Cin, int Ho, int Wo, int Cout, int ks, int stride, long long x_gs) {
    __shared__ double sh[256];
    const int pad = ks / 2, T = ks * ks;
    const long long b = blockIdx.x;
    const int ci = (int)(b % Cin), co = (int)((b / Cin) % Cout), g = (int)(b / ((long long)Cin * Cout));
    const long long npix = (long long)N * Ho * Wo;
    for (int t = 0; t < T; ++t) {
        const int ky = t / ks, kx = t % ks;
        double s = 0;
        for (long long p = threadIdx.x; p < npix; p += 256) {
            const int ox = (int)(p % Wo);
            const long long r = p / Wo;
            const int oy = (int)(r % Ho), n = (int)(r / Ho);
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            s += (double)dy[((long long)g * npix + p) * Cout + co] * x[g * x_gs + (((long long)n * H + iy) * W + ix) * Cin + ci];
        }
        __syncthreads();
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0;
            for (int k = 0; k < 256; ++k) tot += sh[k];
            dw[(((long long)g * Cout + co) * Cin + ci) * T + t] = (float)tot;
        }
    }
}
// per-channel sum over rows (conv bias gradient, pointwise bias gradient)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ d, long long group_stride, long long rows, int C, float* __restrict__ out) {
    __shared__ double sh[256];
    const int c = blockIdx.x % C, g = blockIdx.x / C;
    double s = 0;
    for (long long r = threadIdx.x; r < rows; r += 256) s += d[g * group_stride + r * C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < 256; ++k) t += sh[k];
        out[g * C + c] = (float)t;
    }
}
// stem: dW[co][c][ky][kx] = sum dy[n][y][x][co] * tiles[n][y+ky-3][x+kx-3][c] / 255
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const unsigned char* __restrict__ tiles, const float* __restrict__ dy, float* __restrict__ dw, int N, int H,
                                                         int W) {
    __shared__ double sh[256];
    const int b = blockIdx.x;  // ((co * 3 + c) * 7 + ky) * 7 + kx
    const int kx = b % 7, ky = (b / 7) % 7, c = (b / 49) % 3, co = b / 147;
    const long long npix = (long long)N * H * W;
    double s = 0;
    for (long long p = threadIdx.x; p < npix; p += 256) {
        const int x = (int)(p % W);
        const long long r = p / W;
        const int y = (int)(r % H), n = (int)(r / H);
        const int iy = y + ky - 3, ix = x + kx - 3;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        s += (double)dy[p * 64 + co] * ((float)tiles[(((long long)n * H + iy) * W + ix) * 3 + c] / 255.0f);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < 256; ++k) t += sh[k];
        dw[b] = (float)t;
    }
}
// max-pool 3x3 stride 2 pad 1: an input pixel collects the gradient of every window whose FIRST maximum (row-major scan, as
// torch.nn.functional.max_pool2d records it) it is
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ypool, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    // thread = four channels of an input pixel (C is a multiple of 4): one index decode, 16-byte accesses; per channel the rule is unchanged
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    // (tried, round 5: XCD-contiguous block order through xcd_remap, so that the window rows neighbouring blocks share meet in one L2 -- 1.32 -> 1.40 ms, no gain)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = 4 * (int)(i % C4);
        long long r = i / C4;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H), n = (int)(r / H);
        const long long xi = (((long long)n * H + iy) * W + ix) * C + c;
        const float4 xv4 = *reinterpret_cast<const float4*>(x + xi);
        const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int oy = (iy + 1 - 2 + 1) / 2; oy <= (iy + 1) / 2; ++oy) {      // windows with oy*2-1 <= iy <= oy*2+1
            if (oy < 0 || oy >= Ho) continue;
            for (int ox = (ix + 1 - 2 + 1) / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                const long long oi = (((long long)n * Ho + oy) * Wo + ox) * C + c;
                const float4 yp4 = *reinterpret_cast<const float4*>(ypool + oi);
                const float yp[4] = {yp4.x, yp4.y, yp4.z, yp4.w};
                // the pooled value is the window's maximum: most (pixel, window) pairs end here; a pixel that holds it is the FIRST maximum
                // when no window element before it in the row-major scan holds it too
                bool cand[4], any = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cand[e] = (xv[e] == yp[e]);
                    any |= cand[e];
                }
                if (!any) continue;
                for (int ky = 0; ky < 3; ++ky) {
                    const int y = oy * 2 - 1 + ky;
                    if (y < 0 || y >= H || y > iy) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if (xx < 0 || xx >= W) continue;
                        if (y == iy && xx >= ix) break;
                        const float4 o4 = *reinterpret_cast<const float4*>(x + (((long long)n * H + y) * W + xx) * C + c);
                        if (o4.x == xv[0]) cand[0] = false;
                        if (o4.y == xv[1]) cand[1] = false;
                        if (o4.z == xv[2]) cand[2] = false;
                        if (o4.w == xv[3]) cand[3] = false;
                    }
                }
                const float4 g4 = *reinterpret_cast<const float4*>(dy + oi);
                if (cand[0]) acc[0] += g4.x;
                if (cand[1]) acc[1] += g4.y;
                if (cand[2]) acc[2] += g4.z;
                if (cand[3]) acc[3] += g4.w;
            }
        }
        float4 d = *reinterpret_cast<float4*>(dx + xi);
        d.x += acc[0]; d.y += acc[1]; d.z += acc[2]; d.w += acc[3];
        *reinterpret_cast<float4*>(dx + xi) = d;
    }
}
// out_g = skip + up2(prev_g):  dskip += sum_g dout_g;  dprev_g = transpose of the bilinear x2 (align_corners = False) applied to dout_g
__global__ __launch_bounds__(256) void upadd_bwd_skip_kernel(const float* __restrict__ dout, float* __restrict__ dskip, int G, long long per_group) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per_group; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += dout[g * per_group + i];
        dskip[i] += s;
    }
}
__device__ __forceinline__ void up2_taps(int o, int n_src, int& i0, int& i1, float& w0, float& w1) {  // output index o -> sources and weights
    const int bm = (o + 1) / 2 - 1;  // o = 2 bm + 1 or 2 bm + 2
    i0 = max(bm, 0);
    i1 = min(bm + 1, n_src - 1);
    if (o == 2 * bm + 1) { w0 = 0.75f; w1 = 0.25f; } else { w0 = 0.25f; w1 = 0.75f; }
}
__global__ __launch_bounds__(256) void upadd_bwd_prev_kernel(const float* __restrict__ dout, float* __restrict__ dprev, int G, int N, int H, int W, int C,
                                                             long long prev_gs, int shared_prev) {
    // thread = four channels of a source pixel; the (at most 4 x 4) output pixels it feeds are added in raster order, groups outermost
    const int Hp = H / 2, Wp = W / 2, C4 = C >> 2;
    const long long per_g = (long long)N * Hp * Wp * C4, total = shared_prev ? per_g : per_g * G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = shared_prev ? 0 : (int)(i / per_g);
        long long r = i - (long long)g * per_g;
        const int c = 4 * (int)(r % C4);
        r /= C4;
        const int pn = (int)(r % Wp);
        r /= Wp;
        const int pm = (int)(r % Hp), n = (int)(r / Hp);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int gg = shared_prev ? 0 : g; gg < (shared_prev ? G : g + 1); ++gg)
            for (int y = max(2 * pm - 2, 0); y <= min(2 * pm + 2, H - 1); ++y) {
                int a0, a1;
                float u0, u1;
                up2_taps(y, Hp, a0, a1, u0, u1);
                const float wy = (a0 == pm ? u0 : 0.f) + (a1 == pm ? u1 : 0.f);
                if (wy == 0.f) continue;
                for (int x = max(2 * pn - 2, 0); x <= min(2 * pn + 2, W - 1); ++x) {
                    int b0, b1;
                    float v0, v1;
                    up2_taps(x, Wp, b0, b1, v0, v1);
                    const float wx = (b0 == pn ? v0 : 0.f) + (b1 == pn ? v1 : 0.f);
                    if (wx == 0.f) continue;
                    const float4 d = *reinterpret_cast<const float4*>(dout + ((((long long)gg * N + n) * H + y) * W + x) * C + c);
                    const float wgt = wy * wx;
                    acc.x += wgt * d.x; acc.y += wgt * d.y; acc.z += wgt * d.z; acc.w += wgt * d.w;
                }
            }
        float4* o = reinterpret_cast<float4*>(dprev + (shared_prev ? 0 : g * prev_gs) + (((long long)n * Hp + pm) * Wp + pn) * C + c);
        float4 pv = *o;
        pv.x += acc.x; pv.y += acc.y; pv.z += acc.z; pv.w += acc.w;
        *o = pv;
    }
}
// Both gradients of the decoder entry in ONE pass over dout (round 4: the two kernels above read it twice -- 4.9 ms of a 174 ms step at 0.38 of HBM):
// thread = four channels of a SOURCE pixel (pm, pn) of the level below, all groups: the <= 4 x 4 output pixels it feeds give dprev (same loops, same
// order as upadd_bwd_prev_kernel: identical bits), and the 2 x 2 of them it owns -- rows 2 pm, 2 pm + 1, columns 2 pn, 2 pn + 1 -- give dskip
// (sum over the groups in group order, then added: as upadd_bwd_skip_kernel).
// group_mask: bit g clear = group g's dout counts as ZERO (a decoder that does not train: the caller used to zero-fill its slice first -- 2.5 GB per
// step); skip_assign / prev_assign: the output holds nothing yet (this is its first writer): assign, no zero fill before the launch and no read here
__global__ __launch_bounds__(256) void upadd_bwd_fused_kernel(const float* __restrict__ dout, float* __restrict__ dskip, float* __restrict__ dprev, int G, int N,
                                                              int H, int W, int C, long long prev_gs, int shared_prev, unsigned group_mask, int skip_assign,
                                                              int prev_assign) {
    const int Hp = H / 2, Wp = W / 2, C4 = C >> 2;
    const long long total = (long long)N * Hp * Wp * C4;
    // (tried, round 5: XCD-contiguous block order through xcd_remap -- 1.76 -> 1.87 ms per step, no gain: the 2.9x traffic is not neighbours' rows missing in L2)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = 4 * (int)(r % C4);
        r /= C4;
        const int pn = (int)(r % Wp);
        r /= Wp;
        const int pm = (int)(r % Hp), n = (int)(r / Hp);
        float4 sk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int gg = 0; gg < G; ++gg) {
            if (!shared_prev) acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool live = (group_mask >> gg) & 1u;
            for (int y = max(2 * pm - 2, 0); live && y <= min(2 * pm + 2, H - 1); ++y) {
                int a0, a1;
                float u0, u1;
                up2_taps(y, Hp, a0, a1, u0, u1);
                const float wy = (a0 == pm ? u0 : 0.f) + (a1 == pm ? u1 : 0.f);
                if (wy == 0.f) continue;
                for (int x = max(2 * pn - 2, 0); x <= min(2 * pn + 2, W - 1); ++x) {
                    int b0, b1;
                    float v0, v1;
                    up2_taps(x, Wp, b0, b1, v0, v1);
                    const float wx = (b0 == pn ? v0 : 0.f) + (b1 == pn ? v1 : 0.f);
                    if (wx == 0.f) continue;
                    const float4 d = *reinterpret_cast<const float4*>(dout + ((((long long)gg * N + n) * H + y) * W + x) * C + c);
                    const float wgt = wy * wx;
                    acc.x += wgt * d.x; acc.y += wgt * d.y; acc.z += wgt * d.z; acc.w += wgt * d.w;
                    if ((y >> 1) == pm && (x >> 1) == pn) {
                        float4& t = sk[(y & 1) * 2 + (x & 1)];
                        t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
                    }
                }
            }
            if (!shared_prev && (live || prev_assign)) {
                float4* o = reinterpret_cast<float4*>(dprev + gg * prev_gs + (((long long)n * Hp + pm) * Wp + pn) * C + c);
                float4 pv = prev_assign ? make_float4(0.f, 0.f, 0.f, 0.f) : *o;
                pv.x += acc.x; pv.y += acc.y; pv.z += acc.z; pv.w += acc.w;
                *o = pv;
            }
        }
        if (shared_prev) {
            float4* o = reinterpret_cast<float4*>(dprev + (((long long)n * Hp + pm) * Wp + pn) * C + c);
            float4 pv = prev_assign ? make_float4(0.f, 0.f, 0.f, 0.f) : *o;
            pv.x += acc.x; pv.y += acc.y; pv.z += acc.z; pv.w += acc.w;
            *o = pv;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4* o = reinterpret_cast<float4*>(dskip + (((long long)n * H + 2 * pm + (k >> 1)) * W + 2 * pn + (k & 1)) * C + c);
            float4 pv = skip_assign ? make_float4(0.f, 0.f, 0.f, 0.f) : *o;
            pv.x += sk[k].x; pv.y += sk[k].y; pv.z += sk[k].z; pv.w += sk[k].w;
            *o = pv;
        }
    }
}
// max-pool backward by the recorded positions (net_kernels.hip: maxpool3x3s2_idx_kernel): thread = four channels of an input pixel; of the <= 4 windows
// that contain it, those whose recorded position is this pixel hand their gradient over.  Reads one packed word per window (four channels' bytes) and the
// gradient rows it wins; neither the input nor the pooled map (the scan kernel above reads both and re-walks windows to find first maxima: 5.1 GB, 1.3 ms).
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned* __restrict__ idx, const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W,
                                                              int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long r = i / C4;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H), n = (int)(r / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {  // windows with 2 oy - 1 <= iy <= 2 oy + 1
            if (oy >= Ho) continue;
            const unsigned ky = (unsigned)(iy - (2 * oy - 1));
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const unsigned k = ky * 3u + (unsigned)(ix - (2 * ox - 1));
                const long long o = (((long long)n * Ho + oy) * Wo + ox) * C4 + c4;
                const unsigned w = idx[o];
                const bool h0 = (w & 255u) == k, h1 = ((w >> 8) & 255u) == k, h2 = ((w >> 16) & 255u) == k, h3 = (w >> 24) == k;
                if (h0 | h1 | h2 | h3) {
                    const float4 g = *reinterpret_cast<const float4*>(dy + o * 4);
                    if (h0) acc.x += g.x;
                    if (h1) acc.y += g.y;
                    if (h2) acc.z += g.z;
                    if (h3) acc.w += g.w;
                }
            }
        }
        float4 d = *reinterpret_cast<float4*>(dx + i * 4);  // the stem's output is also the decoders' first skip: their gradient is already here
        d.x += acc.x; d.y += acc.y; d.z += acc.z; d.w += acc.w;
        *reinterpret_cast<float4*>(dx + i * 4) = d;
    }
}
// data gradient of a 1x1 stride-2 conv (the residual downsample branches): only the even input positions receive anything --
// dx[n][2 yo][2 xo][ci] += sum_co dy[n][yo][xo][co] W[co][ci]; thread = (output pixel, ci), the dy row is a broadcast
__global__ __launch_bounds__(256) void conv1x1s2_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int H, int W,
                                                              int Cin, int Cout) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        long long r = i / Cin;
        const int xo = (int)(r % Wo);
        const long long r2 = r / Wo;
        const int yo = (int)(r2 % Ho), n = (int)(r2 / Ho);
        const float* d = dy + r * Cout;
        float acc = 0.f;
        for (int co = 0; co < Cout; ++co) acc = fmaf(d[co], w[(long long)co * Cin + ci], acc);
        dx[(((long long)n * H + 2 * yo) * W + 2 * xo) * Cin + ci] += acc;
    }
}
// Data gradient of a 1x1 convolution (stride 1 or 2) with EIGHT output pixels per workgroup pass: their dy rows are staged in LDS once (coalesced), a
// thread = one input channel then reads four dy values of a row per ds_read_b128 (a broadcast) and four weight values (coalesced, independent) per
// 32 multiply-adds.  The one-row kernels issued a 64-lane broadcast load of dy and a weight load per multiply-add, each behind the previous one's
// wait: 0.6 ms per layer whatever its shape, 3 % of the vector pipe.  Per row the products are added in the same order: identical bits.
__global__ __launch_bounds__(256) void conv1x1_dgrad8_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int H, int W,
                                                             int Cin, int Cout, int stride) {
    extern __shared__ __attribute__((aligned(16))) float dyl[];  // [8][Cout]
    const int Ho = H / stride, Wo = W / stride;
    const long long rows = (long long)N * Ho * Wo, nrb = (rows + 7) / 8;
    for (long long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const long long r0 = rb * 8;
        const int nr = (int)min(8ll, rows - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * Cout; i += blockDim.x) dyl[i] = i < nr * Cout ? dy[r0 * Cout + i] : 0.f;
        __syncthreads();
        for (int ci = threadIdx.x; ci < Cin; ci += blockDim.x) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int co = 0; co < Cout; co += 4) {
                const float w0 = w[(long long)co * Cin + ci], w1 = w[(long long)(co + 1) * Cin + ci], w2 = w[(long long)(co + 2) * Cin + ci],
                            w3 = w[(long long)(co + 3) * Cin + ci];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 d = *reinterpret_cast<const float4*>(dyl + j * Cout + co);
                    acc[j] = fmaf(d.x, w0, acc[j]);
                    acc[j] = fmaf(d.y, w1, acc[j]);
                    acc[j] = fmaf(d.z, w2, acc[j]);
                    acc[j] = fmaf(d.w, w3, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nr) {
                    const long long r = r0 + j;
                    const int xo = (int)(r % Wo);
                    const long long r2 = r / Wo;
                    const int yo = (int)(r2 % Ho), n = (int)(r2 / Ho);
                    dx[(((long long)n * H + stride * yo) * W + stride * xo) * Cin + ci] += acc[j];
                }
        }
    }
}
// pointwise (1x1) backward: dx[r][ci] += scale * sum_co dy[r][co] W[co][ci];  dW[co][ci] = sum_r dy[r][co] x[r][ci] scale
__global__ __launch_bounds__(256) void pointwise_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, long long rows, int cin,
                                                              int cout, const float* __restrict__ in_scale, int assign) {
    const long long total = rows * cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cin;
        const int ci = (int)(i % cin);
        float acc = 0.f;
        for (int co = 0; co < cout; ++co) acc = fmaf(dy[r * cout + co], w[(long long)co * cin + ci], acc);
        const float v = in_scale ? acc * in_scale[i] : acc;
        dx[i] = assign ? v : dx[i] + v;  // assign: dx has no other writer and holds nothing yet
    }
}
__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, long long rows, int cin,
                                                              int cout, const float* __restrict__ in_scale) {
    __shared__ double sh[256];
    const int ci = blockIdx.x % cin, co = blockIdx.x / cin;
    double s = 0;
    for (long long r = threadIdx.x; r < rows; r += 256) {
        float xv = x[r * cin + ci];
        if (in_scale) xv *= in_scale[r * cin + ci];
        s += (double)dy[r * cout + co] * xv;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < 256; ++k) t += sh[k];
        dw[(long long)co * cin + ci] = (float)t;
    }
}
// weight gradient of a pointwise layer with FEW outputs (the heads' 96 -> 3 / 7): thread = input channel, the dy row is a broadcast;
// per-slab partials, added in slab order by the second kernel
__global__ __launch_bounds__(128) void pw_wgrad_small_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                                    long long rows, int cin, int cout, long long rows_per_block) {
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int ci = threadIdx.x; ci < cin; ci += 128) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        long long r = r0;
        for (; r + 4 <= r1; r += 4) {  // four rows in flight: the loop is latency-bound otherwise
            const float x0 = x[r * cin + ci], x1 = x[(r + 1) * cin + ci], x2 = x[(r + 2) * cin + ci], x3 = x[(r + 3) * cin + ci];
            for (int c = 0; c < cout; ++c) {
                acc[c] = fmaf(dy[r * cout + c], x0, acc[c]);
                acc[c] = fmaf(dy[(r + 1) * cout + c], x1, acc[c]);
                acc[c] = fmaf(dy[(r + 2) * cout + c], x2, acc[c]);
                acc[c] = fmaf(dy[(r + 3) * cout + c], x3, acc[c]);
            }
        }
        for (; r < r1; ++r) {
            const float xv = x[r * cin + ci];
            for (int c = 0; c < cout; ++c) acc[c] = fmaf(dy[r * cout + c], xv, acc[c]);
        }
        for (int c = 0; c < cout; ++c) part[((long long)blockIdx.x * cout + c) * cin + ci] = acc[c];
    }
}
// The whole backward of such a layer in ONE pass over its input (round 4: the per-family table of bench.py --mode train put the pointwise backward at
// 17 ms of a 180 ms step, 0.16 of any roof -- three passes over the rows for the heads' 96 -> 3 / 7: weight gradient, data gradient, bias sums).
// thread = input channel: it keeps its weight column (<= 8 values), reads x[r][ci] once, adds dy[r][c] x to its weight-gradient partials and
// writes dx[r][ci] = sum_c dy[r][c] W[c][ci] (same fmaf order as pointwise_dgrad_kernel: bit-identical); thread 0 also sums dy for the bias.
__global__ __launch_bounds__(128) void pw_bwd_small_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                                float* __restrict__ dx, float* __restrict__ part, float* __restrict__ part_b, long long rows,
                                                                int cin, int cout, long long rows_per_block, int assign) {
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int ci = threadIdx.x; ci < cin; ci += 128) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, accb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wc[8];
        for (int c = 0; c < 8; ++c) wc[c] = c < cout ? w[(long long)c * cin + ci] : 0.f;
        long long r = r0;
        for (; r + 4 <= r1; r += 4) {  // four rows in flight
            const float x0 = x[r * cin + ci], x1 = x[(r + 1) * cin + ci], x2 = x[(r + 2) * cin + ci], x3 = x[(r + 3) * cin + ci];
            float p0 = assign ? 0.f : dx[r * cin + ci], p1 = assign ? 0.f : dx[(r + 1) * cin + ci], p2 = assign ? 0.f : dx[(r + 2) * cin + ci],
                  p3 = assign ? 0.f : dx[(r + 3) * cin + ci];
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
            for (int c = 0; c < cout; ++c) {
                const float d0 = dy[r * cout + c], d1 = dy[(r + 1) * cout + c], d2 = dy[(r + 2) * cout + c], d3 = dy[(r + 3) * cout + c];
                acc[c] = fmaf(d0, x0, acc[c]);
                acc[c] = fmaf(d1, x1, acc[c]);
                acc[c] = fmaf(d2, x2, acc[c]);
                acc[c] = fmaf(d3, x3, acc[c]);
                g0 = fmaf(d0, wc[c], g0);
                g1 = fmaf(d1, wc[c], g1);
                g2 = fmaf(d2, wc[c], g2);
                g3 = fmaf(d3, wc[c], g3);
                if (ci == 0) accb[c] += (d0 + d1) + (d2 + d3);
            }
            dx[r * cin + ci] = p0 + g0;
            dx[(r + 1) * cin + ci] = p1 + g1;
            dx[(r + 2) * cin + ci] = p2 + g2;
            dx[(r + 3) * cin + ci] = p3 + g3;
        }
        for (; r < r1; ++r) {
            const float xv = x[r * cin + ci];
            float g = 0.f;
            for (int c = 0; c < cout; ++c) {
                const float d = dy[r * cout + c];
                acc[c] = fmaf(d, xv, acc[c]);
                g = fmaf(d, wc[c], g);
                if (ci == 0) accb[c] += d;
            }
            dx[r * cin + ci] = assign ? g : dx[r * cin + ci] + g;
        }
        for (int c = 0; c < cout; ++c) part[((long long)blockIdx.x * cout + c) * cin + ci] = acc[c];
        if (ci == 0)
            for (int c = 0; c < cout; ++c) part_b[(long long)blockIdx.x * cout + c] = accb[c];
    }
}
// D[n][2 yo][2 xo][:] = dy[n][yo][xo][:], zero elsewhere (H, W even): turns the data gradient of a stride-2 conv into a stride-1 conv
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ dy, float* __restrict__ d, long long n, int H, int W, int C4) {
    const long long total = n * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long long b = r / H;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!((x | y) & 1)) v = reinterpret_cast<const float4*>(dy)[((b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * C4 + c];
        reinterpret_cast<float4*>(d)[i] = v;
    }
}
__global__ void crop_gap_bwd_kernel(const float* __restrict__ dg, float* __restrict__ dx, int N, int H, int W, int C, int y0, int ch, int x0, int cw) {
    const long long total = (long long)N * ch * cw * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % cw);
        r /= cw;
        const int y = (int)(r % ch), n = (int)(r / ch);
        dx[(((long long)n * H + y0 + y) * W + x0 + x) * C + c] += dg[n * C + c] / (float)(ch * cw);
    }
}
// Adam (torch.optim.Adam, no weight decay / amsgrad; models/opt.py:47-58): one launch per parameter tensor in this first version
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n, float lr, float b1,
                            float b2, float eps, float bc1, float bc2) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] = b1 * m[i] + (1.f - b1) * gi;
        const float vi = v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}
// the same update over MANY tensors in one launch: a block takes one 16384-element chunk of one tensor (table built by the launcher)
struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
};
constexpr int ADAM_CHUNK = 16384;
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTensor* __restrict__ tensors, const int2* __restrict__ chunks, float lr, float b1, float b2,
                                                         float eps, float bc1, float bc2) {
    const int2 c = chunks[blockIdx.x];
    const AdamTensor t = tensors[c.x];
    const long long lo = (long long)c.y * ADAM_CHUNK, hi = min(t.n, lo + ADAM_CHUNK);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float gi = t.g[i];
        const float mi = t.m[i] = b1 * t.m[i] + (1.f - b1) * gi;
        const float vi = t.v[i] = b2 * t.v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        t.p[i] -= (lr / bc1) * (mi / denom);
    }
}
// ---- many small device-to-device copies in ONE launch (cerb_net_update_params: ~470 parameter tensors per optimiser step went out as 470 hipMemcpyAsync) ----
struct CopyDesc {
    float* dst;
    const float* src;
    long long n;
};
constexpr int COPY_CHUNK = 16384;
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyDesc* __restrict__ descs, const int2* __restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const CopyDesc d = descs[c.x];
    const long long lo = (long long)c.y * COPY_CHUNK, hi = min(d.n, lo + COPY_CHUNK);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) d.dst[i] = d.src[i];
}
static unsigned gridfor(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (unsigned)b;
}
}  // namespace

hipError_t cerb_launch_bn_bwd(const float* dz, const float* z, const float* y, float* dy, float* dresid, long long group_stride, long long rows, int C, int groups,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, int relu, int dy_assign,
                              void* ws, hipStream_t st, unsigned long long eval_mask, int dresid_assign, const double* pre_part, int pre_bpg) {
    // pre_part: the reduction pass's partials ([groups][pre_bpg][C][2] doubles: sum dz, sum dz xhat over the ReLU-masked gradient) were already
    // produced by the kernels that wrote dz (head_train.hip: head_bwd2 behind a deferred BatchNorm) -- no pass over dz / y for them here
    const int rpb = bn_rpb(rows, groups);
    const int bpg = pre_part ? pre_bpg : (int)((rows + rpb - 1) / rpb);
    // a ReLU behind a BatchNorm WITHOUT a residual: z > 0 <=> bn_out(y) > 0, recomputed from the y both passes read anyway (relu = 2):
    // 5 instead of 7 tensor passes over the activation
    if (relu && !dresid && beta) relu = 2;
    if (!pre_part)
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(groups * bpg), dim3(256), 256 * 2 * sizeof(double), st, dz, z, y, group_stride, rows, C, bpg, mean, rstd, relu,
                           gamma, beta, (double*)ws, rpb);
    {
        const double* part = pre_part ? pre_part : (const double*)ws;
        int blocks = bpg;
        if (blocks > 256) {
            double* fold = pre_part ? (double*)ws : (double*)((char*)ws + bn_partial_bytes(groups, bpg, C));  // (pre_part: ws holds no partials of its own)
            const int per = (blocks + 255) / 256, b2 = (blocks + per - 1) / per;
            const long long total = (long long)groups * b2 * C;
            hipLaunchKernelGGL(bn_partial_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, fold, blocks, b2, per, C, groups);
            part = fold;
            blocks = b2;
        }
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, part, C, blocks, dgamma, dbeta);
    }
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;
    const int nrl = 256 / (C / 4);
    long long ablocks = (rows + nrl - 1) / nrl;
    const long long cap = std::max(1ll, (256ll * 32) / std::max(1, groups));
    if (ablocks > cap) ablocks = cap;
    const dim3 agrid((unsigned)std::max(1ll, ablocks), (unsigned)groups);
    if (dy_assign) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, agrid, dim3(256), 0, st, dz, z, y, dy, dresid, group_stride, rows, C,
                                      groups, mean, rstd, gamma, beta, dgamma, dbeta, relu, eval_mask, dresid_assign);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, agrid, dim3(256), 0, st, dz, z, y, dy, dresid, group_stride, rows, C, groups,
                            mean, rstd, gamma, beta, dgamma, dbeta, relu, eval_mask, dresid_assign);
    return hipGetLastError();
}
// dgamma / dbeta from [blocks][C][2] double partials (sum dz xhat -> dgamma, sum dz -> dbeta): head_train.hip's first backward pass
void cerb_bn_bwd_finalize_launch(const double* partial, int C, int blocks, float* dgamma, float* dbeta, hipStream_t st) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64, 1), dim3(1024), 0, st, partial, C, blocks, dgamma, dbeta);
}
hipError_t cerb_launch_conv_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int G, int N, int H, int W, int Cin, int Cout,
                                int ks, int stride, long long x_gs, hipStream_t st) {
    const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
    if (dx && ks == 1 && G == 1 && Cout % 4 == 0 && Cout * 32 <= 64 * 1024 && (stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0))) {
        const long long nrb = ((long long)N * Ho * Wo + 7) / 8;
        hipLaunchKernelGGL(conv1x1_dgrad8_kernel, dim3((unsigned)std::min<long long>(nrb, 256 * 32)), dim3(Cin >= 256 ? 256 : (Cin >= 128 ? 128 : 64)), (size_t)8 * Cout * sizeof(float), st, dy, w, dx, N, H, W, Cin,
                           Cout, stride);
    } else if (dx && ks == 1 && stride == 2 && G == 1 && H % 2 == 0 && W % 2 == 0)
        hipLaunchKernelGGL(conv1x1s2_dgrad_kernel, dim3(gridfor((long long)N * Ho * Wo * Cin)), dim3(256), 0, st, dy, w, dx, N, H, W, Cin, Cout);
    else if (dx) hipLaunchKernelGGL(conv_dgrad_kernel, dim3(gridfor((long long)G * N * H * W * Cin)), dim3(256), 0, st, dy, w, dx, G, N, H, W, Cin, Ho, Wo, Cout, ks, stride, x_gs);
    if (dw) hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)((long long)G * Cout * Cin)), dim3(256), 0, st, x, dy, dw, N, H, W, Cin, Ho, Wo, Cout, ks, stride, x_gs);
    if (db) hipLaunchKernelGGL(colsum_kernel, dim3(G * Cout), dim3(256), 0, st, dy, (long long)N * Ho * Wo * Cout, (long long)N * Ho * Wo, Cout, db);
    return hipGetLastError();
}
hipError_t cerb_launch_stem_wgrad(const unsigned char* tiles, const float* dy, float* dw, int N, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(64 * 147), dim3(256), 0, st, tiles, dy, dw, N, H, W);
    return hipGetLastError();
}
hipError_t cerb_launch_maxpool_bwd(const float* x, const float* ypool, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(gridfor((long long)N * H * W * (C / 4))), dim3(256), 0, st, x, ypool, dy, dx, N, H, W, C, H / 2, W / 2);
    return hipGetLastError();
}
hipError_t cerb_launch_maxpool_bwd_idx(const unsigned* idx, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(gridfor((long long)N * H * W * (C / 4))), dim3(256), 0, st, idx, dy, dx, N, H, W, C, H / 2, W / 2);
    return hipGetLastError();
}
bool cerb_upadd_bwd_fused_ok(int H, int W, int C, int G) { return C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && G <= 32 && !cerb_dev_getenv("CERB_UPADD_BWD_TWO_PASS"); }
hipError_t cerb_launch_upadd_bwd(const float* dout, float* dskip, float* dprev, int G, int N, int H, int W, int C, long long prev_gs, int shared_prev, hipStream_t st,
                                 unsigned group_mask, int skip_assign, int prev_assign) {
    if (cerb_upadd_bwd_fused_ok(H, W, C, G)) {
        hipLaunchKernelGGL(upadd_bwd_fused_kernel, dim3(gridfor((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, st, dout, dskip, dprev, G, N, H, W, C, prev_gs, shared_prev,
                           group_mask, skip_assign, prev_assign);
        return hipGetLastError();
    }
    if (group_mask != 0xffffffffu || skip_assign || prev_assign) return hipErrorInvalidValue;  // (the two-pass form accumulates all groups: the caller prepares the buffers)
    hipLaunchKernelGGL(upadd_bwd_skip_kernel, dim3(gridfor((long long)N * H * W * C)), dim3(256), 0, st, dout, dskip, G, (long long)N * H * W * C);
    hipLaunchKernelGGL(upadd_bwd_prev_kernel, dim3(gridfor((long long)N * (H / 2) * (W / 2) * (C / 4) * (shared_prev ? 1 : G))), dim3(256), 0, st, dout, dprev, G, N, H, W, C,
                       prev_gs, shared_prev);
    return hipGetLastError();
}
hipError_t cerb_launch_pointwise_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long rows, int cin, int cout,
                                     const float* in_scale, int dx_assign, hipStream_t st) {
    if (dx && !in_scale && rows >= 4096 && cerb_launch_pw_mfma(dy, w, 0, nullptr, dx, rows, cout, cin, dx_assign ? 0 : 1, st) == hipSuccess) dx = nullptr;
    if (dx) hipLaunchKernelGGL(pointwise_dgrad_kernel, dim3(gridfor(rows * cin)), dim3(256), 0, st, dy, w, dx, rows, cin, cout, in_scale, dx_assign);
    if (dw) hipLaunchKernelGGL(pointwise_wgrad_kernel, dim3(cin * cout), dim3(256), 0, st, x, dy, dw, rows, cin, cout, in_scale);
    if (db) hipLaunchKernelGGL(colsum_kernel, dim3(cout), dim3(256), 0, st, dy, 0ll, rows, cout, db);
    return hipGetLastError();
}
size_t cerb_pw_wgrad_small_workspace_bytes(long long rows, int cin, int cout) { return (size_t)((rows + 511) / 512) * cin * cout * 4 + 256; }
hipError_t cerb_launch_pw_wgrad_small(const float* x, const float* dy, float* dw, long long rows, int cin, int cout, void* ws, hipStream_t st) {
    if (cout > 8) return hipErrorInvalidValue;
    const int blocks = (int)((rows + 511) / 512);
    hipLaunchKernelGGL(pw_wgrad_small_partial_kernel, dim3(blocks), dim3(128), 0, st, x, dy, (float*)ws, rows, cin, cout, 512ll);
    (void)cerb_launch_slab_sum((const float*)ws, dw, cin * cout, blocks, 1, st);
    return hipGetLastError();
}
size_t cerb_pw_bwd_small_workspace_bytes(long long rows, int cin, int cout) { return (size_t)((rows + 511) / 512) * (cin + 1) * cout * 4 + 512; }
// dx (assigned or accumulated), dw and db of a pointwise layer with <= 8 outputs, one pass over x / dy / dx; partials per 512-row slab added in slab order
hipError_t cerb_launch_pw_bwd_small(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long rows, int cin, int cout,
                                    int dx_assign, void* ws, hipStream_t st) {
    if (cout > 8 || !dx || !dw || !db) return hipErrorInvalidValue;
    const int blocks = (int)((rows + 511) / 512);
    float* part = (float*)ws;
    float* part_b = part + (size_t)blocks * cin * cout;
    hipLaunchKernelGGL(pw_bwd_small_fused_kernel, dim3(blocks), dim3(128), 0, st, x, dy, w, dx, part, part_b, rows, cin, cout, 512ll, dx_assign);
    (void)cerb_launch_slab_sum(part, dw, cin * cout, blocks, 1, st);
    (void)cerb_launch_slab_sum(part_b, db, cout, blocks, 1, st);
    return hipGetLastError();
}
hipError_t cerb_launch_dilate2(const float* dy, float* d, long long n, int H, int W, int C, hipStream_t st) {
    if (C % 4 || H % 2 || W % 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dilate2_kernel, dim3(gridfor(n * H * W * (C / 4))), dim3(256), 0, st, dy, d, n, H, W, C / 4);
    return hipGetLastError();
}
hipError_t cerb_launch_crop_gap_bwd(const float* dg, float* dx, int N, int H, int W, int C, int y0, int ch, int x0, int cw, hipStream_t st) {
    hipLaunchKernelGGL(crop_gap_bwd_kernel, dim3(gridfor((long long)N * ch * cw * C)), dim3(256), 0, st, dg, dx, N, H, W, C, y0, ch, x0, cw);
    return hipGetLastError();
}
// count copies dst[i] <- src[i] (n[i] floats) in one launch.  The descriptor table lives in *dev_tab (grown on demand) and is uploaded only when
// it differs from the previous call's (host_prev): the optimiser's parameter views and the handle's slots are the same pointers every step.
hipError_t cerb_launch_copy_multi(int count, float* const* dst, const float* const* src, const long long* n, void** dev_tab, size_t* dev_bytes,
                                  std::vector<char>* host_prev, hipStream_t st) {
    std::vector<CopyDesc> dd(count);
    std::vector<int2> ch;
    for (int i = 0; i < count; ++i) {
        dd[i] = CopyDesc{dst[i], src[i], n[i]};
        for (long long c = 0; c * COPY_CHUNK < n[i]; ++c) ch.push_back(make_int2(i, (int)c));
    }
    if (ch.empty()) return hipSuccess;
    const size_t tb = (dd.size() * sizeof(CopyDesc) + 255) & ~(size_t)255, need = tb + ch.size() * sizeof(int2);
    std::vector<char> host(need, 0);
    memcpy(host.data(), dd.data(), dd.size() * sizeof(CopyDesc));
    memcpy(host.data() + tb, ch.data(), ch.size() * sizeof(int2));
    hipError_t e;
    if (need > *dev_bytes || host != *host_prev) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;  // an earlier launch may still be reading the old table
        if (need > *dev_bytes) {
            if (*dev_tab) (void)hipFree(*dev_tab);
            if ((e = hipMalloc(dev_tab, need * 2)) != hipSuccess) return e;
            *dev_bytes = need * 2;
        }
        if ((e = hipMemcpy(*dev_tab, host.data(), need, hipMemcpyHostToDevice)) != hipSuccess) return e;
        host_prev->swap(host);
    }
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)ch.size()), dim3(256), 0, st, (const CopyDesc*)*dev_tab, (const int2*)((const char*)*dev_tab + tb));
    return hipGetLastError();
}
// tables live in one device buffer that grows on demand and is reused by later steps (single optimiser stream assumed, as torch's own)
hipError_t cerb_launch_adam_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* n, float lr, float b1,
                                  float b2, float eps, int step, hipStream_t st) {
    struct Tab {
        void* dev = nullptr;
        size_t bytes = 0;
        std::vector<char> host;
    };
    // One table per (device, stream): a launch still reading its table on ANOTHER stream is never overwritten -- the synchronisations below
    // cover this stream only --, and host threads driving different handles go through the mutex.  (Two optimisers alternating on one stream
    // share a table and re-upload it every step: correct, not fast; torch's own optimisers are one per stream too.)
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Tab> tabs;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    Tab& tab = tabs[std::make_pair(devid, st)];
    void*& dev_tab = tab.dev;
    size_t& dev_bytes = tab.bytes;
    std::vector<char>& host = tab.host;
    std::vector<AdamTensor> tt(count);
    std::vector<int2> ch;
    for (int i = 0; i < count; ++i) {
        tt[i] = AdamTensor{p[i], g[i], m[i], v[i], n[i]};
        for (long long c = 0; c * ADAM_CHUNK < n[i]; ++c) ch.push_back(make_int2(i, (int)c));
    }
    if (ch.empty()) return hipSuccess;
    const size_t tb = (tt.size() * sizeof(AdamTensor) + 255) & ~(size_t)255, need = tb + ch.size() * sizeof(int2);
    hipError_t e;
    if (need > dev_bytes) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if (dev_tab) (void)hipFree(dev_tab);
        if ((e = hipMalloc(&dev_tab, need * 2)) != hipSuccess) return e;
        dev_bytes = need * 2;
        host.clear();
    }
    // the optimiser's tensors are the same pointers every step: the table is uploaded only when it differs from the one on the device (as
    // cerb_launch_copy_multi does).  Round 4 synchronised the stream and uploaded it every step -- the host stood still until the whole backward pass
    // had drained and the device then waited for the host to queue the optimiser, the running statistics and the re-pack.
    std::vector<char> now(need);
    memcpy(now.data(), tt.data(), tt.size() * sizeof(AdamTensor));
    memcpy(now.data() + tb, ch.data(), ch.size() * sizeof(int2));
    if (now != host) {
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;  // an earlier launch may still be reading the old table
        host.swap(now);
        if ((e = hipMemcpyAsync(dev_tab, host.data(), need, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;  // `host` may be rebuilt by the next call
    }
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)ch.size()), dim3(256), 0, st, (const AdamTensor*)dev_tab, (const int2*)((const char*)dev_tab + tb), lr, b1,
                       b2, eps, bc1, bc2);
    return hipGetLastError();
}
hipError_t cerb_launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step, hipStream_t st) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(gridfor(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2);
    return hipGetLastError();
}
