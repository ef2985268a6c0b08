// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the Cerberus tile path, every nn.Conv2d + eval-BatchNorm2d + ReLU triple of
//   reference models/backbone/resnet.py:81-97 (BasicBlock, incl. 1x1 stride-2 downsample and the residual add)
//   reference models/utils/conv_layers.py:24-60 (_ConvLayer, post-activation) used by the decoders
//   reference models/net_desc.py:52,153 (conv_map) and :185-188 (upsample2x + skip add, fused as MODE 1)
//
// Data layout: activations NHWC fp32; weights pre-packed (BN folded) by pack_conv_weights() in cerb_api.hip.
// One workgroup = 256 threads = 4 waves computes a TH x TW output tile (256 pixels) x 64 output channels.
// GEMM is "swapped": D[cout][pixel] += W[cout][k] * X[k][pixel]  (A = weights, B = pixels) so each lane owns
// ONE pixel and 4 consecutive couts per accumulator quad.  K order inside an 8-channel group is permuted so
// that one 16-byte read (LDS for pixels, global for weights) feeds 4 consecutive MFMA k-steps:
//   k-slot h (= lane>>5) at step t  <->  channel  g*8 + 4*h + t.
//
// Input halo tile is staged through registers into LDS in chunks of CB channels (pixel stride CB+4 floats keeps
// the 16-lane groups of ds_read_b128 on distinct 16-B slots).  Weights go global(L2) -> VGPR directly, one tap
// ahead of use: they are tiny, identical for all workgroups of a cout-block, and fp32 MFMA (64 cycles per
// instruction) leaves ~4k cycles per tap to hide the latency.
#include "cerb_common.h"

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int PS = CB + 4;           // LDS pixel stride in floats
    static constexpr int NG = CB / 8;           // 8-channel groups per chunk
    static constexpr int T = KS * KS;           // taps
    static constexpr int LDS_FLOATS = IH * IW * PS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p) {
    using C = ConvCfg<KS, STRIDE, TH, TW, CB, MODE>;
    static_assert(TH * TW == 256, "tile must hold 256 pixels (4 waves x 2 x 32)");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;   // pixel within the 32-pixel subtile / cout within the 32-cout subtile
    const int h = lane >> 5;   // k-slot

    // ---- block -> (group, image, tile, cout-block) -------------------------------------------------
    const int ncb = p.Cout >> 6;
    const int ntile = p.N * p.tiles_y * p.tiles_x;
    const int per_group = ntile * ncb;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int g = L / per_group;
    L -= g * per_group;
    const int cb = L % ncb;
    int t_ = L / ncb;
    const int tx = t_ % p.tiles_x;
    t_ /= p.tiles_x;
    const int ty = t_ % p.tiles_y;
    const int n = t_ / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const float* __restrict__ in = p.in + g * p.in_gs + (long long)n * p.H * p.W * p.Cin;
    const float* __restrict__ wp = p.wpack + g * p.w_gs + (long long)cb * (p.Cin / CB) * C::T * C::NG * 2 * 256;

    // per-lane LDS read base for the two 32-pixel subtiles of this wave
    int ldsb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pl = (wave * 2 + q) * 32 + j;
        const int py = pl / TW, px = pl % TW;
        ldsb[q] = ((py * STRIDE) * C::IW + px * STRIDE) * C::PS + 4 * h;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

    const int nchunk = p.Cin / CB;
    f32x4 wcur[C::NG][2], wnxt[C::NG][2];
    const f32x4* wv = reinterpret_cast<const f32x4*>(wp) + lane;
    // weights of (chunk 0, tap 0)
#pragma unroll
    for (int G = 0; G < C::NG; ++G)
#pragma unroll
        for (int s = 0; s < 2; ++s) wcur[G][s] = wv[(G * 2 + s) * 64];

    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = ch * CB;
        if (ch) __syncthreads();  // previous chunk's LDS reads are done
        // ---- stage the (IH x IW x CB) halo tile ------------------------------------------------------
        {
            constexpr int PARTS = CB / 4;
            constexpr int NF = C::IH * C::IW * PARTS;
            constexpr int ITER = (NF + 255) / 256;
            // batch of loads kept in flight before the LDS writes: everything for the plain mode, 2 iterations
            // (10 float4) for the upsample mode whose 5 loads per element would otherwise spill
            constexpr int UB = (MODE == 1) ? 2 : ITER;
#pragma unroll 1
            for (int it0 = 0; it0 < ITER; it0 += UB) {
                f32x4 v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int f = tid + (it0 + u) * 256;
                    const int pix = f / PARTS, part = f % PARTS;
                    const int iy = pix / C::IW, ix = pix % C::IW;
                    const int gy = oy0 * STRIDE - C::PAD + iy, gx = ox0 * STRIDE - C::PAD + ix;
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (f < NF && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                        const long long off = ((long long)gy * p.W + gx) * p.Cin + c0 + part * 4;
                        x = *reinterpret_cast<const f32x4*>(in + off);
                        if (MODE == 1) {
                            // fused F.interpolate(prev, scale_factor=2, bilinear, align_corners=False) + skip add
                            // (reference net_layers.py:45-46, net_desc.py:185-188). src = 0.5*(dst+0.5)-0.5 clamped at 0.
                            const int Hp = p.H >> 1, Wp = p.W >> 1;
                            const float* __restrict__ pv = p.prev + g * p.prev_gs + (long long)n * Hp * Wp * p.Cin + c0 + part * 4;
                            float sy = 0.5f * (gy + 0.5f) - 0.5f, sx = 0.5f * (gx + 0.5f) - 0.5f;
                            sy = sy < 0.f ? 0.f : sy;
                            sx = sx < 0.f ? 0.f : sx;
                            const int y0 = (int)sy, x0 = (int)sx;
                            const int y1 = y0 + (y0 < Hp - 1), x1 = x0 + (x0 < Wp - 1);
                            const float ly = sy - y0, lx = sx - x0;
                            const float hy = 1.f - ly, hx = 1.f - lx;
                            const f32x4 p00 = *reinterpret_cast<const f32x4*>(pv + ((long long)y0 * Wp + x0) * p.Cin);
                            const f32x4 p01 = *reinterpret_cast<const f32x4*>(pv + ((long long)y0 * Wp + x1) * p.Cin);
                            const f32x4 p10 = *reinterpret_cast<const f32x4*>(pv + ((long long)y1 * Wp + x0) * p.Cin);
                            const f32x4 p11 = *reinterpret_cast<const f32x4*>(pv + ((long long)y1 * Wp + x1) * p.Cin);
                            const f32x4 up = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
                            x = x + up;
                        }
                    }
                    v[u] = x;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int f = tid + (it0 + u) * 256;
                    if (f < NF) {
                        const int pix = f / PARTS, part = f % PARTS;
                        *reinterpret_cast<f32x4*>(lds + pix * C::PS + part * 4) = v[u];
                    }
                }
            }
        }
        __syncthreads();

        // ---- taps ------------------------------------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap) {
            // prefetch the next tap's (or next chunk's first tap's) weights
            {
                const bool last = (tap == C::T - 1);
                const int nt = last ? 0 : tap + 1;
                const int nc = last ? ch + 1 : ch;
                if (nc < nchunk) {
                    const f32x4* src = wv + (long long)(nc * C::T + nt) * C::NG * 2 * 64;
#pragma unroll
                    for (int G = 0; G < C::NG; ++G)
#pragma unroll
                        for (int s = 0; s < 2; ++s) wnxt[G][s] = src[(G * 2 + s) * 64];
                }
            }
            const int ky = tap / KS, kx = tap % KS;
            const int toff = (ky * C::IW + kx) * C::PS;
#pragma unroll
            for (int G = 0; G < C::NG; ++G) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + ldsb[0] + toff + G * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + ldsb[1] + toff + G * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[G][0][t], b0[t], acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[G][1][t], b0[t], acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[G][0][t], b1[t], acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[G][1][t], b1[t], acc[1][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int G = 0; G < C::NG; ++G)
#pragma unroll
                for (int s = 0; s < 2; ++s) wcur[G][s] = wnxt[G][s];
        }
    }

    // ---- epilogue: + bias (+ residual) -> ReLU -> float4 NHWC stores ----------------------------------
    const float* __restrict__ bias = p.bias + g * p.bias_gs + cb * 64;
    float* __restrict__ out = p.out + g * p.out_gs;
    const float* __restrict__ resid = p.resid ? p.resid + g * p.resid_gs : nullptr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pl = (wave * 2 + q) * 32 + j;
        const int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const long long pixoff = (((long long)n * p.Ho + oy) * p.Wo + ox) * p.Cout + cb * 64;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int co = s * 32 + rq * 8 + h * 4;
                f32x4 v = {acc[s][q][rq * 4 + 0], acc[s][q][rq * 4 + 1], acc[s][q][rq * 4 + 2], acc[s][q][rq * 4 + 3]};
                v = v + *reinterpret_cast<const f32x4*>(bias + co);
                if (resid) v = v + *reinterpret_cast<const f32x4*>(resid + pixoff + co);
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f);
                    v[1] = fmaxf(v[1], 0.f);
                    v[2] = fmaxf(v[2], 0.f);
                    v[3] = fmaxf(v[3], 0.f);
                }
                *reinterpret_cast<f32x4*>(out + pixoff + co) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host-side launcher (called from cerb_api.hip)
// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
static hipError_t launch_cfg(ConvParams p, hipStream_t st) {
    using C = ConvCfg<KS, STRIDE, TH, TW, CB, MODE>;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const long long nblk = (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    auto kern = conv_igemm_kernel<KS, STRIDE, TH, TW, CB, MODE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, st, p);
    return hipGetLastError();
}

// Chunk size (channels staged per LDS pass) per kernel family -- also used by the weight packer.
extern "C" int cerb_conv_chunk(int ks, int stride) { return (stride == 2) ? 16 : 32; }

hipError_t cerb_launch_conv(const ConvParams& p, int ks, int stride, int mode, hipStream_t st) {
    const bool small = p.Wo < 32;  // 16x16 tiles for the deepest levels (16^2 / 28^2 maps)
    if (ks == 3 && stride == 1 && mode == 0) return small ? launch_cfg<3, 1, 16, 16, 32, 0>(p, st) : launch_cfg<3, 1, 8, 32, 32, 0>(p, st);
    if (ks == 3 && stride == 1 && mode == 1) return small ? launch_cfg<3, 1, 16, 16, 32, 1>(p, st) : launch_cfg<3, 1, 8, 32, 32, 1>(p, st);
    if (ks == 3 && stride == 2 && mode == 0) return small ? launch_cfg<3, 2, 16, 16, 16, 0>(p, st) : launch_cfg<3, 2, 8, 32, 16, 0>(p, st);
    if (ks == 1 && stride == 1 && mode == 0) return small ? launch_cfg<1, 1, 16, 16, 32, 0>(p, st) : launch_cfg<1, 1, 8, 32, 32, 0>(p, st);
    if (ks == 1 && stride == 2 && mode == 0) return small ? launch_cfg<1, 2, 16, 16, 16, 0>(p, st) : launch_cfg<1, 2, 8, 32, 16, 0>(p, st);
    return hipErrorInvalidValue;
}
