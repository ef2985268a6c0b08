// Non-3x3 kernels of the Cerberus network path for gfx950:
//   stem_conv7x7_kernel  : uint8 tile -> /255 -> 7x7 s1 p3 conv (3->64) + BN + ReLU   (reference resnet.py:195-197,276-278;
//                          net_desc.py:147 for the /255)
//   maxpool3x3s2_kernel  : MaxPool2d(3, stride 2, pad 1)                              (reference resnet.py:201,280)
//   head_kernel          : 1x1 64->96 +BN+ReLU -> 1x1 96->out -> softmax -> crop -> INST probs / TYPE argmax, written
//                          straight into the destination canvas (reference net_layers.py:31-38, run_desc.py:451-491)
//   patch_class_kernel   : centre-crop 9x9 -> avg-pool -> BN-ReLU-1x1-BN-ReLU-1x1 -> argmax(softmax) -> broadcast
//                          (reference net_desc.py:169-180, run_desc.py:457-487)
#include "cerb_common.h"

// ---------------------------------------------------------------------------------------------------------------
// Stem: implicit GEMM with K = 7 rows x 24 (= 7 taps x 3 ch = 21, zero-padded to 24), swapped GEMM as in conv_igemm.
// B operand (pixels) is read from an LDS tile of floats laid out [row][col*3 + c]; for tap-row ky the 21 values a pixel
// needs are contiguous, so k-step t / k-slot h reads LDS[(py+ky)][px*3 + 2t + h].
// ---------------------------------------------------------------------------------------------------------------
struct StemParams {
    const unsigned char* tiles;  // [N][H][W][3]
    const float* tiles_f32;      // alternative input (tiles == nullptr): [N][H][W][3] float pixel values, divided by 255 like the bytes (net_desc.py:147)
    const float* wpack;          // [7 ky][12 t][2 s][64 lane]
    const float* bias;           // [64]
    float* out;                  // [N][H][W][64]
    int N, H, W, tiles_x, tiles_y;
    int relu;                    // 1: inference (BN folded into wpack / bias, ReLU fused); 0: train mode (raw conv, BN and ReLU follow)
};

template <bool F32IN>
__global__ __launch_bounds__(256, 2) void stem_conv7x7_kernel(StemParams p) {
    constexpr int TH = 8, TW = 32, IH = TH + 6, IW = TW + 6, ROW = 120;  // ROW >= IW*3 + 3 (k padding)
    __shared__ __attribute__((aligned(16))) float wl[7 * 12 * 2 * 64];
    __shared__ __attribute__((aligned(16))) float xl[IH * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    for (int i = tid; i < 7 * 12 * 2 * 64; i += 256) wl[i] = p.wpack[i];
    const int ntile = p.N * p.tiles_y * p.tiles_x;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int t_ = tile;
        const int tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y;
        const int n = t_ / p.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        const unsigned char* img = F32IN ? nullptr : p.tiles + (long long)n * p.H * p.W * 3;
        const float* imgf = F32IN ? p.tiles_f32 + (long long)n * p.H * p.W * 3 : nullptr;
        for (int i = tid; i < IH * ROW; i += 256) {
            const int iy = i / ROW, c = i % ROW;
            const int gy = oy0 - 3 + iy, gx = ox0 - 3 + c / 3;
            float v = 0.f;
            if (c < IW * 3 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                const long long at = ((long long)gy * p.W + gx) * 3 + c % 3;
                v = (F32IN ? imgf[at] : (float)img[at]) / 255.0f;
            }
            xl[i] = v;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;
        const int b0 = (wave * 2 + 0) * ROW + j * 3 + h;  // subtile q = row (wave*2+q), pixel col j
        const int b1 = (wave * 2 + 1) * ROW + j * 3 + h;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const float a0 = wl[((ky * 12 + t) * 2 + 0) * 64 + lane];
                const float a1 = wl[((ky * 12 + t) * 2 + 1) * 64 + lane];
                const float x0 = xl[b0 + ky * ROW + 2 * t];
                const float x1 = xl[b1 + ky * ROW + 2 * t];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x1, acc[1][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = oy0 + wave * 2 + q, ox = ox0 + j;
            if (oy >= p.H || ox >= p.W) continue;
            float* o = p.out + (((long long)n * p.H + oy) * p.W + ox) * 64;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = s * 32 + rq * 8 + h * 4;
                    f32x4 v = {acc[s][q][rq * 4 + 0], acc[s][q][rq * 4 + 1], acc[s][q][rq * 4 + 2], acc[s][q][rq * 4 + 3]};
                    v = v + *reinterpret_cast<const f32x4*>(p.bias + co);
                    const float floor_ = p.relu ? 0.f : -3.402823466e38f;
                    v[0] = fmaxf(v[0], floor_);
                    v[1] = fmaxf(v[1], floor_);
                    v[2] = fmaxf(v[2], floor_);
                    v[3] = fmaxf(v[3], floor_);
                    *reinterpret_cast<f32x4*>(o + co) = v;
                }
        }
    }
}

hipError_t cerb_launch_stem(StemParams p, hipStream_t st) {
    p.tiles_x = (p.W + 31) / 32;
    p.tiles_y = (p.H + 7) / 8;
    const long long ntile = (long long)p.N * p.tiles_x * p.tiles_y;
    const unsigned grid = (unsigned)(ntile < 256 * 6 ? ntile : 256 * 6);
    if (p.tiles) hipLaunchKernelGGL(stem_conv7x7_kernel<false>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(stem_conv7x7_kernel<true>, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * (C / 4);
    // Workgroup b runs on XCD b % 8, each with its own L2: give every XCD one CONTIGUOUS eighth of the output (whole images at the batch sizes
    // in use), so that the input rows two neighbouring output rows share are fetched from HBM once, not once per XCD (r04 PMC: 2.5x the input).
    const long long per_xcd = gridDim.x / 8, vb = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    for (long long i = vb * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4));
        long long r = i / (C / 4);
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = oy * 2 - 1 + ky;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int x = ox * 2 - 1 + kx;
                if (x < 0 || x >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((long long)n * H + y) * W + x) * C + c4 * 4);
                m[0] = fmaxf(m[0], v[0]);
                m[1] = fmaxf(m[1], v[1]);
                m[2] = fmaxf(m[2], v[2]);
                m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
    }
}

// Training forward: the same pooling that also records, per output element, WHICH of its nine window positions (3 ky + kx, row-major) held the first
// maximum -- what torch.nn.functional.max_pool2d keeps as its indices (strict comparison in scan order).  One byte per element (51 MB at 16 x 448^2):
// the backward pass (train_kernels.hip: maxpool_bwd_idx_kernel) routes gradients by it and reads neither the input nor the pooled map again.
__global__ void maxpool3x3s2_idx_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * (C / 4);
    const long long per_xcd = gridDim.x / 8, vb = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    for (long long i = vb * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4));
        long long r = i / (C / 4);
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = oy * 2 - 1 + ky;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int x = ox * 2 - 1 + kx;
                if (x < 0 || x >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((long long)n * H + y) * W + x) * C + c4 * 4);
                const unsigned k = (unsigned)(ky * 3 + kx);
                // the value is the plain kernel's (fmaxf); the position moves only on a strictly larger element, the first valid one starts it
                if (first || v[0] > m[0]) w0 = k;
                if (first || v[1] > m[1]) w1 = k;
                if (first || v[2] > m[2]) w2 = k;
                if (first || v[3] > m[3]) w3 = k;
                first = false;
                m[0] = fmaxf(m[0], v[0]);
                m[1] = fmaxf(m[1], v[1]);
                m[2] = fmaxf(m[2], v[2]);
                m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
        idx[i] = w0 | (w1 << 8) | (w2 << 16) | (w3 << 24);
    }
}

hipError_t cerb_launch_maxpool_idx(const float* in, float* out, unsigned* idx, int N, int H, int W, int C, hipStream_t st) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    long long blocks = ((total + 255) / 256 + 7) / 8 * 8;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(maxpool3x3s2_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, idx, N, H, W, C, Ho, Wo);
    return hipGetLastError();
}

hipError_t cerb_launch_maxpool(const float* in, float* out, int N, int H, int W, int C, hipStream_t st) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    long long blocks = ((total + 255) / 256 + 7) / 8 * 8;  // a multiple of the 8 XCDs (the kernel's block -> output mapping relies on it)
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, N, H, W, C, Ho, Wo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Decoder entry: out[g][n] = skip[n] + upsample2x(prev[g][n])   (reference net_layers.py:45-46 bilinear, align_corners=False;
// net_desc.py:185-188).  HBM-bound elementwise pass feeding the Winograd conv (conv_wino.hip).  One thread = one 2x2 output
// block x 4 channels: rows 2m+1, 2m+2 and columns 2n+1, 2n+2 blend exactly prev rows m, m+1 / columns n, n+1 with weights
// {0.75, 0.25}; m = -1 and m = Hp-1 are the clamped edge blocks (one of their two rows lies outside the image).
// ---------------------------------------------------------------------------------------------------------------
// a x + b y with the rounding spelled out -- fma(a, x, b * y) -- so that upsample2_add_kernel and upsample2_add_planar_kernel (two
// different instruction streams for the same blend) cannot be contracted differently by the compiler: their values are bit-identical.
__device__ __forceinline__ f32x4 blend2(float a, const f32x4& x, float b, const f32x4& y) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(a, x[e], b * y[e]);
    return r;
}
__global__ void upsample2_add_kernel(const float* __restrict__ skip, const float* __restrict__ prev, float* __restrict__ out, int groups,
                                     int N, int H, int W, int C, long long prev_gs, int bm_lo, int bm_cnt, int bn_lo, int bn_cnt) {
    // 2x2 output blocks (bm, bn), bm in [-1, H/2): the launcher restricts them to [bm_lo, bm_lo + bm_cnt) x [bn_lo, bn_lo + bn_cnt)
    const int Hp = H >> 1, Wp = W >> 1, C4 = C >> 2;
    const long long total = (long long)N * bm_cnt * bn_cnt * C4;
    const long long out_gs = (long long)N * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long r = i / C4;
        const int bn = (int)(r % bn_cnt) + bn_lo;
        r /= bn_cnt;
        const int bm = (int)(r % bm_cnt) + bm_lo;
        const int n = (int)(r / bm_cnt);
        const int m0 = max(bm, 0), m1 = min(bm + 1, Hp - 1), n0 = max(bn, 0), n1 = min(bn + 1, Wp - 1);
        // the skip pixels are shared by every decoder (group): read once, reuse `groups` times
        f32x4 sk[2][2];
        bool ok[2][2];
        long long o[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * bm + 1 + dy, x = 2 * bn + 1 + dx;
                ok[dy][dx] = y >= 0 && y < H && x >= 0 && x < W;
                o[dy][dx] = (((long long)n * H + y) * W + x) * C + c4 * 4;
                if (ok[dy][dx]) sk[dy][dx] = *reinterpret_cast<const f32x4*>(skip + o[dy][dx]);
            }
        const long long pb = (long long)n * Hp * Wp * C + c4 * 4;
        for (int g = 0; g < groups; ++g) {
            const float* pg = prev + g * prev_gs + pb;
            const f32x4 p00 = *reinterpret_cast<const f32x4*>(pg + ((long long)m0 * Wp + n0) * C);
            const f32x4 p01 = *reinterpret_cast<const f32x4*>(pg + ((long long)m0 * Wp + n1) * C);
            const f32x4 p10 = *reinterpret_cast<const f32x4*>(pg + ((long long)m1 * Wp + n0) * C);
            const f32x4 p11 = *reinterpret_cast<const f32x4*>(pg + ((long long)m1 * Wp + n1) * C);
            const f32x4 top_l = blend2(0.75f, p00, 0.25f, p01), top_r = blend2(0.25f, p00, 0.75f, p01);
            const f32x4 bot_l = blend2(0.75f, p10, 0.25f, p11), bot_r = blend2(0.25f, p10, 0.75f, p11);
            const f32x4 u[2][2] = {{blend2(0.75f, top_l, 0.25f, bot_l), blend2(0.75f, top_r, 0.25f, bot_r)},
                                   {blend2(0.25f, top_l, 0.75f, bot_l), blend2(0.25f, top_r, 0.75f, bot_r)}};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    if (ok[dy][dx]) __builtin_nontemporal_store(sk[dy][dx] + u[dy][dx], reinterpret_cast<f32x4*>(out + g * out_gs + o[dy][dx]));
        }
    }
}

// roi = {y0, y1, x0, x1} in output pixels (nullptr or empty = the whole map): only the 2x2 blocks overlapping it are written
hipError_t cerb_launch_upsample2_add(const float* skip, const float* prev, float* out, int groups, int N, int H, int W, int C,
                                     long long prev_gs, const int* roi, hipStream_t st) {
    int bm_lo = -1, bm_hi = H / 2 - 1, bn_lo = -1, bn_hi = W / 2 - 1;  // block bm covers output rows 2 bm + 1, 2 bm + 2
    if (roi && roi[1] > roi[0] && roi[3] > roi[2]) {
        auto lo = [](int v) { return (v - 2 >= 0 ? (v - 2) / 2 : -1); };  // largest bm with 2 bm + 2 <= v  (>= -1)
        bm_lo = lo(roi[0]);
        bn_lo = lo(roi[2]);
        if ((roi[1] - 1) / 2 < bm_hi) bm_hi = (roi[1] - 1) / 2;  // smallest bm with 2 bm + 1 >= y1 - 1 is enough; one more is harmless
        if ((roi[3] - 1) / 2 < bn_hi) bn_hi = (roi[3] - 1) / 2;
    }
    const int bm_cnt = bm_hi - bm_lo + 1, bn_cnt = bn_hi - bn_lo + 1;
    const long long total = (long long)N * bm_cnt * bn_cnt * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(upsample2_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, skip, prev, out, groups, N, H, W, C, prev_gs, bm_lo, bm_cnt, bn_lo,
                       bn_cnt);
    return hipGetLastError();
}

// skip + upsample2x(prev) written in the TILE-PLANAR layout (cerb_common.h) that conv_wino4p.hip reads: thread = (block, 16-channel plane =
// wave, 4x4 tile m, channel quad) keeps the tile's 16 skip quads and the 4 x 4 source pixels of `prev` its 16 outputs blend, and every
// store instruction of a wave is one contiguous 1-KiB row (pixel position (i, j) of the 16 tiles).  Same expressions, same order as
// upsample2_add_kernel: the values are bit-identical.  Pixels of edge blocks beyond the image are not written (they stay zero).
template <bool PREV_PLANAR>  // `prev` (half resolution) is NHWC, or tile-planar itself (the level below also runs on conv_wino4p.hip)
__global__ __launch_bounds__(256) void upsample2_add_planar_kernel(const float* __restrict__ skip, const float* __restrict__ prev, float* __restrict__ out,
                                                                   int groups, int N, int H, int W, long long prev_gs, long long out_gs, int byp, int bxp,
                                                                   int by_lo, int by_cnt, int bx_lo, int bx_cnt) {
    constexpr int C = 64, NCC = 4;
    const int Hp = H >> 1, Wp = W >> 1;
    const int lane = threadIdx.x & 63, cc = threadIdx.x >> 6, m = lane >> 2, cq = lane & 3, ty = m >> 2, tx = m & 3;
    const int nblk = N * by_cnt * bx_cnt;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int bx = blk % bx_cnt + bx_lo;
        const int r_ = blk / bx_cnt;
        const int by = r_ % by_cnt + by_lo, n = r_ / by_cnt;
        const int y0 = 16 * by + 4 * ty, x0 = 16 * bx + 4 * tx;
        if (y0 >= H || x0 >= W) continue;  // the whole tile lies beyond the image (H, W are multiples of 4: a tile is inside or outside)
        const int ch = cc * 16 + cq * 4;
        f32x4 sk[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sk[i][j] = *reinterpret_cast<const f32x4*>(skip + (((long long)n * H + y0 + i) * W + x0 + j) * C + ch);
        // source rows 2Y - 1 .. 2Y + 2 of tile row Y = y0 / 4 (clamped like the index clamp of align_corners = False), columns alike
        int rr[4], qq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rr[k] = min(max((y0 >> 1) - 1 + k, 0), Hp - 1);
            qq[k] = min(max((x0 >> 1) - 1 + k, 0), Wp - 1);
        }
        float* ob = out + ((((long long)n * byp + by + 1) * bxp + bx + 1) * NCC + cc) * 4096 + lane * 4;
        long long pcol[4];  // PREV_PLANAR: the four source columns' offsets inside a block row (block column, tile column, pixel column)
        if constexpr (PREV_PLANAR) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pcol[k] = (long long)((qq[k] >> 4) + 1) * NCC * 4096 + ((qq[k] & 3) * 16 + ((qq[k] >> 2) & 3)) * 16;
        }
        const int pbyp = cerb_planar_blocks(Hp), pbxp = cerb_planar_blocks(Wp);
        for (int g = 0; g < groups; ++g) {
            const float* pg = PREV_PLANAR ? prev + g * prev_gs + (long long)cc * 4096 + cq * 4 : prev + g * prev_gs + (long long)n * Hp * Wp * C + ch;
            f32x4 h[4][4];  // h[source row k][output column j]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 p0, p1, p2, p3;
                if constexpr (PREV_PLANAR) {  // cerb_planar_offset(n, rr[k], qq[.], cc, ...) split into its row and column parts
                    const float* pr = pg + (((long long)n * pbyp + (rr[k] >> 4) + 1) * pbxp) * NCC * 4096 + (((rr[k] & 3) << 2) * 16 + (((rr[k] >> 2) & 3) << 2)) * 16;
                    p0 = *reinterpret_cast<const f32x4*>(pr + pcol[0]); p1 = *reinterpret_cast<const f32x4*>(pr + pcol[1]);
                    p2 = *reinterpret_cast<const f32x4*>(pr + pcol[2]); p3 = *reinterpret_cast<const f32x4*>(pr + pcol[3]);
                } else {
                    const float* pr = pg + (long long)rr[k] * Wp * C;
                    p0 = *reinterpret_cast<const f32x4*>(pr + (long long)qq[0] * C); p1 = *reinterpret_cast<const f32x4*>(pr + (long long)qq[1] * C);
                    p2 = *reinterpret_cast<const f32x4*>(pr + (long long)qq[2] * C); p3 = *reinterpret_cast<const f32x4*>(pr + (long long)qq[3] * C);
                }
                h[k][0] = blend2(0.25f, p0, 0.75f, p1);
                h[k][1] = blend2(0.75f, p1, 0.25f, p2);
                h[k][2] = blend2(0.25f, p1, 0.75f, p2);
                h[k][3] = blend2(0.75f, p2, 0.25f, p3);
            }
            float* og = ob + g * out_gs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 u0 = blend2(0.25f, h[0][j], 0.75f, h[1][j]), u1 = blend2(0.75f, h[1][j], 0.25f, h[2][j]), u2 = blend2(0.25f, h[1][j], 0.75f, h[2][j]),
                            u3 = blend2(0.75f, h[2][j], 0.25f, h[3][j]);
#ifndef UPP_PLAIN_STORE
                __builtin_nontemporal_store(sk[0][j] + u0, reinterpret_cast<f32x4*>(og + (0 * 4 + j) * 256));
                __builtin_nontemporal_store(sk[1][j] + u1, reinterpret_cast<f32x4*>(og + (1 * 4 + j) * 256));
                __builtin_nontemporal_store(sk[2][j] + u2, reinterpret_cast<f32x4*>(og + (2 * 4 + j) * 256));
                __builtin_nontemporal_store(sk[3][j] + u3, reinterpret_cast<f32x4*>(og + (3 * 4 + j) * 256));
#else
                *reinterpret_cast<f32x4*>(og + (0 * 4 + j) * 256) = sk[0][j] + u0;
                *reinterpret_cast<f32x4*>(og + (1 * 4 + j) * 256) = sk[1][j] + u1;
                *reinterpret_cast<f32x4*>(og + (2 * 4 + j) * 256) = sk[2][j] + u2;
                *reinterpret_cast<f32x4*>(og + (3 * 4 + j) * 256) = sk[3][j] + u3;
#endif
            }
        }
    }
}

// roi = {y0, y1, x0, x1} in output pixels (nullptr or empty = the whole map): the 16 x 16 blocks overlapping it are written whole
hipError_t cerb_launch_upsample2_add_planar(const float* skip, const float* prev, float* out, int groups, int N, int H, int W, int C, long long prev_gs,
                                            long long out_gs, const int* roi, int prev_planar, hipStream_t st) {
    if (C != 64 || (H & 3) || (W & 3)) return hipErrorInvalidValue;
    int by_lo = 0, by_hi = (H + 15) / 16, bx_lo = 0, bx_hi = (W + 15) / 16;
    if (roi && roi[1] > roi[0] && roi[3] > roi[2]) {
        by_lo = roi[0] / 16; by_hi = (roi[1] + 15) / 16;
        bx_lo = roi[2] / 16; bx_hi = (roi[3] + 15) / 16;
    }
    const long long nblk = (long long)N * (by_hi - by_lo) * (bx_hi - bx_lo);
    if (nblk <= 0 || nblk >= (1ll << 31)) return hipErrorInvalidValue;
#ifndef UPP_GRID
#define UPP_GRID 16
#endif
    long long blocks = nblk < 256 * UPP_GRID ? nblk : 256 * UPP_GRID;
    auto kern = prev_planar ? upsample2_add_planar_kernel<true> : upsample2_add_planar_kernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, skip, prev, out, groups, N, H, W, prev_gs, out_gs,
                       cerb_planar_blocks(H), cerb_planar_blocks(W), by_lo, by_hi - by_lo, bx_lo, bx_hi - bx_lo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Output head.  Everything stays in registers: GEMM1 is swapped (hidden x pixel) so its accumulators ARE the B
// operand of GEMM2 (k-slot h at step (blk,r) <-> hidden unit blk*32 + (r&3) + 8*(r>>2) + 4*h).
// ---------------------------------------------------------------------------------------------------------------
#ifndef HEAD_TPW
#define HEAD_TPW 2  // 64-pixel tasks per wave
#endif
typedef float f32x4h __attribute__((ext_vector_type(4)));
// The data-aware precision guard (cerb_forward_io.logit_absmax): a lane keeps the largest |logit| of the pixels it finishes; at the end of its
// walk the wave reduces and ONE lane raises the head's word -- non-negative floats order like their bit patterns, so an integer atomicMax
// does it -- and only when the word is not already as large (a plain load first: after the first waves almost nobody issues the atomic).
__device__ __forceinline__ void head_absmax_commit(unsigned int* word, float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned int b = __float_as_uint(m);
        if (b > __atomic_load_n(word, __ATOMIC_RELAXED)) atomicMax(word, b);
    }
}

template <bool ROI>
__global__ __launch_bounds__(256, 2) void head_kernel(HeadParams p) {
    // v_mfma_f32_16x16x4_f32 throughout: D[16 x 16] += A[16 x 4] B[4 x 16]; lane l supplies row/column l & 15 and k-slot l >> 4,
    // and holds D rows 4 (l >> 4) + r of column l & 15.  A wave owns 64 pixels = 4 pixel blocks of 16.
    //   GEMM1 (hidden 96 = 6 blocks) x (64 channels = 16 k-steps): k-slot ks at step t of 16-channel group g <-> channel 16 g + 4 ks + t,
    //     so one float4 per lane feeds 4 steps (features straight from global memory, W1 pre-packed the same way);
    //   its accumulators (bias in the init, ReLU) are the B operand of GEMM2: step (blk, r), k-slot ks <-> hidden 16 blk + 4 ks + r;
    //   GEMM2 pads out_ch to 16 rows (a 32x32 tile would pad to 32: 30 % of the head's MFMAs were padding before).
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, ks = lane >> 4;
    const f32x4h* w1v = reinterpret_cast<const f32x4h*>(p.w1p) + lane;
    const f32x4h* w2v = reinterpret_cast<const f32x4h*>(p.w2p) + lane;
    // Two walks over the pixels.  Whole map (ROI = false): a wave takes 64 consecutive pixels of the flattened [N][H][W] map.
    // Crop window (ROI = true): 16-pixel blocks (n, row, xb) of the 16-aligned cover of the window in linear order; a wave's
    // block positions are wave-uniform -- one 32-bit decode per wave, then increments with carry.
    struct BPos {
        int n, row, xb;
    };
    auto inc = [&](BPos b) {
        if (++b.xb == p.nxb) {
            b.xb = 0;
            if (++b.row == p.rows) {
                b.row = 0;
                ++b.n;
            }
        }
        return b;
    };
    const long long npix = (long long)p.N * p.H * p.W;
    long long pbase = ((long long)blockIdx.x * 4 + (tid >> 6)) * (64 * HEAD_TPW);  // ROI = false
    BPos cur[4], nxt[4];                                                           // ROI = true
    if constexpr (ROI) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned nblk = (unsigned)p.N * (unsigned)p.rows * (unsigned)p.nxb;  // launcher: < 2^31
        const unsigned b0 = ((unsigned)blockIdx.x * 4u + (unsigned)wave) * (4u * HEAD_TPW);
        const unsigned bc = b0 < nblk ? b0 : nblk;
        const unsigned r = bc / (unsigned)p.nxb;
        cur[0].xb = (int)(bc - r * (unsigned)p.nxb);
        cur[0].n = (int)(r / (unsigned)p.rows);
        cur[0].row = (int)(r - (unsigned)cur[0].n * (unsigned)p.rows);
#pragma unroll
        for (int pb = 1; pb < 4; ++pb) cur[pb] = inc(cur[pb - 1]);
        nxt[0] = inc(cur[3]);
#pragma unroll
        for (int pb = 1; pb < 4; ++pb) nxt[pb] = inc(nxt[pb - 1]);
    }
    auto ptr_of = [&](const BPos& b) {  // blocks past the end (n == N) and pixels past the row end are clamped here, dropped at the store
        const int n = min(b.n, p.N - 1), x = min(p.xa0 + 16 * b.xb + px, p.W - 1);
        return p.feat + (((long long)n * p.H + p.row0 + b.row) * p.W + x) * 64 + 4 * ks;
    };
    auto feat_ptr = [&](int pb, bool next) {
        if constexpr (ROI) return ptr_of(next ? nxt[pb] : cur[pb]);
        const long long P = pbase + (next ? 64 : 0) + pb * 16 + px;
        return p.feat + (P < npix ? P : npix - 1) * 64 + 4 * ks;
    };
    // a wave walks HEAD_TPW consecutive tasks of 4 blocks; the first channel group of the next task is requested before this
    // task's second GEMM and softmax, so the HBM latency of the feature read hides behind them
    f32x4h x[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) x[pb] = *reinterpret_cast<const f32x4h*>(feat_ptr(pb, false));
    float amax = 0.f;  // largest |logit| this lane has finished (head_absmax_commit)
#pragma unroll 1
    for (int task = 0; task < HEAD_TPW; ++task, pbase += 64) {
    f32x4h acc1[6][4];
#pragma unroll
    for (int blk = 0; blk < 6; ++blk) {
        const f32x4h bb = *reinterpret_cast<const f32x4h*>(p.b1 + blk * 16 + ks * 4);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc1[blk][pb] = bb;
    }
    const float* xp[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) xp[pb] = feat_ptr(pb, false);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4h xn[4];
        if (g + 1 < 4) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) xn[pb] = *reinterpret_cast<const f32x4h*>(xp[pb] + (g + 1) * 16);
        } else if (task + 1 < HEAD_TPW) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) xn[pb] = *reinterpret_cast<const f32x4h*>(feat_ptr(pb, true));
        }
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const f32x4h a = w1v[(blk * 4 + g) * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc1[blk][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], x[pb][t], acc1[blk][pb], 0, 0, 0);
        }
        if (g + 1 < 4 || task + 1 < HEAD_TPW) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) x[pb] = xn[pb];
        }
    }
    // ReLU (bias and folded BN are already inside)
#pragma unroll
    for (int blk = 0; blk < 6; ++blk)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[blk][pb][r] = fmaxf(acc1[blk][pb][r], 0.f);
    // GEMM2: logits[out 16][pixel 16] per pixel block; lane (px, ks) ends up with logits 4 ks + r of pixel px
    f32x4h acc2[4];
    {
        const f32x4h b2 = *reinterpret_cast<const f32x4h*>(p.b2 + 4 * ks);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc2[pb] = b2;
    }
#pragma unroll
    for (int blk = 0; blk < 6; ++blk) {
        const f32x4h a = w2v[blk * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc2[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], acc1[blk][pb][r], acc2[pb], 0, 0, 0);
    }
    // lane group ks finishes pixel block ks: fetch its pixel's 8 logits from lanes px (rows 0..3) and 16 + px (rows 4..7)
    float lg[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const float t0 = __shfl(acc2[pb][e], px), t1 = __shfl(acc2[pb][e], 16 + px);
            lo = (ks == pb) ? t0 : lo;
            hi = (ks == pb) ? t1 : hi;
        }
        lg[e] = lo;
        lg[4 + e] = hi;
    }
    // lane group ks finishes block ks of this task
    int n, y_, x_;
    if constexpr (ROI) {
        int xb_ = cur[0].xb;
        n = cur[0].n;
        y_ = cur[0].row;
#pragma unroll
        for (int pb = 1; pb < 4; ++pb) {
            n = (ks == pb) ? cur[pb].n : n;
            y_ = (ks == pb) ? cur[pb].row : y_;
            xb_ = (ks == pb) ? cur[pb].xb : xb_;
        }
        y_ += p.row0;
        x_ = p.xa0 + 16 * xb_ + px;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) cur[pb] = nxt[pb];
        nxt[0] = inc(cur[3]);
#pragma unroll
        for (int pb = 1; pb < 4; ++pb) nxt[pb] = inc(nxt[pb - 1]);
        if (n >= p.N || x_ >= p.W) continue;
    } else {
        const long long P = pbase + ks * 16 + px;
        if (P >= npix) continue;
        x_ = (int)(P % p.W);
        const long long r_ = P / p.W;
        y_ = (int)(r_ % p.H);
        n = (int)(r_ / p.H);
    }
    if (p.logits) {
        const long long P = ((long long)n * p.H + y_) * p.W + x_;
        for (int e = 0; e < p.out_ch; ++e) p.logits[P * p.out_ch + e] = lg[e];
    }
    // softmax over out_ch (max-subtracted, as torch.softmax)
    float mx = lg[0];
#pragma unroll
    for (int e = 1; e < 8; ++e)
        if (e < p.out_ch) mx = fmaxf(mx, lg[e]);
    float ex[8], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ex[e] = (e < p.out_ch) ? expf(lg[e] - mx) : 0.f;
        sum += ex[e];
    }
    const int cy = y_ - p.crop_y0, cx = x_ - p.crop_x0;
    const bool inside = cy >= 0 && cy < p.out_h && cx >= 0 && cx < p.out_w;
    if (inside || p.logits) {  // (a cropped forward only finishes the kept window's features: pixels of the 16-aligned cover outside it do not count)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < p.out_ch) amax = fmaxf(amax, fabsf(lg[e]));
    }
    if (!inside) continue;
    const long long dst = (p.tile_off ? p.tile_off[n] : (long long)n * p.tile_stride) + (long long)cy * p.row_stride + cx;
    if (p.kind == 0) {
        float2 o;
        o.x = ex[1] / sum;
        o.y = ex[2] / sum;
        *reinterpret_cast<float2*>(p.out_inst + dst * 2) = o;
    } else {
        int best = 0;
        float bv = ex[0] / sum;
#pragma unroll
        for (int e = 1; e < 8; ++e) {
            const float pe = ex[e] / sum;
            if (e < p.out_ch && pe > bv) {
                bv = pe;
                best = e;
            }
        }
        if (p.out_type_i64) p.out_type_i64[dst] = best;
        if (p.out_type_u8) p.out_type_u8[dst] = (unsigned char)best;
    }
    }  // task
    if (p.absmax_bits) head_absmax_commit(p.absmax_bits, amax);
}

hipError_t cerb_launch_head(const HeadParams& p_in, hipStream_t st) {
    HeadParams p = p_in;
    if (p.W % 16) return hipErrorInvalidValue;
    p.rows = p.H; p.row0 = 0; p.xa0 = 0; p.nxb = p.W / 16;
    if (p.roi && !p.logits && p.out_h > 0 && p.out_w > 0) {  // only the 16-aligned cover of the crop window
        p.rows = p.out_h; p.row0 = p.crop_y0;
        p.xa0 = p.crop_x0 & ~15;
        p.nxb = (p.crop_x0 + p.out_w - p.xa0 + 15) / 16;
    }
    const long long nblk = (long long)p.N * p.rows * p.nxb;  // 16-pixel blocks; a workgroup takes 16 * HEAD_TPW of them
    const long long blocks = (nblk + 16 * HEAD_TPW - 1) / (16 * HEAD_TPW);
    if (p.rows == p.H && p.nxb * 16 == p.W) hipLaunchKernelGGL(head_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(head_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped output heads: ONE launch for all dense heads of a batch (blockIdx.y = head), same arithmetic as head_kernel, re-balanced
// for latency: round 1 measured head_kernel at ~60 % of the matrix pipe with 242 VGPRs (two waves per SIMD) and the feature
// prefetch only one 16-channel group (3072 MFMA cycles) ahead of an HBM read.  Here
//   * W1 / W2 / biases of the workgroup's head live in LDS (31 KB, filled once per workgroup) instead of streaming through VGPRs,
//   * a task is 32 pixels (two 16-pixel blocks): acc1 is 48 registers instead of 96, so THREE workgroups fit a CU, and the whole
//     feature vector of the NEXT task (32 registers) is requested one task = 7680 MFMA cycles ahead,
//   * every wave walks HEAD_G_TPW consecutive tasks.
// Lane roles and the K permutations are those of head_kernel (lane = (px = l & 15, ks = l >> 4)).
// ---------------------------------------------------------------------------------------------------------------
#ifndef HEAD_G_TPW
#define HEAD_G_TPW 8
#endif
#ifndef HEAD_G_OCC
#define HEAD_G_OCC 3
#endif
struct HeadGroupParams {
    HeadParams h[8];
    int n_heads;
};
// W2_44 (round 4, the default): the second 1x1 (96 -> 3 / 7 logits) on v_mfma_f32_4x4x1_16B_f32 instead of a 16-row matrix instruction whose
// rows 3 (7) .. 15 multiply zeros: 16 independent 4x4 outer products per instruction, block b = lanes 4b .. 4b+3 = four pixels of one k-slot
// group, B = the lane's own hidden value (exactly where the first GEMM left it), A = W2[out l & 3][the block's hidden channel], D = 4 logits per
// lane, partial over the lane's k-slot group; 24 instructions of 2 passes per pixel block and set of 4 logits (48 for the 7-class head) replace
// 24 of 8 passes, then two cross-lane exchanges sum the four k-slot groups.  Matrix-pipe time of a 32-pixel task: 7680 -> 6528 (6912) cycles.
template <bool W2_44>
__global__ __launch_bounds__(256, HEAD_G_OCC) void head_group_kernel(HeadGroupParams gp) {
    __shared__ __attribute__((aligned(16))) float s_w1[6 * 4 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float s_w2[(W2_44 ? 2 : 1) * 6 * 64 * 4];  // W2_44: w2q = [set][blk][lane][r], else w2p = [blk][lane][r]
    __shared__ __attribute__((aligned(16))) float s_b1[96];
    __shared__ __attribute__((aligned(16))) float s_b2[32];
    const HeadParams& p = gp.h[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, ks = lane >> 4;
    {
        const f32x4h* w1g = reinterpret_cast<const f32x4h*>(p.w1p);
        const f32x4h* w2g = reinterpret_cast<const f32x4h*>(W2_44 ? p.w2q : p.w2p);
        for (int i = tid; i < 6 * 4 * 64; i += 256) reinterpret_cast<f32x4h*>(s_w1)[i] = w1g[i];
        for (int i = tid; i < (W2_44 ? 2 : 1) * 6 * 64; i += 256) reinterpret_cast<f32x4h*>(s_w2)[i] = w2g[i];
        if (tid < 96) s_b1[tid] = p.b1[tid];
        if (tid < 32) s_b2[tid] = p.b2[tid];
    }
    __syncthreads();
    const f32x4h* w1v = reinterpret_cast<const f32x4h*>(s_w1) + lane;
    const f32x4h* w2v = reinterpret_cast<const f32x4h*>(s_w2) + lane;
    const bool wide = p.out_ch > 4;  // W2_44: a second set of 4 logits
    const unsigned nblk = (unsigned)p.N * (unsigned)p.rows * (unsigned)p.nxb;  // 16-pixel blocks (launcher: < 2^31)
    const unsigned ntask = (nblk + 1u) >> 1;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned task = ((unsigned)blockIdx.x * 4u + wave) * (unsigned)HEAD_G_TPW;
    if (task >= ntask) return;
    const unsigned task_end = min(task + (unsigned)HEAD_G_TPW, ntask);
    struct BPos {
        int n, row, xb;
    };
    auto decode = [&](unsigned b) {  // wave-uniform
        BPos r;
        const unsigned bc = min(b, nblk);  // blocks past the end decode to n == N: clamped for the loads, dropped at the store
        const unsigned q = bc / (unsigned)p.nxb;
        r.xb = (int)(bc - q * (unsigned)p.nxb);
        r.n = (int)(q / (unsigned)p.rows);
        r.row = (int)(q - (unsigned)r.n * (unsigned)p.rows);
        return r;
    };
    // NHWC: a pixel's 64 channels are contiguous (group stride 16 floats); tile-planar (conv_wino4p.hip's output): a pixel's 16-channel groups
    // are one plane = 4096 floats apart, and the 16 pixels of a row segment are four runs of 256 bytes
    const int gstride = p.feat_planar ? 4096 : 16;
    auto ptr_of = [&](const BPos& b) {
        const int n = min(b.n, p.N - 1), x = min(p.xa0 + 16 * b.xb + px, p.W - 1), y = p.row0 + b.row;
        if (p.feat_planar) return p.feat + cerb_planar_offset(n, y, x, 0, p.pl_byp, p.pl_bxp, 4) + 4 * ks;
        return p.feat + (((long long)n * p.H + y) * p.W + x) * 64 + 4 * ks;
    };
    f32x4h xn[4][2];
    auto request = [&](unsigned t) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const float* fp = ptr_of(decode(2u * t + (unsigned)pb));
#pragma unroll
            for (int g = 0; g < 4; ++g) xn[g][pb] = *reinterpret_cast<const f32x4h*>(fp + gstride * g);
        }
    };
    request(task);
    float amax = 0.f;  // largest |logit| this lane has finished (head_absmax_commit)
#pragma unroll 1
    for (; task < task_end; ++task) {
        f32x4h x[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) x[g][pb] = xn[g][pb];
        if (task + 1 < task_end) request(task + 1);  // a whole task ahead of its use
        f32x4h acc1[6][2];
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const f32x4h bb = *reinterpret_cast<const f32x4h*>(s_b1 + blk * 16 + ks * 4);
            acc1[blk][0] = bb;
            acc1[blk][1] = bb;
        }
        {
            // W1 operands come from LDS one (g, blk) step ahead; the scheduling barriers keep hipcc from hoisting all 24 reads
            // (96 registers) to the top of the task
            f32x4h a_cur = w1v[0];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int blk = 0; blk < 6; ++blk) {
                    const int nx = g * 6 + blk + 1;
                    f32x4h a_nxt = a_cur;
                    if (nx < 24) a_nxt = w1v[((nx % 6) * 4 + nx / 6) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb)
                            acc1[blk][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], x[g][pb][t], acc1[blk][pb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a_cur = a_nxt;
                }
        }
#pragma unroll
        for (int blk = 0; blk < 6; ++blk)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[blk][pb][r] = fmaxf(acc1[blk][pb][r], 0.f);
        float lg[8];
        if constexpr (W2_44) {
            f32x4h lo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, hi[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int blk = 0; blk < 6; ++blk) {
                const f32x4h wa = w2v[blk * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) lo[pb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[r], acc1[blk][pb][r], lo[pb], 0, 0, 0);
            }
            if (wide) {
#pragma unroll
                for (int blk = 0; blk < 6; ++blk) {
                    const f32x4h wb = w2v[(6 + blk) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb) hi[pb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[r], acc1[blk][pb][r], hi[pb], 0, 0, 0);
                }
            }
            // sum over the four k-slot groups; lane groups ks = 0, 1 finish pixel blocks 0, 1: first exchange hands the partner (ks ^ 1) the block
            // it finishes, second adds the pair (ks ^ 2) -- a fixed order, so the logits do not depend on anything but the pixel's features
            const bool odd = ks & 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mine = odd ? lo[1][e] : lo[0][e];
                mine += __shfl_xor(odd ? lo[0][e] : lo[1][e], 16);
                mine += __shfl_xor(mine, 32);
                lg[e] = mine + s_b2[e];
                float mh = 0.f;
                if (wide) {
                    mh = odd ? hi[1][e] : hi[0][e];
                    mh += __shfl_xor(odd ? hi[0][e] : hi[1][e], 16);
                    mh += __shfl_xor(mh, 32);
                }
                lg[4 + e] = mh + s_b2[4 + e];
            }
        } else {
        f32x4h acc2[2];
        {
            const f32x4h b2 = *reinterpret_cast<const f32x4h*>(s_b2 + 4 * ks);
            acc2[0] = b2;
            acc2[1] = b2;
        }
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const f32x4h a = w2v[blk * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) acc2[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], acc1[blk][pb][r], acc2[pb], 0, 0, 0);
        }
        // lane groups ks = 0, 1 finish pixel blocks 0, 1: the pixel's 8 logits sit in lanes px (rows 0..3) and 16 + px (rows 4..7)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = __shfl(acc2[0][e], px), a1 = __shfl(acc2[0][e], 16 + px);
            const float c0 = __shfl(acc2[1][e], px), c1 = __shfl(acc2[1][e], 16 + px);
            lg[e] = (ks & 1) ? c0 : a0;
            lg[4 + e] = (ks & 1) ? c1 : a1;
        }
        }
        if (ks >= 2) continue;
        const BPos bp = decode(2u * task + (unsigned)ks);
        const int n = bp.n, y_ = bp.row + p.row0, x_ = p.xa0 + 16 * bp.xb + px;
        if (2u * task + (unsigned)ks >= nblk || x_ >= p.W) continue;
        if (p.logits) {
            const long long P = ((long long)n * p.H + y_) * p.W + x_;
            for (int e = 0; e < p.out_ch; ++e) p.logits[P * p.out_ch + e] = lg[e];
        }
        float mx = lg[0];
#pragma unroll
        for (int e = 1; e < 8; ++e)
            if (e < p.out_ch) mx = fmaxf(mx, lg[e]);
        float ex[8], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ex[e] = (e < p.out_ch) ? expf(lg[e] - mx) : 0.f;
            sum += ex[e];
        }
        const int cy = y_ - p.crop_y0, cx = x_ - p.crop_x0;
        const bool inside = cy >= 0 && cy < p.out_h && cx >= 0 && cx < p.out_w;
        if (inside || p.logits) {  // (a cropped forward only finishes the kept window's features: pixels of the 16-aligned cover outside it do not count)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < p.out_ch) amax = fmaxf(amax, fabsf(lg[e]));
        }
        if (!inside) continue;
        const long long dst = (p.tile_off ? p.tile_off[n] : (long long)n * p.tile_stride) + (long long)cy * p.row_stride + cx;
        if (p.kind == 0) {
            float2 o;
            o.x = ex[1] / sum;
            o.y = ex[2] / sum;
            *reinterpret_cast<float2*>(p.out_inst + dst * 2) = o;
        } else {
            int best = 0;
            float bv = ex[0] / sum;
#pragma unroll
            for (int e = 1; e < 8; ++e) {
                const float pe = ex[e] / sum;
                if (e < p.out_ch && pe > bv) {
                    bv = pe;
                    best = e;
                }
            }
            if (p.out_type_i64) p.out_type_i64[dst] = best;
            if (p.out_type_u8) p.out_type_u8[dst] = (unsigned char)best;
        }
    }
    if (p.absmax_bits) head_absmax_commit(p.absmax_bits, amax);
}

hipError_t cerb_launch_head_group(const HeadParams* heads, int n_heads, hipStream_t st, int w2_44) {
    if (n_heads < 1 || n_heads > 8) return hipErrorInvalidValue;

    HeadGroupParams gp = {};
    gp.n_heads = n_heads;
    unsigned max_blocks = 0;
    for (int i = 0; i < n_heads; ++i) {
        HeadParams p = heads[i];
        if (p.W % 16) return hipErrorInvalidValue;
        p.rows = p.H; p.row0 = 0; p.xa0 = 0; p.nxb = p.W / 16;
        if (p.roi && !p.logits && p.out_h > 0 && p.out_w > 0) {  // only the 16-aligned cover of the crop window
            p.rows = p.out_h; p.row0 = p.crop_y0;
            p.xa0 = p.crop_x0 & ~15;
            p.nxb = (p.crop_x0 + p.out_w - p.xa0 + 15) / 16;
        }
        const long long nblk = (long long)p.N * p.rows * p.nxb;
        if (nblk >= (1ll << 31)) return hipErrorInvalidValue;
        const long long ntask = (nblk + 1) / 2;
        const long long blocks = (ntask + 4 * HEAD_G_TPW - 1) / (4 * HEAD_G_TPW);
        if (blocks > max_blocks) max_blocks = (unsigned)blocks;
        gp.h[i] = p;
    }
    if (w2_44) hipLaunchKernelGGL(head_group_kernel<true>, dim3(max_blocks, (unsigned)n_heads), dim3(256), 0, st, gp);
    else hipLaunchKernelGGL(head_group_kernel<false>, dim3(max_blocks, (unsigned)n_heads), dim3(256), 0, st, gp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
struct PatchClassParams {
    const float* x4;      // [N][Hf][Wf][512] (pre-conv_map bottom features, net_desc.py:152)
    const float* bn1_s;   // [512] scale
    const float* bn1_b;   // [512] shift
    const float* w1t;     // [512][256]  (conv1 with bn2 folded, transposed for coalescing)
    const float* b1;      // [256]
    const float* w2t;     // [256][16]   (conv2, transposed, padded)
    const float* b2;      // [16]
    int N, Hf, Wf, out_ch;
    int out_h, out_w;
    float* logits;        // optional [N][out_ch]
    float* out;           // optional: class id broadcast, addressed like the heads
    const long long* tile_off;
    long long tile_stride, row_stride;
};

__global__ __launch_bounds__(256) void patch_class_kernel(PatchClassParams p) {
    __shared__ float v[512];
    __shared__ float hid[256];
    __shared__ float lg[16];
    __shared__ int cls_s;
    const int n = blockIdx.x, tid = threadIdx.x;
    int y0 = 0, x0 = 0, ch = p.Hf, cw = p.Wf;
    if (p.Hf != 9 && p.Wf != 9) {  // net_desc.py:173-174 (crops only when both differ from 9)
        // cropping_center (models/utils/misc_utils.py:22-24) is the Python slice x[h0 : h0 + 9] with h0 = int((H - 9) * 0.5): for maps
        // smaller than 9 the start is NEGATIVE and counts from the end (H = 6 -> rows [5, 6)), and the stop is clipped to H
        auto py_slice = [](int len, int& start, int& count) {
            const int h0 = (int)((len - 9) * 0.5);
            const int a0 = h0 < 0 ? max(len + h0, 0) : min(h0, len);
            const int a1 = min(h0 + 9, len);
            start = a0;
            count = max(a1 - a0, 0);
        };
        py_slice(p.Hf, y0, ch);
        py_slice(p.Wf, x0, cw);
    }
    const float* x = p.x4 + (long long)n * p.Hf * p.Wf * 512;
    for (int c = tid; c < 512; c += 256) {
        float s = 0.f;
        for (int yy = 0; yy < ch; ++yy)
            for (int xx = 0; xx < cw; ++xx) s += x[((long long)(y0 + yy) * p.Wf + (x0 + xx)) * 512 + c];
        s = s / (float)(ch * cw);
        s = s * p.bn1_s[c] + p.bn1_b[c];
        v[c] = fmaxf(s, 0.f);
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int c = 0; c < 512; ++c) s = fmaf(p.w1t[c * 256 + tid], v[c], s);
        hid[tid] = fmaxf(s + p.b1[tid], 0.f);
    }
    __syncthreads();
    if (tid < 16) {
        float s = 0.f;
        if (tid < p.out_ch) {
            for (int c = 0; c < 256; ++c) s = fmaf(p.w2t[c * 16 + tid], hid[c], s);
            s += p.b2[tid];
            if (p.logits) p.logits[n * p.out_ch + tid] = s;
        }
        lg[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float mx = lg[0];
        for (int e = 1; e < p.out_ch; ++e) mx = fmaxf(mx, lg[e]);
        float sum = 0.f, ex[16];
        for (int e = 0; e < p.out_ch; ++e) {
            ex[e] = expf(lg[e] - mx);
            sum += ex[e];
        }
        int best = 0;
        float bv = ex[0] / sum;
        for (int e = 1; e < p.out_ch; ++e) {
            const float pe = ex[e] / sum;
            if (pe > bv) {
                bv = pe;
                best = e;
            }
        }
        cls_s = best;
    }
    __syncthreads();
    if (p.out) {
        const float cf = (float)cls_s;
        const long long base = p.tile_off ? p.tile_off[n] : (long long)n * p.tile_stride;
        for (int i = tid; i < p.out_h * p.out_w; i += 256) {
            const int yy = i / p.out_w, xx = i % p.out_w;
            p.out[base + (long long)yy * p.row_stride + xx] = cf;
        }
    }
}

hipError_t cerb_launch_patch_class(const PatchClassParams& p, hipStream_t st) {
    hipLaunchKernelGGL(patch_class_kernel, dim3(p.N), dim3(256), 0, st, p);
    return hipGetLastError();
}
