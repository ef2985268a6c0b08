// conv_wgrad.hip -- weight gradient of the convolutions (training step, BASELINE.json configs[4]; written for the 3x3 stride-1 pad-1 case) on the fp32 matrix
// cores:  dW[g][co][ci][ky][kx] = sum over (n, y, x) of dy[g][n][y][x][co] * x[g][n][y + ky - 1][x + kx - 1][ci]
// a GEMM whose reduction dimension is the 10^6 .. 10^7 pixels of the batch.
//   * a workgroup owns a 64 co x 64 ci x 9 tap tile of dW and a SLICE of the pixels (split-K: every slice leaves a partial tile in a
//     workspace, a second kernel adds the slices in a fixed order -- no float atomics, bitwise reproducible);
//   * pixels come in row segments of 32: the dy segment (32 px x 64 co) and the three input rows around it (3 x 34 px x 64 ci) are
//     staged in LDS (34 KB), zero-filled outside the image;
//   * wave (ch, ih) = (co half, ci half) keeps nine 32 x 32 accumulators (144 VGPRs), one per tap; per pixel PAIR it issues nine
//     v_mfma_f32_32x32x2_f32: A = dy^T (32 co x 2 px, one ds_read_b32, shared by the nine taps), B = x at the tap's offset (2 px x 32 ci).
// Arithmetic intensity: 2 * 64 * 64 * 9 flop per pixel against 34 KB per 32 pixels staged -> 69 flop/B, MFMA-bound like the forward.
#include "cerb_common.h"

// The same kernel, templated on (kernel size, stride), also serves the 3x3 stride-2 and the 1x1 convolutions (stride 1 / 2) and -- with
// N = H = 1, W = rows -- the pointwise layers of the heads; channel counts that are not multiples of 64 (the heads' 96 hidden units) are
// zero-filled at the staging and skipped by the reduction.
namespace {
constexpr int SEG = 32;                       // OUTPUT pixels per row segment
constexpr int DY_FLOATS = SEG * 64;           // 2048

struct WgradParams {
    const float* x;
    const float* dy;
    float* part;   // [slices][tiles][taps][64 co][64 ci]
    float* part_b; // optional [slices][G][ncb * 64]: per-slice sums of dy over the pixels (the bias gradient), written by the cib == 0 tiles
    int G, N, H, W, Ho, Wo, Cin, Cout, slices, segs_per_row;  // H, W: input map; Ho, Wo: output map; Cin / Cout: real channel counts (multiples of 4)
    long long x_gs, dy_gs;
};
}  // namespace

template <int KS, int STRIDE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, XQ = (SEG - 1) * STRIDE + KS, X_FLOATS = KS * XQ * 64;
    // PIPE: the next segment's operands are fetched into registers underneath this segment's MFMAs and dropped into the other LDS buffer
    // (3x3 stride 2 would need 52 more registers for that: it keeps the single-buffer loop)
    constexpr bool PIPE = STRIDE == 1;
    constexpr int NBUF = PIPE ? 2 : 1, ND = DY_FLOATS / 4 / 256, NX = (X_FLOATS / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * (DY_FLOATS + X_FLOATS)];
    float* dyl = lds;
    float* xl = lds + DY_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = wave & 1, ih = wave >> 1, j = lane & 31, k = lane >> 5;
    const int ncb = (p.Cout + 63) >> 6, ncib = (p.Cin + 63) >> 6;
    const int tiles = p.G * ncb * ncib;
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
    const float* xg = p.x + g * p.x_gs + cib * 64;
    const float* dyg = p.dy + g * p.dy_gs + cb * 64;
    const int co_valid = min(64, p.Cout - cb * 64), ci_valid = min(64, p.Cin - cib * 64);
    const long long nseg = (long long)p.N * p.Ho * p.segs_per_row;
    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // the bias gradient rides along: lane (j, k) reads dy[pixel of parity k][co = 32 ch + j] as its A operand anyway -- one more add per nine
    // matrix instructions instead of a pass of its own over dy (double, like the column-sum kernel it replaces)
    double bsum = 0.0;
    if constexpr (PIPE) {
        f32x4 rd[ND], rx[NX];
        auto fetch = [&](long long s) {
            const int sx = (int)(s % p.segs_per_row);
            const long long ry = s / p.segs_per_row;
            const int y = (int)(ry % p.Ho), n = (int)(ry / p.Ho);
            const int x0 = sx * SEG;
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const int i = tid + 256 * u, px = i >> 4, c4 = i & 15;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (x0 + px < p.Wo && 4 * c4 < co_valid) v = *reinterpret_cast<const f32x4*>(dyg + (((long long)n * p.Ho + y) * p.Wo + x0 + px) * p.Cout + 4 * c4);
                rd[u] = v;
            }
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int i = tid + 256 * u, c4 = i & 15, q = (i >> 4) % XQ, r = (i >> 4) / XQ;
                const int yy = y * STRIDE + r - PAD, xx = x0 * STRIDE + q - PAD;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (i < X_FLOATS / 4 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && 4 * c4 < ci_valid)
                    v = *reinterpret_cast<const f32x4*>(xg + (((long long)n * p.H + yy) * p.W + xx) * p.Cin + 4 * c4);
                rx[u] = v;
            }
        };
        auto drop = [&](int buf) {
            float* d = lds + buf * (DY_FLOATS + X_FLOATS);
#pragma unroll
            for (int u = 0; u < ND; ++u) *reinterpret_cast<f32x4*>(d + 4 * (tid + 256 * u)) = rd[u];  // dyl[px * 64 + 4 c4], i = px * 16 + c4
#pragma unroll
            for (int u = 0; u < NX; ++u)
                if (tid + 256 * u < X_FLOATS / 4) *reinterpret_cast<f32x4*>(d + DY_FLOATS + 4 * (tid + 256 * u)) = rx[u];  // xl[(r * XQ + q) * 64 + 4 c4]
        };
        long long s = slice;
        int buf = 0;
        if (s < nseg) {
            fetch(s);
            drop(0);
        }
        __syncthreads();
        for (; s < nseg; s += p.slices) {
            const bool more = s + p.slices < nseg;
            if (more) fetch(s + p.slices);
            const float* dl = lds + buf * (DY_FLOATS + X_FLOATS);
            const float* xq = dl + DY_FLOATS;
            float bseg = 0.f;  // the segment's 16 values in float, the segments in double
#pragma unroll 4
            for (int pp = 0; pp < SEG / 2; ++pp) {
                const float a = dl[(2 * pp + k) * 64 + 32 * ch + j];
                bseg += a;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float b = xq[((t / KS) * XQ + (2 * pp + k) * STRIDE + (t % KS)) * 64 + 32 * ih + j];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
            bsum += (double)bseg;
            if (more) drop(buf ^ 1);
            __syncthreads();  // the other buffer is complete; this one has been read by every wave
            buf ^= 1;
        }
    } else {
    for (long long s = slice; s < nseg; s += p.slices) {
            const int sx = (int)(s % p.segs_per_row);
            const long long ry = s / p.segs_per_row;
            const int y = (int)(ry % p.Ho), n = (int)(ry / p.Ho);
            const int x0 = sx * SEG;
            __syncthreads();  // the previous segment's MFMAs have read their operands
            // ---- stage dy: 32 px x 64 co ------------------------------------------------------------------------------------------------
            for (int i = tid; i < DY_FLOATS / 4; i += 256) {
                const int px = i >> 4, c4 = i & 15;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (x0 + px < p.Wo && 4 * c4 < co_valid) v = *reinterpret_cast<const f32x4*>(dyg + (((long long)n * p.Ho + y) * p.Wo + x0 + px) * p.Cout + 4 * c4);
                *reinterpret_cast<f32x4*>(dyl + px * 64 + 4 * c4) = v;
            }
            // ---- stage x: the KS input rows and (SEG - 1) * STRIDE + KS input pixels the segment's taps read, 64 ci --------------------------------
            for (int i = tid; i < X_FLOATS / 4; i += 256) {
                const int c4 = i & 15, q = (i >> 4) % XQ, r = (i >> 4) / XQ;
                const int yy = y * STRIDE + r - PAD, xx = x0 * STRIDE + q - PAD;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && 4 * c4 < ci_valid)
                    v = *reinterpret_cast<const f32x4*>(xg + (((long long)n * p.H + yy) * p.W + xx) * p.Cin + 4 * c4);
                *reinterpret_cast<f32x4*>(xl + (r * XQ + q) * 64 + 4 * c4) = v;
            }
            __syncthreads();
            // ---- 16 pixel pairs x 9 taps ------------------------------------------------------------------------------------------------------
            float bseg = 0.f;
    #pragma unroll 4
            for (int pp = 0; pp < SEG / 2; ++pp) {
                const float a = dyl[(2 * pp + k) * 64 + 32 * ch + j];
                bseg += a;
    #pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float b = xl[((t / KS) * XQ + (2 * pp + k) * STRIDE + (t % KS)) * 64 + 32 * ih + j];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
            bsum += (double)bseg;
        }
}
    if (p.part_b && cib == 0 && ih == 0) {  // (both ci-halves' waves read the same dy: one of them reports)
        const double other = __shfl_xor(bsum, 32);
        if (k == 0) p.part_b[((long long)slice * p.G + g) * ncb * 64 + cb * 64 + 32 * ch + j] = (float)(bsum + other);
    }
    // ---- this slice's partial tile: part[slice][tile][tap][co 64][ci 64]; MFMA D layout: row = rq*8 + (lane>>5)*4 + e, column = lane & 31 ----
    float* o = p.part + ((long long)slice * tiles + tile) * T * 4096;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 32 * ch + rq * 8 + k * 4 + e, ci = 32 * ih + j;
                o[t * 4096 + co * 64 + ci] = acc[t][rq * 4 + e];
            }
}

// dW[g][cb*64 + co][cib*64 + ci][tap] = sum over slices: 64 outputs x 16 slice chunks per workgroup, a chunk's slices added in order, then the
// chunks in chunk order (fixed order: reproducible)
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G, int Cin, int Cout, int T, int slices) {
    __shared__ float sh[16][64];
    const int ncb = (Cout + 63) >> 6, ncib = (Cin + 63) >> 6, tiles = G * ncb * ncib;
    const long long total = (long long)tiles * T * 4096;
    const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int per = (slices + 15) / 16, s0 = ch * per, s1 = min(slices, s0 + per);
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long i = base + lane;  // total is a multiple of 64
        float s = 0.f;
        int sl = s0;
        for (; sl + 4 <= s1; sl += 4) {
            const float v0 = part[(long long)sl * total + i], v1 = part[(long long)(sl + 1) * total + i], v2 = part[(long long)(sl + 2) * total + i],
                        v3 = part[(long long)(sl + 3) * total + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; sl < s1; ++sl) s += part[(long long)sl * total + i];
        __syncthreads();
        sh[ch][lane] = s;
        __syncthreads();
        if (ch == 0) {
            float t = 0.f;
            for (int k = 0; k < 16; ++k) t += sh[k][lane];
            const int ci = (int)(i & 63), co = (int)((i >> 6) & 63), tp = (int)((i >> 12) % T), tile = (int)(i / ((long long)T * 4096));
            const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
            if (cb * 64 + co < Cout && cib * 64 + ci < Cin) dw[(((long long)g * Cout + cb * 64 + co) * Cin + cib * 64 + ci) * T + tp] = t;
        }
    }
}

static void wgrad_shape(int G, int N, int Ho, int Wo, int Cin, int Cout, int* tiles, long long* nseg) {
    *tiles = G * ((Cout + 63) / 64) * ((Cin + 63) / 64);
    *nseg = (long long)N * Ho * ((Wo + SEG - 1) / SEG);
}
size_t cerb_wgrad_workspace_bytes(int G, int N, int Ho, int Wo, int Cin, int Cout, int ks, int* slices_out) {
    int tiles;
    long long nseg;
    wgrad_shape(G, N, Ho, Wo, Cin, Cout, &tiles, &nseg);
    long long slices = 1536 / tiles;  // about six workgroups per CU in flight over the whole grid
    if (slices < 1) slices = 1;
    if (slices > nseg) slices = nseg;
    if (slices_out) *slices_out = (int)slices;
    return (size_t)slices * tiles * ks * ks * 4096 * 4 + (size_t)slices * G * ((Cout + 63) / 64) * 64 * 4;  // + the bias partials
}

__global__ void slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int blocks);
// out[i] = sum_b part[b][i], i < n, rows of `stride` floats (a channel count padded to whole 64-blocks): same fixed order as slab_sum_kernel
__global__ __launch_bounds__(1024) void slab_sum_strided_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int stride, int blocks) {
    __shared__ double sh[16][64];
    const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const int per = (blocks + 15) / 16, b0 = ch * per, b1 = min(blocks, b0 + per);
    double s = 0;
    if (i < n)
        for (int b = b0; b < b1; ++b) s += part[(long long)b * stride + i];
    sh[ch][lane] = s;
    __syncthreads();
    if (ch == 0 && i < n) {
        double t = 0;
        for (int k = 0; k < 16; ++k) t += sh[k][lane];
        out[i] = (float)t;
    }
}
// x: [G][N][H][W][Cin], dy: [G][N][Ho][Wo][Cout] with Ho = H / stride; dw: [G][Cout][Cin][ks][ks].  ks in {1, 3}, stride in {1, 2}; channel counts
// multiples of 4.  Pointwise layers: N = H = 1, W = rows.
// db != nullptr (Cout a multiple of 64): also db[G][Cout] = sum of dy over the pixels, collected inside the same pass.
hipError_t cerb_launch_wgrad(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, int ks, int stride, long long x_gs, void* ws,
                             hipStream_t st, float* db) {
    if (db && Cout % 64 && G != 1) return hipErrorInvalidValue;  // (padded channel blocks: only without groups -- the pointwise layers)
    if (Cin % 4 || Cout % 4 || (ks != 1 && ks != 3) || (stride != 1 && stride != 2)) return hipErrorInvalidValue;
    WgradParams p;
    p.x = x; p.dy = dy; p.part = (float*)ws;
    p.G = G; p.N = N; p.H = H; p.W = W; p.Ho = stride == 2 ? H / 2 : H; p.Wo = stride == 2 ? W / 2 : W; p.Cin = Cin; p.Cout = Cout;
    p.segs_per_row = (p.Wo + SEG - 1) / SEG;
    p.x_gs = x_gs; p.dy_gs = (long long)N * p.Ho * p.Wo * Cout;
    int slices = 1, tiles;
    long long nseg;
    wgrad_shape(G, N, p.Ho, p.Wo, Cin, Cout, &tiles, &nseg);
    (void)cerb_wgrad_workspace_bytes(G, N, p.Ho, p.Wo, Cin, Cout, ks, &slices);
    p.slices = slices;
    p.part_b = db ? (float*)ws + (size_t)slices * tiles * ks * ks * 4096 : nullptr;
    const dim3 grid((unsigned)(tiles * slices));
    if (ks == 3 && stride == 1) hipLaunchKernelGGL((wgrad_kernel<3, 1>), grid, dim3(256), 0, st, p);
    else if (ks == 3) hipLaunchKernelGGL((wgrad_kernel<3, 2>), grid, dim3(256), 0, st, p);
    else if (stride == 1) hipLaunchKernelGGL((wgrad_kernel<1, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<1, 2>), grid, dim3(256), 0, st, p);
    long long blocks = (long long)tiles * ks * ks * 4096 / 64;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, (const float*)ws, dw, G, Cin, Cout, ks * ks, slices);
    if (db && Cout % 64 == 0) hipLaunchKernelGGL(slab_sum_kernel, dim3((G * Cout + 63) / 64, 1), dim3(1024), 0, st, (const float*)p.part_b, db, G * Cout, slices);
    else if (db) hipLaunchKernelGGL(slab_sum_strided_kernel, dim3((Cout + 63) / 64), dim3(1024), 0, st, (const float*)p.part_b, db, Cout, ((Cout + 63) / 64) * 64, slices);
    return hipGetLastError();
}

// =================================================================================================================
// Stem (7x7, 3 -> 64 channels, stride 1, pad 3; uint8 tiles / 255): dW[co][c][ky][kx] = sum dy[n][y][x][co] * tile[n][y+ky-3][x+kx-3][c] / 255.
// GEMM 64 x 147 over the pixels: the 147 columns are five 32-wide tiles (the last one padded), a wave = (co half, two or three of
// the column tiles); per row segment the 7 x 38 x 3 input strip sits in LDS as floats and the B operand is gathered from it.
// =================================================================================================================
namespace {
constexpr int SROWS = 7, SPX = SEG + 6;
struct StemWgradParams {
    const unsigned char* tiles;
    const float* dy;
    float* part;  // [slices][5 column tiles][64 co][32]
    int N, H, W, slices, segs_per_row;
};
}  // namespace
__global__ __launch_bounds__(256, 2) void stem_wgrad_mfma_kernel(StemWgradParams p) {
    __shared__ __attribute__((aligned(16))) float dyl[DY_FLOATS];
    __shared__ float xs[SROWS * SPX * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = wave & 1, jt0 = wave >> 1, j = lane & 31, k = lane >> 5;
    // this lane's column in each of the wave's tiles: jj = 32 * jt + j -> (c, ky, kx) -> offset in the strip (pixel part added per pair)
    int off[3];
    bool live[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int jt = jt0 + 2 * u, jj = 32 * jt + j;
        live[u] = jt < 5 && jj < 147;
        const int c = jj / 49, ky = (jj % 49) / 7, kx = jj % 7;
        off[u] = live[u] ? (ky * SPX + kx) * 3 + c : 0;
    }
    f32x16 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const long long nseg = (long long)p.N * p.H * p.segs_per_row;
    for (long long s = blockIdx.x; s < nseg; s += p.slices) {
        const int sx = (int)(s % p.segs_per_row);
        const long long ry = s / p.segs_per_row;
        const int y = (int)(ry % p.H), n = (int)(ry / p.H), x0 = sx * SEG;
        __syncthreads();
        for (int i = tid; i < DY_FLOATS / 4; i += 256) {
            const int px = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (x0 + px < p.W) v = *reinterpret_cast<const f32x4*>(p.dy + (((long long)n * p.H + y) * p.W + x0 + px) * 64 + 4 * c4);
            *reinterpret_cast<f32x4*>(dyl + px * 64 + 4 * c4) = v;
        }
        for (int i = tid; i < SROWS * SPX * 3; i += 256) {
            const int c = i % 3, q = (i / 3) % SPX, r = i / (3 * SPX);
            const int yy = y + r - 3, xx = x0 + q - 3;
            float v = 0.f;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = (float)p.tiles[(((long long)n * p.H + yy) * p.W + xx) * 3 + c] / 255.0f;
            xs[i] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int pp = 0; pp < SEG / 2; ++pp) {
            const float a = dyl[(2 * pp + k) * 64 + 32 * ch + j];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (jt0 + 2 * u < 5) {  // wave-uniform
                    const float b = live[u] ? xs[off[u] + (2 * pp + k) * 3] : 0.f;
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
                }
            }
        }
    }
    float* o = p.part + (long long)blockIdx.x * 5 * 64 * 32;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int jt = jt0 + 2 * u;
        if (jt >= 5) continue;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(jt * 64 + 32 * ch + rq * 8 + k * 4 + e) * 32 + j] = acc[u][rq * 4 + e];
    }
}
// dw[co][jj] = sum over the slices, in a fixed order: workgroup = one (column tile jt, co) row of 32 columns, thread = (slice group sg of 8, column j); a thread adds
// the slices sg, sg + 8, .. (coalesced 128-byte rows, eight loads in flight), the eight groups meet in LDS and are added in group order.
// (The first version ran one thread per output over ALL 1536 slices, 40 KB apart: 0.46 ms for a 15-MB read.)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int slices) {
    __shared__ float red[8][32];
    const int jt = blockIdx.x >> 6, co = blockIdx.x & 63, sg = threadIdx.x >> 5, j = threadIdx.x & 31;
    const float* src = part + ((long long)jt * 64 + co) * 32 + j;
    float s = 0.f;
#pragma unroll 8
    for (int sl = sg; sl < slices; sl += 8) s += src[(long long)sl * 5 * 64 * 32];
    red[sg][j] = s;
    __syncthreads();
    if (sg != 0) return;
    float t = red[0][j];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][j];
    const int jj = 32 * jt + j;
    if (jj < 147) dw[co * 147 + jj] = t;  // [co][c][ky][kx] with jj = c * 49 + ky * 7 + kx
}
size_t cerb_stem_wgrad_workspace_bytes() { return (size_t)1536 * 5 * 64 * 32 * 4; }
hipError_t cerb_launch_stem_wgrad_mfma(const unsigned char* tiles, const float* dy, float* dw, int N, int H, int W, void* ws, hipStream_t st) {
    StemWgradParams p;
    p.tiles = tiles; p.dy = dy; p.part = (float*)ws; p.N = N; p.H = H; p.W = W;
    p.segs_per_row = (W + SEG - 1) / SEG;
    const long long nseg = (long long)N * H * p.segs_per_row;
    p.slices = (int)(nseg < 1536 ? nseg : 1536);
    hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3((unsigned)p.slices), dim3(256), 0, st, p);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(5 * 64), dim3(256), 0, st, (const float*)ws, dw, p.slices);
    return hipGetLastError();
}

// per-channel sums over the rows of [G][rows][C] (bias gradients): coalesced partial sums per row slab, then a fixed-order finalise
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ d, long long group_stride, long long rows, int C, int slabs, float* __restrict__ part) {
    __shared__ double sh[256];
    const int g = blockIdx.x / slabs, sl = blockIdx.x % slabs;
    const long long per = (rows + slabs - 1) / slabs, r0 = sl * per, r1 = min(rows, r0 + per);
    const float* dg = d + g * group_stride;
    for (int c0 = 0; c0 < C; c0 += 256) {  // thread = (channel, row lane): all 256 threads load, the row lanes are added in lane order
        const int cw = min(256, C - c0), nrl = 256 / cw, c = c0 + (int)threadIdx.x % cw, rl = (int)threadIdx.x / cw;
        double s = 0;
        if (rl < nrl) {
            long long r = r0 + rl;
            for (; r + 3ll * nrl < r1; r += 4ll * nrl) {
                const float v0 = dg[r * C + c], v1 = dg[(r + nrl) * C + c], v2 = dg[(r + 2ll * nrl) * C + c], v3 = dg[(r + 3ll * nrl) * C + c];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; r < r1; r += nrl) s += dg[r * C + c];
        }
        __syncthreads();
        sh[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0) {
            double t = 0;
            for (int k = 0; k < nrl; ++k) t += sh[k * cw + (threadIdx.x % cw)];
            part[((long long)g * slabs + sl) * C + c] = (float)t;
        }
    }
}
// out[g][i] = sum_b part[g][b][i] in a fixed order: 64 outputs x 16 slab chunks per block, chunks combined in chunk order
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int blocks) {
    __shared__ double sh[16][64];
    const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const float* pg = part + (long long)blockIdx.y * blocks * n;
    const int per = (blocks + 15) / 16, b0 = ch * per, b1 = min(blocks, b0 + per);
    double s = 0;
    if (i < n) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            const float v0 = pg[(long long)b * n + i], v1 = pg[(long long)(b + 1) * n + i], v2 = pg[(long long)(b + 2) * n + i], v3 = pg[(long long)(b + 3) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < b1; ++b) s += pg[(long long)b * n + i];
    }
    sh[ch][lane] = s;
    __syncthreads();
    if (ch == 0 && i < n) {
        double t = 0;
        for (int k = 0; k < 16; ++k) t += sh[k][lane];
        out[(long long)blockIdx.y * n + i] = (float)t;
    }
}
hipError_t cerb_launch_slab_sum(const float* part, float* out, int n, int blocks, int groups, hipStream_t st) {
    hipLaunchKernelGGL(slab_sum_kernel, dim3((n + 63) / 64, groups), dim3(1024), 0, st, part, out, n, blocks);
    return hipGetLastError();
}
hipError_t cerb_launch_colsum(const float* d, long long group_stride, long long rows, int C, int G, float* out, void* ws, hipStream_t st) {
    int slabs = (int)(rows < 2048 ? 1 : (rows / 512 > 2048 ? 2048 : rows / 512));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(G * slabs), dim3(256), 0, st, d, group_stride, rows, C, slabs, (float*)ws);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((C + 63) / 64, G), dim3(1024), 0, st, (const float*)ws, out, C, slabs);
    return hipGetLastError();
}

// Pointwise layer as a streaming GEMM on the matrix pipe: out[r][n] (+)= bias[n] + sum_k a[r][k] B[k][n], K and NC multiples of 32 / 8 (the heads' 64 -> 96
// forward and its 96 -> 64 data gradient).  One wave = 32 rows x NC: v_mfma_f32_32x32x2_f32 with A = the rows (lane = row, half-wave = k parity block), so
// each lane fetches its row as float4s and the k order is permuted to match (k = 8 j + 4 h + e; a sum does not care); B sits in LDS in that order, one
// ds_read_b128 per four MFMAs.  Stores put 32 consecutive floats of one row per half-wave.
// bn_part != nullptr: the BatchNorm that follows the layer gets its statistics HERE -- per wave the (sum, sum of squares) of every output column
// over the rows the workgroup produced, laid out as bn_partial_kernel's partials with one "block" per workgroup ([gridDim.x][nc][2] doubles; the 16
// values of a lane and tile are added in float, tiles in double), so that bn_finalize_kernel takes them as they are and the statistics pass over
// the layer's output (1.2 GB per head at batch 16 x 448^2) does not run.  (One row per workgroup, not per wave: the finalising kernel walks the rows serially.)
template <int K, int NC, bool ACCUM, bool EXACT>
__global__ __launch_bounds__(256) void pw_mfma_kernel(const float* __restrict__ a, const float* __restrict__ w, int w_trans, const float* __restrict__ bias,
                                                      float* __restrict__ out, long long rows, int k_rt, int nc_rt, double* __restrict__ bn_part) {
    const int k_real = EXACT ? K : k_rt, nc_real = EXACT ? NC : nc_rt;  // EXACT: the shape is the template's, all strides are constants
    // K, NC: the padded GEMM (multiples of 8 / 32); k_real <= K columns of `a` and nc_real <= NC outputs exist (the heads' 96 -> 3 / 7 layer and
    // its data gradient run here with zero padding: the layer is HBM-bound, the padded MFMAs are free)
    __shared__ __attribute__((aligned(16))) float Bs[K * NC];
    for (int i = threadIdx.x; i < K * NC; i += 256) {
        const int e = i & 3, n = (i >> 2) % NC, jh = (i >> 2) / NC;  // Bs[jh][n][e], jh = 2 j + h
        const int k = 4 * jh + e;                                    // = 8 j + 4 h + e
        Bs[i] = (k < k_real && n < nc_real) ? (w_trans ? w[n * k_real + k] : w[k * nc_real + n]) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, n0 = lane & 31;
    const long long ntiles = (rows + 31) >> 5;
    const bool vec = EXACT || (k_real & 3) == 0;  // float4 fetches need 16-byte aligned rows
    float bv[NC / 32];
#pragma unroll
    for (int nb = 0; nb < NC / 32; ++nb) bv[nb] = (bias && nb * 32 + n0 < nc_real) ? bias[nb * 32 + n0] : 0.f;
    double bs[NC / 32], bq[NC / 32];
#pragma unroll
    for (int nb = 0; nb < NC / 32; ++nb) bs[nb] = bq[nb] = 0.0;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long row = min(tile * 32 + n0, rows - 1);
        f32x4 av[K / 8];
#pragma unroll
        for (int j = 0; j < K / 8; ++j) {
            const int k0 = 8 * j + 4 * h;
            if (vec) {
                av[j] = k0 < k_real ? *reinterpret_cast<const f32x4*>(a + row * k_real + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const float* ar = a + row * k_real + k0;
                av[j].x = k0 < k_real ? ar[0] : 0.f;
                av[j].y = k0 + 1 < k_real ? ar[1] : 0.f;
                av[j].z = k0 + 2 < k_real ? ar[2] : 0.f;
                av[j].w = k0 + 3 < k_real ? ar[3] : 0.f;
            }
        }
        f32x16 acc[NC / 32];
#pragma unroll
        for (int nb = 0; nb < NC / 32; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
#pragma unroll
        for (int j = 0; j < K / 8; ++j)
#pragma unroll
            for (int nb = 0; nb < NC / 32; ++nb) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(Bs + (((2 * j + h) * NC) + nb * 32 + n0) * 4);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].x, b.x, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].y, b.y, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].z, b.z, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].w, b.w, acc[nb], 0, 0, 0);
            }
        float ts[NC / 32], tq[NC / 32];
#pragma unroll
        for (int nb = 0; nb < NC / 32; ++nb) ts[nb] = tq[nb] = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const long long r = tile * 32 + 8 * (v >> 2) + 4 * h + (v & 3);
            if (r < rows) {
#pragma unroll
                for (int nb = 0; nb < NC / 32; ++nb)
                    if (nb * 32 + n0 < nc_real) {
                        float* p = out + r * nc_real + nb * 32 + n0;
                        const float val = acc[nb][v] + bv[nb];
                        *p = ACCUM ? *p + val : val;
                        ts[nb] += val;
                        tq[nb] = fmaf(val, val, tq[nb]);
                    }
            }
        }
        if (bn_part) {
#pragma unroll
            for (int nb = 0; nb < NC / 32; ++nb) {
                bs[nb] += (double)ts[nb];
                bq[nb] += (double)tq[nb];
            }
        }
    }
    if (bn_part) {  // the two half-waves hold the rows of different parity blocks; the four waves meet in LDS: ONE partial row per workgroup
        __shared__ double red[4][NC][2];
#pragma unroll
        for (int nb = 0; nb < NC / 32; ++nb) {
            const double os = __shfl_xor(bs[nb], 32), oq = __shfl_xor(bq[nb], 32);
            if (h == 0) {
                red[wave][nb * 32 + n0][0] = bs[nb] + os;
                red[wave][nb * 32 + n0][1] = bq[nb] + oq;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < nc_real; c += 256) {
            double* o = bn_part + ((long long)blockIdx.x * nc_real + c) * 2;
            o[0] = ((red[0][c][0] + red[1][c][0]) + red[2][c][0]) + red[3][c][0];
            o[1] = ((red[0][c][1] + red[1][c][1]) + red[2][c][1]) + red[3][c][1];
        }
    }
}
// returns hipErrorNotSupported for shapes the kernel is not built for (the caller keeps its scalar path)
// bn_part (optional, K = 64 / NC = 96 forward only): statistics partials of the output for the BatchNorm behind the layer, *bn_blocks rows of them
hipError_t cerb_launch_pw_mfma(const float* a, const float* w, int w_trans, const float* bias, float* out, long long rows, int K, int NC, int accumulate, hipStream_t st,
                               double* bn_part, int* bn_blocks) {
    const long long ntiles = (rows + 31) / 32;
    const unsigned blocks = (unsigned)(ntiles / 4 < 1 ? 1 : (ntiles / 4 > 2048 ? 2048 : ntiles / 4));
    if (bn_blocks) *bn_blocks = 0;
    if (bn_part && !(K == 64 && NC == 96 && !accumulate)) bn_part = nullptr;
    if (bn_part && bn_blocks) *bn_blocks = (int)blocks;
    if (K == 64 && NC == 96 && !accumulate) hipLaunchKernelGGL((pw_mfma_kernel<64, 96, false, true>), dim3(blocks), dim3(256), 0, st, a, w, w_trans, bias, out, rows, K, NC, bn_part);
    else if (K == 96 && NC == 64 && accumulate) hipLaunchKernelGGL((pw_mfma_kernel<96, 64, true, true>), dim3(blocks), dim3(256), 0, st, a, w, w_trans, bias, out, rows, K, NC, (double*)nullptr);
    else if (K == 96 && NC == 64 && !accumulate) hipLaunchKernelGGL((pw_mfma_kernel<96, 64, false, true>), dim3(blocks), dim3(256), 0, st, a, w, w_trans, bias, out, rows, K, NC, (double*)nullptr);
    else if (K == 96 && NC <= 32 && !accumulate) hipLaunchKernelGGL((pw_mfma_kernel<96, 32, false, false>), dim3(blocks), dim3(256), 0, st, a, w, w_trans, bias, out, rows, K, NC, (double*)nullptr);
    else return hipErrorNotSupported;
    return hipGetLastError();
}
