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
    int G, N, H, W, Ho, Wo, Cin, Cout, slices, segs_per_row;  // H, W: input map; Ho, Wo: output map; Cin / Cout: real channel counts (multiples of 4)
    long long x_gs, dy_gs;
};
}  // namespace

template <int KS, int STRIDE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, XQ = (SEG - 1) * STRIDE + KS, X_FLOATS = KS * XQ * 64;
    __shared__ __attribute__((aligned(16))) float lds[DY_FLOATS + X_FLOATS];
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
#pragma unroll 4
        for (int pp = 0; pp < SEG / 2; ++pp) {
            const float a = dyl[(2 * pp + k) * 64 + 32 * ch + j];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float b = xl[((t / KS) * XQ + (2 * pp + k) * STRIDE + (t % KS)) * 64 + 32 * ih + j];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
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

// dW[g][cb*64 + co][cib*64 + ci][tap] = sum over slices, in slice order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G, int Cin, int Cout, int T, int slices) {
    const int ncb = (Cout + 63) >> 6, ncib = (Cin + 63) >> 6, tiles = G * ncb * ncib;
    const long long total = (long long)tiles * T * 4096;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i & 63), co = (int)((i >> 6) & 63), t = (int)((i >> 12) % T), tile = (int)(i / ((long long)T * 4096));
        const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
        if (cb * 64 + co >= Cout || cib * 64 + ci >= Cin) continue;
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += part[(long long)sl * total + i];
        dw[(((long long)g * Cout + cb * 64 + co) * Cin + cib * 64 + ci) * T + t] = s;
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
    return (size_t)slices * tiles * ks * ks * 4096 * 4;
}

// x: [G][N][H][W][Cin], dy: [G][N][Ho][Wo][Cout] with Ho = H / stride; dw: [G][Cout][Cin][ks][ks].  ks in {1, 3}, stride in {1, 2}; channel counts
// multiples of 4.  Pointwise layers: N = H = 1, W = rows.
hipError_t cerb_launch_wgrad(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, int ks, int stride, long long x_gs, void* ws,
                             hipStream_t st) {
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
    const dim3 grid((unsigned)(tiles * slices));
    if (ks == 3 && stride == 1) hipLaunchKernelGGL((wgrad_kernel<3, 1>), grid, dim3(256), 0, st, p);
    else if (ks == 3) hipLaunchKernelGGL((wgrad_kernel<3, 2>), grid, dim3(256), 0, st, p);
    else if (stride == 1) hipLaunchKernelGGL((wgrad_kernel<1, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<1, 2>), grid, dim3(256), 0, st, p);
    long long blocks = ((long long)tiles * ks * ks * 4096 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, dw, G, Cin, Cout, ks * ks, slices);
    return hipGetLastError();
}
