// conv_wgrad.hip -- weight gradient of the 3x3 stride-1 pad-1 convolutions (training step, BASELINE.json configs[4]) on the fp32 matrix
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

namespace {
constexpr int SEG = 32;                       // pixels per row segment
constexpr int XQ = SEG + 2;                   // staged input pixels per row (halo of one on each side)
constexpr int DY_FLOATS = SEG * 64;           // 2048
constexpr int X_FLOATS = 3 * XQ * 64;         // 6528
constexpr int LDS_FLOATS = DY_FLOATS + X_FLOATS;

struct WgradParams {
    const float* x;
    const float* dy;
    float* part;   // [slices][tiles][9][64 co][64 ci]
    int G, N, H, W, Cin, Cout, slices, segs_per_row;
    long long x_gs, dy_gs;
};
}  // namespace

__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* dyl = lds;
    float* xl = lds + DY_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = wave & 1, ih = wave >> 1, j = lane & 31, k = lane >> 5;
    const int ncb = p.Cout >> 6, ncib = p.Cin >> 6;
    const int tiles = p.G * ncb * ncib;
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
    const float* xg = p.x + g * p.x_gs + cib * 64;
    const float* dyg = p.dy + g * p.dy_gs + cb * 64;
    const long long nseg = (long long)p.N * p.H * p.segs_per_row;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (long long s = slice; s < nseg; s += p.slices) {
        const int sx = (int)(s % p.segs_per_row);
        const long long ry = s / p.segs_per_row;
        const int y = (int)(ry % p.H), n = (int)(ry / p.H);
        const int x0 = sx * SEG;
        __syncthreads();  // the previous segment's MFMAs have read their operands
        // ---- stage dy: 32 px x 64 co ------------------------------------------------------------------------------------------------
        for (int i = tid; i < DY_FLOATS / 4; i += 256) {
            const int px = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (x0 + px < p.W) v = *reinterpret_cast<const f32x4*>(dyg + (((long long)n * p.H + y) * p.W + x0 + px) * p.Cout + 4 * c4);
            *reinterpret_cast<f32x4*>(dyl + px * 64 + 4 * c4) = v;
        }
        // ---- stage x: rows y-1 .. y+1, pixels x0-1 .. x0+32, 64 ci ---------------------------------------------------------------------
        for (int i = tid; i < X_FLOATS / 4; i += 256) {
            const int c4 = i & 15, q = (i >> 4) % XQ, r = (i >> 4) / XQ;
            const int yy = y + r - 1, xx = x0 + q - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = *reinterpret_cast<const f32x4*>(xg + (((long long)n * p.H + yy) * p.W + xx) * p.Cin + 4 * c4);
            *reinterpret_cast<f32x4*>(xl + (r * XQ + q) * 64 + 4 * c4) = v;
        }
        __syncthreads();
        // ---- 16 pixel pairs x 9 taps ------------------------------------------------------------------------------------------------------
#pragma unroll 4
        for (int pp = 0; pp < SEG / 2; ++pp) {
            const float a = dyl[(2 * pp + k) * 64 + 32 * ch + j];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float b = xl[((t / 3) * XQ + 2 * pp + k + (t % 3)) * 64 + 32 * ih + j];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- this slice's partial tile: part[slice][tile][tap][co 64][ci 64]; MFMA D layout: row = rq*8 + (lane>>5)*4 + e, column = lane & 31 ----
    float* o = p.part + ((long long)slice * tiles + tile) * 9 * 4096;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 32 * ch + rq * 8 + k * 4 + e, ci = 32 * ih + j;
                o[t * 4096 + co * 64 + ci] = acc[t][rq * 4 + e];
            }
}

// dW[g][cb*64 + co][cib*64 + ci][tap] = sum over slices, in slice order
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G, int Cin, int Cout, int slices) {
    const int ncb = Cout >> 6, ncib = Cin >> 6, tiles = G * ncb * ncib;
    const long long total = (long long)tiles * 9 * 4096;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += part[(long long)sl * total + i];
        const int ci = (int)(i & 63), co = (int)((i >> 6) & 63), t = (int)((i >> 12) % 9), tile = (int)(i / (9 * 4096));
        const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
        dw[(((long long)g * Cout + cb * 64 + co) * Cin + cib * 64 + ci) * 9 + t] = s;
    }
}

size_t cerb_wgrad3x3_workspace_bytes(int G, int N, int H, int W, int Cin, int Cout, int* slices_out) {
    const int tiles = G * (Cout / 64) * (Cin / 64);
    const long long nseg = (long long)N * H * ((W + SEG - 1) / SEG);
    long long slices = 1536 / tiles;  // about six workgroups per CU in flight over the whole grid
    if (slices < 1) slices = 1;
    if (slices > nseg) slices = nseg;
    if (slices_out) *slices_out = (int)slices;
    return (size_t)slices * tiles * 9 * 4096 * 4;
}

hipError_t cerb_launch_wgrad3x3(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, long long x_gs, void* ws, hipStream_t st) {
    if (Cin % 64 || Cout % 64) return hipErrorInvalidValue;
    WgradParams p;
    p.x = x; p.dy = dy; p.part = (float*)ws;
    p.G = G; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.segs_per_row = (W + SEG - 1) / SEG;
    p.x_gs = x_gs; p.dy_gs = (long long)N * H * W * Cout;
    int slices = 1;
    (void)cerb_wgrad3x3_workspace_bytes(G, N, H, W, Cin, Cout, &slices);
    p.slices = slices;
    const int tiles = G * (Cout / 64) * (Cin / 64);
    hipLaunchKernelGGL(wgrad3x3_kernel, dim3((unsigned)(tiles * slices)), dim3(256), 0, st, p);
    long long blocks = ((long long)tiles * 9 * 4096 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(wgrad3x3_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, dw, G, Cin, Cout, slices);
    return hipGetLastError();
}
