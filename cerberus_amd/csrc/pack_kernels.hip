// Weight packing on the device, for handles packed for training (cerb_net_set_fold_bn(net, 0)): after every optimiser step the raw
// state-dict weights [cout][cin][ks][ks] are re-laid out for the conv kernels without a host round trip of the packed copies.
//   pack_conv_kernel : the implicit-GEMM layout of conv_igemm.hip   [cb][chunk][tap][G][s][lane][t]
//   pack_wino_kernel : Winograd F(2x2,3x3) filter transform U = G g G^T (double, rounded once) in the layout of conv_wino.hip
//                      [cb][chunk][a][b][G][s][lane][t]; with dgrad = 1 the filter is the data-gradient one, W'[ci][co][tap] = W[co][ci][8 - tap]
// Same index maps and the same arithmetic as the host packers in cerb_api.hip (pack_conv / pack_wino), which inference handles keep using.
#include <hip/hip_runtime.h>

#include "cerb_common.h"

namespace {
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int T, int CB) {
    const int NG = CB / 8, nchunk = cin / CB;
    const long long total = (long long)cout * cin * T;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int t = (int)(r & 3); r >>= 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int s = (int)(r & 1); r >>= 1;
        const int G = (int)(r % NG); r /= NG;
        const int tap = (int)(r % T); r /= T;
        const int ch = (int)(r % nchunk);
        const int cb = (int)(r / nchunk);
        const int co = cb * 64 + s * 32 + (lane & 31), ci = ch * CB + G * 8 + 4 * (lane >> 5) + t;
        out[i] = w[((long long)co * cin + ci) * T + tap];
    }
}
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int dgrad) {
#pragma clang fp contract(off)
    const int nchunk = cin / 32;
    const long long total = (long long)cout * cin * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int t = (int)(r & 3); r >>= 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int s = (int)(r & 1); r >>= 1;
        const int G = (int)(r & 3); r >>= 2;
        const int b = (int)(r & 3); r >>= 2;
        const int a = (int)(r & 3); r >>= 2;
        const int ch = (int)(r % nchunk);
        const int cb = (int)(r / nchunk);
        const int co = cb * 64 + s * 32 + (lane & 31), ci = ch * 32 + G * 8 + 4 * (lane >> 5) + t;
        double g[3][3];
        for (int k = 0; k < 9; ++k)
            g[k / 3][k % 3] = dgrad ? (double)w[((long long)ci * cout + co) * 9 + (8 - k)] : (double)w[((long long)co * cin + ci) * 9 + k];
        // rows of G: {1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}; products with 0 are kept so that the sums round as on the host
        const double Ga[3] = {a == 0 ? 1.0 : (a == 3 ? 0.0 : 0.5), a == 1 ? 0.5 : (a == 2 ? -0.5 : 0.0), a == 0 ? 0.0 : (a == 3 ? 1.0 : 0.5)};
        const double Gb[3] = {b == 0 ? 1.0 : (b == 3 ? 0.0 : 0.5), b == 1 ? 0.5 : (b == 2 ? -0.5 : 0.0), b == 0 ? 0.0 : (b == 3 ? 1.0 : 0.5)};
        double tx[3];
        for (int x = 0; x < 3; ++x) tx[x] = Ga[0] * g[0][x] + Ga[1] * g[1][x] + Ga[2] * g[2][x];
        out[i] = (float)(tx[0] * Gb[0] + tx[1] * Gb[1] + tx[2] * Gb[2]);
    }
}
// Winograd F(4x4,3x3) filter transform (points 0, 1, -1, 2, -2, inf; the host packer is cerb_api.hip: pack_wino4) in the layouts of
// conv_wino4.hip  [cb][16-channel chunk][wave a][xi][lane][t]      (chunk32 = 0)  and
// conv_wino4b.hip [cb][32-channel chunk][wave a][xi][G][lane][t]   (chunk32 = 1); `cout` / `cin` are the convolution's own (for the data
// gradient: the forward layer's cin / cout, filter W'[ci][co][tap] = W[co][ci][8 - tap])
__global__ __launch_bounds__(256) void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int dgrad, int chunk32) {
#pragma clang fp contract(off)
    w += (size_t)blockIdx.y * cout * cin * 9;    // one launch packs all the groups of a conv (blockIdx.y)
    out += (size_t)blockIdx.y * cout * cin * 36;
    const double Gm[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const long long total = (long long)cout * cin * 36;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int t = (int)(r & 3); r >>= 2;
        const int lane = (int)(r & 63); r >>= 6;
        int G = 0;
        if (chunk32) { G = (int)(r & 1); r >>= 1; }
        const int xi = (int)(r % 36); r /= 36;
        const int a = (int)(r & 3); r >>= 2;
        const int nchunk = chunk32 ? cin / 32 : cin / 16;
        const int ch = (int)(r % nchunk);
        const int cb = (int)(r / nchunk);
        const int co = cb * 64 + 16 * a + (lane & 15);
        const int ci = chunk32 ? ch * 32 + 16 * G + 4 * (lane >> 4) + t : ch * 16 + 4 * (lane >> 4) + t;
        double g[3][3];
        for (int k = 0; k < 9; ++k)
            g[k / 3][k % 3] = dgrad ? (double)w[((long long)ci * cout + co) * 9 + (8 - k)] : (double)w[((long long)co * cin + ci) * 9 + k];
        const int ya = xi / 6, xb = xi % 6;
        double tx[3];
        for (int x = 0; x < 3; ++x) tx[x] = Gm[ya][0] * g[0][x] + Gm[ya][1] * g[1][x] + Gm[ya][2] * g[2][x];
        out[i] = (float)(tx[0] * Gm[xb][0] + tx[1] * Gm[xb][1] + tx[2] * Gm[xb][2]);
    }
}
// stem 7x7 (cerb_api.hip: cerb_net_finalize): wp[ky 7][t 12][s 2][lane 64] = W[32 s + (lane & 31)][c][ky][kx] with kk = 2 t + (lane >> 5) = 3 kx + c < 21
__global__ void pack_stem_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 7 * 12 * 2 * 64) return;
    const int lane = i & 63, s = (i >> 6) & 1, t = (i >> 7) % 12, ky = (i >> 7) / 12;
    const int co = s * 32 + (lane & 31), kk = 2 * t + (lane >> 5);
    out[i] = kk < 21 ? w[((co * 3 + kk % 3) * 7 + ky) * 7 + kk / 3] : 0.f;
}
unsigned pack_grid(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
}  // namespace

hipError_t cerb_launch_pack_conv(const float* w_raw, float* out, int cout, int cin, int ks, int chunk, hipStream_t st) {
    hipLaunchKernelGGL(pack_conv_kernel, dim3(pack_grid((long long)cout * cin * ks * ks)), dim3(256), 0, st, w_raw, out, cout, cin, ks * ks, chunk);
    return hipGetLastError();
}
// (cout, cin) are those of the conv the packed filter serves: for dgrad = 1 the transposed pair of the raw tensor
hipError_t cerb_launch_pack_wino4(const float* w_raw, float* out, int cout, int cin, int dgrad, int chunk32, int groups, hipStream_t st) {
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(pack_grid((long long)cout * cin * 36), groups), dim3(256), 0, st, w_raw, out, cout, cin, dgrad, chunk32);
    return hipGetLastError();
}
hipError_t cerb_launch_pack_wino(const float* w_raw, float* out, int cout, int cin, int dgrad, hipStream_t st) {
    hipLaunchKernelGGL(pack_wino_kernel, dim3(pack_grid((long long)cout * cin * 16)), dim3(256), 0, st, w_raw, out, cout, cin, dgrad);
    return hipGetLastError();
}
hipError_t cerb_launch_pack_stem(const float* w_raw, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_stem_kernel, dim3((7 * 12 * 2 * 64 + 255) / 256), dim3(256), 0, st, w_raw, out);
    return hipGetLastError();
}
