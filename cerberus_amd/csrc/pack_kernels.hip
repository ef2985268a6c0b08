// Weight packing on the device, for handles packed for training (cerb_net_set_fold_bn(net, 0)): after every optimiser step the raw
// state-dict weights [cout][cin][ks][ks] are re-laid out for the conv kernels without a host round trip of the packed copies.
//   pack_conv_kernel : the implicit-GEMM layout of conv_igemm.hip   [cb][chunk][tap][G][s][lane][t]
//   pack_wino_kernel : Winograd F(2x2,3x3) filter transform U = G g G^T (double, rounded once) in the layout of conv_wino.hip
//                      [cb][chunk][a][b][G][s][lane][t]; with dgrad = 1 the filter is the data-gradient one, W'[ci][co][tap] = W[co][ci][8 - tap]
// Same index maps and the same arithmetic as the host packers in cerb_api.hip (pack_conv / pack_wino), which inference handles keep using.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "cerb_common.h"

// one re-pack job of pack_multi_kernel: kind 0 = implicit-GEMM layout (a = taps, b = chunk), 1 = F(2x2) transform (a = dgrad), 2 = F(4x4) transform
// (a = dgrad, b = chunk32); (cout, cin) as the single-launch packers take them
struct PackJob {
    const float* w;
    float* out;
    long long total;
    int cout, cin, kind, a, b, pad;
};

namespace {
__device__ __forceinline__ void pack_conv_elem(const float* __restrict__ w, float* __restrict__ out, long long i, int cout, int cin, int T, int CB) {
    const int NG = CB / 8, nchunk = cin / CB;
    {
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
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int T, int CB) {
    const long long total = (long long)cout * cin * T;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) pack_conv_elem(w, out, i, cout, cin, T, CB);
}
__device__ __forceinline__ void pack_wino_elem(const float* __restrict__ w, float* __restrict__ out, long long i, int cout, int cin, int dgrad) {
#pragma clang fp contract(off)
    const int nchunk = cin / 32;
    {
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
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int dgrad) {
    const long long total = (long long)cout * cin * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) pack_wino_elem(w, out, i, cout, cin, dgrad);
}
// Winograd F(4x4,3x3) filter transform (points 0, 1, -1, 2, -2, inf; the host packer is cerb_api.hip: pack_wino4) in the layouts of
// conv_wino4.hip  [cb][16-channel chunk][wave a][xi][lane][t]      (chunk32 = 0)  and
// conv_wino4b.hip [cb][32-channel chunk][wave a][xi][G][lane][t]   (chunk32 = 1); `cout` / `cin` are the convolution's own (for the data
// gradient: the forward layer's cin / cout, filter W'[ci][co][tap] = W[co][ci][8 - tap])
__device__ __forceinline__ void pack_wino4_elem(const float* __restrict__ w, float* __restrict__ out, long long i, int cout, int cin, int dgrad, int chunk32) {
#pragma clang fp contract(off)
    const double Gm[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    {
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
// One (cout block cb, channel chunk ch, wave slot a) of the F(4x4) layouts per call: thread = (lane, t) [x G for the 32-channel layout] owns ONE filter
// (co, ci), loads its nine taps ONCE, forms all 36 transformed values with the very same double-precision expressions as pack_wino4_elem (bit-identical),
// and the workgroup writes them position by position as contiguous 1-KiB rows.  The per-element form gathered the nine taps again for every one of the
// 36 outputs: 7 GB of scattered reads per optimiser step, 3.4 ms of a 121 ms training step (round 5).
__device__ __forceinline__ void pack_wino4_block(const float* __restrict__ w, float* __restrict__ out, int blk, int cout, int cin, int dgrad, int chunk32) {
#pragma clang fp contract(off)
    const double Gm[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int nchunk = chunk32 ? cin / 32 : cin / 16;
    const int a = blk & 3, ch = (blk >> 2) % nchunk, cb = (blk >> 2) / nchunk;
    const int tid = threadIdx.x, t = tid & 3, lane = tid >> 2;
    const int co = cb * 64 + 16 * a + (lane & 15);
    for (int G = 0; G < (chunk32 ? 2 : 1); ++G) {
        const int ci = chunk32 ? ch * 32 + 16 * G + 4 * (lane >> 4) + t : ch * 16 + 4 * (lane >> 4) + t;
        double g[3][3];
        for (int k = 0; k < 9; ++k)
            g[k / 3][k % 3] = dgrad ? (double)w[((long long)ci * cout + co) * 9 + (8 - k)] : (double)w[((long long)co * cin + ci) * 9 + k];
        float* o = out + (long long)blk * 36 * (chunk32 ? 512 : 256) + (chunk32 ? G * 256 : 0) + tid;
#pragma unroll
        for (int ya = 0; ya < 6; ++ya) {
            double tx[3];
            for (int x = 0; x < 3; ++x) tx[x] = Gm[ya][0] * g[0][x] + Gm[ya][1] * g[1][x] + Gm[ya][2] * g[2][x];
#pragma unroll
            for (int xb = 0; xb < 6; ++xb) o[(ya * 6 + xb) * (chunk32 ? 512 : 256)] = (float)(tx[0] * Gm[xb][0] + tx[1] * Gm[xb][1] + tx[2] * Gm[xb][2]);
        }
    }
}
__global__ __launch_bounds__(256) void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int dgrad, int chunk32) {
    w += (size_t)blockIdx.y * cout * cin * 9;    // one launch packs all the groups of a conv (blockIdx.y)
    out += (size_t)blockIdx.y * cout * cin * 36;
    const int nblk = (cout / 64) * (chunk32 ? cin / 32 : cin / 16) * 4;
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) pack_wino4_block(w, out, b, cout, cin, dgrad, chunk32);
}
// Every re-pack of an optimiser step in ONE launch (round 5: the per-conv launches -- 87 F(4x4) transforms of ~34 us, 85 implicit-GEMM re-layouts -- were
// 3 ms of a 121 ms training step, all of it launch latency): a table of jobs, a workgroup per (job, 65536-element chunk), the same per-element functions.
constexpr long long PACK_CHUNK = 65536;
constexpr int PACK_W4_BLOCKS = 4;
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, const int2* __restrict__ chunks) {
    const int2 jc = chunks[blockIdx.x];
    const PackJob j = jobs[jc.x];
    if (j.kind == 2) {  // a chunk = PACK_W4_BLOCKS (cb, ch, a) blocks of the F(4x4) layout
        const int nblk = (j.cout / 64) * (j.b ? j.cin / 32 : j.cin / 16) * 4;
        for (int b = jc.y * PACK_W4_BLOCKS; b < min(nblk, (jc.y + 1) * PACK_W4_BLOCKS); ++b) pack_wino4_block(j.w, j.out, b, j.cout, j.cin, j.a, j.b);
        return;
    }
    const long long i0 = (long long)jc.y * PACK_CHUNK, i1 = min(j.total, i0 + PACK_CHUNK);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (j.kind == 0) pack_conv_elem(j.w, j.out, i, j.cout, j.cin, j.a, j.b);
        else pack_wino_elem(j.w, j.out, i, j.cout, j.cin, j.a);
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
    hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)std::min<long long>((long long)cout * cin / (chunk32 ? 512 : 256), 8192), groups), dim3(256), 0, st, w_raw, out, cout, cin, dgrad, chunk32);
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

// jobs: host array; dev_tab / dev_bytes / host_prev: the caller's cached device copy of the table (rebuilt only when the job list changes)
hipError_t cerb_launch_pack_multi(const PackJob* jobs, int count, void** dev_tab, size_t* dev_bytes, std::vector<char>* host_prev, hipStream_t st) {
    std::vector<int2> ch;
    for (int i = 0; i < count; ++i) {
        if (jobs[i].kind == 2) {
            const int nblk = (jobs[i].cout / 64) * (jobs[i].b ? jobs[i].cin / 32 : jobs[i].cin / 16) * 4;
            for (int c = 0; c * PACK_W4_BLOCKS < nblk; ++c) ch.push_back(make_int2(i, c));
        } else {
            for (long long c = 0; c * PACK_CHUNK < jobs[i].total; ++c) ch.push_back(make_int2(i, (int)c));
        }
    }
    if (ch.empty()) return hipSuccess;
    const size_t tb = ((size_t)count * sizeof(PackJob) + 255) & ~(size_t)255, need = tb + ch.size() * sizeof(int2);
    std::vector<char> host(need, 0);
    memcpy(host.data(), jobs, (size_t)count * sizeof(PackJob));
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
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)ch.size()), dim3(256), 0, st, (const PackJob*)*dev_tab, (const int2*)((const char*)*dev_tab + tb));
    return hipGetLastError();
}
