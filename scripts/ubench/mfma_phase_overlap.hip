// micro-benchmark: do two waves on one SIMD overlap one's MFMA phase with the other's VALU / LDS phase?
// Each wave alternates an "MFMA phase" (MF MFMAs on two accumulators) and an "other phase" (V independent v_fma_f32 + W LDS
// round trips).  Reported: matrix-pipe utilisation at 1 and 2 waves per SIMD.  conv_wino.hip's item: MF = 256, V ~ 250.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MF, int V, int W, int PRIO>
__global__ __launch_bounds__(256, 2) void kern(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll 16
        for (int u = 0; u < MF; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k & 15] = __builtin_fmaf(v[k & 15], a, b);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            *reinterpret_cast<f32x4*>(lds + lane * 4 + (threadIdx.x >> 6) * 512) = f32x4{v[0], v[1], v[2], v[3]};
            __syncthreads();
            const f32x4 r = *reinterpret_cast<const f32x4*>(lds + lane * 4 + (((threadIdx.x >> 6) + 1) & 3) * 512);
            v[k & 15] += r[0];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MF, int V, int W, int PRIO>
void run(float* d, int wps) {
    const int iters = 300;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<MF, V, W, PRIO>), grid, block, 0, 0, d, 5, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MF, V, W, PRIO>), grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * wps * 4 * iters * (double)MF * 4096.0;
    printf("waves/SIMD %d  %3d MFMA + %3d VALU + %d LDS exchanges%s : %.3f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", wps, MF, V, W, PRIO ? " [setprio]" : "", ms, fl / ms / 1e9,
           fl / ms / 1e9 / 1.573);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int wps = 1; wps <= 2; ++wps) {
        run<256, 0, 0, 0>(d, wps);
        run<256, 256, 0, 0>(d, wps);
        run<256, 1024, 0, 0>(d, wps);
        run<256, 256, 2, 0>(d, wps);
        run<256, 256, 8, 0>(d, wps);
        run<256, 1024, 0, 1>(d, wps);
        run<256, 256, 8, 1>(d, wps);
    }
    return 0;
}
