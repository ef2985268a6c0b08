// developer probe: nominal shader clock, frequency of the constant "real time" counter (s_memrealtime), and the shader clock measured
// against it while the fp32 matrix pipe is saturated (one wave per SIMD, registers only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 1) void spin(unsigned long long* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)s; }
}
int main() {
    int clk = 0, wall = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("hipDeviceAttributeClockRate %d kHz, hipDeviceAttributeWallClockRate %d kHz\n", clk, wall);
    unsigned long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, 0, d, 20000, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("kernel %.3f ms by events; in-kernel: %llu shader cycles, %llu real-time ticks -> ticks/ms %.1f, cycles per tick %.3f, %.1f TFLOP/s\n", ms, h[0], h[1],
               h[1] / ms, (double)h[0] / h[1], 256.0 * 4 * 20000 * 16 * 4096 / ms / 1e9);
    }
    return 0;
}
