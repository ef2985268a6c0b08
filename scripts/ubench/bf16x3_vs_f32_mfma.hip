// micro-benchmark: fp32-equivalent GEMM rate of the bf16x3 split (six v_mfma_f32_32x32x16_bf16 per K=16 step, fp32 accumulate)
// against the native v_mfma_f32_32x32x2_f32, with and without the VALU work of splitting one operand on the fly.
// Informs DESIGN par.9: it is NOT used by the product (configs[1] of BASELINE.json says fp32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void k_f32(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);  // K = 16 per 8 MFMAs per block
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void k_bf16x3(float* out, int iters, const float* src) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 ah, am, al, bh, bm, bl;
    for (int i = 0; i < 8; ++i) {
        ah[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); am[i] = (__bf16)1e-3f; al[i] = (__bf16)1e-6f;
        bh[i] = (__bf16)0.5f; bm[i] = (__bf16)2e-3f; bl[i] = (__bf16)3e-6f;
    }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = src[threadIdx.x * 8 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // one K = 16 step of one 32x32 block = 6 bf16 MFMAs  (== 8 fp32 MFMAs)
            if (SPLIT) {  // split the B operand (8 fp32 per lane) into three bf16 planes: what a fused conv would do per K step
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = x[i] + (float)u;
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    const float r2 = r1 - (float)m;
                    bh[i] = h; bm[i] = m; bl[i] = (__bf16)r2;
                }
            }
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[u], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipEventRecord(e0);
    launch(2000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *d, *src;
    hipMalloc(&d, 1 << 24); hipMalloc(&src, 1 << 16); hipMemset(src, 0, 1 << 16);
    const dim3 grid(512), block(256);
    const double waves = 512.0 * 4, iters = 2000;
    const double fl = waves * iters * 4 * (32.0 * 32 * 16 * 2);  // fp32-equivalent FLOPs: 4 blocks x K=16 per iteration
    float ms = time_ms([&](int it) { hipLaunchKernelGGL(k_f32, grid, block, 0, 0, d, it, 1.0001f, 0.5f); });
    printf("native fp32 MFMA (32x32x2)         : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    ms = time_ms([&](int it) { hipLaunchKernelGGL(k_bf16x3<0>, grid, block, 0, 0, d, it, src); });
    printf("bf16x3, 6 MFMA (32x32x16) per K=16 : %.3f ms  %.1f TFLOP/s fp32-equivalent\n", ms, fl / ms / 1e9);
    ms = time_ms([&](int it) { hipLaunchKernelGGL(k_bf16x3<1>, grid, block, 0, 0, d, it, src); });
    printf("bf16x3 + on-the-fly split of B      : %.3f ms  %.1f TFLOP/s fp32-equivalent\n", ms, fl / ms / 1e9);
    return 0;
}
