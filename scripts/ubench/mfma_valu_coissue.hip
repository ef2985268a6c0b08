// micro-benchmark: can a wave keep v_mfma_f32_32x32x2_f32 at full rate while issuing K v_pk_fma_f32 per MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__global__ __launch_bounds__(256, 2) void kern(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x * 1e-3f + i, 1.f};
    const f32x2 m = {a, a}, c = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) v[(u * K + k) & 15] = __builtin_elementwise_fma(v[(u * K + k) & 15], m, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K>
void run(float* d, int waves_per_simd) {
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<K>, grid, block, 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<K>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = 256.0 * waves_per_simd * 4;
    const double mfma_fl = waves * iters * 16.0 * 4096.0, valu_fl = waves * iters * 16.0 * K * 64 * 2 * 2;
    printf("waves/SIMD %d  K=%2d pk_fma per MFMA: %.3f ms  MFMA %.1f TF  VALU %.1f TF  sum %.1f\n", waves_per_simd, K, ms, mfma_fl / ms / 1e9, valu_fl / ms / 1e9,
           (mfma_fl + valu_fl) / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int w = 1; w <= 2; ++w) { run<0>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); run<12>(d, w); run<16>(d, w); }
    return 0;
}
