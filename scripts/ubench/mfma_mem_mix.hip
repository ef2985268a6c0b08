// micro-benchmark: fp32 MFMA rate of a wave that, per 8 MFMAs (two alternating accumulators, as conv_wino's step), also issues
// L buffer_load_dwordx4 (L2-resident, consumed as A operands two steps later) and D ds_read_b128 (consumed as B operands next
// step) -- the instruction mix of conv_wino.hip's MFMA phase without its phases.  1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int L, int D>
__global__ __launch_bounds__(256, 2) void kern(float* out, const float* wsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc), 0, -1, 0x00020000);
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    f32x4 w[3][2], b[2];
    for (int s = 0; s < 3; ++s) for (int h = 0; h < 2; ++h) w[s][h] = f32x4{1.f, 2.f, 3.f, 4.f};
    b[0] = b[1] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
    int off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {  // 12 steps so that the 3-slot weight window and the 2-slot B window line up
            if (L >= 1) w[(q + 2) % 3][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off, 0));
            if (L >= 2) w[(q + 2) % 3][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off + 1024, 0));
            if (L >= 3) b[(q + 1) & 1] = b[(q + 1) & 1] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off + 2048, 0));
            if (D >= 1) b[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(lds + ((q * 64 + lane) * 4 & 8188));
            off = (off + 2048) & 0xFFFFF;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q % 3][0][t], b[q & 1][t], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q % 3][1][t], b[q & 1][t], acc[1], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int L, int D>
void run(float* d, const float* w, int wps) {
    const int iters = 400;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<L, D>), grid, block, 0, 0, d, w, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<L, D>), grid, block, 0, 0, d, w, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * wps * 4 * iters * 12.0 * 8 * 4096.0;
    printf("waves/SIMD %d  per 8 MFMAs: %d buffer_load_b128 + %d ds_read_b128 : %.3f ms  %.1f TFLOP/s\n", wps, L, D, ms, fl / ms / 1e9);
}
int main() {
    float *d, *w;
    hipMalloc(&d, 1 << 24); hipMalloc(&w, 2 << 20); hipMemset(w, 0, 2 << 20);
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 0>(d, w, wps); run<0, 1>(d, w, wps); run<1, 1>(d, w, wps); run<2, 1>(d, w, wps); run<3, 1>(d, w, wps); run<2, 0>(d, w, wps);
    }
    return 0;
}
