// micro-benchmark: cost of v_accvgpr_read_b32, v_pk_add_f32, v_max_f32, v_add_f32 issue streams on one wave per SIMD (gfx950)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/accvgpr_read_rate.hip -o /tmp/accrd && /tmp/accrd
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(512) void k(unsigned long long* out, int mode, int iters) {
    unsigned long long t0 = 0, t1 = 0;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {1.f, 2.f}, p3 = {3.f, 4.f};
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0" ::"v"(v0) : "a0", "a1", "a2", "a3");
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {  // 64 independent accvgpr reads
            REP16(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3));)
        } else if (mode == 1) {  // 64 independent packed adds
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2));)
        } else if (mode == 2) {  // 64 independent v_max_f32
            REP16(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v1));)
        } else if (mode == 3) {  // 64 DEPENDENT packed adds (one chain)
            REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));)
        } else if (mode == 4) {  // accvgpr read followed directly by a dependent packed add (x32)
            REP16(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_pk_add_f32 %2, %2, %2\n v_add_f32 %0, %0, %1" : "=v"(v0), "=v"(v1), "+v"(p0));)
        } else if (mode == 6) {  // 64 matrix instructions on four accumulators (the 32-cycle yardstick of the counter)
            REP16(asm volatile("v_mfma_f32_16x16x4_f32 a[4:7], %0, %1, a[4:7]\n v_mfma_f32_16x16x4_f32 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x4_f32 a[12:15], %0, %1, a[12:15]\n v_mfma_f32_16x16x4_f32 a[16:19], %0, %1, a[16:19]" ::"v"(v0), "v"(v1) : "a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");)
        } else if (mode == 7) {  // 4 MFMA + 4 accvgpr_read of OTHER registers interleaved: do the reads hide in the matrix pipe's shadow?
            REP16(asm volatile("v_mfma_f32_16x16x4_f32 a[4:7], %4, %5, a[4:7]\n v_accvgpr_read_b32 %0, a0\n v_mfma_f32_16x16x4_f32 a[8:11], %4, %5, a[8:11]\n v_accvgpr_read_b32 %1, a1\n v_mfma_f32_16x16x4_f32 a[12:15], %4, %5, a[12:15]\n v_accvgpr_read_b32 %2, a2\n v_mfma_f32_16x16x4_f32 a[16:19], %4, %5, a[16:19]\n v_accvgpr_read_b32 %3, a3" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(p2[0]), "v"(p2[1]) : "a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");)
        } else if (mode == 8) {  // 4 MFMA + 4 v_pk_add interleaved
            REP16(asm volatile("v_mfma_f32_16x16x4_f32 a[4:7], %4, %5, a[4:7]\n v_pk_add_f32 %0, %0, %6\n v_mfma_f32_16x16x4_f32 a[8:11], %4, %5, a[8:11]\n v_pk_add_f32 %1, %1, %6\n v_mfma_f32_16x16x4_f32 a[12:15], %4, %5, a[12:15]\n v_pk_add_f32 %2, %2, %6\n v_mfma_f32_16x16x4_f32 a[16:19], %4, %5, a[16:19]\n v_pk_add_f32 %3, %3, %6" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(v2), "v"(v3), "v"(p2) : "a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");)
        } else if (mode == 9) {  // 4 MFMA + 4 scalar v_add_f32 interleaved
            REP16(asm volatile("v_mfma_f32_16x16x4_f32 a[4:7], %4, %5, a[4:7]\n v_add_f32 %0, %0, %5\n v_mfma_f32_16x16x4_f32 a[8:11], %4, %5, a[8:11]\n v_add_f32 %1, %1, %5\n v_mfma_f32_16x16x4_f32 a[12:15], %4, %5, a[12:15]\n v_add_f32 %2, %2, %5\n v_mfma_f32_16x16x4_f32 a[16:19], %4, %5, a[16:19]\n v_add_f32 %3, %3, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(p2[0]), "v"(p2[1]) : "a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");)
        } else if (mode == 10) {  // 4x4x1 matrix instructions, B from an AccVGPR, 2 interleaved accumulate chains (x64)
            REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, a0, v[20:23]\n v_mfma_f32_4x4x1_16b_f32 v[24:27], %0, a1, v[24:27]\n v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, a2, v[20:23]\n v_mfma_f32_4x4x1_16b_f32 v[24:27], %0, a3, v[24:27]" ::"v"(v1) : "v20","v21","v22","v23","v24","v25","v26","v27");)
        } else if (mode == 11) {  // the same with 4 chains
            REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, a0, v[20:23]\n v_mfma_f32_4x4x1_16b_f32 v[24:27], %0, a1, v[24:27]\n v_mfma_f32_4x4x1_16b_f32 v[28:31], %0, a2, v[28:31]\n v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, a3, v[32:35]" ::"v"(v1) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
        } else if (mode == 12) {  // 4 chains, B from VGPRs
            REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, %1, v[20:23]\n v_mfma_f32_4x4x1_16b_f32 v[24:27], %0, %2, v[24:27]\n v_mfma_f32_4x4x1_16b_f32 v[28:31], %0, %3, v[28:31]\n v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, %1, v[32:35]" ::"v"(v1), "v"(v0), "v"(v2), "v"(v3) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");)
        } else if (mode == 13) {  // 8 chains, B from AccVGPRs
            REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 v[20:23], %0, a0, v[20:23]\n v_mfma_f32_4x4x1_16b_f32 v[24:27], %0, a1, v[24:27]\n v_mfma_f32_4x4x1_16b_f32 v[28:31], %0, a2, v[28:31]\n v_mfma_f32_4x4x1_16b_f32 v[32:35], %0, a3, v[32:35]" ::"v"(v1) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
                  asm volatile("v_mfma_f32_4x4x1_16b_f32 v[36:39], %0, a0, v[36:39]\n v_mfma_f32_4x4x1_16b_f32 v[40:43], %0, a1, v[40:43]\n v_mfma_f32_4x4x1_16b_f32 v[44:47], %0, a2, v[44:47]\n v_mfma_f32_4x4x1_16b_f32 v[48:51], %0, a3, v[48:51]" ::"v"(v1) : "v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");)
        } else if (mode == 5) {  // packed fma x64 independent
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2));)
        }
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[mode] = t1 - t0;
    if (v0 + v1 + v2 + v3 + p0[0] + p1[1] + p2[0] + p3[1] == 1.2345e-30f) out[63] = 1;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64 * 8);
    hipMemset(d, 0, 64 * 8);
    const int iters = 200;
    for (int m = 0; m < 14; ++m) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, m, iters);
    hipDeviceSynchronize();
    unsigned long long h[64], h2[64];
    hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    // the same streams with TWO waves per SIMD (512-thread workgroups): per-wave time; equal to the one-wave time = the second wave was free
    for (int m = 0; m < 14; ++m) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, m, iters);
    hipDeviceSynchronize();
    hipMemcpy(h2, d, 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"v_accvgpr_read_b32 (independent)", "v_pk_add_f32 (independent)", "v_max_f32 (independent)", "v_pk_add_f32 (dependent chain)",
                           "2 accvgpr_read + pk_add + dependent add", "v_pk_fma_f32 (independent)", "v_mfma_f32_16x16x4_f32 (4 accumulators)",
                           "MFMA + accvgpr_read pairs (per pair / 2)", "MFMA + v_pk_add_f32 pairs (per pair / 2)", "MFMA + v_add_f32 pairs (per pair / 2)",
                           "v_mfma_f32_4x4x1 B=AGPR, 2 chains", "v_mfma_f32_4x4x1 B=AGPR, 4 chains", "v_mfma_f32_4x4x1 B=VGPR, 4 chains", "v_mfma_f32_4x4x1 B=AGPR, 8 chains (x2 count)"};
    for (int m = 0; m < 14; ++m) printf("%-48s %6.2f ticks per instruction, one wave per SIMD | %6.2f with two (per wave)\n", names[m], (double)h[m] / (iters * 64.0), (double)h2[m] / (iters * 64.0));
    return 0;
}
