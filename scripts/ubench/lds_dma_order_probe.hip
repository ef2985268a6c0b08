// Probe (gfx950): does vmcnt retire IN ORDER across the two kinds of vector load -- `buffer_load ... lds` (global -> LDS) and a plain register
// load?  conv_wino4s.hip's hand-counted waits (and its mid-chunk vmcnt(5)) assume it does.
//   test A: a COLD direct-to-LDS load (a cache line nobody touched), then K HOT register loads, s_waitcnt vmcnt(K): is the LDS data there?
//   test B: a COLD register load, then K HOT direct-to-LDS loads, s_waitcnt vmcnt(K): has the register arrived?
// "stale" counts lanes that saw the sentinel.  0 / 0 = in order (as far as this load pattern can provoke it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ void probe(const float* cold, const float* hot, unsigned long long cold_floats, int iters, unsigned* stale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hot), 0, -1, 0x00020000);
    unsigned badA = 0, badB = 0;
    for (int it = 0; it < iters; ++it) {
        // a fresh 1 KiB per (workgroup, wave, iteration, test), spread over the whole cold buffer
        const unsigned long long slot = ((unsigned long long)blockIdx.x * 4 + wave) * (unsigned long long)iters * 2 + it * 2;
        const float* cA = cold + (slot * 2654435761ull % (cold_floats / 256)) * 256;
        const float* cB = cold + ((slot + 1) * 2654435761ull % (cold_floats / 256)) * 256;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cA), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cB), 0, -1, 0x00020000);
        float* mine = lds + wave * 2048;  // [0, 256): test A's landing zone, [256, 512) + : test B's hot landings
        *reinterpret_cast<f32x4*>(mine + 4 * lane) = f32x4{-1.f, -1.f, -1.f, -1.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- A ----
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"((unsigned)lane * 16u), "s"(rA), "s"((unsigned)(wave * 8192)) : "memory");
        f32x4 h[K];
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(h[k]) : "v"((unsigned)lane * 16u), "s"(rh));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
        const f32x4 got = *reinterpret_cast<volatile f32x4*>(mine + 4 * lane);
        badA += got[0] == -1.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("" ::"v"(h[k]));
        // ---- B ----
        f32x4 c = {-1.f, -1.f, -1.f, -1.f};
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(c) : "v"((unsigned)lane * 16u), "s"(rB));
#pragma unroll
        for (int k = 0; k < K; ++k)
            asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"((unsigned)lane * 16u), "s"(rh), "s"((unsigned)(wave * 8192 + 1024 + 1024 * (k & 3))) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
        float c0;
        asm volatile("v_mov_b32 %0, %1" : "=v"(c0) : "v"(c[0]));  // read the register NOW
        badB += c0 == -1.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (badA) atomicAdd(stale + 0, badA);
    if (badB) atomicAdd(stale + 1, badB);
}
int main() {
    const unsigned long long cold_floats = 1ull << 30;  // 4 GiB
    float *cold, *hot;
    unsigned* stale;
    hipMalloc(&cold, cold_floats * 4); hipMalloc(&hot, 4096); hipMalloc(&stale, 8);
    hipMemset(cold, 0x3f, cold_floats * 4); hipMemset(hot, 0, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(stale, 0, 8);
        if (rep == 0) hipLaunchKernelGGL(probe<4>, dim3(1024), dim3(256), 32768, 0, cold, hot, cold_floats, 64, stale);
        else hipLaunchKernelGGL(probe<12>, dim3(1024), dim3(256), 32768, 0, cold, hot, cold_floats, 64, stale);
        unsigned h[2];
        hipMemcpy(h, stale, 8, hipMemcpyDeviceToHost);
        printf("K=%d younger loads: test A (cold LDS load, hot register loads behind it) stale lanes %u ; test B (cold register load, hot LDS loads behind it) stale lanes %u ; of %d lane-trials each  [%s]\n",
               rep ? 12 : 4, h[0], h[1], 1024 * 256 * 64, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
