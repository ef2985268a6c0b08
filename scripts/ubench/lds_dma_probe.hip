// Probe (gfx950): semantics of `buffer_load_dwordx4 ... offen lds` (global -> LDS without a register) before conv_wino4s.hip relies on it:
//   LDS byte address written by lane i = M0 + 16 * i ?   is the global address voffset(lane) + soffset ?   does `offset:N` move both sides ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* out, int variant) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 256) lds[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, -1, 0x00020000);
    const unsigned voff = (unsigned)(((lane * 7 + 3) & 63) * 16 + wave * 1024);  // a permutation of the 64 16-byte pieces of this wave's 1 KiB
    const unsigned m0v = (unsigned)(wave * 1024 + (variant == 2 ? 4096 : 0));
    if (variant == 1)  // the scalar offset operand moves the GLOBAL address only
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(r), "s"(m0v), "s"(2048) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(r), "s"(m0v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) out[i] = lds[i];
}
int main() {
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 8192 * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; ++variant) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 16384 + 4096 * 4, 0, d, o, variant);
        std::vector<float> r(4096);
        hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
        // expectation A: lds float index (m0/4 + 4*lane + k) holds src[(voff + inst_off)/4 + k], where inst_off also shifts the LDS side by inst_off
        int okA = 0, okB = 0, written = 0;
        for (int i = 0; i < 4096; ++i) written += r[i] >= 0.f;
        for (int wave = 0; wave < 4; ++wave)
            for (int lane = 0; lane < 64; ++lane)
                for (int k = 0; k < 4; ++k) {
                    const int voff = ((lane * 7 + 3) & 63) * 16 + wave * 1024, inst = variant == 1 ? 2048 : 0, m0 = wave * 1024 + (variant == 2 ? 4096 : 0);
                    const float want = (float)((voff + inst) / 4 + k);
                    const int ia = (m0 + inst) / 4 + 4 * lane + k, ib = m0 / 4 + 4 * lane + k;  // (variant 1: inst = the SCALAR offset)
                    okA += ia < 4096 && r[ia] == want;
                    okB += ib < 4096 && r[ib] == want;
                }
        printf("variant %d: floats written %d, lane-linear with the extra offset on BOTH sides %d/1024, on the global side only %d/1024\n", variant, written, okA, okB);
    }
    return 0;
}
