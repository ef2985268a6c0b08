// conv_wino3.hip -- OPT-IN variant of conv_wino.hip (cerb_net_set_conv_algo(net, 2)): the same Winograd F(2x2,3x3) convolution with
// every fp32 product emulated on the bf16 matrix pipe.  NOT the default and NOT what bench.py's headline measures: BASELINE.json
// configs[1] says fp32, and this kernel leaves the fp32 MFMA instruction; it exists to measure what the split buys (DESIGN par.9).
//
//   x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)  (24 mantissa bits: the split is exact)
//   a*b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm      (the three dropped terms are below 2^-24 |a b|)
// six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per K = 16 step instead of eight v_mfma_f32_32x32x2_f32: 2.7x fewer matrix-pipe
// cycles, and the bf16 matrix pipe does not share lanes with the fp32 VALU (scripts/ubench/bf16x3_vs_f32_mfma.hip: 362 vs 147
// TFLOP/s fp32-equivalent, the split's VALU work hidden).
//
// Structure = conv_wino.hip (item = 8 x 16 output pixels x 64 couts, wave a = row a of the transformed patch, persistent
// workgroups, output transform through LDS) with these differences:
//   * V lives in LDS as three bf16 planes [plane][xi][tile][32 ch], 80-byte rows (16-lane ds_read_b128 groups hit distinct banks):
//     120 KiB -> ONE workgroup per CU (launch bounds 256 x 1: the accumulators may spill into AccVGPRs);
//   * the fp32 transformed patch is split at the LDS write (3 x 8 bytes per float4);
//   * weights are split on the host (pack_wino3): per (xi, K-step) a wave streams 3 planes x 2 cout halves x 1 KiB;
//   * a chunk (32 channels) is 8 steps (4 positions x 2 K-steps) of 12 MFMAs.
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int NT = 32, OTH = 8, OTW = 16, CB = 32;  // 4 x 8 Winograd tiles -> 8 x 16 output pixels per item
constexpr int ROWB = 80;                       // bytes of one (xi, tile) row of one plane: 32 bf16 + 16 pad
constexpr int PLANE = 16 * NT * ROWB;          // 40960
constexpr int LDS_BYTES = 3 * PLANE;           // 122880
constexpr int QW = 4 * NT * ROWB;              // a wave's quarter of one plane: 10240 bytes
constexpr int NQ = 8;                          // steps per chunk: 4 positions x 2 K-steps of 16 channels
constexpr int WD = 2, NPRE = 2;
constexpr int STEP_W_BYTES = 3 * 2 * 1024;     // one step's weights of one wave: 3 planes x 2 cout halves x 1 KiB
constexpr int WAVE_W_BYTES = NQ * STEP_W_BYTES;
constexpr int CHUNK_W_BYTES = 4 * WAVE_W_BYTES;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ bf16x8 buf_load_bf(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 0);
    asm volatile("s_nop 1");  // gfx950 store-data hazard hipcc does not pad for SGPR soffsets, see conv_wino.hip
    __builtin_amdgcn_sched_barrier(0);
}
// fp32 -> (hi, mid, lo) bf16, four lanes at a time, packed as 3 x 8 bytes
__device__ __forceinline__ void split3(f32x4 x, u32x2& hi, u32x2& mid, u32x2& lo) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)x[e];
        const float r1 = x[e] - (float)h[e];
        m[e] = (__bf16)r1;
        l[e] = (__bf16)(r1 - (float)m[e]);
    }
    hi = __builtin_bit_cast(u32x2, h);
    mid = __builtin_bit_cast(u32x2, m);
    lo = __builtin_bit_cast(u32x2, l);
}

struct Item {
    int g, cb, n, oy0, ox0, tx, ty;
};
}  // namespace

template <bool HAS_RES>
__global__ __launch_bounds__(256, 1) void conv_wino3_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    float* lds = reinterpret_cast<float*>(ldsb);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int ncb = p.Cout >> 6;
    const int per_group = p.N * p.tiles_y * p.tiles_x * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int base_cnt = total / (int)gridDim.x, rem_cnt = total % (int)gridDim.x;
    int item = lb * base_cnt + min(lb, rem_cnt);
    const int item_end = item + base_cnt + (lb < rem_cnt ? 1 : 0);
    if (item >= item_end) return;

    auto decode = [&](int it) {
        Item w;
        w.g = it / per_group;
        int L = it - w.g * per_group;
        w.cb = L % ncb;
        int t_ = L / ncb;
        w.tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        w.ty = t_ % p.tiles_y;
        w.n = t_ / p.tiles_y;
        w.oy0 = w.ty * OTH;
        w.ox0 = w.tx * OTW;
        return w;
    };
    auto advance = [&](Item w) {
        if (++w.cb == ncb) {
            w.cb = 0;
            if (++w.tx == p.tiles_x) {
                w.tx = 0;
                if (++w.ty == p.tiles_y) {
                    w.ty = 0;
                    if (++w.n == p.N) {
                        w.n = 0;
                        ++w.g;
                    }
                }
            }
        }
        w.oy0 = w.ty * OTH;
        w.ox0 = w.tx * OTW;
        return w;
    };
    auto in_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.in + w.g * p.in_gs) + ((((long long)w.n * p.H + (w.oy0 - 1)) * p.W + (w.ox0 - 1)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {  // wpack is a byte stream of bf16 planes here (pack_wino3)
        return reinterpret_cast<const char*>(p.wpack) + (long long)w.g * p.w_gs + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };

    // ---- lane invariants ------------------------------------------------------------------------------------------------------
    const int t = tid >> 3, c = tid & 7, tty = t >> 3, ttx = t & 7;
    const unsigned ioff = (unsigned)((((2 * tty) * p.W + 2 * ttx) * p.Cin + 4 * c) * 4);
    const int vw = t * ROWB + c * 8;                        // V write (bytes): + plane*PLANE + xi*NT*ROWB
    const int vr = (a * 4 * NT + j) * ROWB + 16 * h;        // V read (bytes):  + plane*PLANE + b*NT*ROWB + 32*s2
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;
    const int cq = tid & 15, pp = tid >> 4;
    const unsigned ooff = (unsigned)((pp * p.Cout + 4 * cq) * 4);
    // T exchange: wave a's 16 blocks of 1056 bytes live in ITS quarters of planes 0 (jj = 0) and 1 (jj = 1)
    constexpr int TBB = 1056, THB = 528;
    static_assert(8 * TBB <= QW, "eight T blocks must fit in a wave's quarter of a plane");
    const int tw = a * QW + h * THB + j * 16;                                   // + jj*PLANE + (s*4+rq)*TBB
    const int tr = (pp & 1) * PLANE + (cq >> 1) * TBB + (cq & 1) * THB + (pp >> 1) * 16;  // + aa*QW + k*128

    f32x4 d[4][4];
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int k) { d[k >> 2][k & 3] = buf_load(r, ioff, chunk_off + (k >> 2) * rowb + (k & 3) * pixb); };
    auto mask_border = [&](const Item& w) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gy = w.oy0 - 1 + 2 * tty + r, gx = w.ox0 - 1 + 2 * ttx + q;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                d[r][q] = ok ? d[r][q] : z;
            }
    };
    auto touches_border = [&](const Item& w) { return w.oy0 < 1 || w.ox0 < 1 || w.oy0 + OTH + 1 > p.H || w.ox0 + OTW + 1 > p.W; };
    auto bt4 = [&](f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3) {
        x0 = x0 - x2;
        x3 = x1 - x3;
        const f32x4 o1 = x1;
        x1 = x1 + x2;
        x2 = x2 - o1;
    };
    auto transform = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) bt4(d[r][0], d[r][1], d[r][2], d[r][3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) bt4(d[0][q], d[1][q], d[2][q], d[3][q]);
    };
    auto write_v = [&]() {  // split the transformed patch into the three bf16 planes
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            u32x2 hi, mid, lo;
            split3(d[xi >> 2][xi & 3], hi, mid, lo);
            *reinterpret_cast<u32x2*>(ldsb + 0 * PLANE + xi * NT * ROWB + vw) = hi;
            *reinterpret_cast<u32x2*>(ldsb + 1 * PLANE + xi * NT * ROWB + vw) = mid;
            *reinterpret_cast<u32x2*>(ldsb + 2 * PLANE + xi * NT * ROWB + vw) = lo;
        }
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w));
#pragma unroll
        for (int k = 0; k < 16; ++k) issue(r0, 0, k);
    }
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    // weight window: step q in slot q & 3; per slot [plane 3][cout half 2]
    bf16x8 wq[4][3][2];
    auto load_w = [&](__amdgpu_buffer_rsrc_t r, int off, int slot) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int s = 0; s < 2; ++s) wq[slot][pl][s] = buf_load_bf(r, wlane, off + (pl * 2 + s) * 1024);
    };
#pragma unroll
    for (int dd = 0; dd < WD; ++dd) load_w(rw, dd * STEP_W_BYTES, dd);
    bf16x8 wpre[NPRE][3][2];
    auto load_pre = [&](__amdgpu_buffer_rsrc_t r) {
#pragma unroll
        for (int dd = 0; dd < NPRE; ++dd)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int s = 0; s < 2; ++s) wpre[dd][pl][s] = buf_load_bf(r, wlane, (WD + dd) * STEP_W_BYTES + (pl * 2 + s) * 1024);
    };
    load_pre(rw);
    f32x4 bnext[2][4];
    auto load_bias = [&](const Item& wi) {
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, a == 1 ? 256 : 0, 0x00020000);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) bnext[s][rq] = buf_load(rb, (unsigned)h * 16u, (32 * s + 8 * rq) * 4);
    };
    load_bias(w);

    for (;;) {
        f32x16 acc[4][2];
        const bool more_items = item + 1 < item_end;
        const Item wnx = more_items ? advance(w) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const bool mask_cur = touches_border(w);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[1][s][rq * 4 + e] = bnext[s][rq][e];

        auto chunk = [&](auto first_tag, int ch) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if (mask_cur) mask_border(w);
            transform();
            __syncthreads();  // every wave finished reading the previous chunk's V (or the previous item's T exchange)
            write_v();
            __syncthreads();

            const bool last_ch = (ch == nchunk - 1);
            const Item wp_ = last_ch ? wnx : w;
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(in_base(wp_));
            const int stage_off = (last_ch ? 0 : ch + 1) * (CB * 4);
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;

            bf16x8 bb[2][3];  // B operand planes of step q in bb[q & 1]
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bb[0][pl] = *reinterpret_cast<const bf16x8*>(ldsb + pl * PLANE + vr);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int q = b * 2 + s2;
                    if (FIRST && q < NPRE) {
                        // requested before the previous item's stores (wpre)
                    } else if (q + WD < NQ) {
                        load_w(rw, wcur_off + (q + WD) * STEP_W_BYTES, (q + WD) & 3);
                    } else {
                        load_w(rw_over, wover_off + (q + WD - NQ) * STEP_W_BYTES, (q + WD) & 3);
                    }
                    if (q + 1 < NQ) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bb[(q + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(ldsb + pl * PLANE + vr + ((q + 1) >> 1) * NT * ROWB + ((q + 1) & 1) * 32);
                    }
                    if (q < 4) {  // next chunk's patch: four loads per step
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) issue(r_stage, stage_off, 4 * q + k4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bool pre = FIRST && q >= WD && q < WD + NPRE;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 ah = pre ? wpre[pre ? q - WD : 0][0][s] : wq[q & 3][0][s];
                        const bf16x8 am = pre ? wpre[pre ? q - WD : 0][1][s] : wq[q & 3][1][s];
                        const bf16x8 al = pre ? wpre[pre ? q - WD : 0][2][s] : wq[q & 3][2][s];
                        const bf16x8 bh = bb[q & 1][0], bm = bb[q & 1][1], bl = bb[q & 1][2];
                        f32x16 c0;
                        if (FIRST && s2 == 0 && b != 1) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            c0 = z;
                        } else {
                            c0 = acc[b][s];
                        }
                        // smallest terms first
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c0, 0, 0, 0);
                        acc[b][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
                    }
                }
            }
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform (as conv_wino.hip) ----------------------------------------------------------------------------------
        {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x16 T0 = acc[0][s] + acc[1][s] + acc[2][s];
                const f32x16 T1 = acc[1][s] - acc[2][s] - acc[3][s];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 v0 = {T0[rq * 4 + 0], T0[rq * 4 + 1], T0[rq * 4 + 2], T0[rq * 4 + 3]};
                    const f32x4 v1 = {T1[rq * 4 + 0], T1[rq * 4 + 1], T1[rq * 4 + 2], T1[rq * 4 + 3]};
                    *reinterpret_cast<f32x4*>(ldsb + 0 * PLANE + (s * 4 + rq) * TBB + tw) = v0;
                    *reinterpret_cast<f32x4*>(ldsb + 1 * PLANE + (s * 4 + rq) * TBB + tw) = v1;
                }
            }
            load_pre(rw_nx);
            load_bias(wnx);
            const long long origin = (((long long)w.n * p.Ho + w.oy0) * p.Wo + w.ox0) * p.Cout + w.cb * 64;
            const unsigned span = (unsigned)(OTH * p.Wo * p.Cout * 4);
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + w.g * p.out_gs + origin, 0, span, 0x00020000);
            const bool partial = (w.oy0 + OTH > p.Ho) || (w.ox0 + OTW > p.Wo);
            const bool col_ok = !partial || (w.ox0 + pp < p.Wo);
            const int rows_ok = partial ? p.Ho - w.oy0 : OTH;
            const unsigned ocol = col_ok ? ooff : 0x80000000u;
            const int orow = p.Wo * p.Cout * 4;
            const float floor_ = p.relu ? 0.f : -3.402823466e38f;
            f32x4 res[8];
            if (HAS_RES) {
                const __amdgpu_buffer_rsrc_t r_res =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + w.g * p.resid_gs + origin), 0, span, 0x00020000);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) res[r8] = buf_load(r_res, (r8 < rows_ok) ? ocol : 0x80000000u, r8 * orow);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 tq[4];
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) tq[aa] = *reinterpret_cast<const f32x4*>(ldsb + tr + aa * QW + k * 128);
                f32x4 y[2];
                y[0] = tq[0] + tq[1] + tq[2];
                y[1] = tq[1] - tq[2] - tq[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned vo = (2 * k + i < rows_ok) ? ocol : 0x80000000u;
                    f32x4 o = HAS_RES ? y[i] + res[2 * k + i] : y[i];
                    o[0] = fmaxf(o[0], floor_);
                    o[1] = fmaxf(o[1], floor_);
                    o[2] = fmaxf(o[2], floor_);
                    o[3] = fmaxf(o[3], floor_);
                    buf_store(o, r_out, vo, (2 * k + i) * orow);
                }
            }
        }
        if (!more_items) break;
        ++item;
        w = wnx;
        rw = rw_nx;
    }
    (void)lds;
}

template <bool HAS_RES>
static hipError_t launch_wino3(ConvParams p, hipStream_t st) {
    p.tiles_x = (p.Wo + OTW - 1) / OTW;
    p.tiles_y = (p.Ho + OTH - 1) / OTH;
    const long long items = (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    auto kern = conv_wino3_kernel<HAS_RES>;
    static bool attr_done[64] = {};
    if (cerb_attr_needed(attr_done)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    long long grid = 256;  // persistent: one workgroup per CU (120 KiB of LDS)
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
    return hipGetLastError();
}

// p.wpack = pack_wino3 byte stream, p.w_gs = BYTES per group
hipError_t cerb_launch_wino3(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino3<true>(p, st) : launch_wino3<false>(p, st);
}
