// conv_wino4q.hip -- the tile-planar Winograd F(4x4,3x3) convolution of conv_wino4p.hip with TWO waves per SIMD.
//
// Why: a lone wave on a gfx950 SIMD issues one VALU-class instruction per ~10 cycles, whatever the instruction (v_pk_add_f32, v_max_f32,
// v_accvgpr_read_b32, even an 8-cycle v_mfma_f32_4x4x1: scripts/ubench/accvgpr_read_rate.hip), and the fp32 matrix instruction shares the
// SIMD's lanes with them -- conv_wino4p.hip's ~1600 VALU instructions per item (input transform 576, output transform + accumulator reads +
// ReLU ~950) therefore cost ~16 k of its 62 k cycles, with nothing to overlap them at one wave per SIMD.  With a second wave on the SIMD the
// same streams run 2-4x cheaper (the ubench: matrix instruction + VALU pairs 50.9 -> 28.6 .. 42 ticks per pair).  The register file allows two
// waves only at 256 registers each, i.e. 144 accumulators per wave instead of 288:
//   * item, V tile, LDS budget, planar layout, weights and B-operand traffic are conv_wino4p.hip's: two 16x16-pixel blocks x 64 output
//     channels, V [36][32 tiles][16 ch] double-buffered (144 KiB), one workgroup per CU -- of EIGHT waves;
//   * wave (a, h): 16 output channels a of BOTH blocks (every weight register still feeds two matrix instructions) for HALF of the 36 Winograd
//     positions -- rows 3h .. 3h+2 of the 6x6: 18 positions x 2 blocks x 4 registers = 144 accumulators, all in AccVGPRs;
//   * the output transform needs all six rows of a tile: each wave transforms its three rows horizontally, forms the two partial vertical
//     combinations its partner needs (s = row1 + row2, d = row1 - row2 of its half), the partners swap them through the free V buffer (8 KiB per
//     wave), and wave h finishes output rows 2h, 2h+1 of every tile: Y0 = (m0 + s1) + s2, Y1 = d1 + 2 d2 | Y2 = s1 + 4 s2, Y3 = (d1 + 8 d2) + m5;
//   * input path: thread = (tile, ONE channel) -- 512 threads cover a chunk's 32 tiles x 16 channels; scalar transform arithmetic (the same
//     fma chain per element as the packed form of conv_wino4p.hip: identical V values).
// Results equal conv_wino4p.hip's up to the summation order of the output transform (horizontal pass first here).
// Reference layers: models/utils/conv_layers.py:24-60 (_ConvLayer) inside models/net_desc.py:182-198 (the decoder loop).
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int NPOS = 36;
constexpr int NT = 32;                        // tiles per item: two blocks of 4x4 tiles
constexpr int BLK = 16;                       // a block is 16x16 output pixels
constexpr int CB = 16;                        // input channels per LDS pass = one plane
constexpr int V_FLOATS = NPOS * NT * CB;      // one V buffer: 72 KiB
constexpr int LDS_BYTES = 2 * V_FLOATS * 4;   // double-buffered: 144 KiB
constexpr int PLANE_BYTES = 16 * 16 * 16 * 4; // one 16-channel plane of a block
constexpr int NS = 18;                        // steps per chunk and wave: one position each (one 16-byte weight load, two ds_read_b128, 8 MFMAs)
#ifndef Q4_RING
#define Q4_RING 6
#endif
constexpr int RING = Q4_RING;                 // weight operand slots (NS % RING == 0)
#ifndef Q4_WD
#define Q4_WD 4
#endif
constexpr int WD = Q4_WD;                     // weight prefetch distance in steps
constexpr int PRE = WD;                       // steps of the NEXT item whose weights are requested before an item's output stores
#ifndef Q4_TQ
#define Q4_TQ 5
#endif
constexpr int TQ = Q4_TQ;                     // the next chunk's patch is transformed at steps TQ+1 .. TQ+12, rows written at TQ+7 .. TQ+12
#ifndef Q4_HQ
#define Q4_HQ 0
#endif
constexpr int HQ = Q4_HQ;                     // halo loads of the next chunk's patch: four per step at steps HQ .. HQ+4
constexpr int halo_r(int k) { return k < 6 ? k : k < 14 ? ((k - 6) & 1 ? 5 : 0) : k - 14; }  // column 0 (6), top / bottom of columns 1..4 (8), column 5 (6)
constexpr int halo_q(int k) { return k < 6 ? 0 : k < 14 ? 1 + (k - 6) / 2 : 5; }
static_assert(NS % RING == 0 && WD + 1 <= RING, "weight ring");
static_assert(HQ + 5 <= TQ + 1 && TQ + 12 < NS && TQ + 8 + 4 < NS + 1, "patch schedule");
constexpr int BIAS_XI = 7;                    // A^T[i][1] A[1][j] = 1 for all 16 outputs: the bias enters through position (1, 1) -- a position of half 0
constexpr int CHUNK_W_BYTES = NPOS * 4 * 1024;  // packed weights of one (cout block, 16-channel chunk): 144 KiB (conv_wino4.hip's layout)
constexpr int WAVE_W_BYTES = NPOS * 1024;       // one cout group's share: 36 positions x 1 KiB

template <int I>
using IC = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 0);
    asm volatile("s_nop 1");  // gfx950 store hazard, see conv_wino.hip buf_store / tests/test_isa_hazard.py
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}

// One step = the 8 matrix instructions of one position (4 k-slots x 2 blocks) in one statement.  hipcc splits a 256-register wave into 128
// architectural + 128 accumulation registers: 16 of the 18 positions keep their accumulators in AccVGPRs, two in VGPRs.
constexpr int NPOS_A = 16;
template <bool AGPR, bool ZERO>
__device__ __forceinline__ void mfma_step(f32x4& c0, f32x4& c1, const f32x4& av, const f32x4& b0, const f32x4& b1) {
#define Q4_STEP_BODY(FIRST0, FIRST1)                               \
    "v_mfma_f32_16x16x4_f32 %0, %2, %6, " FIRST0 "\n\t"             \
    "v_mfma_f32_16x16x4_f32 %1, %2, %10, " FIRST1 "\n\t"            \
    "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\t"                     \
    "v_mfma_f32_16x16x4_f32 %1, %3, %11, %1\n\t"                    \
    "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\t"                     \
    "v_mfma_f32_16x16x4_f32 %1, %4, %12, %1\n\t"                    \
    "v_mfma_f32_16x16x4_f32 %0, %5, %9, %0\n\t"                     \
    "v_mfma_f32_16x16x4_f32 %1, %5, %13, %1"
#define Q4_STEP_IN "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3])
    if constexpr (ZERO) {
        if constexpr (AGPR) asm volatile(Q4_STEP_BODY("0", "0") : "=&a"(c0), "=&a"(c1) : Q4_STEP_IN);
        else asm volatile(Q4_STEP_BODY("0", "0") : "=&v"(c0), "=&v"(c1) : Q4_STEP_IN);
    } else {
        if constexpr (AGPR) asm volatile(Q4_STEP_BODY("%0", "%1") : "+a"(c0), "+a"(c1) : Q4_STEP_IN);
        else asm volatile(Q4_STEP_BODY("%0", "%1") : "+v"(c0), "+v"(c1) : Q4_STEP_IN);
    }
#undef Q4_STEP_BODY
#undef Q4_STEP_IN
}
__device__ __forceinline__ void wait_mfma_results() { asm volatile("s_nop 15\n\ts_nop 3"); }  // 8-pass MFMA D -> VALU reader: 12 states and more

__device__ __forceinline__ int fresh_lane() {  // the hardware lane id, computed where it is used (volatile: not hoisted out of the item loop)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

struct Blk {
    int n, by, bx;  // image, block row / column inside the launch's block grid
};
struct Item {
    int g, cb;
    Blk b0, b1;
    int nvalid;  // 2, or 1 when the launch has an odd number of blocks and this is the last pair (block 1 repeats block 0, stores skipped)
};
}  // namespace

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4q_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = wv & 3;   // this wave's 16 output channels of the item's 64 (= one output plane)
    const int h = wv >> 2;  // its half of the Winograd positions: rows 3h .. 3h+2 of the 6x6 (positions 18h .. 18h+17); output rows 2h, 2h+1 of every tile
    const int m = lane & 15;  // MFMA row (cout) / column (tile within a block)
    const int ks = lane >> 4; // k-slot

    const int ncb = p.Cout >> 6;
    const int nblk = p.N * p.tiles_y * p.tiles_x;  // blocks per group
    const int npair = (nblk + 1) >> 1;
    const int per_group = npair * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;   // = input planes per block
    const int nplane_o = p.Cout / CB;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ISTEP = (int)gridDim.x;  // workgroup lb takes items lb, lb + G, ...: neighbouring blocks run at the same time on one XCD
    int item = lb;
    const int item_end = total;
    if (item >= item_end) return;

    auto decode_blk = [&](int id) {
        Blk b;
        b.bx = id % p.tiles_x;
        const int r = id / p.tiles_x;
        b.by = r % p.tiles_y;
        b.n = r / p.tiles_y;
        return b;
    };
    auto decode = [&](int it) {
        Item w;
        w.g = it / per_group;
        const int L = it - w.g * per_group;
        w.cb = L % ncb;
        const int pr = L / ncb;
        w.nvalid = (2 * pr + 1 < nblk) ? 2 : 1;
        w.b0 = decode_blk(2 * pr);
        w.b1 = w.nvalid == 2 ? decode_blk(2 * pr + 1) : w.b0;
        return w;
    };
    // the block's TOP-LEFT neighbour in the guard-ringed block grid (stored index of block (by, bx) is (by + 1, bx + 1)): every patch offset is >= 0
    auto in_base = [&](int g, const Blk& b) {
        return reinterpret_cast<const char*>(p.in + g * p.in_gs) +
               (((long long)b.n * p.pl_byp + (b.by + p.ty_off)) * p.pl_bxp + (b.bx + p.tx_off)) * (long long)nchunk * PLANE_BYTES;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES + h * (NS * 1024);
    };

    // ---- lane invariants ---------------------------------------------------------------------------------------------------
    // input transform: thread = (tile t of the item, channel c); tiles 0..15 are block 0 (waves 0..3), 16..31 block 1 (waves 4..7)
    const int t = tid >> 4, c = tid & 15, tm = t & 15, tty = tm >> 2, ttx = tm & 3;
    const bool second = (wv >> 2) != 0;
    auto mine = [&](const Item& wi) {
        Blk b;
        b.n = second ? wi.b1.n : wi.b0.n;
        b.by = second ? wi.b1.by : wi.b0.by;
        b.bx = second ? wi.b1.bx : wi.b0.bx;
        return b;
    };
    // patch element (r, q) = pixel (4 tty + r - 1, 4 ttx + q - 1) of the block: row class 0 (r = 0: pixel row 3 of the tile above, in the block
    // above when tty = 0), 1 (r = 1..4: the tile's own rows), 2 (r = 5: row 0 of the tile below); columns alike.  Nine lane offsets, for good.
    unsigned poff[3][3];
    {
        const unsigned rowblk = (unsigned)p.pl_bxp * (unsigned)nchunk * PLANE_BYTES, colblk = (unsigned)nchunk * PLANE_BYTES;
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int qc = 0; qc < 3; ++qc) {
                const int dby = (rc == 0 && tty == 0) ? 0 : (rc == 2 && tty == 3) ? 2 : 1;
                const int dbx = (qc == 0 && ttx == 0) ? 0 : (qc == 2 && ttx == 3) ? 2 : 1;
                const int ty2 = rc == 0 ? ((tty + 3) & 3) : rc == 2 ? ((tty + 1) & 3) : tty;
                const int tx2 = qc == 0 ? ((ttx + 3) & 3) : qc == 2 ? ((ttx + 1) & 3) : ttx;
                poff[rc][qc] = (unsigned)dby * rowblk + (unsigned)dbx * colblk + (unsigned)((4 * ty2 + tx2) * 64 + c * 4);
                asm volatile("" : "+v"(poff[rc][qc]));
            }
    }
    const int vw = t * CB + (((c >> 2) ^ ((tm & 8) ? 3 : 0)) << 2) + (c & 3);   // V write position (floats); position xi adds xi*NT*CB
    const int vr = m * CB + ((ks ^ ((m & 8) ? 3 : 0)) << 2);                     // V read position, block 0; block 1 adds 16*CB
    const unsigned wlane = (unsigned)lane * 16u;
    const int pos0 = h * NS;  // this wave's first position

    float d[6][6];  // raw patch of a coming chunk (one channel), transformed in place
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int rr, int qq) __attribute__((always_inline)) {
        const int ii = rr == 0 ? 3 : rr == 5 ? 0 : rr - 1, jj = qq == 0 ? 3 : qq == 5 ? 0 : qq - 1;
        d[rr][qq] = buf_load1(r, poff[rr == 0 ? 0 : rr == 5 ? 2 : 1][qq == 0 ? 0 : qq == 5 ? 2 : 1], chunk_off + (ii * 4 + jj) * 1024);
    };
    // B^T x for the points (0, 1, -1, 2, -2, inf), in place: the fma chain of conv_wino4p.hip's packed form, element for element
    auto bt6 = [&](float& x0, float& x1, float& x2, float& x3, float& x4, float& x5) __attribute__((always_inline)) {
        const float t0 = __builtin_fmaf(-x2, 4.f, x4), t1 = __builtin_fmaf(-x1, 4.f, x3);
        const float u0 = x4 - x2, u1 = x3 - x1;
        float y0 = __builtin_fmaf(x0, 4.f, x4), y5 = __builtin_fmaf(x1, 4.f, x5);
        y0 = __builtin_fmaf(-x2, 5.f, y0);
        y5 = __builtin_fmaf(-x3, 5.f, y5);
        x0 = y0;
        x5 = y5;
        x1 = t0 + t1;
        x2 = t0 - t1;
        x3 = __builtin_fmaf(u1, 2.f, u0);
        x4 = __builtin_fmaf(-u1, 2.f, u0);
    };
    auto pass_v = [&](int q) __attribute__((always_inline)) { bt6(d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]); };  // down column q
    auto pass_h = [&](int r) __attribute__((always_inline)) { bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]); };  // along row r
    auto write_row = [&](int buf, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 6; ++b) lds[buf * V_FLOATS + (r * 6 + b) * NT * CB + vw] = d[r][b];
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w.g, mine(w)));
#pragma unroll
        for (int k = 0; k < 36; ++k) issue(r0, 0, k / 6, k % 6);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) pass_v(q);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        pass_h(r);
        write_row(0, r);
    }
    {   // the interior of the SECOND chunk's patch (in the steady state the previous chunk requests it)
        const bool one = (nchunk == 1);
        const Item w1 = (one && item + ISTEP < item_end) ? decode(item + ISTEP) : w;
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(in_base(w1.g, mine(w1)));
#pragma unroll
        for (int k = 0; k < 16; ++k) issue(r1, one ? 0 : PLANE_BYTES, 1 + k / 4, 1 + k % 4);
    }
    int vbuf = 0;  // the buffer the CURRENT chunk reads; the next chunk's patch goes to vbuf ^ 1
    __syncthreads();
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    f32x4 wq[RING];
#pragma unroll
    for (int dd = 0; dd < PRE; ++dd) wq[dd] = buf_load(rw, wlane, dd * 1024);
    f32x4 bnext;
    auto load_bias = [&](const Item& wi) {  // half 1 reads through a zero-length descriptor: zeros (position 7 belongs to half 0)
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64 + 16 * a;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, h ? 0 : 64, 0x00020000);
        bnext = buf_load(rb, (unsigned)(fresh_lane() >> 4) * 16u, 0);
    };
    load_bias(w);

    for (;;) {
        f32x4 acc[NS][2];
        const bool more_items = item + ISTEP < item_end;
        const Item wnx = more_items ? decode(item + ISTEP) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const char* in_cur = in_base(w.g, mine(w));
        const char* in_nx = in_base(wnx.g, mine(wnx));
        const bool more2 = item + 2 * ISTEP < item_end;
        const Item wn2 = (nchunk == 1 && more2) ? decode(item + 2 * ISTEP) : wnx;
        const char* in_n2 = in_base(wn2.g, mine(wn2));
        auto in_nx2 = [&](int over) { return (nchunk == 1 && over >= 1) ? in_n2 : in_nx; };
        acc[BIAS_XI][0] = bnext;
        acc[BIAS_XI][1] = bnext;

        auto chunk = [&](auto first_tag, int ch) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool last_ch = (ch == nchunk - 1);
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(last_ch ? in_nx : in_cur);   // the next chunk's patch (halo loads)
            const int stage_off = (last_ch ? 0 : ch + 1) * PLANE_BYTES;
            const bool wrap2 = ch + 2 >= nchunk;                                           // the patch of the chunk after next (interior loads)
            const __amdgpu_buffer_rsrc_t r_stage2 = make_rsrc(wrap2 ? in_nx2(ch + 2 - nchunk) : in_cur);
            const int stage_off2 = (wrap2 ? (ch + 2 - nchunk) % nchunk : ch + 2) * PLANE_BYTES;
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;
            const float* vsrc = lds + vbuf * V_FLOATS + pos0 * NT * CB + vr;
            const int wbuf = vbuf ^ 1;

            f32x4 bb[2][2];  // B operands (blocks 0, 1) of step q in bb[q & 1]
            bb[0][0] = *reinterpret_cast<const f32x4*>(vsrc);
            bb[0][1] = *reinterpret_cast<const f32x4*>(vsrc + 16 * CB);
            static_for<0, NS>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                {
                    constexpr int dd = q + WD;
                    if constexpr (!(FIRST && dd < PRE)) {  // (those were requested before the previous item's stores, or in the prologue)
                        if constexpr (dd < NS) wq[dd % RING] = buf_load(rw, wlane, wcur_off + dd * 1024);
                        else wq[(dd - NS) % RING] = buf_load(rw_over, wlane, wover_off + (dd - NS) * 1024);
                    }
                }
                if constexpr (q + 1 < NS) {
                    bb[(q + 1) & 1][0] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * CB);
                    bb[(q + 1) & 1][1] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * CB + 16 * CB);
                }
#ifndef Q4_ABL_NOIN
                if constexpr (q >= HQ && q < HQ + 5) {  // halo of the next chunk's patch, four pixels per step
#pragma unroll
                    for (int u = 0; u < 4; ++u) issue(r_stage, stage_off, halo_r(4 * (q - HQ) + u), halo_q(4 * (q - HQ) + u));
                }
                // the next chunk's patch landed: B^T d B (one 1-D pass per step) and the V writes into the OTHER buffer
                if constexpr (q > TQ && q <= TQ + 6) pass_v(q - TQ - 1);
                if constexpr (q > TQ + 6 && q <= TQ + 12) {
                    pass_h(q - TQ - 7);
                    write_row(wbuf, q - TQ - 7);
                }
                if constexpr (q >= TQ + 9 && q <= TQ + 12) {  // row q - TQ - 8 was written a step ago: its registers take the interior of the chunk after next
#pragma unroll
                    for (int u = 1; u <= 4; ++u) issue(r_stage2, stage_off2, q - TQ - 8, u);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 av = wq[q % RING];
                const f32x4 b0 = bb[q & 1][0], b1 = bb[q & 1][1];
                mfma_step<(q < NPOS_A), FIRST && q != BIAS_XI>(acc[q][0], acc[q][1], av, b0, b1);
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();  // everybody has read this chunk's V and written the next one's
            vbuf ^= 1;
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform A^T M A, shared between the two waves of a cout group ---------------------------------------------------------
        // vmcnt retires in order across loads AND stores: the next item's first WD weight steps went out during the last chunk, its bias goes now,
        // before this item's stores
        load_bias(wnx);
        wait_mfma_results();
#ifdef Q4_ABL_NOOUT
        if (acc[0][0][0] == 1.2345e-30f)
#endif
        {
            // the ReLU floor as a SCALAR made here, per item: hipcc had put this kernel invariant in a vector register at kernel start, spilled it,
            // and reloaded it in the middle of the output stage -- behind an s_waitcnt vmcnt(0) that drained every load and every store in flight
            int relu_s = p.relu;
            asm volatile("" : "+s"(relu_s));
            const float floor_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(relu_s ? 0 : (int)0xff7fffff));
            const int lane_o = fresh_lane();
            const int m_o = lane_o & 15;
            const unsigned olane = (unsigned)(m_o * 64 + (lane_o >> 4) * 16);  // lane (tile m, channel quad ks) inside a 1-KiB pixel-position row
            // exchange area: the V buffer the last chunk has finished with; [cout group a][sender half][8 values][64 lanes] x 16 bytes = 64 KiB
            float* xch = lds + (vbuf ^ 1) * V_FLOATS;
            float* x_mine = xch + ((a * 2 + h) * 8) * 256 + lane_o * 4;
            const float* x_theirs = xch + ((a * 2 + (h ^ 1)) * 8) * 256 + lane_o * 4;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const Blk bo = tb ? w.b1 : w.b0;
                const bool dead = (tb == 1 && w.nvalid == 1);  // the pair's second block repeats the first: nothing to store (the exchange still runs: barriers)
                const int by_abs = bo.by + p.ty_off, bx_abs = bo.bx + p.tx_off;
                const long long origin = p.out_gs * w.g + ((((long long)bo.n * p.pl_byp + by_abs + 1) * p.pl_bxp + bx_abs + 1) * nplane_o + w.cb * 4 + a) * (long long)(PLANE_BYTES / 4);
                const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + origin, 0, PLANE_BYTES, 0x00020000);
                const bool partial = (by_abs * BLK + BLK > p.Ho) || (bx_abs * BLK + BLK > p.Wo);
                // horizontal pass over this wave's three rows: R[rr][j] = sum_b M[3h + rr][b] A[b][j]
                f32x4 R[3][4];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const f32x4 m0 = acc[rr * 6 + 0][tb], m1 = acc[rr * 6 + 1][tb], m2 = acc[rr * 6 + 2][tb], m3 = acc[rr * 6 + 3][tb], m4 = acc[rr * 6 + 4][tb],
                                m5 = acc[rr * 6 + 5][tb];
                    const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    R[rr][0] = m0 + s1 + s2;
                    R[rr][1] = d1 + 2.f * d2;
                    R[rr][2] = s1 + 4.f * s2;
                    R[rr][3] = (d1 + 8.f * d2) + m5;
                }
                // what the partner needs of my half: half 0 rows (0, 1, 2) -> s1 = r1 + r2, d1 = r1 - r2; half 1 rows (3, 4, 5) -> s2 = r3 + r4, d2 = r3 - r4
                f32x4 S[4], D[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 u = h ? R[0][j] : R[1][j], v = h ? R[1][j] : R[2][j];
                    S[j] = u + v;
                    D[j] = u - v;
                    *reinterpret_cast<f32x4*>(x_mine + j * 256) = S[j];
                    *reinterpret_cast<f32x4*>(x_mine + (4 + j) * 256) = D[j];
                }
                __syncthreads();
                const int rem_y = p.Ho - by_abs * BLK - 4 * (m_o >> 2), rem_x = p.Wo - bx_abs * BLK - 4 * (m_o & 3);  // partial blocks: rows / columns of this lane's tile inside the image
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 so = *reinterpret_cast<const f32x4*>(x_theirs + j * 256), dox = *reinterpret_cast<const f32x4*>(x_theirs + (4 + j) * 256);
                    f32x4 ya, yb;  // output rows 2h, 2h + 1 of the tile
                    if (h == 0) {
                        ya = (R[0][j] + S[j]) + so;   // Y0 = (m0 + s1) + s2
                        yb = D[j] + 2.f * dox;        // Y1 = d1 + 2 d2
                    } else {
                        ya = so + 4.f * S[j];         // Y2 = s1 + 4 s2
                        yb = (dox + 8.f * D[j]) + R[2][j];  // Y3 = (d1 + 8 d2) + m5
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ya[e] = fmaxf(ya[e], floor_);
                        yb[e] = fmaxf(yb[e], floor_);
                    }
                    const int i0 = 2 * h;
                    // pixels of an edge block beyond the image are never written: they stay zero (the next convolution's padding)
                    const unsigned va = (!dead && (!partial || (i0 < rem_y && j < rem_x))) ? olane : 0x80000000u;
                    const unsigned vb = (!dead && (!partial || (i0 + 1 < rem_y && j < rem_x))) ? olane : 0x80000000u;
                    buf_store(ya, r_out, va, (i0 * 4 + j) * 1024);
                    buf_store(yb, r_out, vb, ((i0 + 1) * 4 + j) * 1024);
                }
                __syncthreads();  // the exchange area is rewritten by the next block, then by the next item's second chunk
            }
        }
        if (!more_items) break;
        item += ISTEP;
        w = wnx;
        rw = rw_nx;
    }
}

hipError_t cerb_launch_wino4q(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cin < 2 * CB || p.Cout % 64 || p.resid || p.pl_byp < 3 || p.pl_bxp < 3 || p.H != p.Ho || p.W != p.Wo) return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + BLK - 1) / BLK;  // blocks, not tiles
    p.tiles_y = (p.Ho + BLK - 1) / BLK;
    p.ty_off = p.tx_off = 0;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) {
        p.ty_off = p.roi_y0 / BLK;
        p.tx_off = p.roi_x0 / BLK;
        p.tiles_y = (p.roi_y1 + BLK - 1) / BLK - p.ty_off;
        p.tiles_x = (p.roi_x1 + BLK - 1) / BLK - p.tx_off;
    }
    if (p.pl_byp != (p.Ho + BLK - 1) / BLK + 2 || p.pl_bxp != (p.Wo + BLK - 1) / BLK + 2) return hipErrorInvalidValue;
    const long long nblk = (long long)p.N * p.tiles_x * p.tiles_y;
    const long long items = (long long)p.groups * ((nblk + 1) / 2) * (p.Cout / 64);
    auto kern = conv_wino4q_kernel;
    static bool attr_done[64] = {};
    if (cerb_attr_needed(attr_done)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    long long grid = 256;  // persistent: one workgroup per CU
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS_BYTES, st, p);
    return hipGetLastError();
}
