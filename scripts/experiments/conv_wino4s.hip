// conv_wino4s.hip -- conv_wino4p.hip's F(4x4,3x3) convolution on the tile-planar layout with the RAW INPUT PATCH STAGED THROUGH LDS (round 4;
// VERDICT r3 item 1c, bounded by the ablation in profiles/r04_w4p_ldspatch_ablation.txt before it was built).
//
// conv_wino4p.hip loads a thread's 6x6 patch (two channels) with 36 global loads per 16-channel chunk: 64 lanes x 8 bytes scattered over eight
// 64-byte segments each.  Its ablations priced that shape at ~190 wave cycles per instruction even when it hits -- 16 % of the launch -- while the
// workgroup's UNIQUE pixels are 1.78x fewer (a block of 4x4 tiles covers 18 x 18 pixels, not 16 x 36).  Here
//   * the unique pixels of both blocks of an item (2 x 324 pixels x 16 channels = 41.5 KB) go global -> LDS WITHOUT A REGISTER: eleven 16-byte
//     `buffer_load_dwordx4 ... lds` per thread and chunk (1 KiB contiguous in LDS per wave and instruction), requested during the FIRST half of
//     the chunk before the one that consumes them;
//   * a thread reads its patch from LDS (36 ds_read_b64, free next to the matrix pipe: ablation) during the SECOND half and transforms it there;
//   * to make room for the raw buffer next to V in the CU's 160 KB, V holds 8 channels at a time: the matrix instructions t = 0, 1 of a
//     16-channel chunk (channels {t, 4 + t, 8 + t, 12 + t}) read V_A in the first half, t = 2, 3 read V_B in the second; waves 0-1 transform the
//     channels of V_A, waves 2-3 those of V_B (every lane busy: lane = (tile, k-slot), two channels each).  V_A(c+1) is written while V_B(c) is
//     read, so V_A is single- and V_B double-buffered: 3 x 36.9 KB + 43 KB of raw patch (+ 5 KB of offsets) = 159 KB.  Two barriers per chunk instead of one.
// Each accumulator still receives t = 0, 1, 2, 3 of chunk 0, then of chunk 1, ...: the SAME products in the SAME order as conv_wino4p.hip and
// conv_wino4.hip -- results are bit-identical (tests/test_net_gpu.py::test_planar_last_level_is_bit_identical_to_nhwc runs all three).
// Weights are packed per PAIR of positions so that one 1-KiB load still feeds 8 matrix instructions (cerb_api.hip: pack_wino4 layout 2):
//   [cout block][16-channel chunk][wave a][half h][pair s 18][lane 64][4] = U[pos 2s + (e >> 1)][co 16 a + (l & 15)][ci 16 ch + 4 (l >> 4) + 2 h + (e & 1)]
// Reference layers: models/utils/conv_layers.py:24-60 (_ConvLayer) inside models/net_desc.py:182-198 (the decoder loop).
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int NPOS = 36;
constexpr int NPAIR = 18;                      // position pairs = steps per half chunk
constexpr int NT = 32;                         // tiles per item: two blocks of 4x4 tiles
constexpr int BLK = 16;
constexpr int CB = 16;                         // input channels per raw chunk = one plane
constexpr int VH_FLOATS = NPAIR * NT * 16;     // one V half: [pair][tile][k-slot 4][pos in pair 2][t in half 2] = 36,864 B
constexpr int RAW_PIECES = 21 * 64;            // 16-byte pieces of one block's raw patch: 324 pixels x 4 channel quads = 1296, padded to 21 wave instructions
constexpr int RAW_BLOCK_BYTES = RAW_PIECES * 16;  // 21,504
constexpr int LDS_VA = 0, LDS_VB = VH_FLOATS * 4, LDS_RAW = 3 * VH_FLOATS * 4;   // byte offsets: V_A, V_B[2], raw patch (2 blocks)
constexpr int LDS_TAB = LDS_RAW + 2 * RAW_BLOCK_BYTES;                         // [21 wave instructions of a block][64 lanes] global byte offsets of the raw pieces
constexpr int LDS_BYTES = LDS_TAB + 23 * 64 * 4;                               // 110,592 + 43,008 + 5,888 = 159,488 (rows 21, 22: copies, read by nobody's DMA)
constexpr int PLANE_BYTES = 16 * 16 * 16 * 4;
constexpr int NS = 36;                         // steps per 16-channel chunk: 18 pair steps of half A, 18 of half B (8 matrix instructions each)
constexpr int NPOS_A = 32;                     // positions whose accumulators live in AccVGPRs
#ifndef S4_RING
#define S4_RING 12
#endif
constexpr int RING = S4_RING;
#ifndef S4_WD
#define S4_WD 8
#endif
constexpr int WD = S4_WD;                      // weight prefetch distance in steps
constexpr int PRE = WD;
#ifndef S4_RQ
#define S4_RQ 18
#endif
constexpr int RQ = S4_RQ;                      // second half: patch reads at RQ .. RQ+5 (a row of 6 pixels per step), vertical passes RQ+6 .. +11, horizontal + V writes RQ+12 .. +17
static_assert(NS % RING == 0 && WD + 1 <= RING && RQ >= NPAIR && RQ + 18 <= NS && PRE == WD, "schedule");
constexpr int NDMA = 11;                       // direct-to-LDS loads per thread and chunk (waves 2, 3: ten)
constexpr int BIAS_XI = 7;
constexpr int CHUNK_W_BYTES = NPOS * 4 * 1024;
constexpr int WAVE_W_BYTES = NPOS * 1024;

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
typedef float f32x2 __attribute__((ext_vector_type(2)));

// EXPERIMENT, off by default (-DS4_MANUAL_WAITS): the weight and bias loads as inline assembly with every wait on them spelled out.
// Idea: with the builtin load the compiler counts ITS loads only -- for the operand of step q, requested WD steps earlier, it emits
// s_waitcnt vmcnt(WD) -- but the queue also holds the direct-to-LDS loads below, which it cannot see, and vmcnt retires in order: vmcnt(WD) then
// also demands the patch loads of ~WD/2 steps ago.  With both kinds of load invisible the count can be exact (wait below: WD + the patch loads
// issued since).  Measured (profiles/r04_w4s_ablations.txt): 2.108 ms per launch against 2.155 with the compiler's waits -- the coupling is NOT
// where the 0.28 ms of the patch loads go -- and the results were NOT bit-identical to conv_wino4p on the GPU although the instruction stream
// passes the in-flight-register scan of tests/test_isa_hazard.py: either a direct-to-LDS load can retire behind a younger register load (the
// counter is not in order across the two kinds) or the hand count has a hole.  Not shipped; kept for the record with the scan that guards it.
__device__ __forceinline__ f32x4 weight_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
#ifdef S4_MANUAL_WAITS
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(r), "s"(soff));
    return v;
#else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
#endif
}

// global -> LDS without a register: 64 lanes x 16 bytes land at LDS byte address lds_base + 16 * lane (scripts/ubench/lds_dma_probe.hip); the
// compiler does not see the instruction, so it neither tracks it in vmcnt nor knows it writes LDS -- every wait on it is spelled out below.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_base) : "memory");
}

// One pair step = the 8 matrix instructions of two positions (2 t of this half x 2 blocks each) in one statement (see conv_wino4p.hip mfma_step).
template <bool AGPR, bool Z0, bool Z1>
__device__ __forceinline__ void mfma_pair_step(f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11, const f32x4& av, const f32x4& b0, const f32x4& b1) {
#define S4_BODY(F00, F01, F10, F11)                                  \
    "v_mfma_f32_16x16x4_f32 %0, %4, %8, " F00 "\n\t"                  \
    "v_mfma_f32_16x16x4_f32 %1, %4, %12, " F01 "\n\t"                 \
    "v_mfma_f32_16x16x4_f32 %2, %6, %10, " F10 "\n\t"                 \
    "v_mfma_f32_16x16x4_f32 %3, %6, %14, " F11 "\n\t"                 \
    "v_mfma_f32_16x16x4_f32 %0, %5, %9, %0\n\t"                       \
    "v_mfma_f32_16x16x4_f32 %1, %5, %13, %1\n\t"                      \
    "v_mfma_f32_16x16x4_f32 %2, %7, %11, %2\n\t"                      \
    "v_mfma_f32_16x16x4_f32 %3, %7, %15, %3"
#define S4_IN "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3])
    // operands: %0 / %1 = position 2s (blocks 0 / 1), %2 / %3 = position 2s + 1; av = {pos 2s: t0, t1; pos 2s+1: t0, t1}; b likewise
    if constexpr (Z0 && Z1) {
        if constexpr (AGPR) asm volatile(S4_BODY("0", "0", "0", "0") : "=&a"(c00), "=&a"(c01), "=&a"(c10), "=&a"(c11) : S4_IN);
        else asm volatile(S4_BODY("0", "0", "0", "0") : "=&v"(c00), "=&v"(c01), "=&v"(c10), "=&v"(c11) : S4_IN);
    } else if constexpr (Z0) {  // position 2s starts from zero, 2s + 1 from what it holds (the bias position is odd: 7)
        if constexpr (AGPR) asm volatile(S4_BODY("0", "0", "%2", "%3") : "=&a"(c00), "=&a"(c01), "+a"(c10), "+a"(c11) : S4_IN);
        else asm volatile(S4_BODY("0", "0", "%2", "%3") : "=&v"(c00), "=&v"(c01), "+v"(c10), "+v"(c11) : S4_IN);
    } else {
        static_assert(!Z1, "only the odd position of a pair can carry the bias");
        if constexpr (AGPR) asm volatile(S4_BODY("%0", "%1", "%2", "%3") : "+a"(c00), "+a"(c01), "+a"(c10), "+a"(c11) : S4_IN);
        else asm volatile(S4_BODY("%0", "%1", "%2", "%3") : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11) : S4_IN);
    }
#undef S4_BODY
#undef S4_IN
}
__device__ __forceinline__ void wait_mfma_results() { asm volatile("s_nop 15\n\ts_nop 3"); }
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ f32x4 sub4(const f32x4& x, const f32x4& y) {
    f32x2 lo, hi;
    const f32x2 xl = {x[0], x[1]}, xh = {x[2], x[3]}, yl = {y[0], y[1]}, yh = {y[2], y[3]};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(xl), "v"(yl));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(xh), "v"(yh));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
struct Blk {
    int n, by, bx;
};
struct Item {
    int g, cb;
    Blk b0, b1;
    int nvalid;
};
}  // namespace

template <int LEVEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino4s_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: 16 output channels of the item's 64; input path: half (a >> 1), block (a & 1)
    const int m = lane & 15;                                 // MFMA row (cout) / column (tile within a block); input path: the tile
    const int ks = lane >> 4;                                // k-slot; input path: channels 4 ks + 2 half, + 1

    const int ncb = p.Cout >> 6;
    const int nblk = p.N * p.tiles_y * p.tiles_x;
    const int npair = (nblk + 1) >> 1;
    const int per_group = npair * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;
    const int nplane_o = p.Cout / CB;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ISTEP = (int)gridDim.x;
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
    // the block's TOP-LEFT neighbour in the guard-ringed block grid: every raw-patch offset is >= 0
    auto in_base = [&](int g, const Blk& b) {
        return reinterpret_cast<const char*>(p.in + g * p.in_gs) +
               (((long long)b.n * p.pl_byp + (b.by + p.ty_off)) * p.pl_bxp + (b.bx + p.tx_off)) * (long long)nchunk * PLANE_BYTES;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };

    // ---- lane invariants ---------------------------------------------------------------------------------------------------
    // Raw patch in LDS, per block: 1344 16-byte pieces; piece g = pixel slot (g >> 2), channel quad (g & 3); pixel (y, x) of the 18 x 18 patch
    // (image pixel (16 by + y - 1, 16 bx + x - 1)) sits in slot (18 y + x) ^ ((x >> 2) & 1): neighbouring tiles of a row start on slots of
    // opposite parity, so the 16 tiles of a patch read spread over both 64-byte halves of the LDS bank window.
    // Wave `a` issues wave instructions j = a, a + 4, ..., j < 42: block j / 21, pieces (j % 21) * 64 + lane.
    // The offsets are kernel invariants, but eleven registers kept alive across the whole item loop were spilled (each reload a scratch load followed
    // by s_waitcnt vmcnt(0) in front of the DMA it feeds: 3.9 ms per launch instead of 2.1).  They live in an LDS table instead and are read back
    // at the start of every chunk into registers that die within its first eleven steps -- while the patch registers are dead.
    {
        const unsigned rowblk = (unsigned)p.pl_bxp * (unsigned)nchunk * PLANE_BYTES, colblk = (unsigned)nchunk * PLANE_BYTES;
        unsigned* tab = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + LDS_TAB);
        for (int e = tid; e < 23 * 64; e += 256) {
            int g = e;
            if (g > 1295) g = 1295;  // the padding lanes of a block's last instruction fetch a valid piece into the padding of the buffer
            const int s = g >> 2, quad = g & 3;
            const int x0 = s % 18, pp = s ^ ((x0 >> 2) & 1), y = pp / 18, x = pp % 18;
            const int yy = y + 15, xx = x + 15;  // image pixel relative to the top-left neighbour block's origin
            const int dby = yy >> 4, dbx = xx >> 4, iy = yy & 15, ix = xx & 15;
            tab[e] = (unsigned)dby * rowblk + (unsigned)dbx * colblk +
                     (unsigned)(((((iy & 3) << 2) + (ix & 3)) * 16 + (((iy >> 2) & 3) << 2) + ((ix >> 2) & 3)) * 64 + quad * 16);
        }
    }
    // wave a's instruction k is j = 4 k + a of the 42; its table row is j % 21 = 4 k + a (k < 5, and k = 5 for wave 0), else 4 k + a - 21: a
    // compile-time displacement from ONE lane address (+ one more for k = 5) -- eleven separately computed addresses were hoisted and spilled too
    const char* tabA = reinterpret_cast<const char*>(lds) + LDS_TAB + a * 256 + lane * 4;
    const int tab5_off = __builtin_amdgcn_readfirstlane(a ? -256 : 5 * 1024);
    auto load_doff = [&](unsigned (&doff)[NDMA]) __attribute__((always_inline)) {
        static_for<0, NDMA>([&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < 5) doff[k] = *reinterpret_cast<const unsigned*>(tabA + k * 1024);
            else if constexpr (k == 5) {
                unsigned addr5;  // computed where it is used (volatile: a hoisted copy was kept in a register across the item loop and spilled)
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(addr5) : "s"(tab5_off), "v"((unsigned)(size_t)tabA));
                doff[k] = *reinterpret_cast<const unsigned __attribute__((address_space(3)))*>(addr5);
            }
            else doff[k] = *reinterpret_cast<const unsigned*>(tabA + k * 1024 - 21 * 256);
        });
    };
    // patch reads: thread (tile m of block a & 1, k-slot ks), channels 4 ks + 2 (a >> 1), + 1
    // (input-path lane roles: k-slot fastest -- lane = 4 tile + k-slot -- so that a quarter wave reads 4 tiles x 32 bytes spread over both 64-byte
    // halves of the bank window and a pair write of the wave is one contiguous 1-KiB run)
    const int tm = lane >> 2, tks = lane & 3;
    const int tty = tm >> 2, ttx = tm & 3, half = a >> 1;
    // slot of patch element (r, q) = (18 (4 tty + r) + 4 ttx + q) ^ ((ttx + (q >> 2)) & 1); the sum is even iff q is, so the XOR is +-1:
    //   even q (0, 2, 4): + 1 when flipped, odd q (1, 3, 5): - 1;  flipped  <=>  (ttx odd) for q < 4, (ttx even) for q >= 4
    const int slot0 = 18 * 4 * tty + 4 * ttx;
    const int rd_base = LDS_RAW + (a & 1) * RAW_BLOCK_BYTES + slot0 * 64 + tks * 16 + half * 8;  // bytes
    const int odd = ttx & 1;
    const int rd_lo = rd_base + (odd ? 64 : 0);    // q = 0, 2:  + 64 when ttx is odd
    const int rd_lo_m = rd_base - (odd ? 64 : 0);  // q = 1, 3:  - 64 when ttx is odd
    const int rd_hi = rd_base + (odd ? 0 : 64);    // q = 4:     + 64 when ttx is even (the pixel's 4-group is ttx + 1)
    const int rd_hi_m = rd_base - (odd ? 0 : 64);  // q = 5:     - 64 when ttx is even
    // V write position (floats) inside a half buffer: pair s adds s * NT * 16
    const int vw = (((a & 1) * 16 + tm) * 4 + tks) * 4;
    // V read position (floats), block 0; block 1 adds 16 * 16
    const int vr = (m * 4 + ks) * 4;
    const unsigned wlane = (unsigned)lane * 16u;
    const char* lds_b = reinterpret_cast<const char*>(lds);

    f32x2 d[6][6];
    f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f};
    asm volatile("" : "+v"(k2), "+v"(k4), "+v"(k5));
    auto bt6 = [&](f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4, f32x2& x5) __attribute__((always_inline)) {
        f32x2 t0, t1, u0, u1;
        asm("v_pk_fma_f32 %6, %2, %11, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 %7, %1, %11, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_add_f32 %8, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %9, %3, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %0, %11, %4\n\t"
            "v_pk_fma_f32 %5, %1, %11, %5\n\t"
            "v_pk_fma_f32 %0, %2, %12, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 %5, %3, %12, %5 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_add_f32 %1, %6, %7\n\t"
            "v_pk_add_f32 %2, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %3, %9, %10, %8\n\t"
            "v_pk_fma_f32 %4, %9, %10, %8 neg_lo:[1,0,0] neg_hi:[1,0,0]"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1)
            : "v"(k2), "v"(k4), "v"(k5));
    };
    auto pass_v = [&](int q) __attribute__((always_inline)) { bt6(d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]); };
    auto pass_h = [&](int r) __attribute__((always_inline)) { bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]); };
    auto read_row = [&](int r) __attribute__((always_inline)) {  // the six pixels of patch row r from the raw buffer
        d[r][0] = *reinterpret_cast<const f32x2*>(lds_b + rd_lo + (r * 18 + 0) * 64);
        d[r][1] = *reinterpret_cast<const f32x2*>(lds_b + rd_lo_m + (r * 18 + 1) * 64);
        d[r][2] = *reinterpret_cast<const f32x2*>(lds_b + rd_lo + (r * 18 + 2) * 64);
        d[r][3] = *reinterpret_cast<const f32x2*>(lds_b + rd_lo_m + (r * 18 + 3) * 64);
        d[r][4] = *reinterpret_cast<const f32x2*>(lds_b + rd_hi + (r * 18 + 4) * 64);
        d[r][5] = *reinterpret_cast<const f32x2*>(lds_b + rd_hi_m + (r * 18 + 5) * 64);
    };
    // V destination of this wave's half: V_A (waves 0, 1), or V_B[buf] (waves 2, 3)
    auto write_row = [&](int vb_buf, int r) __attribute__((always_inline)) {
        float* dst = lds + (half ? (LDS_VB / 4 + vb_buf * VH_FLOATS) : LDS_VA / 4) + vw;
        // positions 6 r + 2 j and 6 r + 2 j + 1 are one pair (6 r is even): both go out as ONE 16-byte write, {pos 0: t0, t1; pos 1: t0, t1}
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 v = {d[r][2 * j][0], d[r][2 * j][1], d[r][2 * j + 1][0], d[r][2 * j + 1][1]};
            *reinterpret_cast<f32x4*>(dst + (3 * r + j) * NT * 16) = v;
        }
    };
    // the raw patch of one 16-channel chunk: this wave's share of the 42 wave instructions
    // (a block's pieces fill exactly 21 KiB, so wave instruction j lands at LDS_RAW + 1024 j whichever block it belongs to; only k = 5 straddles:
    // j = 20 is block 0's last instruction, 21 .. 23 are block 1's first)
    // r5 = the block wave instruction j = 20 + a belongs to (block 0 for wave 0, block 1 for the others), chosen by the caller as a VALUE: a branch
    // here would put the two loads in different basic blocks, and the hand-counted waits below want one straight instruction stream
    auto stage = [&](const unsigned (&doff)[NDMA], const __amdgpu_buffer_rsrc_t& r0, const __amdgpu_buffer_rsrc_t& r1, const __amdgpu_buffer_rsrc_t& r5, int chunk_off, auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int j = 4 * k + a;
        if (k < 10 || a < 2) {  // j < 42
            if constexpr (k < 5) dma16(r0, doff[k], chunk_off, (unsigned)(LDS_RAW + j * 1024));
            else if constexpr (k > 5) dma16(r1, doff[k], chunk_off, (unsigned)(LDS_RAW + j * 1024));
            else dma16(r5, doff[k], chunk_off, (unsigned)(LDS_RAW + j * 1024));
        }
    };

    // ---- prologue: the first chunk's raw patch, its transform, V ------------------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w.g, w.b0)), r1 = make_rsrc(in_base(w.g, w.b1));
        const __amdgpu_buffer_rsrc_t r5 = make_rsrc(in_base(w.g, a == 0 ? w.b0 : w.b1));
        __syncthreads();  // the offset table
        unsigned doff[NDMA];
        load_doff(doff);
        static_for<0, NDMA>([&](auto K) __attribute__((always_inline)) { stage(doff, r0, r1, r5, 0, K); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; ++r) read_row(r);
#pragma unroll
    for (int q = 0; q < 6; ++q) pass_v(q);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        pass_h(r);
        write_row(0, r);
    }
    int vbb = 0;  // the V_B buffer the CURRENT chunk reads
    __syncthreads();
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    f32x4 wq[RING];
    f32x4 bnext;
    auto load_bias = [&](const Item& wi) {
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64 + 16 * a;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, 64, 0x00020000);
#ifdef S4_MANUAL_WAITS
        // straight into AccVGPRs, where the value waits until the next item's step BIAS_XI / 2: given a VGPR destination the compiler parks the
        // value in AccVGPRs itself -- with copies placed right behind the load, reading registers the data has not reached (tests/test_isa_hazard.py)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=a"(bnext) : "v"((unsigned)(fresh_lane() >> 4) * 16u), "s"(rb));
#else
        bnext = buf_load(rb, (unsigned)(fresh_lane() >> 4) * 16u, 0);
#endif
    };
    // the bias first: it is consumed at step BIAS_XI / 2 of the item's first chunk behind that step's weight wait, which only covers what is
    // OLDER than the step's own operand (in the output stage below it is younger than the next item's first WD operands, but >= 16 stores follow it)
    load_bias(w);
#pragma unroll
    for (int dd = 0; dd < PRE; ++dd) wq[dd] = weight_load(rw, wlane, dd * 1024);

    for (;;) {
        f32x4 acc[NPOS][2];
        const bool more_items = item + ISTEP < item_end;
        const Item wnx = more_items ? decode(item + ISTEP) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const char* cur0 = in_base(w.g, w.b0);
        const char* cur1 = in_base(w.g, w.b1);
        const char* nx0 = in_base(wnx.g, wnx.b0);
        const char* nx1 = in_base(wnx.g, wnx.b1);
#ifndef S4_MANUAL_WAITS
        acc[BIAS_XI][0] = bnext;
        acc[BIAS_XI][1] = bnext;
#endif

        auto chunk = [&](auto first_tag, int ch) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool last_ch = (ch == nchunk - 1);
            const __amdgpu_buffer_rsrc_t st0 = make_rsrc(last_ch ? nx0 : cur0), st1 = make_rsrc(last_ch ? nx1 : cur1);  // the next chunk's raw patch
            const __amdgpu_buffer_rsrc_t st5 = make_rsrc(a == 0 ? (last_ch ? nx0 : cur0) : (last_ch ? nx1 : cur1));
            const int stage_off = (last_ch ? 0 : ch + 1) * PLANE_BYTES;
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;
            const float* vsrcA = lds + LDS_VA / 4 + vr;
            const float* vsrcB = lds + LDS_VB / 4 + vbb * VH_FLOATS + vr;
            const int wbuf = vbb ^ 1;

            unsigned doff[NDMA];
            load_doff(doff);
            f32x4 bb[2][2];
            bb[0][0] = *reinterpret_cast<const f32x4*>(vsrcA);
            bb[0][1] = *reinterpret_cast<const f32x4*>(vsrcA + 16 * 16);
            static_for<0, NS>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                constexpr int hh = q / NPAIR, s = q % NPAIR;  // half, pair
                constexpr int x0 = 2 * s, x1 = 2 * s + 1;
                constexpr bool AG = x0 < NPOS_A;
                // weight stream: the operand of step q + WD
                {
                    constexpr int dd = q + WD;
                    if constexpr (!(FIRST && dd < PRE)) {
                        if constexpr (dd < NS) wq[dd % RING] = weight_load(rw, wlane, wcur_off + dd * 1024);
                        else wq[(dd - NS) % RING] = weight_load(rw_over, wlane, wover_off + (dd - NS) * 1024);
                    }
                }
                if constexpr (q + 1 < NS) {  // B operands of the next step
                    constexpr int nh = (q + 1) / NPAIR, ns = (q + 1) % NPAIR;
                    const float* src = (nh ? vsrcB : vsrcA) + ns * NT * 16;
                    bb[(q + 1) & 1][0] = *reinterpret_cast<const f32x4*>(src);
                    bb[(q + 1) & 1][1] = *reinterpret_cast<const f32x4*>(src + 16 * 16);
                }
                // first half: the NEXT chunk's raw patch, global -> LDS (everybody finished reading the raw buffer before the barrier that ended the
                // previous chunk)
#ifndef S4_ABL_NODMA
                if constexpr (q < NDMA) stage(doff, st0, st1, st5, stage_off, Q);
#endif
                // second half: this wave's channels of the next chunk: read, B^T d B, V writes
#ifndef S4_ABL_NOREAD
                if constexpr (q >= RQ && q < RQ + 6) read_row(q - RQ);
#endif
#ifndef S4_ABL_NOXF
                if constexpr (q >= RQ + 6 && q < RQ + 12) pass_v(q - RQ - 6);
#endif
                if constexpr (q >= RQ + 12 && q < RQ + 18) {
#ifndef S4_ABL_NOXF
                    pass_h(q - RQ - 12);
#endif
#ifndef S4_ABL_NOVWRITE
                    write_row(wbuf, q - RQ - 12);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef S4_MANUAL_WAITS
                {
                    // loads of THIS kernel's inline assembly younger than the operand of step q (requested at the top of step q - WD): the WD weight
                    // loads of steps q - WD + 1 .. q, and the patch loads of steps max(0, q - WD) .. min(q, 9) (step 10's is issued by two waves
                    // only and not counted).  A lower bound is all a wait needs: whatever else is queued (stores, the bias) only makes it stricter.
                    constexpr int lo = q - WD > 0 ? q - WD : 0, hi = q < 9 ? q : 9;
#if !defined(S4_ABL_NODMA) && !defined(S4_WAIT_IGNORE_DMA)
                    constexpr int nwait = WD + (hi >= lo ? hi - lo + 1 : 0);
#else
                    constexpr int nwait = WD;
#endif
                    static_assert(nwait < 64, "vmcnt is a 6-bit field");
                    if constexpr (FIRST && q == BIAS_XI / 2) {
                        // the item's bias enters position BIAS_XI here: everything but this item's own loads of steps 0 .. q must have landed
                        constexpr int nb = nwait - WD + q + 1 < nwait ? nwait - WD + q + 1 : nwait;
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nb));
                        asm volatile("" : "+a"(bnext));  // a value made AFTER the wait: no copy of it can be scheduled above it
                        acc[BIAS_XI][0] = bnext;
                        acc[BIAS_XI][1] = bnext;
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nwait));
                    }
                }
#endif
                const f32x4 av = wq[q % RING];
                const f32x4 b0 = bb[q & 1][0], b1 = bb[q & 1][1];
                constexpr bool Z = FIRST && hh == 0;
                mfma_pair_step<AG, Z && x0 != BIAS_XI, Z && x1 != BIAS_XI>(acc[x0][0], acc[x0][1], acc[x1][0], acc[x1][1], av, b0, b1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (q == NPAIR - 1) {
                    // mid-chunk: the raw patch requested at steps 0 .. 10 must have landed (at most the 6 youngest loads -- weight operands of
                    // steps 12 .. 17 + WD -- may still be in flight: any count below 7 is safe whatever else the compiler queued), V_A is read
#ifndef S4_ABL_NOWAIT
                    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#endif
#ifndef S4_ABL_NOMIDBAR
                    __syncthreads();
#endif
                }
            });
            __syncthreads();  // everybody has read V_B of this chunk and written V_A / V_B of the next one
            vbb ^= 1;
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform A^T M A (conv_wino4p.hip) ---------------------------------------------------------------------------------
#pragma unroll
        for (int dd = WD; dd < PRE; ++dd) wq[dd % RING] = weight_load(rw_nx, wlane, dd * 1024);
#ifndef S4_MANUAL_WAITS
        load_bias(wnx);
        wait_mfma_results();
#else
        // the asm bias load wants four AccVGPRs and all 256 hold accumulators: the compiler frees them by reading an accumulator out first -- which
        // must not happen before the matrix pipe has drained (it does not know the asm matrix instructions' latency)
        wait_mfma_results();
        load_bias(wnx);
#endif
        {
            int relu_s = p.relu;
            asm volatile("" : "+s"(relu_s));
            const float floor_ = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(relu_s ? 0 : (int)0xff7fffff));
            const int lane_o = fresh_lane();
            const int m_o = lane_o & 15;
            const unsigned olane = (unsigned)(m_o * 64 + (lane_o >> 4) * 16);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const Blk bo = tb ? w.b1 : w.b0;
                if (tb == 1 && w.nvalid == 1) continue;
                const int by_abs = bo.by + p.ty_off, bx_abs = bo.bx + p.tx_off;
                const long long origin = p.out_gs * w.g + ((((long long)bo.n * p.pl_byp + by_abs + 1) * p.pl_bxp + bx_abs + 1) * nplane_o + w.cb * 4 + a) * (long long)(PLANE_BYTES / 4);
                const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + origin, 0, PLANE_BYTES, 0x00020000);
                const bool partial = (by_abs * BLK + BLK > p.Ho) || (bx_abs * BLK + BLK > p.Wo);
                f32x4 T[4][6];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const f32x4 m0 = acc[0 * 6 + b][tb], m1 = acc[1 * 6 + b][tb], m2 = acc[2 * 6 + b][tb], m3 = acc[3 * 6 + b][tb],
                                m4 = acc[4 * 6 + b][tb], m5 = acc[5 * 6 + b][tb];
                    const f32x4 s1 = m1 + m2, d1 = sub4(m1, m2), s2 = m3 + m4, d2 = sub4(m3, m4);
                    T[0][b] = m0 + s1 + s2;
                    T[1][b] = d1 + 2.f * d2;
                    T[2][b] = s1 + 4.f * s2;
                    T[3][b] = (d1 + 8.f * d2) + m5;
                }
                const int rem_y = p.Ho - by_abs * BLK - 4 * (m_o >> 2), rem_x = p.Wo - bx_abs * BLK - 4 * (m_o & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 s1 = T[i][1] + T[i][2], d1 = sub4(T[i][1], T[i][2]), s2 = T[i][3] + T[i][4], d2 = sub4(T[i][3], T[i][4]);
                    f32x4 y[4];
                    y[0] = T[i][0] + s1 + s2;
                    y[1] = d1 + 2.f * d2;
                    y[2] = s1 + 4.f * s2;
                    y[3] = (d1 + 8.f * d2) + T[i][5];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 o = y[j];
                        o[0] = fmaxf(o[0], floor_);
                        o[1] = fmaxf(o[1], floor_);
                        o[2] = fmaxf(o[2], floor_);
                        o[3] = fmaxf(o[3], floor_);
                        const unsigned vo = (!partial || (i < rem_y && j < rem_x)) ? olane : 0x80000000u;
                        buf_store(o, r_out, vo, (i * 4 + j) * 1024);
                    }
                }
            }
        }
        if (!more_items) break;
        item += ISTEP;
        w = wnx;
        rw = rw_nx;
    }
}

hipError_t cerb_launch_wino4s(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64 || p.resid || p.pl_byp < 3 || p.pl_bxp < 3 || p.H != p.Ho || p.W != p.Wo) return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + BLK - 1) / BLK;
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
    auto kern = p.level_tag ? conv_wino4s_kernel<1> : conv_wino4s_kernel<0>;
    static bool attr_done[2][64] = {};
    if (cerb_attr_needed(attr_done[p.level_tag ? 1 : 0])) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    long long grid = 256;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
    return hipGetLastError();
}
