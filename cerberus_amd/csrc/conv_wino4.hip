// conv_wino4.hip -- 3x3 stride-1 convolution (+ folded BN bias, residual, ReLU) as Winograd F(4x4, 3x3) on the gfx950 fp32 matrix
// cores (cerb_net_set_conv_algo(5)).  36 products per 4x4 output pixels and input channel instead of 64 for F(2x2) (144 direct): 1.78x
// fewer matrix instructions than conv_wino.hip AND 1.78x less input-transform volume per output pixel (a 6x6 patch per 16 outputs
// instead of a 4x4 patch per 4) -- the input path is what bounded the three F(2x2) kernels.  Products are exact fp32
// FMAs; the transforms (points 0, +-1, +-2, inf) move the probability maps by < 4e-6 against an fp64 evaluation of the network
// (tests/tools/dev_wino4_numerics.py), 1.5x the direct fp32 convolution's own distance.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d 6x6 input patch, g 3x3 filter (BN folded), Y 4x4 outputs
//
// Work decomposition (conv_wino16d.hip's, grown to 36 positions):
//   * item = two 16x16-pixel blocks (2 x 16 tiles of 4x4 outputs) x 64 output channels; persistent workgroups of 4 waves, ONE per CU;
//   * a wave owns ALL 36 positions of 16 output channels for the 32 tiles on v_mfma_f32_16x16x4_f32: 36 x 2 x 4 = 288 accumulator
//     registers -- one wave per SIMD, accumulators in the AccVGPR half of the 512-register file -- so every weight register feeds two
//     matrix instructions (a 16-tile item would need 1 KiB of weights per wave per 128 cycles = the L2's whole bandwidth) and the
//     output transform A^T M A is additions in registers: no exchange through LDS;
//   * 16-channel chunks, V tile [36][32 tiles][16 ch] = 72 KiB, double-buffered (144 of the CU's 160 KiB): one barrier per chunk;
//     16-byte slots of a tile's channel vector are XOR-swizzled by the tile index so that ds_read_b128 of 16 tiles x 4 k-slots and the
//     8-byte transform writes are bank-conflict-free without padding;
//   * thread = (tile, channel pair): the raw 6x6 patch is 36 float2 loaded straight from global memory (uniform offsets ride in the
//     buffer instruction's scalar offset), transformed in place one 1-D pass per step in the shadow of the matrix pipe.
// Reference layers: models/utils/conv_layers.py:24-60 (_ConvLayer) and models/backbone/resnet.py:81-97 (BasicBlock).
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int NPOS = 36;
constexpr int NT = 32;                        // tiles per item: two blocks of 4x4 tiles
constexpr int BLK = 16;                       // a block is 16x16 output pixels
constexpr int CB = 16;                        // input channels per LDS pass
constexpr int V_FLOATS = NPOS * NT * CB;      // one V buffer: 72 KiB
constexpr int LDS_BYTES = 2 * V_FLOATS * 4;   // double-buffered: 144 KiB
constexpr int OPX = 68;                       // output staging: floats per pixel (64 channels + 4: bank skew)
static_assert(256 * OPX + 16 <= V_FLOATS, "a block's outputs are staged in one V buffer");
#ifdef W4_PROF
constexpr int PROF_BYTES = 16 * 40 * 8;
#else
constexpr int PROF_BYTES = 0;
#endif
constexpr int NS = NPOS;                      // steps per chunk: one position each (one 16-byte weight load, two ds_read_b128, 8 MFMAs)
#ifndef W4_RING
#define W4_RING 12
#endif
constexpr int RING = W4_RING;                 // weight operand slots (NS % RING == 0: the slot of a step does not depend on the chunk)
#ifndef W4_PRE
#define W4_PRE 8
#endif
constexpr int PRE = W4_PRE;                   // steps of the NEXT item whose weights are requested before an item's output stores
#ifndef W4_WD
#define W4_WD 8
#endif
constexpr int WD = W4_WD;                     // weight prefetch distance in steps
#ifndef W4_WB
#define W4_WB 4
#endif
constexpr int WB = W4_WB;                     // weight burst size in steps
#ifndef W4_PL
#define W4_PL 3
#endif
constexpr int PL = W4_PL;                     // patch loads issued per step
#ifndef W4_P0
#define W4_P0 0
#endif
constexpr int P0 = W4_P0;                     // first step that issues patch loads
#ifndef W4_TQ
#define W4_TQ 18
#endif
#ifndef W4_PATCH_AUX
#define W4_PATCH_AUX 0
#endif
#ifndef W4_STORE_AUX
#define W4_STORE_AUX 0
#endif
constexpr int TQ = W4_TQ;                     // the next chunk's patch is masked at TQ, transformed at TQ+1 .. TQ+12, written at TQ+7 .. TQ+12
static_assert(NS % RING == 0 && WD + WB <= RING && NS % WB == 0 && PRE >= WD && PRE <= RING, "weight ring");
#ifdef W4_ABL_HALFPATCH
static_assert(P0 + (W4_ABL_HALFPATCH + PL - 1) / PL <= TQ && TQ + 13 < NS, "ablation: the loads that are issued must precede the transform");
#else
static_assert(P0 + (36 + PL - 1) / PL <= TQ && TQ + 13 < NS, "the patch must be requested before its transform starts");
#endif
constexpr int BIAS_XI = 7;                    // A^T[i][1] A[1][j] = 1 for all 16 outputs: the bias enters through position (1, 1)
constexpr int CHUNK_W_BYTES = NPOS * 4 * 1024;  // packed weights of one (cout block, 16-channel chunk): 144 KiB
constexpr int WAVE_W_BYTES = NPOS * 1024;       // one wave's share: 36 steps x 1 KiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
// the input patch: 2 GiB of range, so that a lane offset of 0x80000000 is out of range and the hardware returns zeros (conv_wino4b.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_lim(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, W4_STORE_AUX);
    asm volatile("s_nop 1");  // gfx950 store hazard, see conv_wino.hip buf_store / tests/test_isa_hazard.py
    __builtin_amdgcn_sched_barrier(0);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, W4_PATCH_AUX));
}

#ifdef W4_PROF
// developer instrumentation (scripts/dev_w4prof.py): wave 0 of workgroup W4_PROF stamps s_memtime at every step of its second item
__device__ unsigned long long w4_prof_buf[16 * 40];
#endif

struct Blk {
    int n, by, bx;  // image, block row / column inside the launch's block grid
};
struct Item {
    int g, cb;
    Blk b0, b1;  // (no array: a dynamically indexed member would keep the whole struct in scratch, i.e. in vector registers)
    int nvalid;  // 2, or 1 when the launch has an odd number of blocks and this is the last pair (block 1 repeats block 0, stores dropped)
};
}  // namespace

// STATS 1 (training forward): the output stage also leaves BatchNorm statistics partials per block (ConvParams::bn_part) -- a separate
// instantiation, so that the inference kernels are the same code as without it.
// STATS 2 (training backward, this launch is a data gradient whose output is the gradient dz behind a BatchNorm + ReLU and its only writer): the output stage
// reads the BatchNorm's INPUT y at its own pixels and leaves that BatchNorm's backward sums per block -- (sum dz', sum dz' xhat), dz' = dz where bn_out(y) > 0 --
// in the same place and layout: the BatchNorm's reduction pass over dz and y (train_kernels.hip: bn_bwd_partial_kernel) does not run (ConvParams::bst_*).
template <bool HAS_RES, int STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino4_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float bnred[STATS ? 4 * 16 * 8 : 4];  // STATS: [wave][channel quad][4 sums, 4 sums of squares]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's 16 output channels of the item's 64; input path: block a >> 1
    const int m = lane & 15;                                 // MFMA row (cout) / column (tile within a block)
    const int ks = lane >> 4;                                // k-slot

    const int ncb = p.Cout >> 6;
    const int nblk = p.N * p.tiles_y * p.tiles_x;  // blocks per group
    const int npair = (nblk + 1) >> 1;
    const int per_group = npair * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int base_cnt = total / (int)gridDim.x, rem_cnt = total % (int)gridDim.x;
    int item = lb * base_cnt + min(lb, rem_cnt);
#ifndef W4_CONTIGUOUS  // workgroup lb takes items lb, lb + G, ...: neighbouring blocks (shared halo lines, the cout blocks of one patch) run at the
                       // same time on one XCD and meet in its L2 (-1 % / -3 % against contiguous item ranges per workgroup)
    const int ISTEP = (int)gridDim.x;
    item = lb;
    const int item_end = total;
#else
    const int ISTEP = 1;
    const int item_end = item + base_cnt + (lb < rem_cnt ? 1 : 0);
#endif
    if (item >= item_end) return;
#ifdef W4_PROF
    const int prof_item = item + ISTEP;
#endif

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
    auto oy0 = [&](const Blk& b) { return (b.by + p.ty_off) * BLK; };
    auto ox0 = [&](const Blk& b) { return (b.bx + p.tx_off) * BLK; };
    auto in_base = [&](int g, const Blk& b) {
        return reinterpret_cast<const char*>(p.in + g * p.in_gs) + ((((long long)b.n * p.H + (oy0(b) - 1)) * p.W + (ox0(b) - 1)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };
    auto hangs_over = [&](const Blk& b) { return oy0(b) + BLK > p.H || ox0(b) + BLK > p.W; };
    auto edge_bits = [&](const Blk& b) {  // 1 top, 2 bottom, 4 left, 8 right
        return (oy0(b) == 0 ? 1 : 0) | (oy0(b) + BLK == p.H ? 2 : 0) | (ox0(b) == 0 ? 4 : 0) | (ox0(b) + BLK == p.W ? 8 : 0);
    };

    // ---- lane invariants ---------------------------------------------------------------------------------------------------
    // input transform: thread = (tile t of the item, channel pair c); tiles 0..15 are block 0 (waves 0, 1), 16..31 block 1 (waves 2, 3)
    const int t = tid >> 3, c = tid & 7, tm = t & 15, tty = tm >> 2, ttx = tm & 3;
    const bool second = (a >> 1) != 0;
    auto mine = [&](const Item& wi) {
        Blk b;
        b.n = second ? wi.b1.n : wi.b0.n;
        b.by = second ? wi.b1.by : wi.b0.by;
        b.bx = second ? wi.b1.bx : wi.b0.bx;
        return b;
    };
    const unsigned ioff = (unsigned)((((4 * tty) * p.W + 4 * ttx) * p.Cin + 2 * c) * 4);
    const int vw = t * CB + (((c >> 1) ^ ((tm & 8) ? 3 : 0)) << 2) + 2 * (c & 1);  // V write position (floats); position xi adds xi*NT*CB
    const int vr = m * CB + ((ks ^ ((m & 8) ? 3 : 0)) << 2);                       // V read position, block 0; block 1 adds 16*CB
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;

    f32x2 d[6][6];  // raw patch of the NEXT chunk (two channels), transformed in place in the shadow of the matrix pipe
    const bool lane_top = (tty == 0), lane_bot = (tty == 3), lane_left = (ttx == 0), lane_right = (ttx == 3);
    // zero padding at the image border by ADDRESS (conv_wino4b.hip): nine lane offsets per chunk instead of ~100 VALU instructions of masking
    struct EdgeOff { unsigned o[3][3]; };
    auto edge_offsets = [&](int bits) __attribute__((always_inline)) {
        EdgeOff e;
        const bool zt = (bits & 1) && lane_top, zb = (bits & 2) && lane_bot, zl = (bits & 4) && lane_left, zr = (bits & 8) && lane_right;
        const unsigned col[3] = {zl ? 0x80000000u : ioff, ioff, zr ? 0x80000000u : ioff};
#pragma unroll
        for (int qc = 0; qc < 3; ++qc) {
            e.o[0][qc] = zt ? 0x80000000u : col[qc];
            e.o[1][qc] = col[qc];
            e.o[2][qc] = zb ? 0x80000000u : col[qc];
        }
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int qc = 0; qc < 3; ++qc) asm volatile("" : "+v"(e.o[rc][qc]));
        return e;
    };
    auto issue = [&](__amdgpu_buffer_rsrc_t r, const EdgeOff& e, int chunk_off, int k) __attribute__((always_inline)) {
        const int rr = k / 6, qq = k % 6;
        d[rr][qq] = buf_load2(r, e.o[rr == 0 ? 0 : rr == 5 ? 2 : 1][qq == 0 ? 0 : qq == 5 ? 2 : 1], chunk_off + rr * rowb + qq * pixb);
    };
    auto mask_border = [&](const Blk& b) __attribute__((always_inline)) {
        const f32x2 z = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int gy = oy0(b) - 1 + 4 * tty + r, gx = ox0(b) - 1 + 4 * ttx + q;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                d[r][q] = ok ? d[r][q] : z;
            }
    };
    // B^T x for the points (0, 1, -1, 2, -2, inf), in place: 12 packed operations.  Written as v_pk_fma_f32 / v_pk_add_f32 by hand: hipcc
    // (ROCm 7.2) scalarises vector subtractions and multiplies by negative literals (116 v_fma_f32 + 44 v_add_f32 + 64 packed instructions
    // per chunk instead of 144 packed ones), and every VALU instruction of this wave is a matrix-pipe cycle lost (one wave per SIMD).
    f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f};
    asm volatile("" : "+v"(k2), "+v"(k4), "+v"(k5));
    auto bt6 = [&](f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4, f32x2& x5) __attribute__((always_inline)) {
#ifdef W4_C_XF
        const f32x2 t0 = x4 - 4.f * x2, t1 = x3 - 4.f * x1;
        const f32x2 u0 = x4 - x2, u1 = x3 - x1;
        x0 = (4.f * x0 + x4) - 5.f * x2;
        x5 = (4.f * x1 + x5) - 5.f * x3;
        x1 = t0 + t1;
        x2 = t0 - t1;
        x3 = u0 + 2.f * u1;
        x4 = u0 - 2.f * u1;
#else
        f32x2 t0, t1, u0, u1;
        asm("v_pk_fma_f32 %6, %2, %11, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // t0 = x4 - 4 x2
            "v_pk_fma_f32 %7, %1, %11, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // t1 = x3 - 4 x1
            "v_pk_add_f32 %8, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"            // u0 = x4 - x2
            "v_pk_add_f32 %9, %3, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"            // u1 = x3 - x1
            "v_pk_fma_f32 %0, %0, %11, %4\n\t"                                  // x0 = 4 x0 + x4
            "v_pk_fma_f32 %5, %1, %11, %5\n\t"                                  // x5 = 4 x1 + x5
            "v_pk_fma_f32 %0, %2, %12, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // x0 -= 5 x2
            "v_pk_fma_f32 %5, %3, %12, %5 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // x5 -= 5 x3
            "v_pk_add_f32 %1, %6, %7\n\t"                                       // x1 = t0 + t1
            "v_pk_add_f32 %2, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"            // x2 = t0 - t1
            "v_pk_fma_f32 %3, %9, %10, %8\n\t"                                  // x3 = u0 + 2 u1
            "v_pk_fma_f32 %4, %9, %10, %8 neg_lo:[1,0,0] neg_hi:[1,0,0]"         // x4 = u0 - 2 u1
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1)
            : "v"(k2), "v"(k4), "v"(k5));
#endif
    };
    auto pass_v = [&](int q) { bt6(d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q]); };  // down column q
    auto pass_h = [&](int r) { bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]); };  // along row r
    auto write_row = [&](int buf, int r) {
#pragma unroll
        for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(lds + buf * V_FLOATS + (r * 6 + b) * NT * CB + vw) = d[r][b];
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------------
    Item w = decode(item);
    {
        const Blk b0_ = mine(w);
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc_lim(in_base(w.g, b0_));
        const EdgeOff e0 = edge_offsets(hangs_over(b0_) ? 0 : edge_bits(b0_));
#pragma unroll
        for (int k = 0; k < 36; ++k) issue(r0, e0, 0, k);
        if (hangs_over(b0_)) mask_border(b0_);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) pass_v(q);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        pass_h(r);
        write_row(0, r);
    }
    int vbuf = 0;  // the buffer the CURRENT chunk reads; the next chunk's patch goes to vbuf ^ 1
    __syncthreads();
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    // Weight stream in bursts (conv_wino16d.hip): every WB steps the operands of steps q + WD .. q + WD + WB - 1 are requested at once; the
    // operand of step q lives in slot q % RING.  Steps past the chunk's 36 belong to the next chunk, or to the next item's first chunk.
    f32x4 wq[RING];
#pragma unroll
    for (int dd = 0; dd < PRE; ++dd) wq[dd] = buf_load(rw, wlane, dd * 1024);
    f32x4 bnext;
    auto load_bias = [&](const Item& wi) {
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64 + 16 * a;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, 64, 0x00020000);
        bnext = buf_load(rb, (unsigned)ks * 16u, 0);
    };
    load_bias(w);

    for (;;) {
        f32x4 acc[NPOS][2];
        const bool more_items = item + ISTEP < item_end;
        const Item wnx = more_items ? decode(item + ISTEP) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const Blk bcur = mine(w), bnx = mine(wnx);
        const bool mask_cur = hangs_over(bcur), mask_next = hangs_over(bnx);
        const int edge_next = edge_bits(bnx), edge_cur = edge_bits(bcur);
        const char* in_cur = in_base(w.g, bcur);
        const char* in_nx = in_base(wnx.g, bnx);
        acc[BIAS_XI][0] = bnext;
        acc[BIAS_XI][1] = bnext;
#ifdef W4_PROF
        const bool prof_on = (blockIdx.x == W4_PROF) && a == 0 && (item == prof_item);
#define W4_STAMP(k) do { if (prof_on && lane == 0) reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[15 * 40 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4_STAMP(k) do { } while (0)
#endif

        auto chunk = [&](auto first_tag, int ch) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool last_ch = (ch == nchunk - 1);
            const Blk bp_ = last_ch ? bnx : bcur;
            const bool mask_nx = last_ch ? mask_next : mask_cur;
            const int edge_nx = last_ch ? edge_next : edge_cur;
#ifdef W4_ABL_PATCHHOT
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc_lim(reinterpret_cast<const char*>(p.in) + (blockIdx.x & 7) * 65536);  // cache-resident
            const int stage_off = 0;
#else
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc_lim(last_ch ? in_nx : in_cur);
            const int stage_off = (last_ch ? 0 : ch + 1) * (CB * 4);
#endif
            const EdgeOff eN = edge_offsets(mask_nx ? 0 : edge_nx);
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;
            const float* vsrc = lds + vbuf * V_FLOATS + vr;
            const int wbuf = vbuf ^ 1;

            f32x4 bb[2][2];  // B operands (blocks 0, 1) of step q in bb[q & 1]
            bb[0][0] = *reinterpret_cast<const f32x4*>(vsrc);
            bb[0][1] = *reinterpret_cast<const f32x4*>(vsrc + 16 * CB);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int xi = q;
#ifdef W4_PROF
                if (prof_on && ch < 16 && lane == 0) reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[ch * 40 + q] = __builtin_readcyclecounter();
#endif
#ifndef W4_ABL_NOWLOAD
                if (q % WB == 0) {
#pragma unroll
                    for (int dd = q + WD; dd < q + WD + WB; ++dd) {
                        if (FIRST && dd < PRE) continue;  // requested before the previous item's stores (or in the prologue)
                        if (dd < NS) wq[dd % RING] = buf_load(rw, wlane, wcur_off + dd * 1024);
                        else wq[(dd - NS) % RING] = buf_load(rw_over, wlane, wover_off + (dd - NS) * 1024);
                    }
                }
#endif
                if (q + 1 < NS) {
                    bb[(q + 1) & 1][0] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * CB);
                    bb[(q + 1) & 1][1] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * CB + 16 * CB);
                }
#ifndef W4_ABL_NOPATCH
                if (q >= P0 && (q - P0) * PL < 36 && q <= TQ) {  // next chunk's patch: PL loads per step
#pragma unroll
                    for (int u = 0; u < PL; ++u)
#ifdef W4_ABL_HALFPATCH
                        if ((q - P0) * PL + u < W4_ABL_HALFPATCH) issue(r_stage, eN, stage_off, (q - P0) * PL + u);
#else
                        if ((q - P0) * PL + u < 36) issue(r_stage, eN, stage_off, (q - P0) * PL + u);
#endif
                }
#endif
                // the next chunk's patch landed: mask, B^T d B (one 1-D pass per step) and the V writes into the OTHER buffer
                if (q == TQ) {
                    if (mask_nx) mask_border(bp_);
                }
#ifndef W4_ABL_NOXF
                if (q > TQ && q <= TQ + 6) pass_v(q - TQ - 1);
                if (q > TQ + 6 && q <= TQ + 12) pass_h(q - TQ - 7);
#endif
#ifndef W4_ABL_NOVWRITE
                if (q > TQ + 6 && q <= TQ + 12) write_row(wbuf, q - TQ - 7);
#endif
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 av = wq[q % RING];
                const f32x4 b0 = bb[q & 1][0], b1 = bb[q & 1][1];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    if (FIRST && tt == 0 && xi != BIAS_XI) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], z, 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], z, 0, 0, 0);
                    } else {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], acc[xi][1], 0, 0, 0);
                    }
                }
            }
#ifndef W4_ABL_NOBAR
            __syncthreads();  // everybody has read this chunk's V and written the next one's
#endif
            vbuf ^= 1;
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform A^T M A, entirely in registers; lane (m, ks): tile m of each block, channels 4 ks .. + 3 of the wave's 16 -------
        // vmcnt retires in order across loads AND stores, and the stores' acknowledgements take microseconds: everything the next item
        // needs during its first PRE steps is requested here, before this item's stores enter the queue (its steps 0 .. WD-1 went out
        // during the last chunk).  The raw-patch registers are dead at this point, so the extra slots cost nothing in the steady state.
        W4_STAMP(0);
#pragma unroll
        for (int dd = WD; dd < PRE; ++dd) wq[dd % RING] = buf_load(rw_nx, wlane, dd * 1024);
        load_bias(wnx);
        {
            // The wave's results are 64-byte pieces (16 channels) of pixels 4 apart: stored directly, one instruction touches 16 partial
            // cache lines and takes ~300 cycles to issue, with the matrix pipe idle (measured: 8 k of an item's 80 k cycles).  The four
            // waves therefore transpose each block through the V buffer the last chunk has finished with (64 KiB + skew of its 72):
            // [pixel][64 channels], and store whole pixel rows -- 1 KiB contiguous (4 pixels x 256 bytes) per instruction.
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));  // recomputed per item: keeps these out of the MFMA phase's register budget
            const int mo = lane_o & 15, kso = lane_o >> 4;
            float* stg = lds + (vbuf ^ 1) * V_FLOATS;
            // write side: lane (tile mo, channel quad kso) owns pixels (4 ty + i, 4 tx + j); pixel stride 68 floats, 4 floats of skew per tile row
            const int sw = ((64 * (mo >> 2) + 4 * (mo & 3)) * OPX + 4 * (mo >> 2) + 16 * a + 4 * kso);
            // read side: wave a stores pixel rows 4 a .. 4 a + 3; lane = (pixel lane_o >> 4 of a group of four, 16-byte piece lane_o & 15)
            const int sr = ((64 * a + (lane_o >> 4)) * OPX + 4 * a + 4 * (lane_o & 15));
            const int orow = p.Wo * p.Cout * 4, opix = p.Cout * 4;
            const unsigned ooff = (unsigned)(((4 * a * p.Wo + (lane_o >> 4)) * p.Cout + 4 * (lane_o & 15)) * 4);
            const float floor_ = p.relu ? 0.f : -3.402823466e38f;
            const unsigned span = (unsigned)(BLK * p.Wo * p.Cout * 4);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const Blk bo = tb ? w.b1 : w.b0;
                const int by0 = oy0(bo), bx0 = ox0(bo);
                const long long origin = (((long long)bo.n * p.Ho + by0) * p.Wo + bx0) * p.Cout + w.cb * 64;  // floats, uniform
                const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + w.g * p.out_gs + origin, 0, span, 0x00020000);
                const bool partial = (by0 + BLK > p.Ho) || (bx0 + BLK > p.Wo);
                const bool dead = (tb == 1 && w.nvalid == 1);
                W4_STAMP(1 + 3 * tb);
                // vertical pass: T[i][b] = sum_a A^T[i][a] M[a][b]
                f32x4 T[4][6];
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const f32x4 m0 = acc[0 * 6 + b][tb], m1 = acc[1 * 6 + b][tb], m2 = acc[2 * 6 + b][tb], m3 = acc[3 * 6 + b][tb],
                                m4 = acc[4 * 6 + b][tb], m5 = acc[5 * 6 + b][tb];
                    const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    T[0][b] = m0 + s1 + s2;
                    T[1][b] = d1 + 2.f * d2;
                    T[2][b] = s1 + 4.f * s2;
                    T[3][b] = (d1 + 8.f * d2) + m5;
                }
                W4_STAMP(2 + 3 * tb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 s1 = T[i][1] + T[i][2], d1 = T[i][1] - T[i][2], s2 = T[i][3] + T[i][4], d2 = T[i][3] - T[i][4];
                    *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 0) * OPX) = T[i][0] + s1 + s2;
                    *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 1) * OPX) = d1 + 2.f * d2;
                    *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 2) * OPX) = s1 + 4.f * s2;
                    *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 3) * OPX) = (d1 + 8.f * d2) + T[i][5];
                }
                __syncthreads();
                // 16 groups of four pixels per wave: row 4 a + (k >> 2), pixels 4 (k & 3) .. + 3
                unsigned vo[4];
#pragma unroll
                for (int x4 = 0; x4 < 4; ++x4) {
                    const bool ok = !dead && (!partial || (bx0 + 4 * x4 + (lane_o >> 4) < p.Wo));
                    vo[x4] = ok ? ooff : 0x80000000u;  // out-of-range offsets: the hardware drops the store / returns 0
                }
                f32x4 res[16];
                if (HAS_RES) {
                    const __amdgpu_buffer_rsrc_t r_res =
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + w.g * p.resid_gs + origin), 0, span, 0x00020000);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool rowok = !partial || (by0 + 4 * a + (k >> 2) < p.Ho);
                        res[k] = buf_load(r_res, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + 4 * (k & 3) * opix);
                    }
                }
                // (requested here, behind the staging barrier; requested before the output transform instead: no faster, and one instantiation of conv_wino4b spilled)
                f32x4 yv[STATS == 2 ? 16 : 1], bm, brs, bga, bbe;  // STATS 2: the BatchNorm's input at this lane's pixels, its parameters for this lane's four channels
                if constexpr (STATS == 2) {
                    const __amdgpu_buffer_rsrc_t r_y =
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bst_y + w.g * p.bst_y_gs + origin), 0, span, 0x00020000);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool rowok = !partial || (by0 + 4 * a + (k >> 2) < p.Ho);
                        yv[k] = buf_load(r_y, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + 4 * (k & 3) * opix);
                    }
                    const int pc = w.g * p.Cout + w.cb * 64 + 4 * (lane_o & 15);
                    bm = *reinterpret_cast<const f32x4*>(p.bst_mean + pc);
                    brs = *reinterpret_cast<const f32x4*>(p.bst_rstd + pc);
                    bga = *reinterpret_cast<const f32x4*>(p.bst_gamma + pc);
                    bbe = *reinterpret_cast<const f32x4*>(p.bst_beta + pc);
                }
                f32x4 bts = {0.f, 0.f, 0.f, 0.f}, btq = {0.f, 0.f, 0.f, 0.f};  // STATS: this lane's 16 pixels x 4 channels
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    f32x4 o = *reinterpret_cast<const f32x4*>(stg + sr + (16 * (k >> 2) + 4 * (k & 3)) * OPX);
                    if (HAS_RES) o = o + res[k];
                    o[0] = fmaxf(o[0], floor_);
                    o[1] = fmaxf(o[1], floor_);
                    o[2] = fmaxf(o[2], floor_);
                    o[3] = fmaxf(o[3], floor_);
                    const bool rowok = !partial || (by0 + 4 * a + (k >> 2) < p.Ho);
                    if constexpr (STATS == 1) {
                        if (rowok && vo[k & 3] != 0x80000000u) {  // pixels inside the image only
                            bts = bts + o;
                            btq[0] = fmaf(o[0], o[0], btq[0]);
                            btq[1] = fmaf(o[1], o[1], btq[1]);
                            btq[2] = fmaf(o[2], o[2], btq[2]);
                            btq[3] = fmaf(o[3], o[3], btq[3]);
                        }
                    }
                    if constexpr (STATS == 2) {
                        if (rowok && vo[k & 3] != 0x80000000u) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {  // the mask by the ONE expression every BatchNorm kernel uses (train_kernels.hip: bn_out): identical ReLU masks
                                const float yy = yv[k][e];
                                const float z = __fmaf_rn(yy - bm[e], brs[e] * bga[e], bbe[e]);
                                const float g = z > 0.f ? o[e] : 0.f;
                                bts[e] += g;
                                btq[e] = fmaf(g, (yy - bm[e]) * brs[e], btq[e]);
                            }
                        }
                    }
#ifndef W4_ABL_NOSTORE
                    buf_store(o, r_out, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + 4 * (k & 3) * opix);
#else
                    if (o[0] == 1.2345e-30f) buf_store(o, r_out, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + 4 * (k & 3) * opix);
#endif
                }
                if constexpr (STATS) {  // the four lanes that hold a channel quad, then (behind the barrier) the four waves = the block's 256 pixels
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bts[e] += __shfl_xor(bts[e], 16);
                        bts[e] += __shfl_xor(bts[e], 32);
                        btq[e] += __shfl_xor(btq[e], 16);
                        btq[e] += __shfl_xor(btq[e], 32);
                    }
                    if (lane_o < 16) {
                        *reinterpret_cast<f32x4*>(bnred + (a * 16 + lane_o) * 8) = bts;
                        *reinterpret_cast<f32x4*>(bnred + (a * 16 + lane_o) * 8 + 4) = btq;
                    }
                }
                __syncthreads();  // the staging buffer is rewritten by the next block, then by the next item's second chunk
                if constexpr (STATS) {
                    if (a == 0 && lane_o < 16 && !dead && p.bn_part) {
                        const long long blk = ((long long)bo.n * p.tiles_y + bo.by) * p.tiles_x + bo.bx;
                        double* dst = p.bn_part + (((long long)w.g * p.bn_bpg + blk) * p.Cout + w.cb * 64 + 4 * lane_o) * 2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dst[2 * e] = (double)(((bnred[lane_o * 8 + e] + bnred[(16 + lane_o) * 8 + e]) + bnred[(32 + lane_o) * 8 + e]) + bnred[(48 + lane_o) * 8 + e]);
                            dst[2 * e + 1] = (double)(((bnred[lane_o * 8 + 4 + e] + bnred[(16 + lane_o) * 8 + 4 + e]) + bnred[(32 + lane_o) * 8 + 4 + e]) + bnred[(48 + lane_o) * 8 + 4 + e]);
                        }
                    }
                }
            }
        }
#ifdef W4_PROF
        W4_STAMP(7);
        if (prof_on) {
            __builtin_amdgcn_s_waitcnt(0);
            W4_STAMP(8);
            if (lane == 0) reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[(nchunk < 16 ? nchunk : 15) * 40 + 39] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0);
            for (int i = lane; i < 16 * 40; i += 64) w4_prof_buf[i] = reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[i];
        }
#endif
        if (!more_items) break;
        item += ISTEP;
        w = wnx;
        rw = rw_nx;
    }
}

template <bool HAS_RES>
static hipError_t launch_wino4(ConvParams p, hipStream_t st) {
    p.tiles_x = (p.Wo + BLK - 1) / BLK;  // blocks, not tiles
    p.tiles_y = (p.Ho + BLK - 1) / BLK;
    p.ty_off = p.tx_off = 0;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) {
        p.ty_off = p.roi_y0 / BLK;
        p.tx_off = p.roi_x0 / BLK;
        p.tiles_y = (p.roi_y1 + BLK - 1) / BLK - p.ty_off;
        p.tiles_x = (p.roi_x1 + BLK - 1) / BLK - p.tx_off;
    }
    const long long nblk = (long long)p.N * p.tiles_x * p.tiles_y;
    const long long items = (long long)p.groups * ((nblk + 1) / 2) * (p.Cout / 64);
    const int stats = p.bn_part == nullptr ? 0 : (p.bst_y ? 2 : 1);
    if (stats && HAS_RES) return hipErrorInvalidValue;
    p.bn_bpg = (int)nblk;
    auto kern = stats == 2 ? conv_wino4_kernel<false, 2> : stats == 1 ? conv_wino4_kernel<false, 1> : conv_wino4_kernel<HAS_RES, 0>;
    static bool attr_done[3][64] = {};
    if (cerb_attr_needed(attr_done[stats])) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + PROF_BYTES);
        if (e != hipSuccess) return e;
    }
    long long grid = 256;  // persistent: one workgroup per CU
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES + PROF_BYTES, st, p);
    return hipGetLastError();
}

#ifdef W4_PROF
extern "C" int cerb_w4_prof_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(w4_prof_buf), sizeof(unsigned long long) * 16 * 40); }
#endif

hipError_t cerb_launch_wino4(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino4<true>(p, st) : launch_wino4<false>(p, st);
}
