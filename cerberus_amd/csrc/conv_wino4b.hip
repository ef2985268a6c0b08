// conv_wino4b.hip -- 3x3 stride-1 convolution (+ folded BN bias, residual, ReLU) as Winograd F(4x4, 3x3) on the gfx950 fp32 matrix
// cores, second work decomposition (cerb_net_set_conv_algo(7)).  Same mathematics and wave ownership as conv_wino4.hip (a wave owns ALL
// 36 positions of 16 output channels on v_mfma_f32_16x16x4_f32, output transform in registers, ONE wave per SIMD), but
//   * item = ONE 16x16-pixel block (16 tiles of 4x4 outputs) x 64 output channels, chunks of 32 input channels: 36 x 4 = 144 accumulator
//     registers (all of them AccVGPRs), which leaves the 256 architectural registers to the input path;
//   * thread = (tile, channel pair of 16): a patch load covers a pixel's 32 channels = one full 128-byte line per 16 lanes (the 16-channel
//     chunks of conv_wino4.hip ask for every line twice, 64 bytes at a time -- the per-CU miss queue, not the latency, is what stalled it);
//   * the raw patch is DOUBLE-BUFFERED in registers: the loads issued during chunk c (two per step, evenly) belong to chunk c + 2 and are
//     transformed during chunk c + 1, so an HBM miss has a whole chunk (9216 matrix-pipe cycles) before anything waits behind it in the
//     in-order vmcnt queue;
//   * weights: two 16-byte loads per position (every weight register feeds ONE matrix instruction: 32 bytes / clock / CU from L2, what
//     conv_wino.hip has always drawn).
// V tile [36][16 tiles][32 ch] = 72 KiB, double-buffered; 16-byte slots XOR-swizzled by the tile index (conflict-free ds_read_b128 /
// ds_write_b64 without padding).  Output rows leave through the free V buffer as whole lines (conv_wino4.hip).
// Reference layers: models/utils/conv_layers.py:24-60 (_ConvLayer) and models/backbone/resnet.py:81-97 (BasicBlock).
#include <stdlib.h>
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int NPOS = 36;
constexpr int NT = 16;                        // tiles per item: one block of 4x4 tiles
constexpr int BLK = 16;                       // a block is 16x16 output pixels
constexpr int CB = 32;                        // input channels per LDS pass
constexpr int V_FLOATS = NPOS * NT * CB;      // one V buffer: 72 KiB
constexpr int LDS_BYTES = 2 * V_FLOATS * 4;   // double-buffered: 144 KiB
constexpr int OPX = 68;                       // output staging: floats per pixel (64 channels + 4: bank skew)
static_assert(256 * OPX + 16 <= V_FLOATS, "a block's outputs are staged in one V buffer");
#ifndef W4B_RP
#define W4B_RP 9
#endif
constexpr int RP = W4B_RP;                    // weight ring, in positions (two 16-byte operands each); 36 % RP == 0
#ifndef W4B_WDP
#define W4B_WDP 4
#endif
constexpr int WDP = W4B_WDP;                  // weight prefetch distance in positions (256 matrix-pipe cycles each)
static_assert(NPOS % RP == 0 && WDP + 2 <= RP && WDP % 2 == 0, "weight ring");
#ifndef W4B_TQ
#define W4B_TQ 1
#endif
constexpr int TQ = W4B_TQ;                    // pair-step at which the NEXT chunk's patch is masked; transformed at TQ+1 .. TQ+12
static_assert(TQ + 12 < 18, "transform schedule");
constexpr int BIAS_XI = 7;                    // A^T[i][1] A[1][j] = 1 for all 16 outputs: the bias enters through position (1, 1)
constexpr int WAVE_W_BYTES = NPOS * 2 * 1024;     // one wave's share of a chunk: 36 positions x 2 KiB
constexpr int CHUNK_W_BYTES = 4 * WAVE_W_BYTES;   // packed weights of one (cout block, 32-channel chunk): 288 KiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
// the input patch: 2 GiB of range, so that a lane offset of 0x80000000 is out of range and the hardware returns zeros (zero padding of the
// image border without a single VALU instruction)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_lim(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef W4B_WAUX
#define W4B_WAUX 0
#endif
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, W4B_WAUX));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 0);
    asm volatile("s_nop 1");  // gfx950 store hazard, see conv_wino.hip buf_store / tests/test_isa_hazard.py
    __builtin_amdgcn_sched_barrier(0);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
}

#ifdef W4_PROF
// developer instrumentation (scripts/dev_w4prof.py): wave 0 of workgroup W4_PROF stamps s_memtime at every pair-step of its second item
__device__ unsigned long long w4_prof_buf[16 * 40];
constexpr int PROF_BYTES = 16 * 40 * 8;
#else
constexpr int PROF_BYTES = 0;
#endif

struct Item {
    int g, cb, n, by, bx;  // group, block of 64 output channels, image, block row / column inside the launch's block grid
};
}  // namespace

// STATS (training forward): BatchNorm statistics partials per block from the output stage (ConvParams::bn_part; see conv_wino4.hip)
// PACKED (maps whose sides are multiples of 4 but not of 16 -- the 28^2 / 56^2 maps of a 448-pixel patch, where whole 16x16 blocks would be 31 %
// padding): an item's 16 tiles are 16 CONSECUTIVE 4x4 tiles of the flattened (image, tile row, tile column) order instead of one 4x4-tile block, so
// only the launch's last item carries empty tiles.  Every tile's arithmetic is the block form's (a matrix-instruction column per tile): same bits.
// The lane's patch offset and edge flags become per-lane values recomputed per item, the output rows leave per tile (4 pixels x 256 bytes).
// STATS: 0 none, 1 training forward (sum, sum of squares of the outputs), 2 training backward (the BatchNorm-backward sums of conv_wino4.hip's STATS 2: ConvParams::bst_*)
template <bool HAS_RES, int STATS, bool PACKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino4b_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float bnred[STATS ? 4 * 16 * 8 : 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's 16 output channels of the item's 64 / its tile row in the input path
    const int m = lane & 15;                                 // MFMA row (cout) / column (tile)
    const int ks = lane >> 4;                                // k-slot

    const int ncb = p.Cout >> 6;
    const int nblk = PACKED ? (p.pk_ntile + NT - 1) / NT : p.N * p.tiles_y * p.tiles_x;  // blocks per group
    const int per_group = nblk * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;  // even (launcher)
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int base_cnt = total / (int)gridDim.x, rem_cnt = total % (int)gridDim.x;
    int item = lb * base_cnt + min(lb, rem_cnt);
#ifndef W4B_CONTIGUOUS  // workgroup lb takes items lb, lb + G, ...: neighbours meet in the XCD's L2 (conv_wino4.hip; -2.6 %)
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

    auto decode = [&](int it) {
        Item w;
        w.g = it / per_group;
        const int L = it - w.g * per_group;
        w.cb = L % ncb;
        const int id = L / ncb;
        if (PACKED) {  // n = the packed block: tiles 16 n .. 16 n + 15 of the group
            w.n = id;
            w.by = w.bx = 0;
            return w;
        }
        w.bx = id % p.tiles_x;
        const int r = id / p.tiles_x;
        w.by = r % p.tiles_y;
        w.n = r / p.tiles_y;
        return w;
    };
    auto oy0 = [&](const Item& b) { return (b.by + p.ty_off) * BLK; };
    auto ox0 = [&](const Item& b) { return (b.bx + p.tx_off) * BLK; };
    auto in_base = [&](const Item& b) {
        // PACKED: the group's tensor, one row + one pixel early, so that a lane's offset of patch element (0, 0) is never negative
        if (PACKED) return reinterpret_cast<const char*>(p.in + b.g * p.in_gs) - (long long)(p.W + 1) * p.Cin * 4;
        return reinterpret_cast<const char*>(p.in + b.g * p.in_gs) + ((((long long)b.n * p.H + (oy0(b) - 1)) * p.W + (ox0(b) - 1)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };
    auto hangs_over = [&](const Item& b) { return !PACKED && (oy0(b) + BLK > p.H || ox0(b) + BLK > p.W); };
    auto edge_bits = [&](const Item& b) {  // 1 top, 2 bottom, 4 left, 8 right
        return (oy0(b) == 0 ? 1 : 0) | (oy0(b) + BLK == p.H ? 2 : 0) | (ox0(b) == 0 ? 4 : 0) | (ox0(b) + BLK == p.W ? 8 : 0);
    };

    // ---- lane invariants ---------------------------------------------------------------------------------------------------
    // input transform: thread = (tile t, channel pair c of the chunk's 16); a wave owns tile row a: 16 lanes = one pixel's 128-byte line
    const int t = tid >> 4, c = tid & 15, tty = t >> 2, ttx = t & 3;
    const unsigned ioff = (unsigned)((((4 * tty) * p.W + 4 * ttx) * p.Cin + 2 * c) * 4);
    const int vw = t * CB + (((c >> 1) ^ (t >> 1)) << 2) + 2 * (c & 1);  // V write position (floats); position xi adds xi*NT*CB
    const int vr = m * CB + ((ks ^ (m >> 1)) << 2);                      // V read position of channel group G = 0; G = 1: XOR 16 floats
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;

    f32x2 d[2][6][6];  // raw patches (two channels) of the chunks with even / odd index: one is being loaded while the other is transformed
    const bool lane_left = (ttx == 0), lane_right = (ttx == 3);
    // Zero padding at the image border by ADDRESS: the lane offset of a patch element that lies outside the image is out of the buffer's
    // range.  Nine offsets (patch row first / inner / last x column first / inner / last) are derived from the block's edge bits once per
    // chunk; the former in-register masking cost ~100 VALU instructions per chunk (selects + the copies that merged its two code paths),
    // each of them matrix-pipe time at one wave per SIMD.  Blocks hanging over the image keep the per-pixel mask (mask_border).
    struct EdgeOff { unsigned o[3][3]; };
    // tile index of the group -> (image, tile row, tile column); quotients by float reciprocal + one correction step (tile counts are far below 2^24)
    const float pk_inv_img = PACKED ? 1.0f / (float)(p.pk_ty * p.pk_tx) : 0.f, pk_inv_tx = PACKED ? 1.0f / (float)p.pk_tx : 0.f;
    auto pk_div = [&](int x, int dv, float inv) __attribute__((always_inline)) {
        int q = (int)((float)x * inv);
        const int r = x - q * dv;
        q += (r >= dv) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    };
    auto pk_decode = [&](int T, int& n, int& ty, int& tx) __attribute__((always_inline)) {
        const int per_img = p.pk_ty * p.pk_tx;
        n = pk_div(T, per_img, pk_inv_img);
        const int r = T - n * per_img;
        ty = pk_div(r, p.pk_tx, pk_inv_tx);
        tx = r - ty * p.pk_tx;
    };
    // PACKED: the lane's tile is tile 16 n + t of the group: its patch offset (from in_base) and its own edge flags; an empty tile's offset is out of range
    struct LaneGeo { unsigned ioff; int eb; };
    auto lane_geo = [&](const Item& b) __attribute__((always_inline)) {
        LaneGeo L;
        L.ioff = ioff;
        L.eb = 0;
        if (PACKED) {
            const int T = b.n * NT + t;
            const bool valid = T < p.pk_ntile;
            const int Tc = valid ? T : 0;
            int n, ty, tx;
            pk_decode(Tc, n, ty, tx);
            L.ioff = valid ? (unsigned)(((((n * p.H + 4 * ty) * p.W) + 4 * tx) * p.Cin + 2 * c) * 4) : 0x80000000u;
            L.eb = (ty == 0 ? 1 : 0) | (ty == p.pk_ty - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == p.pk_tx - 1 ? 8 : 0);
        }
        return L;
    };
    auto edge_offsets = [&](int bits, const LaneGeo& lg) __attribute__((always_inline)) {
        EdgeOff e;
        const bool zt = PACKED ? (lg.eb & 1) != 0 : ((bits & 1) && a == 0), zb = PACKED ? (lg.eb & 2) != 0 : ((bits & 2) && a == 3);  // block form: wave-uniform, a wave is one tile row
        const bool zl = PACKED ? (lg.eb & 4) != 0 : ((bits & 4) && lane_left), zr = PACKED ? (lg.eb & 8) != 0 : ((bits & 8) && lane_right);
        const unsigned io = PACKED ? lg.ioff : ioff;
        const unsigned col[3] = {zl ? 0x80000000u : io, io, zr ? 0x80000000u : io};
#pragma unroll
        for (int qc = 0; qc < 3; ++qc) {
            e.o[0][qc] = zt ? 0x80000000u : col[qc];
            e.o[1][qc] = col[qc];
            e.o[2][qc] = zb ? 0x80000000u : col[qc];
        }
        // pinned in registers: left alone, hipcc re-derives the select in front of every load
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int qc = 0; qc < 3; ++qc) asm volatile("" : "+v"(e.o[rc][qc]));
        return e;
    };
    auto issue = [&](int P, __amdgpu_buffer_rsrc_t r, const EdgeOff& e, int chunk_off, int k) __attribute__((always_inline)) {
        const int rr = k / 6, qq = k % 6;
        d[P][rr][qq] = buf_load2(r, e.o[rr == 0 ? 0 : rr == 5 ? 2 : 1][qq == 0 ? 0 : qq == 5 ? 2 : 1], chunk_off + rr * rowb + qq * pixb);
    };
    auto mask_border = [&](int P, const Item& b) __attribute__((always_inline)) {
        const f32x2 z = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int gy = oy0(b) - 1 + 4 * tty + r, gx = ox0(b) - 1 + 4 * ttx + q;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                d[P][r][q] = ok ? d[P][r][q] : z;
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
    auto pass_v = [&](int P, int q) __attribute__((always_inline)) { bt6(d[P][0][q], d[P][1][q], d[P][2][q], d[P][3][q], d[P][4][q], d[P][5][q]); };  // down column q
    auto pass_h = [&](int P, int r) __attribute__((always_inline)) { bt6(d[P][r][0], d[P][r][1], d[P][r][2], d[P][r][3], d[P][r][4], d[P][r][5]); };  // along row r
#ifdef W4B_ABL_NOVW
    f32x2 abl_sink = {0.f, 0.f};
#endif
    auto write_row = [&](int P, int r) __attribute__((always_inline)) {  // patch of parity P -> V buffer P
#ifdef W4B_ABL_NOVW
#pragma unroll
        for (int b = 0; b < 6; ++b) abl_sink += d[P][r][b];
        if (abl_sink[0] == 1.2345e-30f) lds[vw] = abl_sink[1];
#else
#pragma unroll
        for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(lds + P * V_FLOATS + (r * 6 + b) * NT * CB + vw) = d[P][r][b];
#endif
    };

    // ---- prologue: chunk 0 of the first item into V buffer 0, chunk 1's patch requested ------------------------------------------
    Item w = decode(item);
    {
        const LaneGeo geo_cur = lane_geo(w);
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc_lim(in_base(w));
        const EdgeOff e0 = edge_offsets(hangs_over(w) ? 0 : edge_bits(w), geo_cur);
#pragma unroll
        for (int k = 0; k < 36; ++k) issue(0, r0, e0, 0, k);
#pragma unroll
        for (int k = 0; k < 36; ++k) issue(1, r0, e0, CB * 4, k);
        if (hangs_over(w)) mask_border(0, w);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) pass_v(0, q);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        pass_h(0, r);
        write_row(0, r);
    }
    __syncthreads();
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    // Weight stream: while position xi is multiplied, the two operands of position xi + WDP are requested; the operands of position xi
    // live in slot xi % RP.  Positions past the chunk's 36 belong to the next chunk, or to the next item's first chunk.
    f32x4 wq[RP][2];
#pragma unroll
    for (int dd = 0; dd < WDP; ++dd) {
        wq[dd][0] = buf_load(rw, wlane, dd * 2048);
        wq[dd][1] = buf_load(rw, wlane, dd * 2048 + 1024);
    }
    f32x4 bnext;
    auto load_bias = [&](const Item& wi) {
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64 + 16 * a;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, 64, 0x00020000);
        bnext = buf_load(rb, (unsigned)ks * 16u, 0);
    };
    load_bias(w);

    for (;;) {
        f32x4 acc[NPOS];
        const bool more_items = item + ISTEP < item_end;
        const Item wnx = more_items ? decode(item + ISTEP) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const bool mask_cur = hangs_over(w), mask_next = hangs_over(wnx);
        const int edge_next = edge_bits(wnx), edge_cur = edge_bits(w);
        const char* in_cur = in_base(w);
        const char* in_nx = in_base(wnx);
        acc[BIAS_XI] = bnext;
#ifdef W4_PROF
        const bool prof_on = (blockIdx.x == W4_PROF) && a == 0 && (item == prof_item);
#define W4_STAMP(k) do { if (prof_on && lane == 0) reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[15 * 40 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4_STAMP(k) do { } while (0)
#endif

        // chunk ch (parity P = ch & 1) multiplies V buffer P; meanwhile the patch of chunk ch + 1 (registers d[P ^ 1], requested during
        // chunk ch - 1) is transformed into V buffer P ^ 1, and the patch of chunk ch + 2 is requested into d[P]
        auto chunk = [&](auto first_tag, auto par_tag, int ch) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int P = decltype(par_tag)::value, Q = P ^ 1;
            const bool a_next = (ch + 1 >= nchunk);           // the chunk being transformed belongs to the next item
            const bool b_next = (ch + 2 >= nchunk);           // the chunk being requested belongs to the next item
            const bool mask_a = a_next ? mask_next : mask_cur;
#ifdef W4B_ABL_PATCHHOT
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc_lim(reinterpret_cast<const char*>(p.in) + (blockIdx.x & 7) * 65536);  // cache-resident
            const int stage_off = 0;
#else
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc_lim(b_next ? in_nx : in_cur);
            const EdgeOff eB = edge_offsets((b_next ? mask_next : mask_cur) ? 0 : (b_next ? edge_next : edge_cur), lane_geo(b_next ? wnx : w));  // PACKED: recomputed per chunk (~30 instructions) rather than held in two more registers across the matrix loop
            const int stage_off = (b_next ? ch + 2 - nchunk : ch + 2) * (CB * 4);
#endif
            const int wcur_off = ch * CHUNK_W_BYTES;
            const bool last_ch = (ch == nchunk - 1);
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;
            const float* vsrc = lds + P * V_FLOATS + vr;

            f32x4 bb[2][2][2];  // B operands of the position pair s in bb[s & 1][position][channel group]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bb[0][u][0] = *reinterpret_cast<const f32x4*>(vsrc + u * NT * CB);
                bb[0][u][1] = *reinterpret_cast<const f32x4*>((vsrc + u * NT * CB) + ((vr & 16) ? -16 : 16));
            }
#pragma unroll
            for (int s = 0; s < NPOS / 2; ++s) {
#ifdef W4_PROF
                if (prof_on && ch < 15 && lane == 0) reinterpret_cast<unsigned long long*>(lds + 2 * V_FLOATS)[ch * 40 + s] = __builtin_readcyclecounter();
#endif
                // weights of positions 2 s + WDP, 2 s + 1 + WDP
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int dd = 2 * s + u + WDP;
                    if (dd < NPOS) {
                        wq[dd % RP][0] = buf_load(rw, wlane, wcur_off + dd * 2048);
                        wq[dd % RP][1] = buf_load(rw, wlane, wcur_off + dd * 2048 + 1024);
                    } else {
                        wq[(dd - NPOS) % RP][0] = buf_load(rw_over, wlane, wover_off + (dd - NPOS) * 2048);
                        wq[(dd - NPOS) % RP][1] = buf_load(rw_over, wlane, wover_off + (dd - NPOS) * 2048 + 1024);
                    }
                }
                if (s + 1 < NPOS / 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float* src = vsrc + (2 * s + 2 + u) * NT * CB;
                        bb[(s + 1) & 1][u][0] = *reinterpret_cast<const f32x4*>(src);
                        bb[(s + 1) & 1][u][1] = *reinterpret_cast<const f32x4*>(src + ((vr & 16) ? -16 : 16));
                    }
                }
#ifndef W4B_ABL_NOPATCH
                issue(P, r_stage, eB, stage_off, 2 * s);
#ifndef W4B_ABL_HALFLOADS
                issue(P, r_stage, eB, stage_off, 2 * s + 1);
#endif
#endif
                // the patch of chunk ch + 1 landed long ago: mask, B^T d B (one 1-D pass per step) and the V writes into the OTHER buffer
                if (s == TQ) {
                    if (mask_a) mask_border(Q, a_next ? wnx : w);
                }
#ifndef W4B_ABL_NOXF
#ifndef W4B_ABL_XFLITE
                if (s > TQ && s <= TQ + 6) pass_v(Q, s - TQ - 1);
#endif
                if (s > TQ + 6 && s <= TQ + 12) {
#ifndef W4B_ABL_XFLITE
                    pass_h(Q, s - TQ - 7);
#endif
                    write_row(Q, s - TQ - 7);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int G = 0; G < 2; ++G)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int xi = 2 * s + u;
                            const float av = wq[xi % RP][G][tt], bv = bb[s & 1][u][G][tt];
                            if (FIRST && G == 0 && tt == 0 && xi != BIAS_XI) {
                                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, z, 0, 0, 0);
                            } else {
                                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[xi], 0, 0, 0);
                            }
                        }
            }
            __syncthreads();  // everybody has read this chunk's V and written the next one's
        };
        chunk(std::true_type{}, std::integral_constant<int, 0>{}, 0);
        chunk(std::false_type{}, std::integral_constant<int, 1>{}, 1);
        for (int ch = 2; ch < nchunk; ch += 2) {
            chunk(std::false_type{}, std::integral_constant<int, 0>{}, ch);
            chunk(std::false_type{}, std::integral_constant<int, 1>{}, ch + 1);
        }

        // ---- output transform A^T M A in registers; the block's rows leave through V buffer 1 (the last chunk's, free now) as whole lines
        W4_STAMP(0);
        load_bias(wnx);  // before this item's stores enter the in-order vmcnt queue (the next item's first WDP positions already went out)
        {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));  // recomputed per item: keeps these out of the MFMA phase's register budget
            const int mo = lane_o & 15, kso = lane_o >> 4;
            float* stg = lds + V_FLOATS;
            // write side: lane (tile mo, channel quad kso) owns pixels (4 ty + i, 4 tx + j); pixel stride 68 floats, 4 floats of skew per tile row
            const int sw = ((64 * (mo >> 2) + 4 * (mo & 3)) * OPX + 4 * (mo >> 2) + 16 * a + 4 * kso);
            // read side: wave a stores pixel rows 4 a .. 4 a + 3; lane = (pixel lane_o >> 4 of a group of four, 16-byte piece lane_o & 15)
            const int sr = ((64 * a + (lane_o >> 4)) * OPX + 4 * a + 4 * (lane_o & 15));
            const int orow = p.Wo * p.Cout * 4, opix = p.Cout * 4;
            const unsigned ooff = (unsigned)((((PACKED ? 0 : 4 * a * p.Wo) + (lane_o >> 4)) * p.Cout + 4 * (lane_o & 15)) * 4);
            const float floor_ = p.relu ? 0.f : -3.402823466e38f;
            const unsigned span = PACKED ? (unsigned)((long long)p.N * p.Ho * p.Wo * p.Cout * 4) : (unsigned)(BLK * p.Wo * p.Cout * 4);
            const int by0 = oy0(w), bx0 = ox0(w);
            const long long origin = PACKED ? (long long)w.cb * 64 : (((long long)w.n * p.Ho + by0) * p.Wo + bx0) * p.Cout + w.cb * 64;  // floats, uniform
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + w.g * p.out_gs + origin, 0, span, 0x00020000);
            const bool partial = !PACKED && ((by0 + BLK > p.Ho) || (bx0 + BLK > p.Wo));
            // PACKED: this wave stores tiles 4 a .. 4 a + 3 of the item; a tile's byte offset in the group's tensor joins the lane's own offset
            unsigned toff[4] = {0u, 0u, 0u, 0u};
            bool tvalid[4] = {true, true, true, true};
            if (PACKED) {
#pragma unroll
                for (int x4 = 0; x4 < 4; ++x4) {
                    const int T = w.n * NT + 4 * a + x4;
                    tvalid[x4] = T < p.pk_ntile;
                    int n, ty, tx;
                    pk_decode(tvalid[x4] ? T : 0, n, ty, tx);
                    toff[x4] = (unsigned)(((n * p.Ho + 4 * ty) * p.Wo + 4 * tx) * p.Cout * 4);
                }
            }
            // vertical pass: T[i][b] = sum_a A^T[i][a] M[a][b]
            f32x4 T[4][6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const f32x4 m0 = acc[0 * 6 + b], m1 = acc[1 * 6 + b], m2 = acc[2 * 6 + b], m3 = acc[3 * 6 + b], m4 = acc[4 * 6 + b], m5 = acc[5 * 6 + b];
                const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                T[0][b] = m0 + s1 + s2;
                T[1][b] = d1 + 2.f * d2;
                T[2][b] = s1 + 4.f * s2;
                T[3][b] = (d1 + 8.f * d2) + m5;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 s1 = T[i][1] + T[i][2], d1 = T[i][1] - T[i][2], s2 = T[i][3] + T[i][4], d2 = T[i][3] - T[i][4];
                *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 0) * OPX) = T[i][0] + s1 + s2;
                *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 1) * OPX) = d1 + 2.f * d2;
                *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 2) * OPX) = s1 + 4.f * s2;
                *reinterpret_cast<f32x4*>(stg + sw + (16 * i + 3) * OPX) = (d1 + 8.f * d2) + T[i][5];
            }
            W4_STAMP(1);
            __syncthreads();
            W4_STAMP(2);
            // 16 groups of four pixels per wave: row 4 a + (k >> 2), pixels 4 (k & 3) .. + 3
            unsigned vo[4];
#pragma unroll
            for (int x4 = 0; x4 < 4; ++x4) {
                const bool ok = PACKED ? tvalid[x4] : (!partial || (bx0 + 4 * x4 + (lane_o >> 4) < p.Wo));
                vo[x4] = ok ? ooff + toff[x4] : 0x80000000u;  // out-of-range offsets: the hardware drops the store / returns 0
            }
            f32x4 res[16];
            if (HAS_RES) {
                const __amdgpu_buffer_rsrc_t r_res =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + w.g * p.resid_gs + origin), 0, span, 0x00020000);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const bool rowok = !partial || (by0 + 4 * a + (k >> 2) < p.Ho);
                    res[k] = buf_load(r_res, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + (PACKED ? 0 : 4 * (k & 3) * opix));
                }
            }
            f32x4 yv[STATS == 2 ? 16 : 1], bm, brs, bga, bbe;  // STATS 2: the BatchNorm's input at this lane's pixels, its parameters for this lane's four channels
            if constexpr (STATS == 2) {
                const __amdgpu_buffer_rsrc_t r_y =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bst_y + w.g * p.bst_y_gs + origin), 0, span, 0x00020000);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const bool rowok = !partial || (by0 + 4 * a + (k >> 2) < p.Ho);
                    yv[k] = buf_load(r_y, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + (PACKED ? 0 : 4 * (k & 3) * opix));
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
                    if (rowok && vo[k & 3] != 0x80000000u) {
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
                        for (int e = 0; e < 4; ++e) {  // the mask by the ONE expression every BatchNorm kernel uses (train_kernels.hip: bn_out)
                            const float yy = yv[k][e];
                            const float z = __fmaf_rn(yy - bm[e], brs[e] * bga[e], bbe[e]);
                            const float g = z > 0.f ? o[e] : 0.f;
                            bts[e] += g;
                            btq[e] = fmaf(g, (yy - bm[e]) * brs[e], btq[e]);
                        }
                    }
                }
                buf_store(o, r_out, rowok ? vo[k & 3] : 0x80000000u, (k >> 2) * orow + (PACKED ? 0 : 4 * (k & 3) * opix));
            }
            if constexpr (STATS) {
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
            W4_STAMP(3);
            __syncthreads();  // the staging buffer is V buffer 1: the next item's first chunk writes it
            W4_STAMP(4);
            if constexpr (STATS) {
                if (a == 0 && lane_o < 16 && p.bn_part) {
                    const long long blk = PACKED ? (long long)w.n : ((long long)w.n * p.tiles_y + w.by) * p.tiles_x + w.bx;
                    double* dst = p.bn_part + (((long long)w.g * p.bn_bpg + blk) * p.Cout + w.cb * 64 + 4 * lane_o) * 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dst[2 * e] = (double)(((bnred[lane_o * 8 + e] + bnred[(16 + lane_o) * 8 + e]) + bnred[(32 + lane_o) * 8 + e]) + bnred[(48 + lane_o) * 8 + e]);
                        dst[2 * e + 1] = (double)(((bnred[lane_o * 8 + 4 + e] + bnred[(16 + lane_o) * 8 + 4 + e]) + bnred[(32 + lane_o) * 8 + 4 + e]) + bnred[(48 + lane_o) * 8 + 4 + e]);
                    }
                }
            }
        }
#ifdef W4_PROF
        if (prof_on) {
            __builtin_amdgcn_s_waitcnt(0);
            W4_STAMP(5);
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

// Packed items: maps whose sides are multiples of 4 but not both of 16, whole map (no region of interest), tensors below 2 GiB per group (a lane's
// offset is 32 bits from the group's base).  cerb_net_set_packed_items(net, 0) / CERB_W4B_PACKED=0 keep the block form (A/B: bitwise the same results).
bool cerb_wino4b_packed(const ConvParams& p) {
    static const bool off = [] { const char* e = cerb_dev_getenv("CERB_W4B_PACKED"); return e && e[0] == '0'; }();
    if (off || p.pk_off || p.H != p.Ho || p.W != p.Wo) return false;
    if (p.Ho % 4 || p.Wo % 4 || (p.Ho % 16 == 0 && p.Wo % 16 == 0)) return false;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) return false;
    const long long px = (long long)p.N * p.Ho * p.Wo;
    return (px + p.W + 1) * p.Cin * 4 < 0x7f000000ll && px * p.Cout * 4 < 0x7f000000ll;  // (the input's base sits one row + one pixel early)
}
// blocks per group of the BatchNorm partials the kernel leaves (ConvParams::bn_part): what cerb_api.hip sizes the buffer by and the finalize kernel walks
int cerb_wino4b_bn_blocks(const ConvParams& p) {
    if (cerb_wino4b_packed(p)) return (p.N * (p.Ho / 4) * (p.Wo / 4) + NT - 1) / NT;
    return p.N * ((p.Ho + BLK - 1) / BLK) * ((p.Wo + BLK - 1) / BLK);
}

template <bool HAS_RES>
static hipError_t launch_wino4b(ConvParams p, hipStream_t st) {
    p.tiles_x = (p.Wo + BLK - 1) / BLK;  // blocks, not tiles
    p.tiles_y = (p.Ho + BLK - 1) / BLK;
    p.ty_off = p.tx_off = 0;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) {
        p.ty_off = p.roi_y0 / BLK;
        p.tx_off = p.roi_x0 / BLK;
        p.tiles_y = (p.roi_y1 + BLK - 1) / BLK - p.ty_off;
        p.tiles_x = (p.roi_x1 + BLK - 1) / BLK - p.tx_off;
    }
    const bool packed = cerb_wino4b_packed(p);
    if (packed) {
        p.pk_ty = p.Ho / 4;
        p.pk_tx = p.Wo / 4;
        p.pk_ntile = p.N * p.pk_ty * p.pk_tx;
    }
    const long long items = packed ? (long long)p.groups * ((p.pk_ntile + NT - 1) / NT) * (p.Cout / 64) : (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    const int stats = p.bn_part == nullptr ? 0 : (p.bst_y ? 2 : 1);
    if (stats && HAS_RES) return hipErrorInvalidValue;
    p.bn_bpg = packed ? cerb_wino4b_bn_blocks(p) : p.N * p.tiles_x * p.tiles_y;
    auto kern = packed ? (stats == 2 ? conv_wino4b_kernel<false, 2, true> : stats == 1 ? conv_wino4b_kernel<false, 1, true> : conv_wino4b_kernel<HAS_RES, 0, true>)
                       : (stats == 2 ? conv_wino4b_kernel<false, 2, false> : stats == 1 ? conv_wino4b_kernel<false, 1, false> : conv_wino4b_kernel<HAS_RES, 0, false>);
    static bool attr_done[6][64] = {};
    if (cerb_attr_needed(attr_done[stats + (packed ? 3 : 0)])) {
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

hipError_t cerb_launch_wino4b(ConvParams p, hipStream_t st) {
    if (p.Cin % (2 * CB) || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino4b<true>(p, st) : launch_wino4b<false>(p, st);
}
