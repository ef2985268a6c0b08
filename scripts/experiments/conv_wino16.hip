// conv_wino16.hip -- 3x3 stride-1 convolution (+ folded BN bias, residual, ReLU) as Winograd F(2x2, 3x3) on the gfx950 fp32
// matrix cores, second work decomposition (cerb_net_set_conv_algo(3); conv_wino.hip is algorithm 1).  Same layers, same math:
//   reference models/utils/conv_layers.py:24-60 (_ConvLayer: Conv2d 3x3 pad 1 -> BatchNorm2d -> ReLU, eval mode) and
//   reference models/backbone/resnet.py:81-97 (BasicBlock conv3x3 + bn (+ identity) + relu)
//
// Why a second decomposition.  In conv_wino.hip wave a owns row a of the 4x4 transformed patch for all 64 output channels, so the
// output transform Y = A^T M A needs the four waves to exchange their partial sums through LDS behind a barrier -- round 1's cycle
// counters put 20 % of a wave's time in that output stage.  Here wave a owns ALL 16 positions for 16 of the item's 64 output
// channels, on v_mfma_f32_16x16x4_f32 (same FLOP rate as the 32x32x2 form):
//   * item = 8 x 16 output pixels (32 Winograd tiles = two 16-column MFMA blocks) x 64 output channels, as in conv_wino.hip;
//   * accumulators: 16 positions x 2 tile blocks x 4 registers = 128 VGPRs -- lane (m = l & 15, ks = l >> 4) holds, for every
//     position, output channels 16 a + 4 ks .. + 3 of tile 16 tb + m: the whole 4x4 of M for its (tile, channel quad), so
//     Y = A^T M A is 24 float4 additions per tile block IN REGISTERS: no LDS exchange, no exchange barrier, and the waves of a
//     workgroup only meet at the two chunk-boundary barriers;
//   * a step = (position, 16-channel group): one 16-byte weight load (A operand of 4 k-steps: k-slot ks at sub-step t <-> channel
//     16 G + 4 ks + t), two ds_read_b128 of V (B operands of the two tile blocks, same permutation), 8 MFMAs = 256 cycles; 32 steps
//     per 32-channel chunk.  Every wave reads the whole V tile (4x the LDS read traffic of conv_wino.hip, ~25 % of the LDS peak);
//   * the weight stream is 4 registers per step instead of 8, so the same register budget holds a prefetch distance of W16_WD = 6
//     steps (1536 cycles with the pipe to itself; conv_wino.hip: 1024);
//   * input transform, V layout [xi][tile][36], persistent XCD-aware item ranges, edge masking, the bias through position (1,1),
//     the requests for the next item's first steps before this item's stores (in-order vmcnt), the s_nop behind SGPR-soffset
//     stores: all as in conv_wino.hip.
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int WTY = 4, WTX = 8;
constexpr int NT = WTY * WTX;                // 32 tiles per item
constexpr int OTH = 2 * WTY, OTW = 2 * WTX;  // 8 x 16 output pixels
constexpr int CB = 32;                       // input channels per LDS pass
constexpr int PS = CB + 4;                   // LDS stride of one tile's channel vector (floats)
constexpr int V_FLOATS = 16 * NT * PS;       // 72 KiB -> two workgroups per CU
constexpr int LDS_BYTES = V_FLOATS * 4;
constexpr int NS = 32;                       // steps per chunk: 16 positions x 2 sixteen-channel groups
#ifndef W16_WD
#define W16_WD 6
#endif
constexpr int WD = W16_WD;                   // weight prefetch distance in steps (ring of 8 names)
#ifndef W16_NPRE
#define W16_NPRE 8
#endif
constexpr int NPRE = W16_NPRE;               // steps of the NEXT item requested before an item's output stores
#ifndef W16_PL
#define W16_PL 1
#endif
constexpr int PL = W16_PL;                   // patch loads issued per step (over the first 16 / PL steps of a chunk)
constexpr int CHUNK_W_BYTES = 16 * 2 * 4 * 1024;  // packed weights of one (cout block, chunk): 128 KiB
constexpr int WAVE_W_BYTES = 16 * 2 * 1024;       // one wave's share: 32 steps x 1 KiB
static_assert(WD >= 1 && WD <= 7, "the slot ring has eight names");

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

struct Item {
    int g, cb, n, oy0, ox0, tx, ty;
};
}  // namespace

template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv_wino16_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's 16 output channels of the item's 64
    const int m = lane & 15;                                 // MFMA row (cout) / column (tile within a 16-tile block)
    const int ks = lane >> 4;                                // k-slot

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
        w.oy0 = (w.ty + p.ty_off) * OTH;
        w.ox0 = (w.tx + p.tx_off) * OTW;
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
        w.oy0 = (w.ty + p.ty_off) * OTH;
        w.ox0 = (w.tx + p.tx_off) * OTW;
        return w;
    };
    auto in_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.in + w.g * p.in_gs) + ((((long long)w.n * p.H + (w.oy0 - 1)) * p.W + (w.ox0 - 1)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };
    auto hangs_over = [&](const Item& w) { return w.oy0 + OTH > p.H || w.ox0 + OTW > p.W; };
    auto edge_bits = [&](const Item& w) {  // 1 top, 2 bottom, 4 left, 8 right
        return (w.oy0 == 0 ? 1 : 0) | (w.oy0 + OTH == p.H ? 2 : 0) | (w.ox0 == 0 ? 4 : 0) | (w.ox0 + OTW == p.W ? 8 : 0);
    };

    // ---- lane invariants ---------------------------------------------------------------------------------------------------
    const int t = tid >> 3, c = tid & 7, tty = t >> 3, ttx = t & 7;  // input transform: thread = (tile t, channel quad c)
    const unsigned ioff = (unsigned)((((2 * tty) * p.W + 2 * ttx) * p.Cin + 4 * c) * 4);
    const int vw = t * PS + 4 * c;    // V write position (floats); position xi adds xi*NT*PS
    const int vr = m * PS + 4 * ks;   // V read position for xi = 0, tb = 0, G = 0; (xi, tb, G) adds xi*NT*PS + tb*16*PS + 16 G
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;

    f32x4 d[4][4];  // raw patch of the NEXT chunk, transformed in place in the shadow of the matrix pipe
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int k) { d[k >> 2][k & 3] = buf_load(r, ioff, chunk_off + (k >> 2) * rowb + (k & 3) * pixb); };
    const bool lane_top = (tty == 0), lane_bot = (tty == WTY - 1), lane_left = (ttx == 0), lane_right = (ttx == WTX - 1);
    auto mask_edges = [&](int bits) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (bits & 3) {
            const bool zt = (bits & 1) && lane_top, zb = (bits & 2) && lane_bot;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[0][q] = zt ? z : d[0][q];
                d[3][q] = zb ? z : d[3][q];
            }
        }
        if (bits & 12) {
            const bool zl = (bits & 4) && lane_left, zr = (bits & 8) && lane_right;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r][0] = zl ? z : d[r][0];
                d[r][3] = zr ? z : d[r][3];
            }
        }
    };
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
    auto bt4 = [&](f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3) {  // (x0, x1, x2, x3) -> (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
        x0 = x0 - x2;
        x3 = x1 - x3;
        const f32x4 o1 = x1;
        x1 = x1 + x2;
        x2 = x2 - o1;
    };
    auto transform_rows = [&](int r0) {
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) bt4(d[r][0], d[r][1], d[r][2], d[r][3]);
    };
    auto transform_cols = [&](int q0) {
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) bt4(d[0][q], d[1][q], d[2][q], d[3][q]);
    };
    auto write_v = [&]() {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(lds + xi * NT * PS + vw) = d[xi >> 2][xi & 3];
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w));
#pragma unroll
        for (int k = 0; k < 16; ++k) issue(r0, 0, k);
    }
    if (!hangs_over(w)) {
        mask_edges(edge_bits(w));
        transform_rows(0);
        transform_rows(2);
        transform_cols(0);
        transform_cols(2);
    }
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    f32x4 wq[8];  // weight ring: the operand of step q lives in slot q & 7
#pragma unroll
    for (int dd = 0; dd < WD; ++dd) wq[dd] = buf_load(rw, wlane, dd * 1024);
    f32x4 wpre[NPRE];  // steps WD .. WD+NPRE-1 of an item's first chunk, requested before the previous item's stores
#pragma unroll
    for (int dd = 0; dd < NPRE; ++dd) wpre[dd] = buf_load(rw, wlane, (WD + dd) * 1024);
    // folded-BN bias through position (1,1) (A^T[i][1] A[1][j] = 1 for all four outputs): its accumulators start at the bias
    f32x4 bnext;
    auto load_bias = [&](const Item& wi) {
        const float* bias = p.bias + wi.g * p.bias_gs + wi.cb * 64 + 16 * a;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, 64, 0x00020000);
        bnext = buf_load(rb, (unsigned)ks * 16u, 0);
    };
    load_bias(w);

    for (;;) {
        f32x4 acc[16][2];
        const bool more_items = item + 1 < item_end;
        const Item wnx = more_items ? advance(w) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const bool mask_cur = hangs_over(w), mask_next = hangs_over(wnx);
        const int edge_next = edge_bits(wnx), edge_cur = edge_bits(w);
        acc[5][0] = bnext;
        acc[5][1] = bnext;

        auto chunk = [&](auto first_tag, int ch) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if (mask_cur) {  // item hanging over the image: per-pixel mask of the raw patch, then transform
                mask_border(w);
                transform_rows(0);
                transform_rows(2);
                transform_cols(0);
                transform_cols(2);
            }
            __syncthreads();  // every wave finished reading the previous chunk's V
            write_v();
            __syncthreads();

            const bool last_ch = (ch == nchunk - 1);
            const Item wp_ = last_ch ? wnx : w;
            const bool mask_nx = last_ch ? mask_next : mask_cur;
            const int edge_nx = last_ch ? edge_next : edge_cur;
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(in_base(wp_));
            const int stage_off = (last_ch ? 0 : ch + 1) * (CB * 4);
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;

            f32x4 bb[2][2];  // B operands (tile blocks 0, 1) of step q in bb[q & 1]
            bb[0][0] = *reinterpret_cast<const f32x4*>(lds + vr);
            bb[0][1] = *reinterpret_cast<const f32x4*>(lds + vr + 16 * PS);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int xi = q >> 1;
                if (FIRST && q < NPRE) {
                    // steps WD .. WD+NPRE-1 of an item's first chunk were requested before the previous item's stores (wpre)
                } else if (q + WD < NS) {
                    wq[(q + WD) & 7] = buf_load(rw, wlane, wcur_off + (q + WD) * 1024);
                } else {
                    wq[(q + WD) & 7] = buf_load(rw_over, wlane, wover_off + (q + WD - NS) * 1024);
                }
                if (q + 1 < NS) {
                    const int vo = vr + ((q + 1) >> 1) * NT * PS + ((q + 1) & 1) * 16;
                    bb[(q + 1) & 1][0] = *reinterpret_cast<const f32x4*>(lds + vo);
                    bb[(q + 1) & 1][1] = *reinterpret_cast<const f32x4*>(lds + vo + 16 * PS);
                }
                if (q * PL < 16) {  // next chunk's patch: PL loads per step from the start of the chunk
#pragma unroll
                    for (int u = 0; u < PL; ++u) issue(r_stage, stage_off, q * PL + u);
                }
                if (!mask_nx) {
                    if (q == 22 && edge_nx) mask_edges(edge_nx);
                    if (q == 24 || q == 26) transform_rows(q - 24);
                    if (q == 28 || q == 30) transform_cols(q - 28);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool pre = FIRST && q >= WD && q < WD + NPRE;  // compile-time after unrolling
                const f32x4 av = pre ? wpre[pre ? q - WD : 0] : wq[q & 7];
                const f32x4 b0 = bb[q & 1][0], b1 = bb[q & 1][1];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    if (FIRST && (q & 1) == 0 && tt == 0 && xi != 5) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], z, 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], z, 0, 0, 0);
                    } else {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], acc[xi][1], 0, 0, 0);
                    }
                }
            }
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform, entirely in registers -----------------------------------------------------------------------------
        {
            f32x4 y[2][2][2];  // [tile block][i][j]
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                f32x4 T0[4], T1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    T0[b] = acc[0 + b][tb] + acc[4 + b][tb] + acc[8 + b][tb];
                    T1[b] = acc[4 + b][tb] - acc[8 + b][tb] - acc[12 + b][tb];
                }
                y[tb][0][0] = T0[0] + T0[1] + T0[2];
                y[tb][0][1] = T0[1] - T0[2] - T0[3];
                y[tb][1][0] = T1[0] + T1[1] + T1[2];
                y[tb][1][1] = T1[1] - T1[2] - T1[3];
            }
            // acc is dead: request what the next item's first steps need before this item's stores enter the vmcnt queue
#pragma unroll
            for (int dd = 0; dd < NPRE; ++dd) wpre[dd] = buf_load(rw_nx, wlane, (WD + dd) * 1024);
            load_bias(wnx);
            const long long origin = (((long long)w.n * p.Ho + w.oy0) * p.Wo + w.ox0) * p.Cout + w.cb * 64 + 16 * a;  // floats, uniform
            const unsigned span = (unsigned)(OTH * p.Wo * p.Cout * 4);
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + w.g * p.out_gs + origin, 0, span, 0x00020000);
            // lane (m, ks): tile m of each tile block -> tile row m >> 3 (+ 2 tb), tile column m & 7; channels 4 ks .. + 3 of the wave's 16
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));  // recomputed per item: keeps these out of the MFMA phase's register budget
            const int mo = lane_o & 15, kso = lane_o >> 4;
            const int ry = 2 * (mo >> 3), rx = 2 * (mo & 7);  // pixel of output (i, j) = (0, 0) inside the item, tile block 0
            const unsigned ooff = (unsigned)(((ry * p.Wo + rx) * p.Cout + 4 * kso) * 4);
            const bool partial = (w.oy0 + OTH > p.Ho) || (w.ox0 + OTW > p.Wo);
            const int orow = p.Wo * p.Cout * 4, opix = p.Cout * 4;
            const float floor_ = p.relu ? 0.f : -3.402823466e38f;
            unsigned vo[2][2][2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const bool ok = !partial || ((w.oy0 + ry + 4 * tb + i < p.Ho) && (w.ox0 + rx + jj < p.Wo));
                        vo[tb][i][jj] = ok ? ooff : 0x80000000u;  // out-of-range offsets: the hardware drops the store / returns 0
                    }
            f32x4 res[2][2][2];
            if (HAS_RES) {
                const __amdgpu_buffer_rsrc_t r_res =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + w.g * p.resid_gs + origin), 0, span, 0x00020000);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) res[tb][i][jj] = buf_load(r_res, vo[tb][i][jj], (4 * tb + i) * orow + jj * opix);
            }
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        f32x4 o = HAS_RES ? y[tb][i][jj] + res[tb][i][jj] : y[tb][i][jj];
                        o[0] = fmaxf(o[0], floor_);
                        o[1] = fmaxf(o[1], floor_);
                        o[2] = fmaxf(o[2], floor_);
                        o[3] = fmaxf(o[3], floor_);
                        buf_store(o, r_out, vo[tb][i][jj], (4 * tb + i) * orow + jj * opix);
                    }
        }
        if (!more_items) break;
        ++item;
        w = wnx;
        rw = rw_nx;
    }
}

template <bool HAS_RES>
static hipError_t launch_wino16(ConvParams p, hipStream_t st) {
    p.tiles_x = (p.Wo + OTW - 1) / OTW;
    p.tiles_y = (p.Ho + OTH - 1) / OTH;
    p.ty_off = p.tx_off = 0;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) {
        p.ty_off = p.roi_y0 / OTH;
        p.tx_off = p.roi_x0 / OTW;
        p.tiles_y = (p.roi_y1 + OTH - 1) / OTH - p.ty_off;
        p.tiles_x = (p.roi_x1 + OTW - 1) / OTW - p.tx_off;
    }
    const long long items = (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    auto kern = conv_wino16_kernel<HAS_RES>;
    static bool attr_done[64] = {};
    if (cerb_attr_needed(attr_done)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    long long grid = 512;  // persistent: two workgroups per CU
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
    return hipGetLastError();
}

hipError_t cerb_launch_wino16(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino16<true>(p, st) : launch_wino16<false>(p, st);
}
