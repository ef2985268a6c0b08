// conv_wino16d.hip -- 3x3 stride-1 convolution (+ folded BN bias, residual, ReLU) as Winograd F(2x2, 3x3) on the gfx950 fp32
// matrix cores, third work decomposition (cerb_net_set_conv_algo(4)): conv_wino16.hip's wave ownership (a wave owns ALL 16
// positions of 16 output channels on v_mfma_f32_16x16x4_f32, output transform in registers, no exchange through LDS) with
//   * 16-channel chunks and a DOUBLE-BUFFERED V tile (2 x 40 KiB = the 80 KiB a workgroup may use at two per CU): a thread writes its
//     transformed patch of chunk c+1 into the other buffer at the end of chunk c's MFMA phase, so a chunk boundary is ONE barrier
//     instead of barrier - 16 LDS writes - barrier;
//   * thread = (tile, channel PAIR): the raw patch is 16 float2 = 32 registers instead of 64, which pays for a weight prefetch
//     distance of 7 steps (1792 cycles with the pipe to itself) without spills -- the in-order vmcnt queue no longer makes a weight
//     wait for an HBM patch load issued just before it;
//   * a step = one position (16 channels): one 16-byte weight load, two ds_read_b128, 8 MFMAs; 16 steps per chunk.
// Reference layers: models/utils/conv_layers.py:24-60 (_ConvLayer) and models/backbone/resnet.py:81-97 (BasicBlock).
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int WTY = 4, WTX = 8;
constexpr int NT = WTY * WTX;                // 32 tiles per item
constexpr int OTH = 2 * WTY, OTW = 2 * WTX;  // 8 x 16 output pixels
constexpr int CB = 16;                       // input channels per LDS pass
constexpr int PS = CB + 4;                   // LDS stride of one tile's channel vector (floats)
constexpr int V_FLOATS = 16 * NT * PS;       // one V buffer: 40 KiB
constexpr int LDS_BYTES = 2 * V_FLOATS * 4;  // double-buffered: 80 KiB -> two workgroups per CU use the whole LDS
constexpr int NS = 16;                       // steps per chunk: the 16 positions
#ifndef W16_WD
#define W16_WD 7
#endif
constexpr int WD = W16_WD;                   // weight prefetch distance in steps (ring of 8 names)
#ifndef W16_NPRE
#define W16_NPRE 8
#endif
constexpr int NPRE = W16_NPRE;               // steps of the NEXT item requested before an item's output stores
#ifndef W16_PL
#define W16_PL 4
#endif
#ifndef W16_WB
#define W16_WB 4
#endif
constexpr int WB = W16_WB;                   // weight burst size in steps (divides 8)
#ifndef W16_TQ
#define W16_TQ 10
#endif
constexpr int TQ = W16_TQ;                   // step at which the next chunk's patch is masked; transformed at TQ+1 .. TQ+4; written at 15
constexpr int PL = W16_PL;                   // patch loads issued per step (over the first 16 / PL steps of a chunk)
static_assert(16 / PL <= TQ && TQ + 4 < 15, "the patch must be requested before its transform starts");
constexpr int CHUNK_W_BYTES = 16 * 4 * 1024;  // packed weights of one (cout block, 16-channel chunk): 64 KiB
constexpr int WAVE_W_BYTES = 16 * 1024;       // one wave's share: 16 steps x 1 KiB
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, 0));
}

struct Item {
    int g, cb, n, oy0, ox0, tx, ty;
};
}  // namespace

template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv_wino16d_kernel(ConvParams p) {
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
    const int t = tid >> 3, c = tid & 7, tty = t >> 3, ttx = t & 7;  // input transform: thread = (tile t, channel pair c)
    const unsigned ioff = (unsigned)((((2 * tty) * p.W + 2 * ttx) * p.Cin + 2 * c) * 4);
    const int vw = t * PS + 2 * c;    // V write position (floats); position xi adds xi*NT*PS
    const int vr = m * PS + 4 * ks;   // V read position for xi = 0, tb = 0; (xi, tb) adds xi*NT*PS + tb*16*PS
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;

    f32x2 d[4][4];  // raw patch of the NEXT chunk (two channels), transformed in place in the shadow of the matrix pipe
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int k) { d[k >> 2][k & 3] = buf_load2(r, ioff, chunk_off + (k >> 2) * rowb + (k & 3) * pixb); };
    const bool lane_top = (tty == 0), lane_bot = (tty == WTY - 1), lane_left = (ttx == 0), lane_right = (ttx == WTX - 1);
    auto mask_edges = [&](int bits) {
        const f32x2 z = {0.f, 0.f};
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
        const f32x2 z = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gy = w.oy0 - 1 + 2 * tty + r, gx = w.ox0 - 1 + 2 * ttx + q;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                d[r][q] = ok ? d[r][q] : z;
            }
    };
    auto bt4 = [&](f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3) {  // (x0, x1, x2, x3) -> (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
        x0 = x0 - x2;
        x3 = x1 - x3;
        const f32x2 o1 = x1;
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
    auto write_v = [&](int buf) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x2*>(lds + buf * V_FLOATS + xi * NT * PS + vw) = d[xi >> 2][xi & 3];
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w));
#pragma unroll
        for (int k = 0; k < 16; ++k) issue(r0, 0, k);
    }
    if (hangs_over(w)) mask_border(w);
    else mask_edges(edge_bits(w));
    transform_rows(0);
    transform_rows(2);
    transform_cols(0);
    transform_cols(2);
    write_v(0);
    int vbuf = 0;  // the buffer the CURRENT chunk reads; the next chunk's patch goes to vbuf ^ 1
    __syncthreads();
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));
    // Weight stream in BURSTS: every WB steps the operands of steps q + 8 .. q + 8 + WB - 1 are requested at once, and the patch
    // loads of the next chunk go out right behind the step-0 burst.  vmcnt retires in order, so a load only delays the waits of
    // loads issued AFTER it: with WB = 4 the HBM-latency patch loads have until the step-4 burst is needed (step 12 = 3072
    // matrix-pipe cycles later) instead of one rolling prefetch distance (7 steps = 1792 cycles; the measured cost of the patch
    // loads there was 14 % of the kernel, W16_ABL_NOPATCH).
    f32x4 wq[16];  // the operand of step q lives in slot q
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) wq[dd] = buf_load(rw, wlane, dd * 1024);
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
            const bool last_ch = (ch == nchunk - 1);
            const Item wp_ = last_ch ? wnx : w;
            const bool mask_nx = last_ch ? mask_next : mask_cur;
            const int edge_nx = last_ch ? edge_next : edge_cur;
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(in_base(wp_));
            const int stage_off = (last_ch ? 0 : ch + 1) * (CB * 4);
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;
            const float* vsrc = lds + vbuf * V_FLOATS + vr;

            f32x4 bb[2][2];  // B operands (tile blocks 0, 1) of step q in bb[q & 1]
            bb[0][0] = *reinterpret_cast<const f32x4*>(vsrc);
            bb[0][1] = *reinterpret_cast<const f32x4*>(vsrc + 16 * PS);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int xi = q;
#ifndef W16_ABL_NOWLOAD
                if (q % WB == 0) {  // burst: the operands of steps q + 8 .. q + 8 + WB - 1 (of this chunk, or of the next one)
#pragma unroll
                    for (int dd = q + 8; dd < q + 8 + WB; ++dd) {
                        if (dd < NS) wq[dd] = buf_load(rw, wlane, wcur_off + dd * 1024);
                        else wq[dd - NS] = buf_load(rw_over, wlane, wover_off + (dd - NS) * 1024);
                    }
                }
#endif
#ifndef W16_ABL_NOLDS
                if (q + 1 < NS) {
                    bb[(q + 1) & 1][0] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * PS);
                    bb[(q + 1) & 1][1] = *reinterpret_cast<const f32x4*>(vsrc + (q + 1) * NT * PS + 16 * PS);
                }
#else
                if (q + 1 < NS) { bb[(q + 1) & 1][0] = bb[q & 1][0]; bb[(q + 1) & 1][1] = bb[q & 1][1]; }
#endif
#ifndef W16_ABL_NOPATCH
                if (q * PL < 16) {  // next chunk's patch: PL loads per step from the start of the chunk
#pragma unroll
                    for (int u = 0; u < PL; ++u) issue(r_stage, stage_off, q * PL + u);
                }
#endif
                // the next chunk's patch landed: mask, B^T d B and the V writes into the OTHER buffer run in the shadow of the pipe
                if (q == TQ) {
                    if (mask_nx) mask_border(wp_);
                    else if (edge_nx) mask_edges(edge_nx);
                }
                if (q == TQ + 1 || q == TQ + 2) transform_rows((q - TQ - 1) * 2);  // one half per step: a two-step version of the same
                if (q == TQ + 3 || q == TQ + 4) transform_cols((q - TQ - 3) * 2);  // work cost 10 % (the VALU burst holds back the MFMAs)
#ifndef W16_ABL_NOVWRITE
                if (q == 15) write_v(vbuf ^ 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 av = wq[q];
                const f32x4 b0 = bb[q & 1][0], b1 = bb[q & 1][1];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    if (FIRST && tt == 0 && xi != 5) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], z, 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], z, 0, 0, 0);
                    } else {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b0[tt], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b1[tt], acc[xi][1], 0, 0, 0);
                    }
                }
            }
#ifndef W16_ABL_NOBAR
            __syncthreads();  // everybody has read this chunk's V and written the next one's
#endif
            vbuf ^= 1;
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
            // the next item's first eight steps were requested at step 8 of this item's last chunk; its bias goes out here, before
            // this item's stores enter the in-order vmcnt queue
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
#ifndef W16_ABL_NOSTORE
                        buf_store(o, r_out, vo[tb][i][jj], (4 * tb + i) * orow + jj * opix);
#else
                        if (o[0] == 1.2345e-30f) buf_store(o, r_out, vo[tb][i][jj], (4 * tb + i) * orow + jj * opix);
#endif
                    }
        }
        if (!more_items) break;
        ++item;
        w = wnx;
        rw = rw_nx;
    }
}

template <bool HAS_RES>
static hipError_t launch_wino16d(ConvParams p, hipStream_t st) {
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
    auto kern = conv_wino16d_kernel<HAS_RES>;
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

hipError_t cerb_launch_wino16d(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino16d<true>(p, st) : launch_wino16d<false>(p, st);
}
