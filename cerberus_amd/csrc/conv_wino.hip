// conv_wino.hip -- 3x3 stride-1 convolution (+ folded BN bias, residual, ReLU) as Winograd F(2x2, 3x3) on the gfx950 fp32
// matrix cores.  Replaces, for the layers that dominate the forward pass:
//   reference models/utils/conv_layers.py:24-60 (_ConvLayer: Conv2d 3x3 pad 1 -> BatchNorm2d -> ReLU, eval mode) and
//   reference models/backbone/resnet.py:81-97 (BasicBlock conv3x3 + bn (+ identity) + relu)
//
// Why: fp32 MFMA and fp32 VALU share the SIMD's lanes, so the direct implicit GEMM (conv_igemm.hip) tops out near the
// 157 TFLOP/s matrix roof.  F(2x2,3x3) needs 16 multiplies per 2x2 output patch instead of 36 (2.25x fewer MFMA cycles); the
// transforms are additions only (B^T d B on the input, A^T m A on the output) and cost ~6 % of the MFMA time.  Products are
// exact fp32 FMAs; the only numerical difference to the direct form is the summation order (error stays ~1e-6 relative).
//
//   Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A            d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Work decomposition (one workgroup = 4 waves):
//   * item = 8 x 16 output pixels (4 x 8 Winograd tiles = the 32 columns of an MFMA) x 64 output channels;
//   * wave a owns row a of the 4x4 transformed patch: positions xi = (a, b), b = 0..3.  For each xi it runs the GEMM
//     M_xi[cout 64][tile 32] += U_xi[cout][cin] V_xi[cin][tile] with v_mfma_f32_32x32x2_f32 -> 4 x 2 accumulators (128 VGPRs);
//   * input transform: thread (tile t, channel quad c) loads its 4x4 patch of float4 straight from global memory (per-lane
//     offset is kernel-invariant, the (row, col) part rides in the scalar offset), transforms it in registers and writes the
//     16 V values to LDS [xi][tile][36] -- the B operand is then one ds_read_b128 per 4 MFMA k-steps, as in conv_igemm;
//   * weights are pre-transformed on the host (cerb_api.hip: pack_wino) and streamed from L2 as 1 KiB blocks per wave;
//   * output transform: each wave reduces over b in registers (T_a[j] = sum_b M[a][b] A[b][j]), the four waves exchange
//     T through LDS (64 KiB, reusing the V region) and each finishes one (column parity, cout half) quarter of the outputs:
//     Y[i][j] = sum_a A^T[i][a] T_a[j], + bias (+ residual), ReLU, float4 NHWC stores.
// Zero padding comes from the zero guard band around activation buffers (cerb_api.hip: DevBuf) for rows above/below the whole
// tensor and from an explicit mask (uniform branch, border items only) for everything else.
#include <type_traits>

#include "cerb_common.h"

namespace {
constexpr int WTY = 4, WTX = 8;          // Winograd tiles per workgroup
constexpr int NT = WTY * WTX;            // 32 = N of the MFMA
constexpr int OTH = 2 * WTY, OTW = 2 * WTX;  // output pixels per workgroup: 8 x 16
constexpr int CB = 32;                   // input channels per LDS pass
constexpr int PS = CB + 4;               // LDS stride of one tile's channel vector (floats)
constexpr int V_FLOATS = 16 * NT * PS;   // 18432 floats = 72 KiB -> two workgroups per CU
constexpr int LDS_BYTES = V_FLOATS * 4;
constexpr int NQ = 16;                   // steps per chunk for one wave: 4 positions x 4 eight-channel groups
#ifndef WINO_WD
#define WINO_WD 2
#endif
constexpr int WD = WINO_WD;              // weight prefetch distance in steps
constexpr int NPRE = 4;                  // extra weight steps of the NEXT item requested before an item's output stores
constexpr int CHUNK_W_BYTES = 16 * 4 * 2 * 1024;  // packed weights of one (cout block, chunk): 128 KiB
constexpr int WAVE_W_BYTES = 4 * 4 * 2 * 1024;    // one wave's share of it

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
#ifndef WINO_STORE_AUX
#define WINO_STORE_AUX 0
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, WINO_STORE_AUX);
    // gfx950 hazard hipcc (ROCm 7.2) does not pad: buffer_store_dwordx4 whose soffset is an SGPR, followed directly by a VALU
    // write of its data VGPRs, stores corrupted data (the compiler only inserts wait states for the immediate-soffset form).
    // Found as run-to-run differing outputs; two wait states pinned behind the store cure it (scripts/dev_wrace.sh).
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

struct Item {
    int g, cb, n, oy0, ox0, tx, ty;
};
}  // namespace

#ifdef WPROF
// developer instrumentation (scripts/dev_wprof.sh, dev_wclock.py, dev_wlog.py): cycles per phase summed over wave 0 of every
// workgroup, the same span in 100 MHz real-time ticks, and a log of every workgroup's start / end tick
__device__ unsigned long long g_wprof[8];
__device__ unsigned long long g_wlog[4 * 32768];  // per workgroup: start tick, end tick (100 MHz), (H << 32 | Cin), blockIdx
__device__ unsigned int g_wlog_n;
extern "C" int cerb_dev_wlog(unsigned long long* out, unsigned* n, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wlog), sizeof(g_wlog)) != hipSuccess) return 1;
    if (n && hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wlog_n), 4) != hipSuccess) return 1;
    if (reset) {
        unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wlog_n), &z, 4) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int cerb_dev_wprof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wprof), sizeof(g_wprof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#define WPROF_T() (__builtin_readcyclecounter())
#define WPROF_ACC(k, t0) prof_acc[k] += WPROF_T() - (t0)
#else
#define WPROF_T() 0ull
#define WPROF_ACC(k, t0) ((void)(t0))
#endif

template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef WPROF
    const unsigned long long t_kernel = WPROF_T();
    const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz counter: shader clock = cycles / ticks * 100 MHz
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's row of the transformed patch (scalar)
    const int j = lane & 31;                                 // MFMA column = Winograd tile / row = cout within a half
    const int h = lane >> 5;                                 // k-slot

    // ---- this workgroup's contiguous item range (same scheme as conv_igemm) ------------------------------------------------
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
    auto in_base = [&](const Item& w) {  // top-left input pixel of the item's patch grid (may lie in the guard band)
        return reinterpret_cast<const char*>(p.in + w.g * p.in_gs) + ((((long long)w.n * p.H + (w.oy0 - 1)) * p.W + (w.ox0 - 1)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {  // this wave's slice of the item's weight stream
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs) + (long long)w.cb * nchunk * CHUNK_W_BYTES + a * WAVE_W_BYTES;
    };
    // An item that hangs over the image (H % 8 or W % 16 != 0: only the small odd maps) takes the generic path: per-pixel mask and
    // transform at the chunk boundary.  Every other item is masked by edge: the zero padding is exactly patch row 0 of the top
    // tile row, patch row 3 of the bottom tile row, patch column 0 / 3 of the left / right tile column.
    auto hangs_over = [&](const Item& w) { return w.oy0 + OTH > p.H || w.ox0 + OTW > p.W; };
    auto edge_bits = [&](const Item& w) {  // 1 top, 2 bottom, 4 left, 8 right
        return (w.oy0 == 0 ? 1 : 0) | (w.oy0 + OTH == p.H ? 2 : 0) | (w.ox0 == 0 ? 4 : 0) | (w.ox0 + OTW == p.W ? 8 : 0);
    };

    // ---- lane invariants -----------------------------------------------------------------------------------------------------
    // input transform: thread = (tile t, channel quad c)
    const int t = tid >> 3, c = tid & 7, tty = t >> 3, ttx = t & 7;
    const unsigned ioff = (unsigned)((((2 * tty) * p.W + 2 * ttx) * p.Cin + 4 * c) * 4);
    const int vw = t * PS + 4 * c;                 // V write position (floats); position xi adds xi*NT*PS
    const int vr = (a * 4 * NT + j) * PS + 4 * h;  // V read position for b = 0, G = 0; (b, G) adds b*NT*PS + 8 G
    const unsigned wlane = (unsigned)lane * 16u;
    const int rowb = p.W * p.Cin * 4, pixb = p.Cin * 4;
    // output stage: thread = (column pp of the 16-pixel-wide item, cout quad cq) for all 8 rows -> a wave's store covers 1 KiB
    // (cq, pp, the store offset and the T-exchange positions are recomputed inside the output stage: five lane invariants less
    // to keep in registers through the MFMA phase)
    // T exchange (floats): wave a writes into ITS OWN quarter of the V region (only wave a ever reads V[xi = (a, *)], so no
    // barrier is needed between its last MFMA and its T stores): block (jj, s, rq) at a*VW + (jj*8 + s*4+rq) * TB, lane (j, h)
    // at h*TH + j*4.  The 16-byte skews make both the per-wave writes (16 lanes = 16 tiles) and the remapped reads (16 lanes =
    // 16 cout quads) hit distinct banks.
    constexpr int TB = 264, TH = 132, VW = 4 * NT * PS;
    static_assert(16 * TB <= VW, "a wave's T blocks must fit in its quarter of V");

    f32x4 d[4][4];  // raw patch of the NEXT chunk while the matrix pipe works, transformed in place at the chunk boundary
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int k) { d[k >> 2][k & 3] = buf_load(r, ioff, chunk_off + (k >> 2) * rowb + (k & 3) * pixb); };
    const bool lane_top = (tty == 0), lane_bot = (tty == WTY - 1), lane_left = (ttx == 0), lane_right = (ttx == WTX - 1);
    auto mask_edges = [&](int bits) {  // 48 v_cndmask at most, none for interior items
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
    // B^T d B in place with one float4 of temporaries per 1-D transform: (x0, x1, x2, x3) -> (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
#ifdef WINO_SCALAR_XF
    // scalar v_add / v_sub instead of the v_pk_add_f32 hipcc forms from float4 arithmetic (the microarchitecture guide lists packed
    // f32 VALU beside MFMAs as an anti-lever): experiment switch
    auto sadd = [](float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto ssub = [](float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto bt4 = [&](f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = x0[e], a1 = x1[e], a2 = x2[e], a3 = x3[e];
            x0[e] = ssub(a0, a2);
            x3[e] = ssub(a1, a3);
            x1[e] = sadd(a1, a2);
            x2[e] = ssub(a2, a1);
        }
    };
#else
    auto bt4 = [&](f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3) {
        x0 = x0 - x2;
        x3 = x1 - x3;
        const f32x4 o1 = x1;
        x1 = x1 + x2;
        x2 = x2 - o1;
    };
#endif
    auto transform_rows = [&](int r0) {  // rows r0, r0+1 of d <- d B
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) bt4(d[r][0], d[r][1], d[r][2], d[r][3]);
    };
    auto transform_cols = [&](int q0) {  // columns q0, q0+1 of d <- B^T d
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) bt4(d[0][q], d[1][q], d[2][q], d[3][q]);
    };
    auto write_v = [&]() {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(lds + xi * NT * PS + vw) = d[xi >> 2][xi & 3];
    };

    // ---- prologue --------------------------------------------------------------------------------------------------------------
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
    // weight stream window: the operands of step q live in slot q & 7 (WD + 1 slots are live at a time; eight names so that the
    // slot of a step is the same in every chunk -- 16 steps per chunk -- and the unrolled body needs no register moves)
    static_assert(WD >= 1 && WD <= 7, "the slot ring has eight names");
    f32x4 wq[8][2];
#pragma unroll
    for (int dd = 0; dd < WD; ++dd)
#pragma unroll
        for (int s = 0; s < 2; ++s) wq[dd][s] = buf_load(rw, wlane, (dd * 2 + s) * 1024);
    // gfx9-family vmcnt retires loads AND stores in issue order: a load issued after an item's output stores cannot be
    // waited for before those stores are acknowledged.  Everything the first steps of the next item need (weights of steps
    // WD .. WD+NPRE-1, the bias) is therefore requested BEFORE the stores, into registers the dead accumulators free up.
    f32x4 wpre[NPRE][2];
#pragma unroll
    for (int dd = 0; dd < NPRE; ++dd)
#pragma unroll
        for (int s = 0; s < 2; ++s) wpre[dd][s] = buf_load(rw, wlane, ((WD + dd) * 2 + s) * 1024);
    // The folded-BN bias enters through position (1,1): A^T[i][1] * A[1][j] = 1 for all four outputs, so the accumulator of
    // xi = (1,1) starts at the bias and every other accumulator starts at the MFMA's constant-zero C operand.  Waves a != 1
    // read through a zero-length buffer descriptor (out-of-range buffer loads return 0).
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

#ifdef WPROF
    unsigned long long prof_acc[4] = {0, 0, 0, 0};
    const unsigned long long t_first = WPROF_T();
#endif
    for (;;) {
        f32x16 acc[4][2];
        const bool more_items = item + 1 < item_end;
        const Item wnx = more_items ? advance(w) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const bool mask_cur = hangs_over(w), mask_next = hangs_over(wnx);
        const int edge_next = edge_bits(wnx), edge_cur = edge_bits(w);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[1][s][rq * 4 + e] = bnext[s][rq][e];

        auto chunk = [&](auto first_tag, int ch) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const unsigned long long t_x = WPROF_T();
            if (mask_cur) {  // item hanging over the image: per-pixel mask of the raw patch, then transform
                mask_border(w);
                transform_rows(0);
                transform_rows(2);
                transform_cols(0);
                transform_cols(2);
            }
            WPROF_ACC(3, t_x);
            const unsigned long long t_b = WPROF_T();
            __syncthreads();  // every wave finished reading the previous chunk's V (or the previous item's T exchange)
            write_v();
            __syncthreads();
            WPROF_ACC(0, t_b);
            const unsigned long long t_m = WPROF_T();

            const bool last_ch = (ch == nchunk - 1);
            const Item wp_ = last_ch ? wnx : w;
            const bool mask_nx = last_ch ? mask_next : mask_cur;
            const int edge_nx = last_ch ? edge_next : edge_cur;
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(in_base(wp_));
            const int stage_off = (last_ch ? 0 : ch + 1) * (CB * 4);
            const int wcur_off = ch * CHUNK_W_BYTES;
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * CHUNK_W_BYTES;

#ifdef WINO_BB1
            f32x4 bb[1];  // B operand of the current step; the next one is read right behind the step's MFMAs (they hold their operands)
#define WINO_BBI(q) 0
#else
            f32x4 bb[2];  // B operand of step q in bb[q & 1]
#define WINO_BBI(q) ((q) & 1)
#endif
            bb[0] = *reinterpret_cast<const f32x4*>(lds + vr);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    const int q = b * 4 + G;
                    if (FIRST && q < NPRE) {
                        // steps WD .. WD+NPRE-1 of an item's first chunk were requested before the previous item's stores (wpre)
                    } else if (q + WD < NQ) {
                        wq[(q + WD) & 7][0] = buf_load(rw, wlane, wcur_off + ((q + WD) * 2 + 0) * 1024);
                        wq[(q + WD) & 7][1] = buf_load(rw, wlane, wcur_off + ((q + WD) * 2 + 1) * 1024);
                    } else {
                        wq[(q + WD) & 7][0] = buf_load(rw_over, wlane, wover_off + ((q + WD - NQ) * 2 + 0) * 1024);
                        wq[(q + WD) & 7][1] = buf_load(rw_over, wlane, wover_off + ((q + WD - NQ) * 2 + 1) * 1024);
                    }
#ifndef WINO_BB1
                    if (q + 1 < NQ) bb[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(lds + vr + ((q + 1) >> 2) * NT * PS + ((q + 1) & 3) * 8);
#endif
                    if (q < 8) {  // next chunk's patch: two loads per step, all in flight half a chunk before the transform
                        issue(r_stage, stage_off, 2 * q);
                        issue(r_stage, stage_off, 2 * q + 1);
                    }
                    // the next chunk's patch landed (requested in steps 0..7): B^T d B runs here, in the shadow of the matrix pipe,
                    // so that the chunk boundary is only barrier - 16 LDS writes - barrier
                    // (items hanging over the image keep the raw patch: they are masked and transformed at the boundary instead)
                    if (!mask_nx) {
                        if (q == 11 && edge_nx) mask_edges(edge_nx);
                        if (q == 12 || q == 13) transform_rows((q - 12) * 2);
                        if (q == 14 || q == 15) transform_cols((q - 14) * 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bool pre = FIRST && q >= WD && q < WD + NPRE;  // compile-time after unrolling
                    const f32x4 a0 = pre ? wpre[pre ? q - WD : 0][0] : wq[q & 7][0], a1 = pre ? wpre[pre ? q - WD : 0][1] : wq[q & 7][1];
                    const f32x4 bq = bb[WINO_BBI(q)];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        if (FIRST && G == 0 && tt == 0 && b != 1) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[tt], bq[tt], z, 0, 0, 0);
                            acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[tt], bq[tt], z, 0, 0, 0);
                        } else {
                            acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[tt], bq[tt], acc[b][0], 0, 0, 0);
                            acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[tt], bq[tt], acc[b][1], 0, 0, 0);
                        }
                    }
#ifdef WINO_BB1
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 1 < NQ) bb[0] = *reinterpret_cast<const f32x4*>(lds + vr + ((q + 1) >> 2) * NT * PS + ((q + 1) & 3) * 8);
#endif
                }
            }
            WPROF_ACC(1, t_m);
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);

        // ---- output transform ---------------------------------------------------------------------------------------------------
        const unsigned long long t_e = WPROF_T();
        {
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));  // keeps the compiler from hoisting these out of the item loop
            const int cq = tid_o & 15, pp = tid_o >> 4;
            const unsigned ooff = (unsigned)((pp * p.Cout + 4 * cq) * 4);
            const int tw = a * VW + ((tid_o >> 5) & 1) * TH + (tid_o & 31) * 4;
            const int tr = ((pp & 1) * 8 + (cq >> 1)) * TB + (cq & 1) * TH + (pp >> 1) * 4;  // + aa*VW + k*32
            // over b, in registers: T[0] = M0 + M1 + M2, T[1] = M1 - M2 - M3
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x16 T0 = acc[0][s] + acc[1][s] + acc[2][s];
                const f32x16 T1 = acc[1][s] - acc[2][s] - acc[3][s];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 v0 = {T0[rq * 4 + 0], T0[rq * 4 + 1], T0[rq * 4 + 2], T0[rq * 4 + 3]};
                    const f32x4 v1 = {T1[rq * 4 + 0], T1[rq * 4 + 1], T1[rq * 4 + 2], T1[rq * 4 + 3]};
                    *reinterpret_cast<f32x4*>(lds + (0 * 8 + s * 4 + rq) * TB + tw) = v0;
                    *reinterpret_cast<f32x4*>(lds + (1 * 8 + s * 4 + rq) * TB + tw) = v1;
                }
            }
            // acc is dead: request what the next item's first steps need before this item's stores enter the vmcnt queue
#pragma unroll
            for (int dd = 0; dd < NPRE; ++dd)
#pragma unroll
                for (int s = 0; s < 2; ++s) wpre[dd][s] = buf_load(rw_nx, wlane, ((WD + dd) * 2 + s) * 1024);
            load_bias(wnx);
            const long long origin = (((long long)w.n * p.Ho + w.oy0) * p.Wo + w.ox0) * p.Cout + w.cb * 64;  // floats, uniform
            // Branch-free output stage: every lane issues every load / store.  hipcc's s_waitcnt insertion merges control-flow
            // paths conservatively, so a store inside a branch makes every later wait on an OLDER load a vmcnt(0) -- i.e. a wait for
            // the stores.  Pixels outside the image get an offset past the descriptor's range instead (the hardware drops
            // out-of-range buffer stores and returns 0 for out-of-range loads).
            const unsigned span = (unsigned)(OTH * p.Wo * p.Cout * 4);  // bytes from the item origin to past its last row
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + w.g * p.out_gs + origin, 0, span, 0x00020000);
            const bool partial = (w.oy0 + OTH > p.Ho) || (w.ox0 + OTW > p.Wo);
            const bool col_ok = !partial || (w.ox0 + pp < p.Wo);
            const int rows_ok = partial ? p.Ho - w.oy0 : OTH;  // uniform: output rows of this item inside the image
            const unsigned ocol = col_ok ? ooff : 0x80000000u;
            const int orow = p.Wo * p.Cout * 4;
            const float floor_ = p.relu ? 0.f : -3.402823466e38f;
            // the residual (BasicBlock identity) is requested here, before the exchange barrier, so that its latency hides behind
            // the barrier and the LDS reads -- 32 registers the dead accumulators leave free
            f32x4 res[8];
            if (HAS_RES) {
                const __amdgpu_buffer_rsrc_t r_res =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + w.g * p.resid_gs + origin), 0, span, 0x00020000);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) res[r8] = buf_load(r_res, (r8 < rows_ok) ? ocol : 0x80000000u, r8 * orow);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // tile row k -> output rows 2k, 2k+1
                f32x4 tq[4];
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) tq[aa] = *reinterpret_cast<const f32x4*>(lds + tr + aa * VW + k * 32);
                f32x4 y[2];
                y[0] = tq[0] + tq[1] + tq[2];
                y[1] = tq[1] - tq[2] - tq[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned vo = (2 * k + i < rows_ok) ? ocol : 0x80000000u;  // scalar compare + one v_cndmask (or none)
                    f32x4 o = HAS_RES ? y[i] + res[2 * k + i] : y[i];
                    o[0] = fmaxf(o[0], floor_);
                    o[1] = fmaxf(o[1], floor_);
                    o[2] = fmaxf(o[2], floor_);
                    o[3] = fmaxf(o[3], floor_);
                    buf_store(o, r_out, vo, (2 * k + i) * orow);
                }
            }
        }
        WPROF_ACC(2, t_e);
        if (!more_items) break;
        ++item;
        w = wnx;
        rw = rw_nx;
    }
#ifdef WPROF
    if (tid == 0) {
        atomicAdd(&g_wprof[0], prof_acc[0]);
        atomicAdd(&g_wprof[1], prof_acc[1]);
        atomicAdd(&g_wprof[2], prof_acc[2]);
        atomicAdd(&g_wprof[5], prof_acc[3]);
        atomicAdd(&g_wprof[6], t_first - t_kernel);
        atomicAdd(&g_wprof[3], WPROF_T() - t_kernel);
        atomicAdd(&g_wprof[4], 1ull);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_wprof[7], t_end - t_real);
        const unsigned slot = atomicAdd(&g_wlog_n, 1u);
        if (slot < 32768u) {
            g_wlog[4 * slot + 0] = t_real;
            g_wlog[4 * slot + 1] = t_end;
            g_wlog[4 * slot + 2] = ((unsigned long long)p.H << 32) | (unsigned)p.Cin;
            g_wlog[4 * slot + 3] = blockIdx.x;
        }
    }
#endif
}

// Host-side launcher (called from cerb_api.hip).  p.wpack must hold the Winograd-packed weights (pack_wino).
template <bool HAS_RES>
static hipError_t launch_wino(ConvParams p, hipStream_t st) {
    p.tiles_x = (p.Wo + OTW - 1) / OTW;
    p.tiles_y = (p.Ho + OTH - 1) / OTH;
    p.ty_off = p.tx_off = 0;
    if (p.roi_y1 > p.roi_y0 && p.roi_x1 > p.roi_x0) {  // only the items overlapping the region of interest
        p.ty_off = p.roi_y0 / OTH;
        p.tx_off = p.roi_x0 / OTW;
        p.tiles_y = (p.roi_y1 + OTH - 1) / OTH - p.ty_off;
        p.tiles_x = (p.roi_x1 + OTW - 1) / OTW - p.tx_off;
    }
    const long long items = (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    auto kern = conv_wino_kernel<HAS_RES>;
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

hipError_t cerb_launch_wino(ConvParams p, hipStream_t st) {
    if (p.Cin % CB || p.Cout % 64) return hipErrorInvalidValue;
    return p.resid ? launch_wino<true>(p, st) : launch_wino<false>(p, st);
}
