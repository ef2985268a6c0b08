// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the Cerberus tile path, every nn.Conv2d + eval-BatchNorm2d + ReLU triple of
//   reference models/backbone/resnet.py:81-97 (BasicBlock, incl. 1x1 stride-2 downsample and the residual add)
//   reference models/utils/conv_layers.py:24-60 (_ConvLayer, post-activation) used by the decoders
//   reference models/net_desc.py:52,153 (conv_map) and :185-188 (upsample2x + skip add, fused as MODE 1)
//
// Data layout: activations NHWC fp32 (every activation buffer of the library has a zero-filled guard band in front and
// behind, so halo reads never need address clamping); weights pre-packed (BN folded) by pack_conv() in cerb_api.hip.
// One workgroup = 256 threads = 4 waves; a work ITEM is a TH x TW output tile (256 pixels) x 64 output channels.
// GEMM is "swapped": D[cout][pixel] += W[cout][k] * X[k][pixel]  (A = weights, B = pixels) so each lane owns ONE pixel
// and 4 consecutive couts per accumulator quad.  K order inside an 8-channel group is permuted so that one 16-byte
// read (LDS for pixels, global for weights) feeds 4 consecutive MFMA k-steps:
//   k-slot h (= lane>>5) at step t  <->  channel  g*8 + 4*h + t.
//
// Pipeline: workgroups are PERSISTENT over a contiguous range of items and software-pipelined in registers -- while
// the matrix pipe works on chunk c (CB input channels x all taps), the global loads of the next chunk (of this item or
// of the next item) are issued one 256-element slice per 8-channel step into VGPRs; after the chunk: barrier,
// ~11 ds_write_b128 per lane, barrier.  MODE 1 (decoder entry) additionally prefetches the half-resolution `prev` tile
// ((TH/2+2) x (TW/2+2) pixels) into a small auxiliary LDS tile mid-chunk and folds bilinear_x2(prev) into the prefetched
// skip values with LDS reads -- 5 global loads per element never exist.
//
// VALU diet.  On gfx950 the fp32 MFMA runs on the SIMD's fp32 lanes: a micro-benchmark (scripts/ubench/
// mfma_valu_coissue.hip) shows every VALU instruction issued next to v_mfma_f32_32x32x2_f32 costs its own issue time in
// matrix throughput (2 v_pk_fma per MFMA: 147 -> 117 TFLOP/s).  So the chunk body contains NO per-element address
// arithmetic: every global access is  uniform base (SGPRs, advanced by the scalar unit)  +  a per-lane 32-bit byte offset
// computed ONCE per kernel  (+ immediate), every LDS access is  per-lane base + immediate,  and image-border handling
// (zero padding) is a uniform branch taken only by items that touch the border.
#include "cerb_common.h"

// Buffer-resource addressing: address = descriptor base (4 SGPRs, built by the scalar unit per item) + per-lane 32-bit byte
// offset (VGPR, kernel-invariant) + uniform byte offset (SGPR) -- no VALU instruction per access.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);  // raw buffer, no range clipping
}
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 0);
    // gfx950 hazard hipcc (ROCm 7.2) does not pad: buffer_store_dwordx4 whose soffset is an SGPR, followed directly by a VALU
    // write of its data VGPRs, stores corrupted data (the compiler only inserts wait states for the immediate-soffset form).
    // Found as run-to-run differing outputs; two wait states pinned behind the store cure it (scripts/dev_wrace.sh).
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ f32x4 splat4(float x) {
    f32x4 r = {x, x, x, x};
    return r;
}

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    // A strided 1x1 convolution only ever reads the pixels its outputs sit on: it stages THOSE (every STRIDE-th pixel of every STRIDE-th row, a
    // TH x TW tile) instead of the whole (TH-1)*STRIDE+1 window -- a quarter of the loads and of the LDS tile at stride 2.
    static constexpr int GSTEP = (KS == 1) ? STRIDE : 1;  // global pixels between neighbouring staged pixels
    static constexpr int LSTEP = (KS == 1) ? 1 : STRIDE;  // staged pixels between neighbouring output pixels
    static constexpr int IH = (KS == 1) ? TH : (TH - 1) * STRIDE + KS;
    static constexpr int IW = (KS == 1) ? TW : (TW - 1) * STRIDE + KS;
    static constexpr int PS = CB + 4;  // LDS pixel stride in floats (16-lane ds_read_b128 groups hit distinct 16-B slots)
    static constexpr int NG = CB / 8;  // 8-channel groups per chunk
    static constexpr int T = KS * KS;  // taps
    static constexpr int NQ = T * NG;  // weight-stream steps per chunk
    static constexpr int PARTS = CB / 4;
    static constexpr int PPS = 256 / PARTS;        // pixels covered by one staging slice
    static constexpr int NF = IH * IW * PARTS;     // float4 elements of one staged chunk
    // MODE 1 stages by rows (see the kernel): NM slices of RPS halo rows x TW columns, NE slices for the 2 right-hand columns
    static constexpr int RPS = 32 / TW;
    static constexpr int NM = (IH + RPS - 1) / RPS;
    static constexpr int NE = (2 * IH * 8 + 255) / 256;
    static constexpr int ITER = (MODE == 1) ? NM + NE : (NF + 255) / 256;  // staging slices per thread
    static constexpr int MAIN_FLOATS = IH * IW * PS;
    // MODE 1: half-resolution tile of `prev` covering every bilinear source of the halo tile
    static constexpr int AR = TH / 2 + 2, AC = TW / 2 + 2;
    static constexpr int NA = AR * AC * PARTS;
    static constexpr int AITER = (NA + 255) / 256;
    static constexpr int AUX_FLOATS = (MODE == 1) ? AR * AC * PS : 0;
    static constexpr int LDS_FLOATS = MAIN_FLOATS + AUX_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    // MODE 1 schedule inside a chunk (in 8-channel steps): skip slices at steps 0..ITER-1, prev slices next, then the
    // auxiliary tile is written + one barrier, then one combine per step
    static constexpr int S_PREV = ITER;
    static constexpr int S_BAR = ITER + AITER + 6;
    static constexpr int S_COMB = S_BAR + 1;
    static constexpr int WD = (MODE == 0 && STRIDE == 1) ? 3 : 2;  // weight prefetch distance in steps (8 VGPRs per step)
};

struct Item {
    int g, cb, n, oy0, ox0, tx, ty;
};

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p) {
    using C = ConvCfg<KS, STRIDE, TH, TW, CB, MODE>;
    constexpr int WD = C::WD;
    static_assert(TH * TW == 256, "tile must hold 256 pixels (4 waves x 2 x 32)");
    static_assert(MODE == 0 || (KS == 3 && STRIDE == 1 && CB == 32 && C::IH % C::RPS == 0 && C::S_COMB + C::ITER <= C::NQ),
                  "MODE 1 schedule must fit in one chunk");
    static_assert(C::ITER <= 2 * C::NQ || KS == 1, "every staging slice needs a step");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;  // pixel within the 32-pixel subtile / cout within the 32-cout subtile
    const int h = lane >> 5;  // k-slot

    // ---- this workgroup's contiguous item range (XCD-aware: neighbouring ranges live on one XCD's L2) ----------------
    const int ncb = p.Cout >> 6;
    const int ntile = p.N * p.tiles_y * p.tiles_x;
    const int per_group = ntile * ncb;
    const int total = per_group * p.groups;
    const int nchunk = p.Cin / CB;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int base_cnt = total / (int)gridDim.x, rem_cnt = total % (int)gridDim.x;
    int item = lb * base_cnt + min(lb, rem_cnt);
    const int item_end = item + base_cnt + (lb < rem_cnt ? 1 : 0);
    if (item >= item_end) return;

    auto decode = [&](int it) {  // uniform (scalar unit)
        Item w;
        w.g = it / per_group;
        int L = it - w.g * per_group;
        w.cb = L % ncb;
        int t_ = L / ncb;
        w.tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        w.ty = t_ % p.tiles_y;
        w.n = t_ / p.tiles_y;
        w.oy0 = w.ty * TH;
        w.ox0 = w.tx * TW;
        return w;
    };
    auto advance = [&](Item w) {  // item + 1 without integer divisions (they would run as VALU sequences)
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
        w.oy0 = w.ty * TH;
        w.ox0 = w.tx * TW;
        return w;
    };
    // uniform base addresses (bytes) of an item: halo-tile origin in `in`, first weight step, prev-tile origin
    auto in_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.in + w.g * p.in_gs) +
               ((((long long)w.n * p.H + (w.oy0 * STRIDE - C::PAD)) * p.W + (w.ox0 * STRIDE - C::PAD)) * p.Cin) * 4;
    };
    auto w_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.wpack + w.g * p.w_gs + (long long)w.cb * nchunk * C::NQ * 2 * 256);
    };
    auto touches_border = [&](const Item& w) {  // some halo / tile element lies outside the image -> masked LDS write
        const int y0 = w.oy0 * STRIDE - C::PAD, x0 = w.ox0 * STRIDE - C::PAD;
        return y0 < 0 || x0 < 0 || y0 + (C::IH - 1) * C::GSTEP + 1 > p.H || x0 + (C::IW - 1) * C::GSTEP + 1 > p.W;
    };

    // ---- per-lane invariants, computed ONCE ------------------------------------------------------------------------------
    // MODE 0 staging: slice s handles element f = tid + 256 s: pixel f / PARTS of the halo tile, channels 4 (f % PARTS) ..
    // MODE 1 staging: main slice s = halo rows [s RPS, s RPS + RPS) x columns [0, TW) (the per-slice part of the address is
    //   uniform and rides in the scalar offset; the row parity that picks the bilinear weights is a compile-time constant
    //   or a lane constant); extra slice k = rows 16 k + (tid >> 4), columns TW + ((tid >> 3) & 1).
    constexpr int RPS = C::RPS, NM = C::NM, NE = C::NE;
    unsigned soff[MODE == 1 ? 1 + NE : C::ITER];  // byte offset from the halo-tile origin (same for every item and chunk)
    int ldsw0, ldswE = 0;                         // LDS write position (floats) of slice 0 / extra slice 0
    bool e_last_ok = true;                        // MODE 1: lane holds a real element in the last extra slice
    const int rowb = p.W * p.Cin * 4;             // bytes per input row (uniform)
    if (MODE == 1) {
        const int mrow = (tid >> 3) / TW, mcol = (tid >> 3) % TW, part = tid & 7;
        soff[0] = (unsigned)(((mrow * p.W + mcol) * p.Cin + part * 4) * 4);
        ldsw0 = (mrow * C::IW + mcol) * C::PS + part * 4;
        const int e_iy = tid >> 4, e_ix = TW + ((tid >> 3) & 1);
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int iy = min(e_iy + 16 * k, C::IH - 1);  // lanes past the last row re-read it (never written)
            soff[1 + k] = (unsigned)(((iy * p.W + e_ix) * p.Cin + part * 4) * 4);
        }
        ldswE = (e_iy * C::IW + e_ix) * C::PS + part * 4;
        e_last_ok = e_iy + 16 * (NE - 1) < C::IH;
    } else {
#pragma unroll
        for (int s = 0; s < C::ITER; ++s) {
            const int f = min(tid + s * 256, C::NF - 1);  // the tail of the last slice re-reads the last element (never written)
            const int pix = f / C::PARTS, part = f % C::PARTS;
            soff[s] = (unsigned)((((pix / C::IW) * C::GSTEP * p.W + (pix % C::IW) * C::GSTEP) * p.Cin + part * 4) * 4);
        }
        ldsw0 = (tid / C::PARTS) * C::PS + (tid % C::PARTS) * 4;  // slice s: + s*PPS*PS
    }
    const unsigned wlane = (unsigned)lane * 16u;  // byte offset of this lane inside a 1 KiB weight block
    // per-lane LDS read base for the two 32-pixel subtiles of this wave
    int ldsb[2];
    unsigned ooff[2];  // byte offset of the lane's output pixel (+ 4 h couts) from the item's output origin
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pl = (wave * 2 + q) * 32 + j;
        const int py = pl / TW, px = pl % TW;
        ldsb[q] = ((py * C::LSTEP) * C::IW + px * C::LSTEP) * C::PS + 4 * h;
        ooff[q] = (unsigned)(((py * p.Wo + px) * p.Cout + 4 * h) * 4);
    }
    float* aux = lds + C::MAIN_FLOATS;
    // MODE 1 bilinear invariants.  dst (gy, gx) = (oy0 - 1 + iy, ox0 - 1 + ix) with oy0, ox0 even: src0 = ((g - 1) >> 1) ->
    // aux row/col = i >> 1, and the weight of the second source is 0.25 for even i and 0.75 for odd i.  The aux tile is
    // loaded with coordinates clamped into `prev`, which reproduces align_corners=False edge clamping.
    int aoffM = 0, aoffE[NE];  // LDS offset (floats) of the top-left source: main slices (+ per-slice constant), extra slices
    // blend weights {w00, w01, w10, w11} of the four sources.  Main slices with RPS == 1 have a compile-time row parity:
    // wM is stored for even rows and odd rows use it with the two source rows swapped.
    float wM[4] = {0.f, 0.f, 0.f, 0.f}, wE[4] = {0.f, 0.f, 0.f, 0.f};
    int auxw0 = 0;
    if (MODE == 1) {
        const int mrow = (tid >> 3) / TW, mcol = (tid >> 3) % TW, part = tid & 7;
        aoffM = (mcol >> 1) * C::PS + part * 4;
        const float lxM = (mcol & 1) ? 0.75f : 0.25f;
        const float lyM = (RPS == 2 && (mrow & 1)) ? 0.75f : 0.25f;  // RPS == 2: row parity = mrow; RPS == 1: parity = s & 1
        wM[0] = (1.f - lyM) * (1.f - lxM), wM[1] = (1.f - lyM) * lxM, wM[2] = lyM * (1.f - lxM), wM[3] = lyM * lxM;
        const int e_iy = tid >> 4, e_b = (tid >> 3) & 1;
#pragma unroll
        for (int k = 0; k < NE; ++k) aoffE[k] = ((min(e_iy + 16 * k, C::IH - 1) >> 1) * C::AC + TW / 2) * C::PS + part * 4;
        const float lxE = e_b ? 0.75f : 0.25f, lyE = (e_iy & 1) ? 0.75f : 0.25f;
        wE[0] = (1.f - lyE) * (1.f - lxE), wE[1] = (1.f - lyE) * lxE, wE[2] = lyE * (1.f - lxE), wE[3] = lyE * lxE;
        auxw0 = (tid / C::PARTS) * C::PS + (tid % C::PARTS) * 4;
    }

    f32x4 v[C::ITER];
    f32x4 pvv[C::AITER > 0 ? C::AITER : 1];  // MODE 1: in-flight slices of the half-resolution `prev` tile
    unsigned poff[MODE == 1 ? C::AITER : 1];  // MODE 1: per-item byte offsets of those slices (clamped into `prev`)

    auto issue = [&](__amdgpu_buffer_rsrc_t r, int chunk_off, int s) {
        if (MODE == 1) {
            if (s < NM) v[s] = buf_load(r, soff[0], chunk_off + s * RPS * rowb);
            else v[s] = buf_load(r, soff[1 + s - NM], chunk_off);
        } else {
            v[s] = buf_load(r, soff[s], chunk_off);
        }
    };
    auto prev_offsets = [&](const Item& w) {  // once per item
        const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
        for (int k = 0; k < C::AITER; ++k) {
            const int f = min(tid + k * 256, C::NA - 1);
            const int apix = f / C::PARTS, part = f % C::PARTS;
            const int py = min(max((w.oy0 >> 1) - 1 + apix / C::AC, 0), Hp - 1), px = min(max((w.ox0 >> 1) - 1 + apix % C::AC, 0), Wp - 1);
            poff[k] = (unsigned)(((py * Wp + px) * p.Cin + part * 4) * 4);
        }
    };
    auto prev_base = [&](const Item& w) {
        return reinterpret_cast<const char*>(p.prev + w.g * p.prev_gs + (long long)w.n * (p.H >> 1) * (p.W >> 1) * p.Cin);
    };
    auto write_prev = [&]() {
#pragma unroll
        for (int k = 0; k < C::AITER; ++k)
            if (tid + k * 256 < C::NA) *reinterpret_cast<f32x4*>(aux + auxw0 + k * C::PPS * C::PS) = pvv[k];
    };
    auto combine = [&](int s) {  // MODE 1: v[s] += bilinear_x2(prev)  (net_layers.py:45-46, net_desc.py:188)
        const bool main = s < NM;
        const float* a = main ? aux + aoffM + ((s * RPS) >> 1) * C::AC * C::PS : aux + aoffE[main ? 0 : s - NM];
        const f32x4 p00 = *reinterpret_cast<const f32x4*>(a);
        const f32x4 p01 = *reinterpret_cast<const f32x4*>(a + C::PS);
        const f32x4 p10 = *reinterpret_cast<const f32x4*>(a + C::AC * C::PS);
        const f32x4 p11 = *reinterpret_cast<const f32x4*>(a + C::AC * C::PS + C::PS);
        const bool swap = main && RPS == 1 && (s & 1);
        const float* wt = main ? wM : wE;
        f32x4 r = v[s];
        r = __builtin_elementwise_fma(splat4(wt[swap ? 2 : 0]), p00, r);
        r = __builtin_elementwise_fma(splat4(wt[swap ? 3 : 1]), p01, r);
        r = __builtin_elementwise_fma(splat4(wt[swap ? 0 : 2]), p10, r);
        r = __builtin_elementwise_fma(splat4(wt[swap ? 1 : 3]), p11, r);
        v[s] = r;
    };
    // halo-tile coordinates of the element slice s holds (slow path only: border tiles)
    auto slice_iy = [&](int s) { return MODE == 1 ? (s < NM ? s * RPS + (tid >> 3) / TW : (tid >> 4) + 16 * (s - NM)) : ((tid + s * 256) / C::PARTS) / C::IW; };
    auto slice_ix = [&](int s) { return MODE == 1 ? (s < NM ? (tid >> 3) % TW : TW + ((tid >> 3) & 1)) : ((tid + s * 256) / C::PARTS) % C::IW; };
    auto slice_ptr = [&](int s) { return MODE == 1 ? (s < NM ? lds + ldsw0 + s * RPS * C::IW * C::PS : lds + ldswE + (s - NM) * 16 * C::IW * C::PS) : lds + ldsw0 + s * C::PPS * C::PS; };
    auto slice_live = [&](int s) { return MODE == 1 ? (s < C::ITER - 1 || e_last_ok) : (tid + s * 256 < C::NF); };
    auto write_tile = [&](const Item& w, bool masked) {  // registers -> LDS; conv zero padding only where needed
        if (!masked) {
#pragma unroll
            for (int s = 0; s < C::ITER; ++s)
                if (slice_live(s)) *reinterpret_cast<f32x4*>(slice_ptr(s)) = v[s];
        } else {
#pragma unroll
            for (int s = 0; s < C::ITER; ++s) {
                const int gy = w.oy0 * STRIDE - C::PAD + slice_iy(s) * C::GSTEP, gx = w.ox0 * STRIDE - C::PAD + slice_ix(s) * C::GSTEP;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (slice_live(s)) *reinterpret_cast<f32x4*>(slice_ptr(s)) = ok ? v[s] : z;
            }
        }
    };

    // ---- prologue: first chunk of the first item, synchronously ----------------------------------------------------------
    Item w = decode(item);
    {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(in_base(w));
#pragma unroll
        for (int s = 0; s < C::ITER; ++s) issue(r0, 0, s);
    }
    if (MODE == 1) {
        prev_offsets(w);
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(prev_base(w));
#pragma unroll
        for (int k = 0; k < C::AITER; ++k) pvv[k] = buf_load(rp, poff[k], 0);
        write_prev();
        __syncthreads();
#pragma unroll
        for (int s = 0; s < C::ITER; ++s) combine(s);
    }
    __amdgpu_buffer_rsrc_t rw = make_rsrc(w_base(w));  // this item's weight stream
    f32x4 wq[WD + 1][2];                                // weight stream window: steps q .. q+WD
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) wq[d][s] = buf_load(rw, wlane, (d * 2 + s) * 1024);

    for (;;) {
        f32x16 acc[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

        const bool more_items = item + 1 < item_end;
        const Item wnx = more_items ? advance(w) : w;
        const __amdgpu_buffer_rsrc_t rw_nx = more_items ? make_rsrc(w_base(wnx)) : rw;
        const bool mask_cur = touches_border(w);

        for (int ch = 0; ch < nchunk; ++ch) {
            __syncthreads();  // every wave finished reading the previous chunk from LDS
            write_tile(w, mask_cur);
            __syncthreads();

            const bool last_ch = (ch == nchunk - 1);
            // what the staging of this chunk prefetches: the next chunk of this item, or chunk 0 of the next item (at the very
            // end: this item's chunk 0 again, harmless)
            const Item wp_ = last_ch ? wnx : w;
            const int chp = last_ch ? 0 : ch + 1;
            const __amdgpu_buffer_rsrc_t r_stage = make_rsrc(in_base(wp_));
            const int stage_off = chp * (CB * 4);
            __amdgpu_buffer_rsrc_t r_prev = r_stage;
            if (MODE == 1) {
                r_prev = make_rsrc(prev_base(wp_));
                if (last_ch) prev_offsets(wp_);  // tile position changes only between items
            }
            // weight stream: steps of this chunk, then the next chunk (contiguous) or the next item's first steps
            const int wcur_off = ch * (C::NQ * 2048);
            const __amdgpu_buffer_rsrc_t rw_over = last_ch ? rw_nx : rw;
            const int wover_off = last_ch ? 0 : (ch + 1) * (C::NQ * 2048);

            f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + ldsb[0]), b1 = *reinterpret_cast<const f32x4*>(lds + ldsb[1]);
            f32x4 bn0 = b0, bn1 = b1;
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap) {
#pragma unroll
                for (int G = 0; G < C::NG; ++G) {
                    const int q = tap * C::NG + G;
                    {  // weights WD steps ahead (issued BEFORE this step's staging loads: vmcnt retires in order, so a wait
                       // for weights also waits for every older staging load), pixels (LDS) one step ahead
                        if (q + WD < C::NQ) {
                            wq[WD][0] = buf_load(rw, wlane, wcur_off + ((q + WD) * 2 + 0) * 1024);
                            wq[WD][1] = buf_load(rw, wlane, wcur_off + ((q + WD) * 2 + 1) * 1024);
                        } else {
                            wq[WD][0] = buf_load(rw_over, wlane, wover_off + ((q + WD - C::NQ) * 2 + 0) * 1024);
                            wq[WD][1] = buf_load(rw_over, wlane, wover_off + ((q + WD - C::NQ) * 2 + 1) * 1024);
                        }
                        if (q + 1 < C::NQ) {
                            const int tap1 = (q + 1) / C::NG, G1 = (q + 1) % C::NG;
                            const int toff1 = ((tap1 / KS) * C::IW + (tap1 % KS)) * C::PS + G1 * 8;
                            bn0 = *reinterpret_cast<const f32x4*>(lds + ldsb[0] + toff1);
                            bn1 = *reinterpret_cast<const f32x4*>(lds + ldsb[1] + toff1);
                        }
                    }
                    // ---- staging of the next chunk in the shadow of this step's MFMAs: one load per slice, no address math ----
#pragma unroll
                    for (int s = q; s < C::ITER; s += C::NQ) issue(r_stage, stage_off, s);
                    if (MODE == 1) {
                        if (q >= C::S_PREV && q < C::S_PREV + C::AITER) pvv[q - C::S_PREV] = buf_load(r_prev, poff[q - C::S_PREV], stage_off);
                        if (q == C::S_BAR) {
                            write_prev();
                            __syncthreads();
                        }
                        if (q >= C::S_COMB && q - C::S_COMB < C::ITER) combine(q - C::S_COMB);
                    }
                    // hipcc otherwise sinks these loads to just before their first use and waits vmcnt(0) there
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][0][t], b0[t], acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][1][t], b0[t], acc[1][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][0][t], b1[t], acc[0][1], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[0][1][t], b1[t], acc[1][1], 0, 0, 0);
                    }
#pragma unroll
                    for (int d = 0; d < WD; ++d) {
                        wq[d][0] = wq[d + 1][0];
                        wq[d][1] = wq[d + 1][1];
                    }
                    b0 = bn0;
                    b1 = bn1;
                }
            }
        }

        // ---- epilogue: + bias (+ residual) -> ReLU -> float4 NHWC stores ------------------------------------------------
        {
            const float* bias = p.bias + w.g * p.bias_gs + w.cb * 64 + h * 4;
            const long long origin = (((long long)w.n * p.Ho + w.oy0) * p.Wo + w.ox0) * p.Cout + w.cb * 64;  // floats, uniform
            const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out + w.g * p.out_gs + origin);
            const __amdgpu_buffer_rsrc_t r_res = make_rsrc(p.resid ? p.resid + w.g * p.resid_gs + origin : p.out);
            const bool has_res = p.resid != nullptr;
            const bool partial = (w.oy0 + TH > p.Ho) || (w.ox0 + TW > p.Wo);  // uniform
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (partial) {
                    const int pl = (wave * 2 + q) * 32 + j;
                    if (w.oy0 + pl / TW >= p.Ho || w.ox0 + pl % TW >= p.Wo) continue;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = s * 32 + rq * 8;  // cout quad (+ 4 h in the lane offsets)
                        f32x4 o = {acc[s][q][rq * 4 + 0], acc[s][q][rq * 4 + 1], acc[s][q][rq * 4 + 2], acc[s][q][rq * 4 + 3]};
                        o = o + *reinterpret_cast<const f32x4*>(bias + co);
                        if (has_res) o = o + buf_load(r_res, ooff[q], co * 4);
                        if (p.relu) {
                            o[0] = fmaxf(o[0], 0.f);
                            o[1] = fmaxf(o[1], 0.f);
                            o[2] = fmaxf(o[2], 0.f);
                            o[3] = fmaxf(o[3], 0.f);
                        }
                        buf_store(o, r_out, ooff[q], co * 4);
                    }
                }
            }
        }
        if (!more_items) break;
        ++item;
        w = wnx;
        rw = rw_nx;
    }
}

// ------------------------------------------------------------------------------------------------
// Host-side launcher (called from cerb_api.hip)
// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
static hipError_t launch_cfg(ConvParams p, hipStream_t st) {
    using C = ConvCfg<KS, STRIDE, TH, TW, CB, MODE>;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const long long items = (long long)p.groups * p.N * p.tiles_x * p.tiles_y * (p.Cout / 64);
    auto kern = conv_igemm_kernel<KS, STRIDE, TH, TW, CB, MODE>;
    static bool attr_done[64] = {};
    if (cerb_attr_needed(attr_done)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    // persistent grid: every workgroup resident at once (2 per CU when LDS allows), each walks a contiguous item range
    const int blocks_per_cu = (C::LDS_BYTES * 2 <= 160 * 1024) ? 2 : 1;
    long long grid = 256ll * blocks_per_cu;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, st, p);
    return hipGetLastError();
}

// Chunk size (channels staged per LDS pass) per kernel family -- also used by the weight packer.
extern "C" int cerb_conv_chunk(int ks, int stride) { return (stride == 2) ? 16 : 32; }

// Bytes of zero-filled guard band every activation buffer needs in front of and behind its payload.  Halo tiles / Winograd
// patch grids are read with unclamped addresses: up to one row + one pixel before a tensor, and after it up to (tile rows
// hanging over the last image + 1) rows -- at most 16 for the 16x16 tiles of the deepest level.  The widest row of any level of
// a tile_w-wide input is tile_w x 64 channels x 4 B (levels 2..4 are tile_w/4 x 128, /8 x 256, /16 x 512 = half of that).
extern "C" size_t cerb_conv_guard_bytes(int tile_w) { return (size_t)17 * (size_t)tile_w * 64 * 4 + (64u << 10); }

hipError_t cerb_launch_conv(const ConvParams& p, int ks, int stride, int mode, hipStream_t st) {
    const bool small = p.Wo < 32;  // 16x16 tiles for the deepest levels (16^2 / 28^2 maps)
    if (ks == 3 && stride == 1 && mode == 0) return small ? launch_cfg<3, 1, 16, 16, 32, 0>(p, st) : launch_cfg<3, 1, 8, 32, 32, 0>(p, st);
    if (ks == 3 && stride == 1 && mode == 1) return small ? launch_cfg<3, 1, 16, 16, 32, 1>(p, st) : launch_cfg<3, 1, 8, 32, 32, 1>(p, st);
    if (ks == 3 && stride == 2 && mode == 0) return small ? launch_cfg<3, 2, 16, 16, 16, 0>(p, st) : launch_cfg<3, 2, 8, 32, 16, 0>(p, st);
    if (ks == 1 && stride == 1 && mode == 0) return small ? launch_cfg<1, 1, 16, 16, 32, 0>(p, st) : launch_cfg<1, 1, 8, 32, 32, 0>(p, st);
    if (ks == 1 && stride == 2 && mode == 0) return small ? launch_cfg<1, 2, 16, 16, 16, 0>(p, st) : launch_cfg<1, 2, 8, 32, 16, 0>(p, st);
    return hipErrorInvalidValue;
}
