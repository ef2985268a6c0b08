// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the Cerberus tile path, every nn.Conv2d + eval-BatchNorm2d + ReLU triple of
//   reference models/backbone/resnet.py:81-97 (BasicBlock, incl. 1x1 stride-2 downsample and the residual add)
//   reference models/utils/conv_layers.py:24-60 (_ConvLayer, post-activation) used by the decoders
//   reference models/net_desc.py:52,153 (conv_map) and :185-188 (upsample2x + skip add, fused as MODE 1)
//
// Data layout: activations NHWC fp32; weights pre-packed (BN folded) by pack_conv() in cerb_api.hip.
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
// skip values with LDS reads issued in the shadow of the MFMAs -- 5 global loads per element never exist.  Co-resident
// workgroups run in lockstep on identical work, so occupancy alone never overlapped the staging; this does.  The
// weight stream of an item is ONE linear array ([chunk][tap][group][half][lane][4]) read two 8-channel groups (2 KiB per
// wave) ahead of use, straight from L2 into VGPRs.
#include "cerb_common.h"

// CERB_WLDS = 1 (experiment, OFF): the weights of a chunk are staged into LDS next to the pixel tile so the MFMA loop
// waits only on LDS (lgkmcnt) -- tests the idea that in-order vmcnt couples weight waits to older HBM loads / stores.
// It needs 123-138 KB of LDS => one workgroup per CU, one wave per SIMD, and measured 116 vs 122 TFLOP/s for the
// global->VGPR weight stream with two co-resident workgroups, so the default stays 0 (DESIGN.md "what did not work").
#ifndef CERB_WLDS
#define CERB_WLDS 0
#endif
#ifndef CERB_SCHED_GROUPS
#define CERB_SCHED_GROUPS (CERB_WLDS ? 3 : 0)
#endif

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int IW = (TW - 1) * STRIDE + KS;
    static constexpr int PS = CB + 4;  // LDS pixel stride in floats (16-lane ds_read_b128 groups hit distinct 16-B slots)
    static constexpr int NG = CB / 8;  // 8-channel groups per chunk
    static constexpr int T = KS * KS;  // taps
    static constexpr int NQ = T * NG;  // weight-stream steps per chunk
    static constexpr int PARTS = CB / 4;
    static constexpr int NF = IH * IW * PARTS;     // float4 elements of one staged chunk
    static constexpr int ITER = (NF + 255) / 256;  // staging slices per thread
    static constexpr int MAIN_FLOATS = IH * IW * PS;
    // MODE 1: half-resolution tile of `prev` covering every bilinear source of the halo tile
    static constexpr int AR = TH / 2 + 2, AC = TW / 2 + 2;
    static constexpr int NA = AR * AC * PARTS;
    static constexpr int AITER = (NA + 255) / 256;
    static constexpr int AUX_FLOATS = (MODE == 1) ? AR * AC * PS : 0;
    static constexpr int W_FLOATS = CERB_WLDS ? NQ * 2 * 256 : 0;  // one chunk of packed weights
    static constexpr int WITER = NQ / 2;                            // float4 weight-staging slices per thread
    static constexpr int LDS_FLOATS = MAIN_FLOATS + AUX_FLOATS + W_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    // MODE 1 schedule inside a chunk (in 8-channel steps): skip slices at steps 0..ITER-1, prev slices next, then the
    // auxiliary tile is written + one barrier, then one combine per step
    static constexpr int S_PREV = ITER;
    static constexpr int S_BAR = ITER + AITER + 6;
    static constexpr int S_COMB = S_BAR + 1;
};

struct Item {
    int g, cb, n, oy0, ox0;
};

template <int KS, int STRIDE, int TH, int TW, int CB, int MODE>
__global__ __launch_bounds__(256, CERB_WLDS ? 1 : 2) void conv_igemm_kernel(ConvParams p) {
    using C = ConvCfg<KS, STRIDE, TH, TW, CB, MODE>;
    static_assert(TH * TW == 256, "tile must hold 256 pixels (4 waves x 2 x 32)");
    static_assert(MODE == 0 || (KS == 3 && STRIDE == 1 && C::S_COMB + C::ITER <= C::NQ), "MODE 1 schedule must fit in one chunk");
    static_assert(C::ITER <= 2 * C::NQ || KS == 1, "every staging slice needs a step");
    static_assert(C::NQ % 2 == 0, "weight staging assumes an even number of steps per chunk");
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

    auto decode = [&](int it) {
        Item w;
        w.g = it / per_group;
        int L = it - w.g * per_group;
        w.cb = L % ncb;
        int t_ = L / ncb;
        const int tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y;
        w.n = t_ / p.tiles_y;
        w.oy0 = ty * TH;
        w.ox0 = tx * TW;
        return w;
    };
    auto wbase = [&](const Item& w) {
        return reinterpret_cast<const f32x4*>(p.wpack + w.g * p.w_gs + (long long)w.cb * nchunk * C::NQ * 2 * 256) + (CERB_WLDS ? tid : lane);
    };

    // ---- staging of one slice (256 float4 elements) of chunk (w, ch) into registers --------------------------------------
    // MODE 0: v = in[gy][gx][c0 + 4 part ..]   (zero outside the image = conv zero padding)
    // MODE 1: v = skip + bilinear_up2(prev)    (reference net_layers.py:45-46 + net_desc.py:188); loads now, math later
    f32x4 v[C::ITER];
    f32x4 pvv[C::AITER > 0 ? C::AITER : 1];  // MODE 1: in-flight slices of the half-resolution `prev` tile
    float* aux = lds + C::MAIN_FLOATS;
#if CERB_WLDS
    f32x4 wst[C::WITER];                                                  // in-flight slices of the next chunk's weights
    f32x4* wl = reinterpret_cast<f32x4*>(lds + C::MAIN_FLOATS + C::AUX_FLOATS);  // [step][half][lane] float4, as packed
#endif
    auto slice_coords = [&](const Item& w, int s, int& part, int& gy, int& gx) {
        const int f = tid + s * 256;
        const int pix = f / C::PARTS;
        part = f % C::PARTS;
        const int iy = pix / C::IW, ix = pix % C::IW;
        gy = w.oy0 * STRIDE - C::PAD + iy;
        gx = w.ox0 * STRIDE - C::PAD + ix;
        return f < C::NF && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    };
    // The loads are UNCONDITIONAL (addresses clamped into the tensor) so that the chunk body stays one basic block the
    // scheduler can interleave with the MFMAs; out-of-image / out-of-tile elements are zeroed when the slice is written to LDS.
    auto issue = [&](const Item& w, int ch, int s) {  // skip / plain input slice -> v[s]
        int part, gy, gx;
        slice_coords(w, s, part, gy, gx);
        const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        const float* in = p.in + w.g * p.in_gs + (long long)w.n * p.H * p.W * p.Cin;
#ifdef CERB_ABL_NOSTAGE
        v[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        (void)in; (void)cy; (void)cx; (void)ch;
#else
        v[s] = *reinterpret_cast<const f32x4*>(in + ((long long)cy * p.W + cx) * p.Cin + ch * CB + part * 4);
#endif
    };
    auto issue_prev = [&](const Item& w, int ch, int k) {  // MODE 1: slice k of the clamped half-resolution tile
        const int f = min(tid + k * 256, C::NA - 1);
        const int apix = f / C::PARTS, part = f % C::PARTS;
        const int ar = apix / C::AC, ac = apix % C::AC;
        const int Hp = p.H >> 1, Wp = p.W >> 1;
        const int py = min(max((w.oy0 >> 1) - 1 + ar, 0), Hp - 1), px = min(max((w.ox0 >> 1) - 1 + ac, 0), Wp - 1);
        const float* pv = p.prev + w.g * p.prev_gs + (long long)w.n * Hp * Wp * p.Cin;
        pvv[k] = *reinterpret_cast<const f32x4*>(pv + ((long long)py * Wp + px) * p.Cin + ch * CB + part * 4);
    };
    auto write_prev = [&]() {
#pragma unroll
        for (int k = 0; k < C::AITER; ++k) {
            const int f = tid + k * 256;
            if (f < C::NA) *reinterpret_cast<f32x4*>(aux + (f / C::PARTS) * C::PS + (f % C::PARTS) * 4) = pvv[k];
        }
    };
    auto combine = [&](const Item& w, int s) {  // MODE 1: v[s] += bilinear_x2(prev)  (net_layers.py:45-46, net_desc.py:188)
        int part, gy, gx;
        slice_coords(w, s, part, gy, gx);
        const int Hp = p.H >> 1, Wp = p.W >> 1;
        // src = 0.5*(dst+0.5)-0.5 clamped at 0 (align_corners=False): y0 = (gy-1)>>1 for gy>=1, 0 for gy==0;
        // fractional offset 0 (clamped edge), 0.25 (odd dst) or 0.75 (even dst).  Out-of-image elements compute garbage from
        // clamped window indices and are zeroed at the LDS write.
        const int y0 = gy > 0 ? (gy - 1) >> 1 : 0, x0 = gx > 0 ? (gx - 1) >> 1 : 0;
        const int y1 = y0 + (y0 < Hp - 1), x1 = x0 + (x0 < Wp - 1);
        const float ly = gy > 0 ? ((gy & 1) ? 0.25f : 0.75f) : 0.f;
        const float lx = gx > 0 ? ((gx & 1) ? 0.25f : 0.75f) : 0.f;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const int ry = (w.oy0 >> 1) - 1, rx = (w.ox0 >> 1) - 1;
        const int a0_ = min(max(y0 - ry, 0), C::AR - 1), a1_ = min(max(y1 - ry, 0), C::AR - 1);
        const int c0_ = min(max(x0 - rx, 0), C::AC - 1), c1_ = min(max(x1 - rx, 0), C::AC - 1);
        const float* a = aux + part * 4;
        const f32x4 p00 = *reinterpret_cast<const f32x4*>(a + (a0_ * C::AC + c0_) * C::PS);
        const f32x4 p01 = *reinterpret_cast<const f32x4*>(a + (a0_ * C::AC + c1_) * C::PS);
        const f32x4 p10 = *reinterpret_cast<const f32x4*>(a + (a1_ * C::AC + c0_) * C::PS);
        const f32x4 p11 = *reinterpret_cast<const f32x4*>(a + (a1_ * C::AC + c1_) * C::PS);
        v[s] = v[s] + (hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11));
    };

    // per-lane LDS read base for the two 32-pixel subtiles of this wave
    int ldsb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pl = (wave * 2 + q) * 32 + j;
        const int py = pl / TW, px = pl % TW;
        ldsb[q] = ((py * STRIDE) * C::IW + px * STRIDE) * C::PS + 4 * h;
    }

    // ---- prologue: first chunk of the first item, synchronously ----------------------------------------------------------
    Item w = decode(item);
#pragma unroll
    for (int s = 0; s < C::ITER; ++s) issue(w, 0, s);
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < C::AITER; ++k) issue_prev(w, 0, k);
        write_prev();
        __syncthreads();
#pragma unroll
        for (int s = 0; s < C::ITER; ++s) combine(w, s);
    }
    const f32x4* wv = wbase(w);
#if CERB_WLDS
#pragma unroll
    for (int k = 0; k < C::WITER; ++k) wst[k] = wv[k * 256];
#else
    constexpr int WD = (MODE == 0 && STRIDE == 1) ? 3 : 2;  // weight prefetch distance in steps (registers: 8 per step)
    f32x4 wq[WD + 1][2];  // weight stream window: steps q .. q+WD
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) wq[d][s] = wv[(d * 2 + s) * 64];
#endif

    for (;;) {
        f32x16 acc[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

        const bool more_items = item + 1 < item_end;
        const Item wnx = more_items ? decode(item + 1) : w;
        const f32x4* wv_nx = more_items ? wbase(wnx) : wv;

        for (int ch = 0; ch < nchunk; ++ch) {
            __syncthreads();  // every wave finished reading the previous chunk from LDS
#pragma unroll
            for (int s = 0; s < C::ITER; ++s) {
                const int f = tid + s * 256;
                int part, gy, gx;
                const bool ok = slice_coords(w, s, part, gy, gx);  // conv zero padding / outside the tile
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (f < C::NF) *reinterpret_cast<f32x4*>(lds + (f / C::PARTS) * C::PS + (f % C::PARTS) * 4) = ok ? v[s] : z;
            }
#if CERB_WLDS
#pragma unroll
            for (int k = 0; k < C::WITER; ++k) wl[tid + k * 256] = wst[k];
#endif
            __syncthreads();

            const bool last_ch = (ch == nchunk - 1);
            const Item wp_ = last_ch ? wnx : w;
            const int chp = last_ch ? 0 : ch + 1;
            // weight stream pointer for steps beyond this chunk: the next chunk is contiguous; the next item restarts
#if CERB_WLDS
            const f32x4* wnext = last_ch ? wv_nx : wv + (long long)(ch + 1) * C::NQ * 128;  // next chunk's packed weights (+tid)
            f32x4 a0 = wl[lane], a1 = wl[64 + lane], an0 = a0, an1 = a1;
#else
            const f32x4* wcur = wv + (long long)ch * C::NQ * 128;
            const f32x4* wover = last_ch ? (wv_nx - (long long)C::NQ * 128) : wcur;
#endif

            f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + ldsb[0]), b1 = *reinterpret_cast<const f32x4*>(lds + ldsb[1]);
            f32x4 bn0 = b0, bn1 = b1;
#pragma unroll
            for (int tap = 0; tap < C::T; ++tap) {
#pragma unroll
                for (int G = 0; G < C::NG; ++G) {
                    const int q = tap * C::NG + G;
                    {  // weights WD steps ahead (issued BEFORE this step's staging loads: vmcnt retires in order, so a
                       // wait for weights also waits for every older staging load), pixels (LDS) one step ahead
#if CERB_WLDS
                        if (q + 1 < C::NQ) {
                            an0 = wl[((q + 1) * 2 + 0) * 64 + lane];
                            an1 = wl[((q + 1) * 2 + 1) * 64 + lane];
                        }
                        if ((q & 1) == 0 && q / 2 < C::WITER) wst[q / 2] = wnext[(q / 2) * 256];
#elif defined(CERB_ABL_NOWLOAD)
                        wq[WD][0] = wq[0][1];
                        wq[WD][1] = wq[0][0];
#else
                        const f32x4* src = (q + WD < C::NQ) ? wcur : wover;
                        wq[WD][0] = src[(long long)((q + WD) * 2 + 0) * 64];
                        wq[WD][1] = src[(long long)((q + WD) * 2 + 1) * 64];
#endif
                        if (q + 1 < C::NQ) {
                            const int tap1 = (q + 1) / C::NG, G1 = (q + 1) % C::NG;
                            const int toff1 = ((tap1 / KS) * C::IW + (tap1 % KS)) * C::PS + G1 * 8;
                            bn0 = *reinterpret_cast<const f32x4*>(lds + ldsb[0] + toff1);
                            bn1 = *reinterpret_cast<const f32x4*>(lds + ldsb[1] + toff1);
                        }
                    }
                    // ---- staging of the next chunk in the shadow of this step's MFMAs -----------------------------------
                    {
#pragma unroll
                        for (int s = q; s < C::ITER; s += C::NQ) issue(wp_, chp, s);
                        if (MODE == 1) {
                            if (q >= C::S_PREV && q < C::S_PREV + C::AITER) issue_prev(wp_, chp, q - C::S_PREV);
                            if (q == C::S_BAR) {
                                write_prev();
                                __syncthreads();
                            }
                            if (q >= C::S_COMB && q - C::S_COMB < C::ITER) combine(wp_, q - C::S_COMB);
                        }
                    }
                    // hipcc otherwise sinks these loads to just before their first use and waits vmcnt(0) there.  Inside the
                    // region (previous step's 16 MFMAs + this step's address math / loads) ask for one MFMA, then up to
                    // three non-MFMA instructions, so a single wave keeps the matrix pipe fed (co-resident workgroups run in
                    // lockstep and would otherwise do their non-MFMA bursts at the same time).
#if CERB_SCHED_GROUPS
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x096, CERB_SCHED_GROUPS, 0);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
#if CERB_WLDS
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
                    }
                    a0 = an0;
                    a1 = an1;
#else
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
#endif
                    b0 = bn0;
                    b1 = bn1;
                }
            }
        }

        // ---- epilogue: + bias (+ residual) -> ReLU -> float4 NHWC stores ------------------------------------------------
        {
            const float* __restrict__ bias = p.bias + w.g * p.bias_gs + w.cb * 64;
            float* __restrict__ out = p.out + w.g * p.out_gs;
            const float* __restrict__ resid = p.resid ? p.resid + w.g * p.resid_gs : nullptr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pl = (wave * 2 + q) * 32 + j;
                const int oy = w.oy0 + pl / TW, ox = w.ox0 + pl % TW;
#ifdef CERB_ABL_NOEPI
                if (acc[0][q][0] != 123.456f) continue;
#endif
                if (oy >= p.Ho || ox >= p.Wo) continue;
                const long long pixoff = (((long long)w.n * p.Ho + oy) * p.Wo + ox) * p.Cout + w.cb * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = s * 32 + rq * 8 + h * 4;
                        f32x4 o = {acc[s][q][rq * 4 + 0], acc[s][q][rq * 4 + 1], acc[s][q][rq * 4 + 2], acc[s][q][rq * 4 + 3]};
                        o = o + *reinterpret_cast<const f32x4*>(bias + co);
                        if (resid) o = o + *reinterpret_cast<const f32x4*>(resid + pixoff + co);
                        if (p.relu) {
                            o[0] = fmaxf(o[0], 0.f);
                            o[1] = fmaxf(o[1], 0.f);
                            o[2] = fmaxf(o[2], 0.f);
                            o[3] = fmaxf(o[3], 0.f);
                        }
                        *reinterpret_cast<f32x4*>(out + pixoff + co) = o;
                    }
                }
            }
        }
        if (!more_items) break;
        ++item;
        w = wnx;
        wv = wv_nx;
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
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    // persistent grid: every workgroup resident at once (2 per CU when LDS allows), each walks a contiguous item range
    const int blocks_per_cu = (!CERB_WLDS && C::LDS_BYTES * 2 <= 160 * 1024) ? 2 : 1;
    long long grid = 256ll * blocks_per_cu;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, st, p);
    return hipGetLastError();
}

// Chunk size (channels staged per LDS pass) per kernel family -- also used by the weight packer.
extern "C" int cerb_conv_chunk(int ks, int stride) { return (stride == 2) ? 16 : 32; }

hipError_t cerb_launch_conv(const ConvParams& p, int ks, int stride, int mode, hipStream_t st) {
    const bool small = p.Wo < 32;  // 16x16 tiles for the deepest levels (16^2 / 28^2 maps)
    if (ks == 3 && stride == 1 && mode == 0) return small ? launch_cfg<3, 1, 16, 16, 32, 0>(p, st) : launch_cfg<3, 1, 8, 32, 32, 0>(p, st);
    if (ks == 3 && stride == 1 && mode == 1) return small ? launch_cfg<3, 1, 16, 16, 32, 1>(p, st) : launch_cfg<3, 1, 8, 32, 32, 1>(p, st);
    if (ks == 3 && stride == 2 && mode == 0) return small ? launch_cfg<3, 2, 16, 16, 16, 0>(p, st) : launch_cfg<3, 2, 8, 32, 16, 0>(p, st);
    if (ks == 1 && stride == 1 && mode == 0) return small ? launch_cfg<1, 1, 16, 16, 32, 0>(p, st) : launch_cfg<1, 1, 8, 32, 32, 0>(p, st);
    if (ks == 1 && stride == 2 && mode == 0) return small ? launch_cfg<1, 2, 16, 16, 16, 0>(p, st) : launch_cfg<1, 2, 8, 32, 16, 0>(p, st);
    return hipErrorInvalidValue;
}
