// conv_wgrad_wino.hip -- weight gradient of the 3x3 stride-1 pad-1 convolutions in the WINOGRAD DOMAIN (training step, BASELINE.json configs[4];
// VERDICT r4 item 1a).  With the forward's F(4x4,3x3) factorisation  Y = A^T [ U .* V ] A,  U = G g G^T,  V = B^T d B  (conv_wino4.hip):
//
//     dU[xi][co][ci] = sum over the 4x4-output tiles t of  Z_t[xi][co] * V_t[xi][ci],      Z_t = A dY_t A^T  (6x6 from the 4x4 output gradients)
//     dg[co][ci]     = G^T dU G                                                           (3x3 from 6x6, once per layer)
//
// 36 position-GEMMs (M = co, N = ci, K = tiles) = 36 products per 16 pixels and channel pair instead of the direct form's 144: a quarter of the
// matrix instructions of conv_wgrad.hip's wgrad_kernel<3,1>, which round 4 left at 50 ms of a 163 ms step on the direct fp32 roof (0.71).
//
// Work decomposition
//   * a workgroup (8 waves, one per CU: 144 KiB of LDS) owns 18 of the 36 positions (HALF: the rows 0-2 or 3-5 of the 6x6 position grid) x 64 co x 64 ci
//     of one (group, co block, ci block) and a SLICE of the tiles (split-K: per-slice partial dU in a workspace, summed in slice order by
//     wgrad_wino_reduce_kernel, which also applies G^T . G -- no float atomics, bitwise reproducible);
//   * tiles come in chunks of 8 (flattened (n, ty, tx) order).  Element-wise phase: thread = (tile = wave, channel = lane) loads its raw 6x6 input
//     patch (36 coalesced 256-byte rows of a wave) and 4x4 output gradients, transforms them -- the row pass restricted to the workgroup's three
//     position rows FIRST, so that both halves together do exactly the work of one full transform -- and writes V / Z into LDS as
//     [position 18][tile pair 4][channel 64][2]: the matrix phase's operand reads are conflict-free ds_read_b64 (lane = (channel l & 15, pair l >> 4),
//     the two values = the two k-steps of the chunk);
//   * matrix phase: wave = (9 positions, 32 co, 32 ci): 36 accumulators of v_mfma_f32_16x16x4_f32 (144 registers), 72 instructions per chunk;
//   * per chunk every wave runs matrix phase -> element-wise phase of the NEXT chunk (into the other LDS buffer) -> raw loads of the chunk after
//     that -> barrier; the loads travel underneath the next matrix phase.  (-DWW_SKEW: the two waves of a SIMD in opposite phase order -- built
//     first, 7 % slower: what bounds the kernel is the CU's L2 fill path, 106 KB of raw patches per chunk at the ~23 B/clk it sustains = as
//     long as the chunk's 4608 matrix cycles, see the ablation table in profiles/r05_wgrad_wino_ablations.txt.)
// Bias gradient (sum of dy over the pixels) rides along in the element-wise phase of the (ci block 0, half 0) workgroups.
#include "cerb_common.h"
#include <algorithm>
#include <type_traits>

namespace {
constexpr int CT = 8;                        // tiles per chunk
constexpr int XH = 18;                       // positions per workgroup
constexpr int OPER = XH * 4 * 64 * 2;        // floats of one operand array (Z or V) of one chunk: 36 KiB
constexpr int BUF = 2 * OPER;                // Z then V
constexpr int WW_LDS_BYTES = 2 * BUF * 4;    // double-buffered: 144 KiB

struct WwParams {
    const float* x;     // [G][N][H][W][Cin]
    const float* dy;    // [G][N][H][W][Cout]
    float* part;        // [slices][tiles][36][64 co][64 ci]
    float* part_b;      // optional [slices][G][ncb * 64]
    int G, N, H, W, Cin, Cout, slices, TX, TY;
    long long x_gs, dy_gs, ntile;  // ntile = N * TY * TX tiles per group
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int HALF>
__device__ __forceinline__ void wgrad_wino_body(const WwParams& p, float* lds, int pair) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncb = p.Cout >> 6, ncib = p.Cin >> 6, tiles = p.G * ncb * ncib;
    const int tile = pair % tiles, slice = pair / tiles;
    const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
    const long long nchunk = (p.ntile + CT - 1) / CT;
    // matrix-phase role
    const int gq = w & 1, cob = (w >> 1) & 1, cih = w >> 2, c = lane & 15, kq = lane >> 4;
    f32x4 acc[9][2][2];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[i][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rd[36], ry[16];
    float bsum = 0.f;
    const bool want_bias = p.part_b && cib == 0 && HALF == 0;

    // raw loads as buffer instructions: the tile's base address is wave-uniform (tile = wave), so the 52 loads of a fetch share ONE offset register
    // (lane * 4) and carry their pixel offset in the scalar operand -- 52 flat loads would pin ~100 address registers next to 144 accumulators
    const unsigned voff = (unsigned)lane * 4u;
    const int xrow = p.W * p.Cin * 4, xpix = p.Cin * 4, yrow = p.W * p.Cout * 4, ypix = p.Cout * 4;
    auto fetch = [&](long long ch) {
#ifdef WW_ABL_NOLOAD
        if (ch != slice) return;
#endif
        const long long T = ch * CT + w;
        if (T < p.ntile) {
            const int tx = (int)(T % p.TX);
            const long long r = T / p.TX;
            const int ty = (int)(r % p.TY), n = (int)(r / p.TY);
            const float* xb = p.x + g * p.x_gs + cib * 64 + (((long long)n * p.H + 4 * ty - 1) * p.W + 4 * tx - 1) * p.Cin;
            const float* yb = p.dy + g * p.dy_gs + cb * 64 + (((long long)n * p.H + 4 * ty) * p.W + 4 * tx) * p.Cout;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(yb), 0, 0x7fffffff, 0x00020000);
            const bool top = ty == 0, bot = ty == p.TY - 1, left = tx == 0, right = tx == p.TX - 1;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
#if defined(WW_ABL_HALFX)  // ablations (scripts/dev_wwabl.sh): how the launch time follows the bytes a chunk pulls -- half the patch / no patch / no gradients
                    const bool out = i >= 3 || (i == 0 && top) || (j == 0 && left) || (j == 5 && right);
#elif defined(WW_ABL_NOX)
                    const bool out = true;
#else
                    const bool out = (i == 0 && top) || (i == 5 && bot) || (j == 0 && left) || (j == 5 && right);  // wave-uniform
#endif
                    rd[i * 6 + j] = out ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)voff, i * xrow + j * xpix, 0));
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
#ifdef WW_ABL_NOY
                for (int j = 0; j < 4; ++j) ry[i * 4 + j] = 0.f;
#else
                for (int j = 0; j < 4; ++j) ry[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, (int)voff, i * yrow + j * ypix, 0));
#endif
        } else {
#pragma unroll
            for (int k = 0; k < 36; ++k) rd[k] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) ry[k] = 0.f;
        }
    };
    // The same fetch in NINE parts, one behind each position of the matrix phase (WW_INTERLEAVE, default): the CU's address path takes a wave's load every
    // ~12 cycles (368 loads per chunk = the 4608 matrix cycles of a chunk), and a wave whose load is not accepted yet issues nothing else -- 52 loads in a
    // row in front of the barrier stalled every wave for about the time the matrix phase then took (round 5 ablations: matrix 2.5 + loads 2.1 + element-wise
    // 0.5 us per chunk measured 4.4 together).  Spread between the matrix instructions they are accepted while the matrix pipe works.
    struct FetchDesc {
        __amdgpu_buffer_rsrc_t rx, ry;
        unsigned xo[3][3], yo;  // lane offsets by patch row (first / inner / last) and column, and of the gradients: voff, or out of range (zero padding / no tile)
    };
    auto fetch_desc = [&](long long ch) {
        FetchDesc f;
        const long long T = ch * CT + w;
        const bool valid = ch < nchunk && T < p.ntile;
        const long long Tc = valid ? T : 0;
        const int tx = (int)(Tc % p.TX);
        const long long r = Tc / p.TX;
        const int ty = (int)(r % p.TY), n = (int)(r / p.TY);
        const float* xb = p.x + g * p.x_gs + cib * 64 + (((long long)n * p.H + 4 * ty - 1) * p.W + 4 * tx - 1) * p.Cin;
        const float* yb = p.dy + g * p.dy_gs + cb * 64 + (((long long)n * p.H + 4 * ty) * p.W + 4 * tx) * p.Cout;
        f.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
        f.ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(yb), 0, 0x7fffffff, 0x00020000);
        // zero padding (and chunks past the end) by ADDRESS: an offset beyond the descriptor's range returns 0.  Nine offsets pinned in registers once per
        // chunk (conv_wino4b.hip's edge offsets): a select or a move in front of every load made the compiler wait for the loads in flight (its temporaries
        // reuse the loads' destination registers) in the middle of the matrix phase
        const bool top = ty == 0, bot = ty == p.TY - 1, left = tx == 0, right = tx == p.TX - 1;
        const unsigned oor = 0x80000000u;
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int qc = 0; qc < 3; ++qc) {
                const bool out = !valid || (rc == 0 && top) || (rc == 2 && bot) || (qc == 0 && left) || (qc == 2 && right);  // wave-uniform
                f.xo[rc][qc] = out ? oor : voff;
                asm volatile("" : "+v"(f.xo[rc][qc]));
            }
        f.yo = valid ? voff : oor;
        asm volatile("" : "+v"(f.yo));
        return f;
    };
#ifndef WW_LPP
#define WW_LPP 9  // loads behind each of the first six positions of the matrix phase; the last three cover the tail's latency (6 / 7 / 8 / 9 / 10 / 11 / 13 / 26 per position: 20.4 / 19.6 / 19.4 / 19.4 / 19.4 / 19.5 / 20.0 / 20.7 ms per step; one load behind every matrix instruction: 19.2; two: 20.3; all 52 in front of the barrier, round 5's first form: 23.1)
#endif
    auto fetch_part = [&](const FetchDesc& f, int part) __attribute__((always_inline)) {  // loads 6 part .. 6 part + 5 of the 52 (patch first, then the gradients)
#pragma unroll
        for (int l = 0; l < 52; ++l) {
            if (l / WW_LPP != part) continue;
            if (l < 36) {
                const int i = l / 6, j = l % 6;
                rd[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(f.rx, (int)f.xo[i == 0 ? 0 : i == 5 ? 2 : 1][j == 0 ? 0 : j == 5 ? 2 : 1], i * xrow + j * xpix, 0));
            } else {
                const int i = (l - 36) / 4, j = (l - 36) % 4;
                ry[l - 36] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(f.ry, (int)f.yo, i * yrow + j * ypix, 0));
            }
        }
    };
    // write position of (local position xi, this thread's tile w and channel lane): [xi][pair w >> 1][channel][w & 1]
    const int wpos = ((w >> 1) * 64 + lane) * 2 + (w & 1);
    auto estage = [&](int buf, const float (&ryv)[16]) __attribute__((always_inline)) {
        float* Zl = lds + buf * BUF;
        float* Vl = Zl + OPER;
#ifdef WW_ABL_NOE
        if (buf >= 0 && rd[0] != 1.2345e-30f) return;
#endif
        if (want_bias) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += ryv[k];
            bsum += s;
        }
        // ---- V = B^T d B, rows 3 HALF .. 3 HALF + 2: the row pass (down each column) first, then the full column pass on those three rows ----
        float m[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float x0 = rd[j], x1 = rd[6 + j], x2 = rd[12 + j], x3 = rd[18 + j], x4 = rd[24 + j], x5 = rd[30 + j];
            if (HALF == 0) {
                const float t0 = fmaf(-4.f, x2, x4), t1 = fmaf(-4.f, x1, x3);
                m[0][j] = fmaf(-5.f, x2, fmaf(4.f, x0, x4));
                m[1][j] = t0 + t1;
                m[2][j] = t0 - t1;
            } else {
                const float u0 = x4 - x2, u1 = x3 - x1;
                m[0][j] = fmaf(2.f, u1, u0);
                m[1][j] = fmaf(-2.f, u1, u0);
                m[2][j] = fmaf(-5.f, x3, fmaf(4.f, x1, x5));
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float x0 = m[r][0], x1 = m[r][1], x2 = m[r][2], x3 = m[r][3], x4 = m[r][4], x5 = m[r][5];
            const float t0 = fmaf(-4.f, x2, x4), t1 = fmaf(-4.f, x1, x3), u0 = x4 - x2, u1 = x3 - x1;
            float* o = Vl + (r * 6) * 512 + wpos;
            o[0 * 512] = fmaf(-5.f, x2, fmaf(4.f, x0, x4));
            o[1 * 512] = t0 + t1;
            o[2 * 512] = t0 - t1;
            o[3 * 512] = fmaf(2.f, u1, u0);
            o[4 * 512] = fmaf(-2.f, u1, u0);
            o[5 * 512] = fmaf(-5.f, x3, fmaf(4.f, x1, x5));
        }
        // ---- Z = A dY A^T, the same three rows ------------------------------------------------------------------------------------
        float zm[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y0 = ryv[j], y1 = ryv[4 + j], y2 = ryv[8 + j], y3 = ryv[12 + j];
            if (HALF == 0) {
                const float s = y0 + y2, t = y1 + y3;
                zm[0][j] = y0;
                zm[1][j] = s + t;
                zm[2][j] = s - t;
            } else {
                const float pp = fmaf(4.f, y2, y0), qq = fmaf(8.f, y3, 2.f * y1);
                zm[0][j] = pp + qq;
                zm[1][j] = pp - qq;
                zm[2][j] = y3;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float z0 = zm[r][0], z1 = zm[r][1], z2 = zm[r][2], z3 = zm[r][3];
            const float s = z0 + z2, t = z1 + z3, pp = fmaf(4.f, z2, z0), qq = fmaf(8.f, z3, 2.f * z1);
            float* o = Zl + (r * 6) * 512 + wpos;
            o[0 * 512] = z0;
            o[1 * 512] = s + t;
            o[2 * 512] = s - t;
            o[3 * 512] = pp + qq;
            o[4 * 512] = pp - qq;
            o[5 * 512] = z3;
        }
    };
    auto mstage = [&](int buf, auto&& hook) __attribute__((always_inline)) {
#ifdef WW_ABL_NOM
        if (buf >= 0 && rd[0] != 1.2345e-30f) return;
#endif
        const float* Zl = lds + buf * BUF + (kq * 64 + 32 * cob + c) * 2;
        const float* Vl = lds + buf * BUF + OPER + (kq * 64 + 32 * cih + c) * 2;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int xi = 9 * gq + i;
#ifdef WW_ABL_NOLDSR  // ablation: the matrix phase without its operand reads (do LDS returns and global-load returns share a path?)
            f32x2 a0 = {(float)xi, 1.f}, a1 = {2.f, (float)lane}, b0 = {3.f, (float)i}, b1 = {(float)buf, 4.f};
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
#else
            const f32x2 a0 = *reinterpret_cast<const f32x2*>(Zl + xi * 512), a1 = *reinterpret_cast<const f32x2*>(Zl + xi * 512 + 32);
            const f32x2 b0 = *reinterpret_cast<const f32x2*>(Vl + xi * 512), b1 = *reinterpret_cast<const f32x2*>(Vl + xi * 512 + 32);
#endif
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[i][0][0], 0, 0, 0);
                acc[i][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[i][0][1], 0, 0, 0);
                acc[i][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[i][1][0], 0, 0, 0);
                acc[i][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[i][1][1], 0, 0, 0);
            }
#ifndef WW_ABL_NOLOAD
            hook(i);
#endif
        }
    };

    // Waves 0-3 (X) and 4-7 (Y) share the four SIMDs pairwise.  X: M(i) E(i+1) | M(i+1) E(i+2) | ...   Y: E(i+1) M(i) | E(i+2) M(i+1) | ...  ( | = barrier):
    // between two barriers X's matrix phase runs beside Y's element-wise phase and vice versa.  E(k) writes buffer k & 1, M(k) reads it; every E(k) lies
    // between barrier k - 2 and barrier k - 1, every M(k) between k - 1 and k: no buffer is read and written in the same interval.
    const long long S = p.slices;
    long long ch = slice;  // chunk i of this slice = slice + i * S
#ifndef WW_BULK_FETCH
    // M(i) with the loads of chunk i + 1 between its matrix instructions -> E(i + 1) -> barrier.  (Tried: the 16 gradient loads one MORE chunk ahead in a second
    // register set, so that E waits for the 36 patch loads only -- 18.7 -> 20.6 ms per step, slower; scheduling barriers around the loads: no difference.)
    if (ch < nchunk) {
        fetch(ch);
        estage(0, ry);
    }
    __syncthreads();
    {
        int buf = 0;
        for (; ch < nchunk; ch += S) {
            const bool more = ch + S < nchunk;
            const FetchDesc nf = fetch_desc(ch + S);
            mstage(buf, [&](int i) __attribute__((always_inline)) { if (more) fetch_part(nf, i); });
            if (more) estage(buf ^ 1, ry);
            __syncthreads();
            buf ^= 1;
        }
    }
#else
#ifdef WW_SKEW
    const bool grpY = w >= 4;
#else
    const bool grpY = false;  // measured (profiles/r05_wgrad_wino_ablations.txt): all eight waves in X order 4.42 ms, skewed 4.75 ms on the 448^2 level
#endif
    if (ch < nchunk) {
        fetch(ch);
        estage(0, ry);
        if (ch + S < nchunk) fetch(ch + S);
    }
    __syncthreads();
    if (grpY && ch + S < nchunk) {
        estage(1, ry);
        if (ch + 2 * S < nchunk) fetch(ch + 2 * S);
    }
    int buf = 0;
    for (; ch < nchunk; ch += S) {
        mstage(buf, [](int) {});
        if (!grpY && ch + S < nchunk) {
            estage(buf ^ 1, ry);
            if (ch + 2 * S < nchunk) fetch(ch + 2 * S);
        }
        __syncthreads();
        if (grpY && ch + 2 * S < nchunk) {
            estage(buf, ry);
            if (ch + 3 * S < nchunk) fetch(ch + 3 * S);
        }
        buf ^= 1;
    }
#endif
    // ---- this slice's partial dU: part[slice][tile][xi 36][co 64][ci 64]; D[co 4 kq + e][ci c] ----------------------------------------------
    float* o = p.part + ((long long)slice * tiles + tile) * 36 * 4096 + (long long)(18 * HALF + 9 * gq) * 4096;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[i * 4096 + (32 * cob + 16 * a + 4 * kq + e) * 64 + 32 * cih + 16 * b + c] = acc[i][a][b][e];
    if (want_bias) {  // the eight tiles' sums of a channel, in wave order
        __syncthreads();
        lds[w * 64 + lane] = bsum;
        __syncthreads();
        if (w == 0) {
            float s = 0.f;
            for (int k = 0; k < 8; ++k) s += lds[k * 64 + lane];
            p.part_b[((long long)slice * p.G + g) * ncb * 64 + cb * 64 + lane] = s;
        }
    }
}

// The two halves of a (tile, slice) pair sit 8 workgroup ids apart: dispatched back to back onto the SAME XCD (id % 8), so that the raw patch / dy
// rows both of them read meet in that XCD's L2.
__global__ __launch_bounds__(512) void wgrad_wino_kernel(WwParams p, int npair) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, half = (b >> 3) & 1, pair = (b >> 4) * 8 + (b & 7);
    if (pair >= npair) return;
    if (half == 0) wgrad_wino_body<0>(p, lds, pair);
    else wgrad_wino_body<1>(p, lds, pair);
}

// dw[g][co][ci][ky][kx] = G^T (sum over slices of dU) G.  Workgroup = (tile, co): thread = (slice group sg of 4, ci of 64) -- a wave reads whole 256-byte rows of a
// partial; a thread adds the slices sg, sg + 4, .. in order, the four groups meet in LDS and are added in group order (fixed order: reproducible), 64 threads apply G^T . G.
// (First version: one thread per (co, ci) over ALL slices, 40 us per launch; second: 16 groups x 16 ci, 64-byte pieces, 1.47 ms per step over the 37 launches.)
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G, int Cin, int Cout, int slices) {
    __shared__ float red[4][36][64];
    const int ncb = Cout >> 6, ncib = Cin >> 6, tiles = G * ncb * ncib;
    const int co = blockIdx.x & 63, tile = blockIdx.x >> 6;
    const int sg = threadIdx.x >> 6, cl = threadIdx.x & 63, ci = cl;
    float u[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) u[xi] = 0.f;
    for (int s = sg; s < slices; s += 4) {
        const float* src = part + ((long long)s * tiles + tile) * 36 * 4096 + co * 64 + ci;
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) u[xi] += src[xi * 4096];
    }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) red[sg][xi][cl] = u[xi];
    __syncthreads();
    if (sg != 0) return;
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) u[xi] = ((red[0][xi][cl] + red[1][xi][cl]) + red[2][xi][cl]) + red[3][xi][cl];
    // t[x][b] = sum_a G[a][x] u[a][b];  dg[x][y] = sum_b t[x][b] G[b][y]
    float t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float u0 = u[b], u1 = u[6 + b], u2 = u[12 + b], u3 = u[18 + b], u4 = u[24 + b], u5 = u[30 + b];
        t[0][b] = 0.25f * u0 - (u1 + u2) * (1.f / 6.f) + (u3 + u4) * (1.f / 24.f);
        t[1][b] = (u2 - u1) * (1.f / 6.f) + (u3 - u4) * (1.f / 12.f);
        t[2][b] = -(u1 + u2) * (1.f / 6.f) + (u3 + u4) * (1.f / 6.f) + u5;
    }
    const int cib = tile % ncib, cb = (tile / ncib) % ncb, g = tile / (ncib * ncb);
    float* o = dw + (((long long)g * Cout + cb * 64 + co) * Cin + cib * 64 + ci) * 9;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const float u0 = t[x][0], u1 = t[x][1], u2 = t[x][2], u3 = t[x][3], u4 = t[x][4], u5 = t[x][5];
        o[x * 3 + 0] = 0.25f * u0 - (u1 + u2) * (1.f / 6.f) + (u3 + u4) * (1.f / 24.f);
        o[x * 3 + 1] = (u2 - u1) * (1.f / 6.f) + (u3 - u4) * (1.f / 12.f);
        o[x * 3 + 2] = -(u1 + u2) * (1.f / 6.f) + (u3 + u4) * (1.f / 6.f) + u5;
    }
}
}  // namespace

hipError_t cerb_launch_slab_sum(const float* part, float* out, int n, int blocks, int groups, hipStream_t st);

bool cerb_wgrad_wino_supported(int H, int W, int Cin, int Cout) { return H % 4 == 0 && W % 4 == 0 && Cin % 64 == 0 && Cout % 64 == 0 && H >= 8 && W >= 8; }

static int ww_slices(int G, int Cin, int Cout, long long ntile) {
    const int tiles = G * (Cout / 64) * (Cin / 64);
    long long s = 128 / tiles;  // x 2 halves: one workgroup per CU over the whole grid, all of them resident at once
    if (s < 1) s = 1;
    const long long nchunk = (ntile + CT - 1) / CT;
    if (s > nchunk) s = nchunk;
    return (int)s;
}
size_t cerb_wgrad_wino_workspace_bytes(int G, int N, int H, int W, int Cin, int Cout) {
    const long long ntile = (long long)N * (H / 4) * (W / 4);
    const int tiles = G * (Cout / 64) * (Cin / 64), slices = ww_slices(G, Cin, Cout, ntile);
    return (size_t)slices * tiles * 36 * 4096 * 4 + (size_t)slices * G * Cout * 4 + 256;
}
// x: [G][N][H][W][Cin] (group stride x_gs), dy: [G][N][H][W][Cout] contiguous; dw: [G][Cout][Cin][3][3]; db (optional): [G][Cout]
hipError_t cerb_launch_wgrad_wino(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, long long x_gs, void* ws, hipStream_t st,
                                  float* db) {
    if (!cerb_wgrad_wino_supported(H, W, Cin, Cout)) return hipErrorInvalidValue;
    WwParams p;
    p.x = x; p.dy = dy; p.part = (float*)ws;
    p.G = G; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.TX = W / 4; p.TY = H / 4;
    p.ntile = (long long)N * p.TY * p.TX;
    p.x_gs = x_gs; p.dy_gs = (long long)N * H * W * Cout;
    const int tiles = G * (Cout / 64) * (Cin / 64);
    p.slices = ww_slices(G, Cin, Cout, p.ntile);
    p.part_b = db ? (float*)ws + (size_t)p.slices * tiles * 36 * 4096 : nullptr;
    static bool attr0[64];
    if (cerb_attr_needed(attr0)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    const int npair = tiles * p.slices;
    const dim3 grid((unsigned)(((npair + 7) / 8) * 16));
    hipLaunchKernelGGL(wgrad_wino_kernel, grid, dim3(512), WW_LDS_BYTES, st, p, npair);
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)(tiles * 64)), dim3(256), 0, st, (const float*)ws, dw, G, Cin, Cout, p.slices);
    if (db) (void)cerb_launch_slab_sum(p.part_b, db, G * Cout, p.slices, 1, st);
    return hipGetLastError();
}
