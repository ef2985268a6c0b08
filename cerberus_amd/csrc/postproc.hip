// Instance post-processing of the Cerberus tile path on gfx950, entirely on device.
//
// Replaces reference loader/postproc.py:268-407 (PostProcInstErodedContourMap.__proc_nuclei / __proc_gland /
// __proc_lumen) and the third-party routines it calls (scipy.ndimage.label / binary_fill_holes,
// skimage.morphology.remove_small_objects, skimage.segmentation.watershed, cv2.erode / dilate /
// getStructuringElement).  Integer work, HBM/latency bound -- no MFMA here.
//
// Building blocks
//   * 4-connected component labelling: lock-free union-find (atomicMin on roots), root = smallest raster index of
//     the component, so "rank of the root among roots in raster order" reproduces scipy's label numbering.
//   * component areas: wave-aggregated atomics; removal of small objects = drop roots whose area < min_size.
//   * fill holes = background components that do not touch the (image or crop) border.
//   * marker-controlled watershed: skimage's priority flood is sequential per 4-connected MASK component (floods of
//     different mask components never interact), so one WAVEFRONT owns one component and runs the exact flood with
//     a wave-cooperative 64-ary min-heap keyed by (priority value, push age, pixel index): the 64 children of a node
//     are compared with one wave-wide argmin.  Ages are assigned in skimage's neighbour order (up, left, right,
//     down), so equal-priority plateaus are split exactly as the reference splits them.  The only freedom skimage
//     leaves to its binary heap's internal layout -- the order between SEED pixels with bit-identical priority --
//     is resolved by raster index here; the floods PROVE per component whether that order can reach the label map
//     ("uncertain age" propagation, see ws_flood_kernel) and count the components where it can in n_ambiguous.
//     When the count is not zero the map is re-flooded by ws_exact_kernel: a literal emulation of skimage's ONE
//     global binary heap (all markers pushed in raster order, strict-less sift-up / sift-down) -- slow, exact.
//   * gland / lumen: per instance crop (bounding box + conditional padding) -> elliptical dilation clipped to the
//     crop -> fill holes inside the crop -> paste with "later id wins" (atomicMax).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <string>

#include "../../include/cerberus_hip.h"
#include "cerb_dev.h"

int cerb_set_error(const std::string& m);  // cerb_api.hip
#define PP_OK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return cerb_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define KCHECK() PP_OK(hipGetLastError())

typedef unsigned long long u64;
typedef unsigned int u32;

static inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// =================================================================================================================
// Exclusive prefix sum over int32 (three-kernel, recursive on the block sums)
// =================================================================================================================
#define SCAN_ITEMS 1024  // per block: 256 threads x 4
// ROOTS: the scanned value is "pixel j is a root of the union-find map `in`" (in[j] == j), computed here instead of by a flag pass of its own
template <bool ROOTS>
__global__ __launch_bounds__(256) void scan_block_kernel(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ sums, int n) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SCAN_ITEMS + tid * 4;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (base + i < n) ? (ROOTS ? (in[base + i] == (int)(base + i) ? 1 : 0) : in[base + i]) : 0;
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int inc = tsum;  // inclusive scan across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + inc - tsum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (tid == 255 && sums) sums[blockIdx.x] = woff + inc;
}
__global__ void scan_add_kernel(int* __restrict__ out, const int* __restrict__ sums, int n) {
    const long long i = (long long)blockIdx.x * SCAN_ITEMS + threadIdx.x;
    const int add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long j = i + k * 256;
        if (j < n) out[j] += add;
    }
}
// tmp must hold at least n/1024 + n/1024^2 + ... + 4 ints.  total (sum of all elements) is written to *total_dev.
// block_offsets != nullptr: the last pass (adding every block's offset to its 1024 outputs: a read-modify-write of the whole array) is left to
// the READERS -- scan[j] = out[j] + (*block_offsets)[j >> 10], with *block_offsets == nullptr when one block held everything.
static int scan_exclusive(const int* in, int* out, int n, int* tmp, hipStream_t st, bool roots = false, const int** block_offsets = nullptr) {
    const int nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    auto kern = roots ? scan_block_kernel<true> : scan_block_kernel<false>;
    if (block_offsets) *block_offsets = nullptr;
    if (nb <= 1) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, st, in, out, (int*)nullptr, n);
        KCHECK();
        return 0;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, in, out, tmp, n);
    KCHECK();
    if (scan_exclusive(tmp, tmp, nb, tmp + ((nb + 3) & ~3), st)) return 1;
    if (block_offsets) {
        *block_offsets = tmp;
        return 0;
    }
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, st, out, tmp, n);
    KCHECK();
    return 0;
}

// ---- root bitmaps (round 5) -------------------------------------------------------------------------------------
// The roots of a labelling are ~1 pixel in 1000; the passes that only want THEM (ranks = scipy's label ids, per-component setup, heap offsets, work
// lists) used to scan the whole pixel map (0.12 - 0.19 ms each at 8192^2).  The last per-pixel pass of a labelling leaves one bit per pixel instead
// ("is a kept root": a wave ballot, one 8-byte store per 64 pixels), and those passes walk n / 64 words.
//   rank of root r = wpre[r >> 6] (+ block offset) + popcount(bits[r >> 6] below bit r & 63)
__global__ __launch_bounds__(256) void scan_block_popc_kernel(const u64* __restrict__ bits, int* __restrict__ out, int* __restrict__ sums, int nw) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SCAN_ITEMS + tid * 4;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (base + i < nw) ? __popcll(bits[base + i]) : 0;
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + inc - tsum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < nw) out[base + i] = run;
        run += v[i];
    }
    if (tid == 255 && sums) sums[blockIdx.x] = woff + inc;
}
// wpre[w] (+ (*block_offsets)[w >> 10] when that is not nullptr) = number of set bits in the words before w
static int scan_exclusive_popc(const u64* bits, int* wpre, int nw, int* tmp, hipStream_t st, const int** block_offsets) {
    const int nb = (nw + SCAN_ITEMS - 1) / SCAN_ITEMS;
    *block_offsets = nullptr;
    hipLaunchKernelGGL(scan_block_popc_kernel, dim3(nb), dim3(256), 0, st, bits, wpre, nb > 1 ? tmp : (int*)nullptr, nw);
    KCHECK();
    if (nb > 1) {
        if (scan_exclusive(tmp, tmp, nb, tmp + ((nb + 3) & ~3), st)) return 1;
        *block_offsets = tmp;
    }
    return 0;
}

// (row, column) of a linear pixel index p < 2^31 without the 64-bit integer division a per-pixel `p / W`, `p % W` costs (~100 VALU
// instructions): one fp64 multiply by a per-thread reciprocal, exact after a one-step fix-up.
__device__ __forceinline__ void pix_yx(long long p, int W, double invW, int& y, int& x) {
    int q = (int)((double)p * invW);
    int r = (int)(p - (long long)q * W);
    if (r < 0) {
        --q;
        r += W;
    } else if (r >= W) {
        ++q;
        r -= W;
    }
    y = q;
    x = r;
}

// =================================================================================================================
// Union-find connected components (4-connectivity) on an implicit grid
// =================================================================================================================
__device__ __forceinline__ int uf_find(const int* L, int x) {
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    bool done;
    do {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = (old == a);
            a = old;
        } else
            done = true;
    } while (!done);
}

// fg: foreground predicate bytes compared against `val` (lets one byte plane serve a mask and its complement).
// Run-based initialisation: a wave owns 64 consecutive pixels; every pixel is linked straight to the first pixel of its
// horizontal run inside that 64-pixel chunk (ballot + count-leading-zeros, no atomics), so the merge pass only needs one
// union per run and row pair instead of two per pixel.
__global__ void ccl_init_kernel(const uint8_t* __restrict__ fg, uint8_t val, int* __restrict__ L, int n, int W) {
    const int lane = threadIdx.x & 63;
    const double invW = 1.0 / (double)W;
    for (long long p0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; p0 < n; p0 += (long long)gridDim.x * blockDim.x) {
        const long long p = p0 + lane;
        const bool f = p < n && fg[p] == val;
        int y_, x_;
        pix_yx(p < n ? p : 0, W, invW, y_, x_);
        const bool link = f && lane > 0 && x_ != 0 && fg[p - 1] == val;  // joined to the previous lane's pixel
        const u64 starts = __ballot(f && !link);
        if (f) {
            const u64 below = starts & ((2ull << lane) - 1);  // run starts at or below this lane
            L[p] = (int)(p0 + 63 - __clzll((long long)below));
        } else if (p < n)
            L[p] = -1;
    }
}
__global__ void ccl_merge_kernel(const uint8_t* __restrict__ fg, uint8_t val, int* L, int H, int W) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (fg[p] != val) continue;
        int y_, x;
        pix_yx(p, W, invW, y_, x);
        const bool left = x > 0 && fg[p - 1] == val;
        const bool chunk_start = (p & 63) == 0 || !left;  // first pixel of its run inside the 64-pixel chunk
#ifndef PP_ABL_NOUNION
        if (left && (p & 63) == 0) uf_union(L, (int)p, (int)p - 1);  // run continues across the chunk boundary
#endif
        if (p >= W && fg[p - W] == val) {
            // one union per (upper run, lower run) pair: at the first column where they overlap either the lower run starts
            // here, or the upper run does (its left neighbour is background)
            const bool up_starts = !(x > 0 && fg[p - W - 1] == val);
#ifndef PP_ABL_NOUNION
            if (chunk_start || up_starts) uf_union(L, (int)p, (int)(p - W));
#else
            if ((chunk_start || up_starts) && L[p] == -12345) L[p] = 0;
#endif
        }
    }
}
__global__ void ccl_flatten_kernel(int* L, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (L[p] < 0) continue;
        L[p] = uf_find(L, (int)p);
    }
}
// flatten + component areas in one pass over L: area[root] += the length of each RUN of equal roots inside the wave (64 consecutive pixels of a
// row hold a few runs; the separate ccl_area_kernel pass re-read all of L for the same atomics).  `area` is zeroed by the caller.
__global__ void ccl_flatten_area_kernel(int* L, int* __restrict__ area, int n) {
    const int lane = threadIdx.x & 63;
    for (long long base = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + lane;
        int r = -1;
        if (p < n && L[p] >= 0) {
            r = uf_find(L, (int)p);
            L[p] = r;
        }
        const int prev = __shfl_up(r, 1);
        const bool head = r >= 0 && (lane == 0 || prev != r);
        const unsigned long long bounds = __ballot(head || r < 0);
        if (head) {
            const unsigned long long after = lane == 63 ? 0ull : (bounds >> (lane + 1));
            atomicAdd(&area[r], after ? __ffsll((long long)after) : 64 - lane);
        }
    }
}
static unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}
// Tile-local labelling (round 2): a workgroup labels a 64 x 32-pixel tile with its union-find in LDS (an LDS atomic is ~20x cheaper than
// the L2 round trips of a global uf_union, and a run-pair union of the global merge pass cost 2-3 us of serial latency), writes every
// pixel's tile root as a GLOBAL index, and only the pixel pairs across tile borders (1/32 of the rows, 1/64 of the columns) go through
// the global union-find.  Roots stay "smallest raster index of the component", so the labels are the ones ccl_init / ccl_merge gave.
constexpr int CT_W = 64, CT_H = 32;
// first slot of a tile's range in the root list: the pixels of the tile rows above it + of the tiles to its left in its own tile row
// (a tile gets as many slots as it has pixels, the list as many as the map)
__device__ __forceinline__ int tile_list_base(int tx0, int ty0, int H, int W) {
    const int rows = H - ty0 < CT_H ? H - ty0 : CT_H;
    return ty0 * W + tx0 * rows;
}
__device__ __forceinline__ int lds_find(volatile int* L, int x) {
    int p = L[x];
    while (p != x) {
        x = p;
        p = L[x];
    }
    return x;
}
__device__ __forceinline__ void lds_union(int* L, int a, int b) {
    bool done;
    do {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = (old == a);
            a = old;
        } else
            done = true;
    } while (!done);
}
// ROOTS (round 5): every tile-local root is appended to a compact list (one atomic per tile) and area[root] = the pixels of its tile-local set (counted
// per run in LDS) -- with the list flattened after the seam unions (ccl2_flatten_roots_kernel) L[L[p]] is a pixel's root and roots_area_merge_kernel
// adds the merged sets' counts up over the LIST: no pass over the pixel map for the flatten, none for the areas, no zero fill of `area`.
template <bool ROOTS>
__global__ __launch_bounds__(256) void ccl_tile_kernel(const uint8_t* __restrict__ fg, uint8_t val, int* __restrict__ L, int H, int W, int tiles_x, int n_tiles,
                                                       int* __restrict__ roots, int* __restrict__ cnt, int* __restrict__ area, int* __restrict__ colbuf = nullptr) {
    __shared__ int sl[CT_H * CT_W];
    __shared__ u64 smask[CT_H];
    __shared__ int scnt[ROOTS ? CT_H * CT_W : 1];
    __shared__ int swtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx0 = (tile % tiles_x) * CT_W, ty0 = (tile / tiles_x) * CT_H;
        const int x = tx0 + lane;
        // rows wave*8 .. +7: every pixel points at the first pixel of its horizontal run inside the tile (ballot, no atomics)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, y = ty0 + r;
            const bool f = x < W && y < H && fg[(long long)y * W + x] == val;
            const u64 m = __ballot(f);
            const u64 starts = m & ~(m << 1);
            sl[r * CT_W + lane] = f ? r * CT_W + 63 - __clzll((long long)(starts & ((2ull << lane) - 1))) : -1;
            if (ROOTS) scnt[r * CT_W + lane] = 0;
            if (lane == 0) smask[r] = m;
        }
        __syncthreads();
        // one union per pair of vertically adjacent runs: at the first column where they overlap
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i;
            if (r == 0) continue;
            const u64 m = smask[r], up = smask[r - 1];
            const u64 both = m & up;
            const u64 first = both & (~(m << 1) | ~(up << 1));  // the lower run starts here, or the upper one does
            if ((first >> lane) & 1) lds_union(sl, r * CT_W + lane, (r - 1) * CT_W + lane);
        }
        __syncthreads();
        u64 rmv[8] = {};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, y = ty0 + r;
            bool is_root = false;
            if (x < W && y < H) {
                const int l = sl[r * CT_W + lane];
                int g = -1;
                if (l >= 0) {
                    const int root = lds_find(sl, l);
                    g = (ty0 + (root >> 6)) * W + tx0 + (root & 63);
                    if (ROOTS) {
                        is_root = root == r * CT_W + lane;
                        const u64 mrow = smask[r];
                        if (((mrow & ~(mrow << 1)) >> lane) & 1ull) {  // first pixel of a run: the run's length goes to the set's counter
                            const u64 rest = ~(mrow >> lane);
                            atomicAdd(&scnt[root], rest ? __ffsll((long long)rest) - 1 : 64 - lane);
                        }
                    }
                }
                L[(long long)y * W + x] = g;
                // the tile's border columns, compactly: what the seam pass wants of this tile (seam_cols: [right-of-seam | left-of-seam][seam k][y])
                if (colbuf) {
                    if (lane == 0 && tx0 > 0) colbuf[(long long)(tx0 / CT_W - 1) * H + y] = g;
                    if (lane == CT_W - 1 && tx0 + CT_W < W) colbuf[(long long)(tiles_x - 1) * H + (long long)(tx0 / CT_W) * H + y] = g;
                }
            }
            if (ROOTS) rmv[i] = __ballot(is_root);
        }
        if (ROOTS) {
            __syncthreads();  // every set's counter is complete
            {   // the tile's roots go to the tile's OWN range of the list (tile_list_base: as many slots as the tile has pixels) and their number to
                // cnt[tile] -- a shared list head took one returning atomic per tile on ONE address: 33 k of them cost 0.1 ms, 131 k (one per wave) 0.4 ms
                int tot = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) tot += __popcll(rmv[i]);
                if (lane == 0) swtot[wave] = tot;
                __syncthreads();
                int base = tile_list_base(tx0, ty0, H, W);
                for (int w = 0; w < wave; ++w) base += swtot[w];
                if (tid == 0) cnt[tile] = swtot[0] + swtot[1] + swtot[2] + swtot[3];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = wave * 8 + i, y = ty0 + r;
                    if ((rmv[i] >> lane) & 1ull) {
                        roots[base + __popcll(rmv[i] & ((1ull << lane) - 1))] = y * W + x;
                        area[(long long)y * W + x] = scnt[r * CT_W + lane];
                    }
                    base += __popcll(rmv[i]);
                }
            }
            __syncthreads();  // (sl / scnt / swtot are rewritten by the next tile)
        } else {
            __syncthreads();
        }
    }
}
// The list kernels: one wave per tile walks the tile's range of the root list (cnt[tile] entries from tile_list_base).
// area[root] += the counts of the tile-local sets that were united into it (needs the flattened list: L[t] is t's root)
__global__ void roots_area_merge_kernel(const int* __restrict__ L, const int* __restrict__ roots, const int* __restrict__ cnt, int n_tiles, int tiles_x, int H,
                                        int W, int* __restrict__ area) {
    const int lane = threadIdx.x & 63, nwaves = gridDim.x * (blockDim.x >> 6);
    for (int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += nwaves) {
        const int c = cnt[tile], base = tile_list_base((tile % tiles_x) * CT_W, (tile / tiles_x) * CT_H, H, W);
        for (int k = lane; k < c; k += 64) {
            const int t = roots[base + k];
            const int f = L[t];
            if (f != t) {
                const int a = area[t];
                if (a) atomicAdd(&area[f], a);
            }
        }
    }
}
// unions across tile borders: vertical seams (x a multiple of 64: the run continues), horizontal seams (y a multiple of 32: one union per
// pair of tile-clipped runs, at the first column where they overlap)
// cols (round 6): the tile kernel's compact copy of its border columns -- [0, nv): the label RIGHT of seam k at row y (the right tile's first
// column), [nv, 2 nv): left of it (the left tile's last column); -1 = not `val`.  A vertical seam then reads two coalesced ints per pixel pair instead of two
// bytes and two labels out of four different 64-byte lines (4 B/px of HBM traffic for 3 % of the pixels: profiles/r06_postproc_nuclei_8192_bytes_per_pass.txt).
// What a column entry holds is the pixel's tile root as the tile pass left it: an ancestor of the pixel whatever unions have happened since -- the same
// starting point L[pixel] gives.
__global__ void ccl_seam_kernel(const uint8_t* __restrict__ fg, uint8_t val, int* L, int H, int W, int tiles_x, int tiles_y, const int* __restrict__ cols = nullptr) {
    const unsigned sx = (unsigned)(tiles_x - 1), sy = (unsigned)(tiles_y - 1);
    const unsigned nv = sx * (unsigned)H, nh = sy * (unsigned)W;  // < 2^31 / 32
    const unsigned total = nv + nh, stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x - lane; i0 < total; i0 += stride) {  // wave-uniform trip count (shuffles below)
        const unsigned i = i0 + lane;
        int a = -1, b = -1;
        int ca = -1, cb = -1;  // vertical seam through the compact columns: the two roots themselves
        if (i < nv && cols) {
            ca = cols[i];
            cb = cols[nv + i];
            if (ca < 0 || cb < 0) ca = cb = -1;
        } else if (i < nv) {  // consecutive threads walk down one seam (a row-major order was 2.6x slower)
            const unsigned k = i / (unsigned)H, y = i - k * (unsigned)H, x = (k + 1) * CT_W;
            const long long p = (long long)y * W + x;
            if (fg[p] == val && fg[p - 1] == val) {
                a = (int)p;
                b = (int)p - 1;
            }
        } else if (i < total) {  // consecutive threads: consecutive pixels of one seam row
            const unsigned j = i - nv;
            const unsigned k = j / (unsigned)W, x = j - k * (unsigned)W, y = (k + 1) * CT_H;
            const long long p = (long long)y * W + x;
            if (fg[p] == val && fg[p - W] == val) {
                const bool edge = (x % CT_W) == 0;
                const bool low_starts = edge || fg[p - 1] != val, up_starts = edge || fg[p - W - 1] != val;
                if (low_starts || up_starts) {
                    a = (int)p;
                    b = (int)(p - W);
                }
            }
        }
        // a component that crosses many seams (the background of a marker image is ONE component) would hammer one root with atomics:
        // the tile pass left every pixel pointing at its tile root, so neighbouring lanes mostly ask for the same (root, root) pair --
        // only the first lane of each run of equal pairs performs the union
        const int ra = ca >= 0 ? ca : a >= 0 ? L[a] : -1, rb = ca >= 0 ? cb : b >= 0 ? L[b] : -2;
        const int pa = __shfl_up(ra, 1), pb = __shfl_up(rb, 1);
        if ((a >= 0 || ca >= 0) && ra != rb && !(lane > 0 && pa == ra && pb == rb)) uf_union(L, ra, rb);
    }
}
static int ccl_run(const uint8_t* fg, uint8_t val, int* L, int H, int W, hipStream_t st, int* area = nullptr) {
    const int n = H * W;
#ifdef PP_CCL_GLOBAL  // round 1's labelling: run-based init + one global union per run pair
    hipLaunchKernelGGL(ccl_init_kernel, dim3(grid_for(n)), dim3(256), 0, st, fg, val, L, n, W);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(grid_for(n)), dim3(256), 0, st, fg, val, L, H, W);
#else
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H;
    const int n_tiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(ccl_tile_kernel<false>, dim3(n_tiles < 256 * 16 ? n_tiles : 256 * 16), dim3(256), 0, st, fg, val, L, H, W, tiles_x, n_tiles, (int*)nullptr,
                       (int*)nullptr, (int*)nullptr);
    const long long seams = (long long)(tiles_x - 1) * H + (long long)(tiles_y - 1) * W;
    if (seams > 0) hipLaunchKernelGGL(ccl_seam_kernel, dim3(grid_for(seams)), dim3(256), 0, st, fg, val, L, H, W, tiles_x, tiles_y);
#endif
    if (area) hipLaunchKernelGGL(ccl_flatten_area_kernel, dim3(grid_for(n)), dim3(256), 0, st, L, area, n);
    else hipLaunchKernelGGL(ccl_flatten_kernel, dim3(grid_for(n)), dim3(256), 0, st, L, n);
    KCHECK();
    return 0;
}

// =================================================================================================================
// Two-colour labelling of the nuclei MARKER image (round 5).  loader/postproc.py:370-377 is
//     label(inner > 0.5) -> remove_small_objects(4) -> binary_fill_holes -> label
// which round 4 ran as THREE union-find labellings (markers; their background, for the holes; the filled markers).  Here the marker image is
// labelled ONCE in both colours (a pixel is united with its 4-neighbours of the SAME colour), and the two later steps are edits of that one forest:
//   * a foreground component smaller than min_size turns into background: its pixels are united with their background neighbours;
//   * a background component that does not touch the image border is a hole: its pixels turn into foreground and are united with their
//     foreground neighbours (this is also what merges two markers that only touched diagonally around a hole).
// Roots stay "smallest raster index of the set", so the rank of a final root among all roots is scipy's label id as before.
// =================================================================================================================
// `roots` / `cnt`: every pixel that is the root of its tile-local set goes to its tile's range of a root list (tile_list_base; cnt[tile] entries).  All later unions link
// roots under roots, so the nodes of the forest above the pixel level are exactly these: flattening the LIST (ccl2_flatten_roots_kernel) makes
// L[L[p]] the set's root for every pixel p -- two loads, no pointer chase, no pass that walks 67 M background pixels up to one giant root.
__global__ __launch_bounds__(256) void ccl2_tile_kernel(const uint8_t* __restrict__ fg, int* __restrict__ L, int H, int W, int tiles_x, int n_tiles,
                                                        int* __restrict__ roots, int* __restrict__ cnt, int* __restrict__ area, int* __restrict__ colbuf = nullptr) {
    __shared__ int swtot[4];
    __shared__ int sl[CT_H * CT_W];
    __shared__ int scnt[CT_H * CT_W];  // FOREGROUND pixels of every tile-local set (background sets stay at 0: ccl2_drop_small_kernel relies on it)
    __shared__ u64 sc[CT_H], sv[CT_H], ss[CT_H];  // per row: colour bits, valid bits, run starts
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx0 = (tile % tiles_x) * CT_W, ty0 = (tile / tiles_x) * CT_H;
        const int x = tx0 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, y = ty0 + r;
            const bool v = x < W && y < H;
            const bool f = v && fg[(long long)y * W + x] != 0;
            const u64 c = __ballot(f), vm = __ballot(v);
            const u64 starts = ((c ^ (c << 1)) | 1ull) & vm;  // a run starts where the colour changes (pixels beyond the image sit at the right end only)
            sl[r * CT_W + lane] = v ? r * CT_W + 63 - __clzll((long long)(starts & ((2ull << lane) - 1))) : -1;
            scnt[r * CT_W + lane] = 0;
            if (lane == 0) {
                sc[r] = c;
                sv[r] = vm;
                ss[r] = starts;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i;
            if (r == 0) continue;
            const u64 both = ~(sc[r] ^ sc[r - 1]) & sv[r] & sv[r - 1];
            const u64 first = both & (ss[r] | ss[r - 1]);  // the lower run starts here, or the upper one does
            if ((first >> lane) & 1) lds_union(sl, r * CT_W + lane, (r - 1) * CT_W + lane);
        }
        __syncthreads();
        u64 rmv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, y = ty0 + r;
            bool is_root = false;
            if (x < W && y < H) {
                const int root = lds_find(sl, sl[r * CT_W + lane]);
                const int g = (ty0 + (root >> 6)) * W + tx0 + (root & 63);
                L[(long long)y * W + x] = g;
                is_root = root == r * CT_W + lane;
                const u64 c = sc[r];
                if (colbuf) {  // border columns for the seam pass (see ccl_seam_kernel): the tile root, the pixel's colour in bit 31 (indices stay below 2^31)
                    const int gc = g | (int)(((c >> lane) & 1ull) << 31);
                    if (lane == 0 && tx0 > 0) colbuf[(long long)(tx0 / CT_W - 1) * H + y] = gc;
                    if (lane == CT_W - 1 && tx0 + CT_W < W) colbuf[(long long)(tiles_x - 1) * H + (long long)(tx0 / CT_W) * H + y] = gc;
                }
                if (((c & ss[r]) >> lane) & 1ull) {  // first pixel of a foreground run: its length (the colour bits beyond the image are 0)
                    const u64 rest = ~(c >> lane);
                    atomicAdd(&scnt[root], rest ? __ffsll((long long)rest) - 1 : 64 - lane);
                }
            }
            rmv[i] = __ballot(is_root);
        }
        __syncthreads();  // every set's counter is complete
        {   // the tile's roots go to the tile's OWN range of the list (tile_list_base: as many slots as the tile has pixels) and their number to
            // cnt[tile] -- a shared list head took one returning atomic per tile on ONE address: 33 k of them cost 0.1 ms, 131 k (one per wave) 0.4 ms
            int tot = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) tot += __popcll(rmv[i]);
            if (lane == 0) swtot[wave] = tot;
            __syncthreads();
            int base = tile_list_base(tx0, ty0, H, W);
            for (int w = 0; w < wave; ++w) base += swtot[w];
            if (tid == 0) cnt[tile] = swtot[0] + swtot[1] + swtot[2] + swtot[3];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wave * 8 + i, y = ty0 + r;
                if ((rmv[i] >> lane) & 1ull) {
                    roots[base + __popcll(rmv[i] & ((1ull << lane) - 1))] = y * W + x;
                    area[(long long)y * W + x] = scnt[r * CT_W + lane];
                }
                base += __popcll(rmv[i]);
            }
        }
        __syncthreads();  // (sl / scnt / swtot are rewritten by the next tile)
    }
}
// L[r] = root of r for every node of the forest above the pixel level (see ccl2_tile_kernel)
// zero (optional): an int map that is only ever read at roots (the border flags of the two-colour labelling) is cleared at the LIST's entries here -- every
// root there will ever be is one of them -- instead of by a whole-map fill (4 B/px of a call's 82: round 6)
__global__ void ccl2_flatten_roots_kernel(int* L, const int* __restrict__ roots, const int* __restrict__ cnt, int n_tiles, int tiles_x, int H, int W,
                                          int* __restrict__ zero = nullptr) {
    const int lane = threadIdx.x & 63, nwaves = gridDim.x * (blockDim.x >> 6);
    for (int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += nwaves) {
        const int c = cnt[tile], base = tile_list_base((tile % tiles_x) * CT_W, (tile / tiles_x) * CT_H, H, W);
        for (int k = lane; k < c; k += 64) {
            const int r = roots[base + k];
            if (zero) zero[r] = 0;
            const int f = uf_find(L, r);
            if (f != r) __hip_atomic_store(&L[r], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Seam unions with CACHED finds (0.27 -> 0.17 ms for the marker image's seams at 8192^2; CERB_PP_SEAM_STRICT=1 keeps uf_union): uf_find's agent-scope
// loads go past the L2 on every step; a plain load may be
// stale, but a stale parent is still an ancestor (links only ever move towards smaller indices inside one set) and the returning atomicMin that
// closes a union validates the root it acts on -- a root that was no root any more hands back its real parent and the loop continues from there.
// Every step also pulls the node it leaves one level up (path halving by a non-returning atomicMin).
__device__ __forceinline__ int uf_find_relaxed(int* L, int x) {
    int p = L[x];
    while (p != x) {
        const int gp = L[p];
        if (gp != p) atomicMin(&L[x], gp);
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union_relaxed(int* L, int a, int b) {
    bool done;
    do {
        a = uf_find_relaxed(L, a);
        b = uf_find_relaxed(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = (old == a);
            a = old;
        } else
            done = true;
    } while (!done);
}
template <bool RELAXED>
__global__ void ccl2_seam_kernel(const uint8_t* __restrict__ fg, int* L, int H, int W, int tiles_x, int tiles_y, const int* __restrict__ cols = nullptr) {
    const unsigned sx = (unsigned)(tiles_x - 1), sy = (unsigned)(tiles_y - 1);
    const unsigned nv = sx * (unsigned)H, nh = sy * (unsigned)W;
    const unsigned total = nv + nh, stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x - lane; i0 < total; i0 += stride) {
        const unsigned i = i0 + lane;
        int a = -1, b = -1;
        int ca = -1, cb = -1;  // vertical seam through the compact columns (two colours: bit 31)
        if (i < nv && cols) {
            const int va = cols[i], vb = cols[nv + i];
            if ((va ^ vb) >= 0) {  // same colour
                ca = va & 0x7fffffff;
                cb = vb & 0x7fffffff;
            }
        } else if (i < nv) {
            const unsigned k = i / (unsigned)H, y = i - k * (unsigned)H, x = (k + 1) * CT_W;
            const long long p = (long long)y * W + x;
            if ((fg[p] != 0) == (fg[p - 1] != 0)) {
                a = (int)p;
                b = (int)p - 1;
            }
        } else if (i < total) {
            const unsigned j = i - nv;
            const unsigned k = j / (unsigned)W, x = j - k * (unsigned)W, y = (k + 1) * CT_H;
            const long long p = (long long)y * W + x;
            const bool c = fg[p] != 0;
            if (c == (fg[p - W] != 0)) {
                const bool edge = (x % CT_W) == 0;
                const bool low_starts = edge || (fg[p - 1] != 0) != c, up_starts = edge || (fg[p - W - 1] != 0) != c;
                if (low_starts || up_starts) {
                    a = (int)p;
                    b = (int)(p - W);
                }
            }
        }
        const int ra = ca >= 0 ? ca : a >= 0 ? L[a] : -1, rb = ca >= 0 ? cb : b >= 0 ? L[b] : -2;  // (see ccl_seam_kernel: one union per run of equal root pairs)
        const int pa = __shfl_up(ra, 1), pb = __shfl_up(rb, 1);
        if ((a >= 0 || ca >= 0) && ra != rb && !(lane > 0 && pa == ra && pb == rb)) {
            if (RELAXED) uf_union_relaxed(L, ra, rb);
            else uf_union(L, ra, rb);
        }
    }
}
// foreground components below min_size turn into background and join the background sets around them
__global__ void ccl2_drop_small_kernel(uint8_t* m, int* L, const int* __restrict__ area, int min_size, int H, int W) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (!m[p]) continue;
        const int r = L[L[p]];  // the component's root as the area merge saw it -- or, once this small component has been united away, a
        if (area[r] >= min_size) continue;  // background index, whose area is 0: "small" either way
        int y, x;
        pix_yx(p, W, invW, y, x);
        m[p] = 0;
        // (a neighbour that reads as background is background or a pixel of this same small component: uniting with either is right)
        if (x > 0 && !m[p - 1]) uf_union(L, (int)p, (int)p - 1);
        if (x < W - 1 && !m[p + 1]) uf_union(L, (int)p, (int)p + 1);
        if (y > 0 && !m[p - W]) uf_union(L, (int)p, (int)(p - W));
        if (y < H - 1 && !m[p + W]) uf_union(L, (int)p, (int)(p + W));
    }
}
__global__ void ccl2_mark_border_kernel(const uint8_t* __restrict__ m, const int* L, int* __restrict__ border, int H, int W) {
    const int per = 2 * (H + W);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        long long p;
        if (i < W) p = i;
        else if (i < 2 * W) p = (long long)(H - 1) * W + (i - W);
        else if (i < 2 * W + H) p = (long long)(i - 2 * W) * W;
        else p = (long long)(i - 2 * W - H) * W + (W - 1);
        if (!m[p]) border[uf_find(L, (int)p)] = 1;
    }
}
// background sets that do not reach the border are holes: their pixels become foreground and join the foreground sets around them
// (border[] is 1 only at the roots of border-touching background sets, which nothing is united with here; every other index reads 0)
__global__ void ccl2_fill_kernel(uint8_t* m, int* L, const int* __restrict__ border, int H, int W) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (m[p]) continue;
        if (border[uf_find(L, (int)p)]) continue;
        int y, x;
        pix_yx(p, W, invW, y, x);
        m[p] = 1;
        // (a neighbour that reads as foreground is foreground or a pixel of this same hole that was filled a moment ago: uniting with either is right)
        if (x > 0 && m[p - 1]) uf_union(L, (int)p, (int)p - 1);
        if (x < W - 1 && m[p + 1]) uf_union(L, (int)p, (int)p + 1);
        if (y > 0 && m[p - W]) uf_union(L, (int)p, (int)(p - W));
        if (y < H - 1 && m[p + W]) uf_union(L, (int)p, (int)(p + W));
    }
}
// final marker labels: root for foreground pixels, -1 for the background
__global__ void ccl2_final_kernel(int* L, const uint8_t* __restrict__ m, int n) {
    // (flattened roots: L[L[p]] is the root; a foreground node that other pixels still point at keeps the value it already holds)
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) L[p] = m[p] ? L[L[p]] : -1;
}
// the same, and bits[p >> 6] |= "p is the root of a marker" (whole waves step together: the ballot covers 64 consecutive pixels)
__global__ void ccl2_final_bits_kernel(int* L, const uint8_t* __restrict__ m, int n, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    for (long long base = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + lane;
        int r = -1;
        if (p < n) {
            r = m[p] ? L[L[p]] : -1;
            L[p] = r;
        }
        const u64 b = __ballot(r == (int)p && p < n);
        if (lane == 0) bits[base >> 6] = b;
    }
}
// ---- four pixels per thread (see nuc_marker_out4_kernel for why) -------------------------------------------------
// 16 consecutive lanes hold 64 consecutive pixels: their 4-bit root flags make one bitmap word (every lane of the wave calls this)
__device__ __forceinline__ void store_root_bits4(u64* __restrict__ bits, long long q, long long nq, unsigned nib, int lane) {
    u64 v = (u64)nib << (4 * (lane & 15));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    v |= __shfl_xor(v, 8);
    if ((lane & 15) == 0 && q < nq) bits[q >> 4] = v;
}
__global__ void ccl2_drop_small4_kernel(uint32_t* m4, int4* L4, const int* __restrict__ area, int min_size, int H, int W) {
    const long long nq = (long long)H * W / 4;
    const int qw = W / 4;
    const double invQ = 1.0 / (double)qw;
    uint8_t* m = (uint8_t*)m4;
    int* L = (int*)L4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        const uint32_t mw = m4[q];
        if (!mw) continue;
        const int4 l = L4[q];
        const int t[4] = {l.x, l.y, l.z, l.w};
        int pt = -1;
        bool small = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!((mw >> (8 * i)) & 0xffu)) continue;
            if (t[i] != pt) {  // (pixels of one tile-local set share the decision; see ccl2_drop_small_kernel for the races that do not matter)
                pt = t[i];
                small = area[L[pt]] < min_size;
            }
            if (!small) continue;
            int y, xq;
            pix_yx(q, qw, invQ, y, xq);
            const int x = xq * 4 + i;
            const long long p = q * 4 + i;
            m[p] = 0;
            if (x > 0 && !m[p - 1]) uf_union(L, (int)p, (int)p - 1);
            if (x < W - 1 && !m[p + 1]) uf_union(L, (int)p, (int)p + 1);
            if (y > 0 && !m[p - W]) uf_union(L, (int)p, (int)(p - W));
            if (y < H - 1 && !m[p + W]) uf_union(L, (int)p, (int)(p + W));
        }
    }
}
__global__ void ccl2_fill4_kernel(uint32_t* m4, int4* L4, const int* __restrict__ border, int H, int W) {
    const long long nq = (long long)H * W / 4;
    const int qw = W / 4;
    const double invQ = 1.0 / (double)qw;
    uint8_t* m = (uint8_t*)m4;
    int* L = (int*)L4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        const uint32_t mw = m4[q];
        if (mw == 0x01010101u) continue;
        const int4 l = L4[q];
        const int t[4] = {l.x, l.y, l.z, l.w};
        int pt = -1;
        bool hole = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((mw >> (8 * i)) & 0xffu) continue;
            if (t[i] != pt) {
                pt = t[i];
                hole = !border[uf_find(L, pt)];
            }
            if (!hole) continue;
            int y, xq;
            pix_yx(q, qw, invQ, y, xq);
            const int x = xq * 4 + i;
            const long long p = q * 4 + i;
            m[p] = 1;
            if (x > 0 && m[p - 1]) uf_union(L, (int)p, (int)p - 1);
            if (x < W - 1 && m[p + 1]) uf_union(L, (int)p, (int)p + 1);
            if (y > 0 && m[p - W]) uf_union(L, (int)p, (int)(p - W));
            if (y < H - 1 && m[p + W]) uf_union(L, (int)p, (int)(p + W));
        }
    }
}
// The same fill, gated per TILE (round 6): a hole is a background set that does not reach the border, and every pixel of it belongs to a tile-local set whose
// tile root sits in that tile's range of the root list -- so a tile none of whose list entries is background with an unflagged root holds no hole pixel and is
// skipped without touching its labels (holes are rare: 5.0 -> ~0.5 B/px of a call, the pass 0.155 -> ~0.03 ms at 8192^2).  One wave per tile; the list is flat
// here (ccl2_flatten_roots_kernel ran after the last unions), so L[entry] is the entry's root.
__global__ void ccl2_fill_tiles4_kernel(uint32_t* m4, int4* L4, const int* __restrict__ border, const int* __restrict__ roots, const int* __restrict__ cnt, int n_tiles,
                                        int tiles_x, int H, int W) {
    const int lane = threadIdx.x & 63, nwaves = gridDim.x * (blockDim.x >> 6);
    const int qw = W / 4;
    uint8_t* m = (uint8_t*)m4;
    int* L = (int*)L4;
    for (int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += nwaves) {
        const int tx0 = (tile % tiles_x) * CT_W, ty0 = (tile / tiles_x) * CT_H;
        const int c = cnt[tile], base = tile_list_base(tx0, ty0, H, W);
        bool any = false;
        for (int k0 = 0; k0 < c; k0 += 64) {  // (wave-uniform trip count)
            const int k = k0 + lane;
            bool h = false;
            if (k < c) {
                const int r = roots[base + k];
                h = !m[r] && !border[L[r]];
            }
            any = any || (__ballot(h) != 0ull);
        }
        if (!any) continue;
        const int rows = H - ty0 < CT_H ? H - ty0 : CT_H, qpr = (W - tx0 < CT_W ? W - tx0 : CT_W) / 4;  // quads per tile row (W % 4 == 0)
        for (int idx = lane; idx < rows * qpr; idx += 64) {
            const int y = ty0 + idx / qpr, xq = tx0 / 4 + idx % qpr;
            const long long q = (long long)y * qw + xq;
            const uint32_t mw = m4[q];
            if (mw == 0x01010101u) continue;
            const int4 l = L4[q];
            const int t[4] = {l.x, l.y, l.z, l.w};
            int pt = -1;
            bool hole = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((mw >> (8 * i)) & 0xffu) continue;
                if (t[i] != pt) {
                    pt = t[i];
                    hole = !border[uf_find(L, pt)];
                }
                if (!hole) continue;
                const int x = xq * 4 + i;
                const long long p = q * 4 + i;
                m[p] = 1;
                if (x > 0 && m[p - 1]) uf_union(L, (int)p, (int)p - 1);
                if (x < W - 1 && m[p + 1]) uf_union(L, (int)p, (int)p + 1);
                if (y > 0 && m[p - W]) uf_union(L, (int)p, (int)(p - W));
                if (y < H - 1 && m[p + W]) uf_union(L, (int)p, (int)(p + W));
            }
        }
    }
}
__global__ void ccl2_final_bits4_kernel(int4* L4, const uint32_t* __restrict__ m4, long long nq, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int* L = (const int*)L4;
    for (long long q0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; q0 < nq; q0 += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + lane;
        unsigned nib = 0;
        if (q < nq) {
            const uint32_t mw = m4[q];
            int r[4] = {-1, -1, -1, -1};
            if (mw) {
                const int4 l = L4[q];
                const int t[4] = {l.x, l.y, l.z, l.w};
                int pt = -1, pr = -1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!((mw >> (8 * i)) & 0xffu)) continue;
                    if (t[i] != pt) {
                        pt = t[i];
                        pr = L[pt];
                    }
                    r[i] = pr;
                    if (pr == (int)(q * 4 + i)) nib |= 1u << i;
                }
                L4[q] = make_int4(r[0], r[1], r[2], r[3]);  // (round 6: quads without a marker pixel keep whatever they held -- nuc_marker_out4_kernel asks the marker bytes first)
            }
        }
        store_root_bits4(bits, q, nq, nib, lane);
    }
}
// mask side: m = "the pixel's component has at least min_size pixels", L[p] <- its root (the list is flat: L[L[p]]), bitmap of the kept roots
__global__ void apply_min_area_bits4_kernel(uint32_t* __restrict__ m4, int4* L4, const int* __restrict__ area, int min_size, long long nq, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int* L = (const int*)L4;
    for (long long q0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; q0 < nq; q0 += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + lane;
        unsigned nib = 0;
        if (q < nq && m4[q]) {  // (round 6: the eroded mask's own bytes say where the labelling has labels -- the int32 map is read for those quads only)
            const int4 l = L4[q];
            uint32_t mw = 0;
            if (l.x >= 0 || l.y >= 0 || l.z >= 0 || l.w >= 0) {
                const int t[4] = {l.x, l.y, l.z, l.w};
                int r[4] = {-1, -1, -1, -1};
                int pt = -1, pr = -1;
                bool keep = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (t[i] < 0) continue;
                    if (t[i] != pt) {
                        pt = t[i];
                        pr = L[pt];
                        keep = area[pr] >= min_size;
                    }
                    r[i] = pr;
                    if (keep) {
                        mw |= 1u << (8 * i);
                        if (pr == (int)(q * 4 + i)) nib |= 1u << i;
                    }
                }
                L4[q] = make_int4(r[0], r[1], r[2], r[3]);
            }
            m4[q] = mw;
        }
        store_root_bits4(bits, q, nq, nib, lane);
    }
}
// roots: H * W ints (every tile its own range), n_roots: one int per tile
static int markers_two_colour(uint8_t* mrk, int* L, int* area, int* border, int* roots, int* n_roots, int min_size, int H, int W, hipStream_t st,
                              u64* root_bits = nullptr, bool wide = false, int* colbuf = nullptr) {
    const int n = H * W;
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, n_tiles = tiles_x * tiles_y;
    const unsigned g = grid_for(n);
    // (area: written by the tile kernel at every tile-local root, read at roots only -- no zero fill; border: read at roots only as well, cleared at the
    //  root list's entries by the first flatten pass below)
    hipLaunchKernelGGL(ccl2_tile_kernel, dim3(n_tiles < 256 * 16 ? n_tiles : 256 * 16), dim3(256), 0, st, mrk, L, H, W, tiles_x, n_tiles, roots, n_roots, area, colbuf);
    const long long seams = (long long)(tiles_x - 1) * H + (long long)(tiles_y - 1) * W;
    static const bool seam_relaxed = cerb_dev_getenv("CERB_PP_SEAM_STRICT") == nullptr;
    if (seams > 0) hipLaunchKernelGGL(seam_relaxed ? ccl2_seam_kernel<true> : ccl2_seam_kernel<false>, dim3(grid_for(seams)), dim3(256), 0, st, mrk, L, H, W, tiles_x, tiles_y, (const int*)colbuf);
    const unsigned gl = nblk(n_tiles, 4) < 4096 ? nblk(n_tiles, 4) : 4096;  // one wave per tile
    auto flatten_roots = [&]() { hipLaunchKernelGGL(ccl2_flatten_roots_kernel, dim3(gl), dim3(256), 0, st, L, (const int*)roots, (const int*)n_roots, n_tiles, tiles_x, H, W, (int*)nullptr); };
    hipLaunchKernelGGL(ccl2_flatten_roots_kernel, dim3(gl), dim3(256), 0, st, L, (const int*)roots, (const int*)n_roots, n_tiles, tiles_x, H, W, border);
    hipLaunchKernelGGL(roots_area_merge_kernel, dim3(gl), dim3(256), 0, st, (const int*)L, (const int*)roots, (const int*)n_roots, n_tiles, tiles_x, H, W, area);
    wide = wide && W % 4 == 0;
    const unsigned g4 = grid_for(n / 4);
    if (wide) hipLaunchKernelGGL(ccl2_drop_small4_kernel, dim3(g4), dim3(256), 0, st, (uint32_t*)mrk, (int4*)L, area, min_size, H, W);
    else hipLaunchKernelGGL(ccl2_drop_small_kernel, dim3(g), dim3(256), 0, st, mrk, L, area, min_size, H, W);
    flatten_roots();
    hipLaunchKernelGGL(ccl2_mark_border_kernel, dim3(nblk(2 * (H + W), 256)), dim3(256), 0, st, mrk, L, border, H, W);
    static const bool fill_by_tile = cerb_dev_getenv("CERB_PP_FILL_WHOLE_MAP") == nullptr;  // developers' build: =1 keeps round 5's pass over every quad
    if (wide && fill_by_tile) hipLaunchKernelGGL(ccl2_fill_tiles4_kernel, dim3(gl), dim3(256), 0, st, (uint32_t*)mrk, (int4*)L, border, (const int*)roots, (const int*)n_roots, n_tiles, tiles_x, H, W);
    else if (wide) hipLaunchKernelGGL(ccl2_fill4_kernel, dim3(g4), dim3(256), 0, st, (uint32_t*)mrk, (int4*)L, border, H, W);
    else hipLaunchKernelGGL(ccl2_fill_kernel, dim3(g), dim3(256), 0, st, mrk, L, border, H, W);
    flatten_roots();
    if (root_bits && wide) hipLaunchKernelGGL(ccl2_final_bits4_kernel, dim3(g4), dim3(256), 0, st, (int4*)L, (const uint32_t*)mrk, (long long)n / 4, root_bits);
    else if (root_bits) hipLaunchKernelGGL(ccl2_final_bits_kernel, dim3(g), dim3(256), 0, st, L, mrk, n, root_bits);
    else hipLaunchKernelGGL(ccl2_final_kernel, dim3(g), dim3(256), 0, st, L, mrk, n);
    KCHECK();
    return 0;
}

// area[root] += 1 for every labelled pixel (wave-aggregated when the whole wave sits in one component)
__global__ void ccl_area_kernel(const int* __restrict__ L, int* __restrict__ area, int n) {
    for (long long p0 = blockIdx.x * (long long)blockDim.x; p0 < n; p0 += (long long)gridDim.x * blockDim.x) {
        const long long p = p0 + threadIdx.x;
        const int r = (p < n) ? L[p] : -1;
        const int r0 = __shfl(r, __ffsll((long long)__ballot(r >= 0)) - 1);
        const u64 same = __ballot(r >= 0 && r == r0);
        if (r >= 0) {
            if (r == r0) {
                if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&area[r], __popcll(same));
            } else
                atomicAdd(&area[r], 1);
        }
    }
}
// flag[p] = 1 for roots of components with area >= min_size (0 otherwise / non-roots)
__global__ void ccl_keep_roots_kernel(const int* __restrict__ L, const int* __restrict__ area, int min_size, int* __restrict__ flag, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        flag[p] = (L[p] == (int)p && area[p] >= min_size) ? 1 : 0;
}
// out[p] = 1 + rank[root] for pixels of kept components (rank = exclusive scan of the root flags), 0 elsewhere
__global__ void ccl_relabel_kernel(const int* __restrict__ L, const int* __restrict__ flag, const int* __restrict__ rank, int* __restrict__ out, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int r = L[p];
        out[p] = (r >= 0 && flag[r]) ? rank[r] + 1 : 0;
    }
}
// the watershed's start map in one pass: out[p] = mask ? 1 + rank[root of p in the marker labelling] : 0 (rank = exclusive scan of "is a root");
// replaces the flag pass, the relabel pass into a marker map and the mask pass over it
// (boff: the scan's per-1024 block offsets, left unadded by scan_exclusive(..., &boff); nullptr = already complete)
__global__ void nuc_marker_out_kernel(const int* __restrict__ L, const int* __restrict__ rank, const int* __restrict__ boff, const uint8_t* __restrict__ mask,
                                      int* __restrict__ out, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int r = L[p];
        out[p] = (r >= 0 && mask[p]) ? rank[r] + (boff ? boff[r / SCAN_ITEMS] : 0) + 1 : 0;
    }
}
// the same from the root bitmap: label = 1 + number of roots before the pixel's root in raster order.
// Also the smallest / largest marker label inside every mask component (lmin / lmax at the component's root LA[p]; roots_setup_kernel has initialised
// them): one pair of atomics per RUN of equal (label, component) inside a wave, and only where a plain read does not already cover the label (the
// extremes are monotone, a stale read costs an atomic, never a miss).  Taken over all labelled pixels this equals the extremes over the SEEDS (labelled
// pixels with an unlabelled mask neighbour) in every component that has an unlabelled pixel at all: a marker none of whose pixels touches an unlabelled
// mask pixel has only itself and non-mask around it, i.e. it IS its component.  Knowing them here lets ws_seed_kernel skip the single-label components.
__global__ void nuc_marker_out_bits_kernel(const int* __restrict__ L, const u64* __restrict__ bits, const int* __restrict__ wpre, const int* __restrict__ boff,
                                           const uint8_t* __restrict__ mask, int* __restrict__ out, int n, const int* __restrict__ LA, int* lmin, int* lmax) {
    const int lane = threadIdx.x & 63;
    for (long long base = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + lane;
        int v = 0, ra = -1;
        if (p < n) {
            const int r = L[p];
            if (r >= 0 && mask[p]) {
                const int w = r >> 6;
                v = wpre[w] + (boff ? boff[w / SCAN_ITEMS] : 0) + __popcll(bits[w] & ((1ull << (r & 63)) - 1ull)) + 1;
                ra = LA[p];
            }
            out[p] = v;
        }
        const int pv = __shfl_up(v, 1), pr = __shfl_up(ra, 1);
        if (v > 0 && (lane == 0 || pv != v || pr != ra)) {
            const volatile int* vmin = lmin + ra;
            const volatile int* vmax = lmax + ra;
            if (v < *vmin) atomicMin(lmin + ra, v);
            if (v > *vmax) atomicMax(lmax + ra, v);
        }
    }
}
// Four pixels per thread (W % 4 == 0, 16-byte aligned maps).  The per-pixel passes of this file are LATENCY-bound, not bandwidth-bound: 8192 resident
// waves x one dependent chain of 3 - 4 loads (~3 us under load) per 64 pixels = 0.3 - 0.5 ms per pass over 67 Mpx, whatever the bytes.  One 16-byte load
// per map and thread instead of four dword loads quarters the chains per pixel, and the common case (background) ends after the first load.
__global__ void nuc_marker_out4_kernel(const int4* __restrict__ L4, const u64* __restrict__ bits, const int* __restrict__ wpre, const int* __restrict__ boff,
                                       const uint32_t* __restrict__ mask4, int4* __restrict__ out4, long long nq, const int4* __restrict__ LA4, int* lmin,
                                       int* lmax, const uint32_t* __restrict__ mrk4) {
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        // a start-map pixel is a marker pixel inside the mask: both byte maps are asked before any int32 map (the markers' map holds roots only in quads
        // that have a marker pixel: ccl2_final_bits4_kernel)
        const uint32_t m = mask4[q] & (mrk4[q] * 0xffu);
        if (!m) {
            out4[q] = make_int4(0, 0, 0, 0);
            continue;
        }
        const int4 l = L4[q];
        const int4 a = LA4[q];  // (same level of the load chain as L4: only labelled pixels want it, but waiting for the labels to ask costs a round trip)
        const int r[4] = {l.x, l.y, l.z, l.w};
        int v[4];
        int pr = -1, pv = 0;
        bool any = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = 0;
            if (r[i] >= 0 && ((m >> (8 * i)) & 0xffu)) {
                if (r[i] != pr) {
                    const int w = r[i] >> 6;
                    pv = wpre[w] + (boff ? boff[w / SCAN_ITEMS] : 0) + __popcll(bits[w] & ((1ull << (r[i] & 63)) - 1ull)) + 1;
                    pr = r[i];
                }
                v[i] = pv;
                any = true;
            }
        }
        out4[q] = make_int4(v[0], v[1], v[2], v[3]);
        if (!any) continue;
        const int ra[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (v[i] > 0 && (i == 0 || v[i] != v[i - 1] || ra[i] != ra[i - 1])) {
                const volatile int* vmin = lmin + ra[i];
                const volatile int* vmax = lmax + ra[i];
                if (v[i] < *vmin) atomicMin(lmin + ra[i], v[i]);
                if (v[i] > *vmax) atomicMax(lmax + ra[i], v[i]);
            }
    }
}
__global__ void count_roots_from_bits_kernel(const u64* __restrict__ bits, const int* __restrict__ wpre, const int* __restrict__ boff, int nw, int* __restrict__ out,
                                             const int* __restrict__ any) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *out = (any && !*any) ? -1 : (nw > 0 ? wpre[nw - 1] + (boff ? boff[(nw - 1) / SCAN_ITEMS] : 0) + __popcll(bits[nw - 1]) : 0);
}
__global__ void count_roots_from_scan_kernel(const int* __restrict__ L, const int* __restrict__ rank, const int* __restrict__ boff, int n, int* __restrict__ out,
                                             const int* __restrict__ any) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *out = (any && !*any) ? -1 : (n > 0 ? rank[n - 1] + (boff ? boff[(n - 1) / SCAN_ITEMS] : 0) + (L[n - 1] == n - 1 ? 1 : 0) : 0);
}
__global__ void count_from_scan_kernel(const int* __restrict__ flag, const int* __restrict__ rank, int n, int* __restrict__ out,
                                       const int* __restrict__ any) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = (any && !*any) ? -1 : (n > 0 ? rank[n - 1] + flag[n - 1] : 0);
}

// =================================================================================================================
// Nuclei front end
// =================================================================================================================
__global__ void nuc_threshold_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride, int H, int W,
                                     uint8_t* __restrict__ msk0, uint8_t* __restrict__ mrk0, int* __restrict__ any) {
    const long long n = (long long)H * W;
    int local = 0;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y, x;
        pix_yx(p, W, invW, y, x);
        const float* s = inst + y * row_stride + (long long)x * pix_stride;
        const float inner = s[0], cnt = s[1];
        const float raw = inner + cnt;  // float32 add, as numpy (postproc.py:360)
        const uint8_t m = raw > 0.5f;
        msk0[p] = m;
        mrk0[p] = inner > 0.5f;
        local |= m;
    }
    if (__any(local) && (threadIdx.x & 63) == 0) atomicOr(any, 1);
}
// Four pixels per thread (W % 4 == 0, so a quad never straddles a row and the byte maps take one 32-bit store per thread instead of four
// single-byte ones): the r04 profile had the one-pixel kernels at 2.7 TB/s of their 10 bytes per pixel.
template <bool PACKED>
__global__ void nuc_threshold4_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride, int H, int W,
                                      uint32_t* __restrict__ msk0, uint32_t* __restrict__ mrk0, int* __restrict__ any) {
    const long long nq = (long long)H * W / 4;
    const int qw = W / 4;
    int local = 0;
    const double invQ = 1.0 / (double)qw;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        int y, xq;
        pix_yx(q, qw, invQ, y, xq);
        const float* s = inst + y * row_stride + (long long)xq * 4 * pix_stride;
        uint32_t m = 0, k = 0;
        float px[8];
        if (PACKED) {  // [H][W][2] floats, 16-byte aligned rows: the quad is two 16-byte loads
            const float4 a = reinterpret_cast<const float4*>(s)[0], b = reinterpret_cast<const float4*>(s)[1];
            px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w; px[4] = b.x; px[5] = b.y; px[6] = b.z; px[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                px[2 * i] = s[(long long)i * pix_stride];
                px[2 * i + 1] = s[(long long)i * pix_stride + 1];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float inner = px[2 * i], cnt = px[2 * i + 1];
            const float raw = inner + cnt;  // float32 add, as numpy (postproc.py:360)
            m |= (uint32_t)(raw > 0.5f) << (8 * i);
            k |= (uint32_t)(inner > 0.5f) << (8 * i);
        }
        msk0[q] = m;
        mrk0[q] = k;
        local |= (int)m;
    }
    if (__any(local) && (threadIdx.x & 63) == 0) atomicOr(any, 1);
}
// the cross erosion on four 0/1 bytes at once: AND of the word with its row neighbours and with itself shifted one byte either way (the byte
// that shifts in comes from the neighbouring word, or is 1 at the image border: the constant border never wins the min)
__global__ void erode_cross4_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int H, int W) {
    const long long nq = (long long)H * W / 4;
    const int qw = W / 4;
    const double invQ = 1.0 / (double)qw;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        int y, xq;
        pix_yx(q, qw, invQ, y, xq);
        const uint32_t c = src[q];
        uint32_t v = c;
        if (y > 0) v &= src[q - qw];
        if (y < H - 1) v &= src[q + qw];
        const uint32_t left = xq > 0 ? src[q - 1] >> 24 : 1u, right = xq < qw - 1 ? src[q + 1] << 24 : 0x01000000u;
        v &= (c << 8) | left;
        v &= (c >> 8) | right;
        dst[q] = v;
    }
}
// cv2.erode with the 3x3 MORPH_ELLIPSE (= cross); the constant border never wins the min
__global__ void erode_cross_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y, x;
        pix_yx(p, W, invW, y, x);
        uint8_t v = src[p];
        if (y > 0) v &= src[p - W];
        if (y < H - 1) v &= src[p + W];
        if (x > 0) v &= src[p - 1];
        if (x < W - 1) v &= src[p + 1];
        dst[p] = v;
    }
}
// m[p] &= area[root(p)] >= min_size
__global__ void apply_min_area_kernel(uint8_t* __restrict__ m, const int* __restrict__ L, const int* __restrict__ area, int min_size, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int r = L[p];
        m[p] = (r >= 0 && area[r] >= min_size) ? 1 : 0;
    }
}
// the same, and bits[p >> 6] |= "p is the root of a kept component"
__global__ void apply_min_area_bits_kernel(uint8_t* __restrict__ m, const int* __restrict__ L, const int* __restrict__ area, int min_size, int n,
                                           u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    for (long long base = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + lane;
        bool root = false;
        if (p < n) {
            const int r = L[p];
            const bool keep = r >= 0 && area[r] >= min_size;
            m[p] = keep ? 1 : 0;
            root = keep && r == (int)p;
        }
        const u64 b = __ballot(root);
        if (lane == 0) bits[base >> 6] = b;
    }
}
// border[root] = 1 for background components touching the border of the H x W domain
__global__ void mark_border_kernel(const int* __restrict__ L, int* __restrict__ border, int H, int W) {
    const int per = 2 * (H + W);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        long long p;
        if (i < W) p = i;
        else if (i < 2 * W) p = (long long)(H - 1) * W + (i - W);
        else if (i < 2 * W + H) p = (long long)(i - 2 * W) * W;
        else p = (long long)(i - 2 * W - H) * W + (W - 1);
        const int r = L[p];
        if (r >= 0) border[r] = 1;
    }
}
// m[p] |= (background component of p does not touch the border)     (scipy.ndimage.binary_fill_holes)
__global__ void fill_holes_apply_kernel(uint8_t* __restrict__ m, const int* __restrict__ L, const int* __restrict__ border, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int r = L[p];
        if (r >= 0 && !border[r]) m[p] = 1;
    }
}

// =================================================================================================================
// Watershed
// =================================================================================================================
__device__ __forceinline__ u32 order_key(float v) {  // monotone map float -> uint32 (with -0.0 == +0.0)
    if (v == 0.0f) v = 0.0f;
    const u32 b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// "Uncertain age" flag of a queue entry (top bit of its pixel / window index): the entry's position among entries of EQUAL
// priority value is not determined by (value, age) alone -- it is a seed (all seeds enter with age 0 and skimage's heap layout
// orders equal ones), or it was pushed by an entry popped inside such a tie (its age then inherits the tie's order).
#define WS_UNC32 0x80000000u
#define WS_UNC16 0x8000u

// out[p] = mask ? marker : 0 ; per mask component: count of pixels that can ever enter the queue (upper bound of the
// heap size = component area) is already known (area).  Seeds = labelled pixels with an unlabelled in-mask neighbour.
__global__ void ws_init_out_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ marker, int* __restrict__ out, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        out[p] = mask[p] ? marker[p] : 0;
}
// heap capacity per component root = its area (kept components only)
__global__ void ws_cap_kernel(const int* __restrict__ L, const int* __restrict__ area, const uint8_t* __restrict__ mask, int* __restrict__ cap, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        cap[p] = (L[p] == (int)p && mask[p]) ? area[p] : 0;
}
__global__ void ws_seed_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride, const uint8_t* __restrict__ mask,
                               const int* __restrict__ out, const int* __restrict__ L, const int* __restrict__ hoff, int* __restrict__ hcnt,
                               u64* __restrict__ hkey, u32* __restrict__ hidx, int H, int W, int* __restrict__ unl, int* __restrict__ lmin,
                               int* __restrict__ lmax, int known) {
    // known: lmin / lmax are final already (nuc_marker_out_bits_kernel): components whose markers carry ONE label never reach a flood
    // (ws_fill_single_kernel paints them), so neither their seeds nor their floodable-pixel counts are wanted -- isolated nuclei issue no atomic here
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    const int lane = threadIdx.x & 63;
    // whole waves step together (no lane leaves the loop early): the count of floodable pixels is added once per RUN of equal roots inside the
    // wave -- 64 consecutive pixels of a row hold a few runs, where one atomic per pixel had every ring pixel of a nucleus hit one address
    for (long long base = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; base < n; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + lane;
        const bool valid = p < n;
        const int o = valid ? out[p] : 0;
        const int key = (valid && !o && mask[p]) ? L[p] : -1;  // root of a floodable pixel
        const int prev = __shfl_up(key, 1);
        const bool head = key >= 0 && (lane == 0 || prev != key);
        const unsigned long long bounds = __ballot(head || key < 0);
        if (head && (!known || lmin[key] < lmax[key])) {
            const unsigned long long after = lane == 63 ? 0ull : (bounds >> (lane + 1));
            const int run = after ? __ffsll((long long)after) : 64 - lane;
            atomicAdd(&unl[key], run);
        }
        if (!o) continue;
        int y, x;
        pix_yx(p, W, invW, y, x);
        bool active = false;
        if (y > 0 && mask[p - W] && !out[p - W]) active = true;
        if (x > 0 && mask[p - 1] && !out[p - 1]) active = true;
        if (x < W - 1 && mask[p + 1] && !out[p + 1]) active = true;
        if (y < H - 1 && mask[p + W] && !out[p + W]) active = true;
        if (!active) continue;
        const int root = L[p];
        if (known) {
            if (lmin[root] >= lmax[root]) continue;
        } else {
            atomicMin(&lmin[root], out[p]);
            atomicMax(&lmax[root], out[p]);
        }
        const int slot = hoff[root] + atomicAdd(&hcnt[root], 1);
        const float v = -inst[y * row_stride + (long long)x * pix_stride];  // watershed(-inst_inner_raw, ...)
        hkey[slot] = ((u64)order_key(v) << 32);  // age 0
        hidx[slot] = (u32)p | WS_UNC32;          // every seed's pop position among equal-valued seeds is skimage's heap layout's choice
    }
}
// bounding box of every kept mask component, keyed by its root pixel (only outline pixels issue atomics)
struct CBox {
    int y1, y2, x1, x2;  // inclusive
};
__global__ void ws_bbox_init_kernel(const int* __restrict__ L, const uint8_t* __restrict__ mask, CBox* __restrict__ bb, int H, int W) {
    const long long n = (long long)H * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        if (L[p] == (int)p && mask[p]) bb[p] = CBox{H, -1, W, -1};
}
// Only components that reach a priority flood need their box (ws_worklist_kernel: seeds of at least two labels, lmin < lmax; no seed at all
// leaves lmin = 0x7f7f7f7f > lmax = 0): isolated nuclei -- the common case -- skip the outline test and the atomics.
__global__ void ws_bbox_kernel(const int* __restrict__ L, const uint8_t* __restrict__ mask, CBox* bb, int H, int W, const int* __restrict__ lmin,
                               const int* __restrict__ lmax) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (!mask[p]) continue;
        int y, x;
        pix_yx(p, W, invW, y, x);
        // (outline test first: its loads are neighbours of the pixel's own; the per-root gathers are then paid by ~1 mask pixel in 6)
        if (y > 0 && y < H - 1 && x > 0 && x < W - 1 && mask[p - W] && mask[p + W] && mask[p - 1] && mask[p + 1]) continue;
        const int root = L[p];
        if (lmin[root] >= lmax[root]) continue;
        CBox* b = bb + root;
        const volatile CBox* vb = b;  // monotone extremes: skip the atomic when a plain read already covers the pixel
        if (y < vb->y1) atomicMin(&b->y1, y);
        if (y > vb->y2) atomicMax(&b->y2, y);
        if (x < vb->x1) atomicMin(&b->x1, x);
        if (x > vb->x2) atomicMax(&b->x2, x);
    }
}
// floodable = in the mask and unlabelled; a seed = labelled with a floodable 4-neighbour
__device__ __forceinline__ uint32_t floodable4(uint32_t m, const int4& o) {  // byte i = 1 when pixel i is in the mask and unlabelled
    return m & ((o.x ? 0u : 1u) | (o.y ? 0u : 0x100u) | (o.z ? 0u : 0x10000u) | (o.w ? 0u : 0x1000000u));
}
// ONE pass for the seeds, the floodable-pixel counts and the boxes (round 5): all three want the same neighbourhood of a mask pixel, and the passes are
// latency-bound -- a second kernel costs its whole chain of dependent loads again (0.34 ms for the boxes alone), a few more loads in the same level of an
// existing chain cost next to nothing.  Level 1: the mask word; level 2 (threads inside the mask, 1 in 5): start map, component roots and mask words of
// the neighbourhood; level 3: the label extremes of the roots met.  Box atomics per RUN of lanes that hold outline pixels of one component in one row (the
// run's first lane carries the row and the smallest column, its last lane the largest column).
__global__ void ws_seed_bbox4_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride, const uint32_t* __restrict__ mask4,
                                     const int4* __restrict__ out4, const int4* __restrict__ L4, const int* __restrict__ hoff, int* __restrict__ hcnt,
                                     u64* __restrict__ hkey, u32* __restrict__ hidx, int H, int W, int* __restrict__ unl, const int* __restrict__ lmin,
                                     const int* __restrict__ lmax, CBox* bb) {
    const long long nq = (long long)H * W / 4;
    const int qw = W / 4;
    const double invQ = 1.0 / (double)qw;
    const uint8_t* mask = (const uint8_t*)mask4;
    const int* out = (const int*)out4;
    const int lane = threadIdx.x & 63;
    for (long long q0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; q0 < nq; q0 += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + lane;
        int key = -1, y = 0, xa = 0, xb = 0;  // this thread's outline pixels of a flooded component (box run)
        const uint32_t m = q < nq ? mask4[q] : 0u;
        if (m) {  // (a label outside the mask does not exist: the start map is mask ? marker : 0)
            int xq;
            pix_yx(q, qw, invQ, y, xq);
            const bool up = y > 0, down = y < H - 1, lft = xq > 0, rgt = xq < qw - 1;
            const int4 o = out4[q];
            const int4 l = L4[q];
            const uint32_t mu = up ? mask4[q - qw] : 0u, md = down ? mask4[q + qw] : 0u;
            const uint32_t ml = lft ? mask4[q - 1] : 0u, mr = rgt ? mask4[q + 1] : 0u;
            const int root[4] = {l.x, l.y, l.z, l.w};
            const uint32_t f = floodable4(m, o);
            const uint32_t lab = m & ~f;  // bytes are 0 / 1
            // multi[i]: pixel i belongs to a component whose markers carry more than one label (the only ones that reach a flood)
            bool multi[4];
            {
                int pr = -1;
                bool pm = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    multi[i] = false;
                    if (!((m >> (8 * i)) & 1u)) continue;
                    if (root[i] != pr) {
                        pr = root[i];
                        pm = lmin[pr] < lmax[pr];
                    }
                    multi[i] = pm;
                }
            }
            if (multi[0] || multi[1] || multi[2] || multi[3]) {
                // floodable pixels, counted per run of equal roots inside the thread
                int i = 0;
                while (i < 4) {
                    if (!((f >> (8 * i)) & 1u) || !multi[i]) {
                        ++i;
                        continue;
                    }
                    int j = i + 1;
                    while (j < 4 && ((f >> (8 * j)) & 1u) && root[j] == root[i]) ++j;
                    atomicAdd(&unl[root[i]], j - i);
                    i = j;
                }
                // outline = mask minus its cross erosion (the image border counts as outside)
                uint32_t v = m & mu & md;
                v &= (m << 8) | (ml >> 24);
                v &= (m >> 8) | (mr << 24);
                const uint32_t outline = m & ~v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!((outline >> (8 * k)) & 1u) || !multi[k]) continue;
                    const int x = xq * 4 + k;
                    if (key < 0) {
                        key = root[k];
                        xa = xb = x;
                    } else if (root[k] == key) {
                        xb = x;
                    } else {  // a second component inside one thread's four pixels: on its own
                        CBox* b = bb + root[k];
                        const volatile CBox* vb = b;
                        if (y < vb->y1) atomicMin(&b->y1, y);
                        if (y > vb->y2) atomicMax(&b->y2, y);
                        if (x < vb->x1) atomicMin(&b->x1, x);
                        if (x > vb->x2) atomicMax(&b->x2, x);
                    }
                }
                // seeds: labelled pixels with a floodable 4-neighbour
                if (lab) {
                    uint32_t nb = (f << 8) | (f >> 8);  // left / right neighbours inside the word
                    // (fetched here, one round trip later than the rest: hoisting the two 16-byte loads to every mask thread measured 0.07 ms slower)
                    if (mu) nb |= floodable4(mu, out4[q - qw]);
                    if (md) nb |= floodable4(md, out4[q + qw]);
                    const long long p0 = q * 4;
                    if ((lab & 1u) && (ml >> 24) && !out[p0 - 1]) nb |= 1u;
                    if ((lab & 0x1000000u) && (mr & 0xffu) && !out[p0 + 4]) nb |= 0x1000000u;
                    const uint32_t act = lab & nb;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!((act >> (8 * k)) & 1u) || !multi[k]) continue;
                        const int r = root[k];
                        const int slot = hoff[r] + atomicAdd(&hcnt[r], 1);
                        const float val = -inst[y * row_stride + (long long)(xq * 4 + k) * pix_stride];  // watershed(-inst_inner_raw, ...)
                        hkey[slot] = ((u64)order_key(val) << 32);  // age 0
                        hidx[slot] = (u32)(p0 + k) | WS_UNC32;     // every seed's pop position among equal-valued seeds is skimage's heap layout's choice
                    }
                }
            }
        }
        const int pk = __shfl_up(key, 1), py = __shfl_up(y, 1), nk = __shfl_down(key, 1), ny = __shfl_down(y, 1);
        if (key >= 0) {
            CBox* b = bb + key;
            const volatile CBox* vb = b;
            if (lane == 0 || pk != key || py != y) {
                if (y < vb->y1) atomicMin(&b->y1, y);
                if (y > vb->y2) atomicMax(&b->y2, y);
                if (xa < vb->x1) atomicMin(&b->x1, xa);
            }
            if (lane == 63 || nk != key || ny != y)
                if (xb > vb->x2) atomicMax(&b->x2, xb);
        }
    }
    (void)mask;
}
// Compact lists of component roots that own at least one seed, in three tiers:
//   window tier : bounding box (+1 px ring) fits WS_WIN_CAP pixels and area <= WS_LDS_CAP -> whole flood in LDS   (front of wl)
//   LDS-heap tier: area <= WS_LDS_CAP                                                     -> heap in LDS           (wl2)
//   global tier : everything else                                                        -> heap in global memory (back of wl)
//   big-window  : bounding box fits WS_BIGWIN_CAP pixels and (seeds + unlabelled mask pixels) <= WS_BIGHEAP_CAP: one wave per
//                 workgroup with ~150 KB of LDS                                                                  (wl3)
#define WS_LDS_CAP 1024
#define WS_WIN_CAP 2304
#define WS_TINY_CAP 512    // tiny-window tier: pairs / small clusters; 13 KB of LDS per wave -> 12 waves per CU instead of 4
#define WS_TINY_WIN 1024
#define WS_BIGWIN_CAP 16384
#define WS_BIGHEAP_CAP 2048
// A component all of whose seeds carry ONE label needs no priority flood: every unlabelled mask pixel of a 4-connected mask
// component is reachable from a seed through unlabelled mask pixels, so the flood can only ever assign that label
// (ws_fill_single_kernel).  Isolated nuclei -- the common case -- take this path; only touching clusters reach the heaps.
// First kernel of the heap-offset scan with the per-root state set up on the way: the scanned value is the heap capacity of a kept mask component
// (its area, at its root pixel; 0 elsewhere), and every kept root gets its counters / label extremes / box initialised HERE -- the per-root arrays are
// indexed by root pixel and only ever read at kept roots, so the four whole-map fills, the capacity pass and the box-init pass are not needed.
__global__ __launch_bounds__(256) void scan_block_cap_kernel(const int* __restrict__ L, const int* __restrict__ area, const uint8_t* __restrict__ mask,
                                                            int* __restrict__ out, int* __restrict__ sums, int n, int* __restrict__ hcnt, int* __restrict__ unl,
                                                            int* __restrict__ lmin, int* __restrict__ lmax, CBox* __restrict__ bb, int H, int W) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SCAN_ITEMS + tid * 4;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long j = base + i;
        v[i] = 0;
        if (j < n && L[j] == (int)j && mask[j]) {
            v[i] = area[j];
            hcnt[j] = 0;
            unl[j] = 0;
            lmin[j] = 0x7f7f7f7f;
            lmax[j] = 0;
            bb[j] = CBox{H, -1, W, -1};
        }
    }
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + inc - tsum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (tid == 255 && sums) sums[blockIdx.x] = woff + inc;
}
__global__ void ws_worklist_kernel(const int* __restrict__ hcnt, const int* __restrict__ area, const int* __restrict__ unl, const CBox* __restrict__ bb,
                                   const int* __restrict__ lmin, const int* __restrict__ lmax, int* __restrict__ wl, int* __restrict__ wl2,
                                   int* __restrict__ wl3, int* __restrict__ wl4, int* __restrict__ counts, int n, const int* __restrict__ L,
                                   const uint8_t* __restrict__ mask) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        if (L[p] == (int)p && mask[p] && hcnt[p] > 0 && lmin[p] != lmax[p]) {  // (the per-root arrays hold something at kept roots only)
            const CBox b = bb[p];
            const long long win = (long long)(b.y2 - b.y1 + 3) * (b.x2 - b.x1 + 3);
            const int need = hcnt[p] + unl[p];  // every queue entry is a seed or a pixel that was unlabelled at the start
            if (need <= WS_TINY_CAP && win <= WS_TINY_WIN) wl4[atomicAdd(counts + 4, 1)] = (int)p;
            else if (need <= WS_LDS_CAP && win <= WS_WIN_CAP) wl[atomicAdd(counts + 0, 1)] = (int)p;
            else if (need <= WS_BIGHEAP_CAP && win <= WS_BIGWIN_CAP) wl3[atomicAdd(counts + 3, 1)] = (int)p;
            else if (area[p] <= WS_LDS_CAP) wl2[atomicAdd(counts + 1, 1)] = (int)p;
            else wl[n - 1 - atomicAdd(counts + 2, 1)] = (int)p;
        }
}

// Per-component setup over the root bitmap (one thread per 64-pixel word): counters, label extremes, box, and the heap offset -- any disjoint
// partition of the heap arrays will do (the floods address their heap as hoff[root] + i), so a block adds the areas of its roots up and takes its
// range with ONE atomic; the exclusive scan over the pixel map (scan_block_cap_kernel + two small scans + a whole-map add) is not needed.
__global__ __launch_bounds__(256) void roots_setup_kernel(const u64* __restrict__ bits, int nw, const int* __restrict__ area, int* __restrict__ hoff,
                                                          int* __restrict__ hcnt, int* __restrict__ unl, int* __restrict__ lmin, int* __restrict__ lmax,
                                                          CBox* __restrict__ bb, int H, int W, int* __restrict__ total) {
    __shared__ int wsum[4];
    __shared__ int bbase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x * 256 + tid;
    const u64 word = w < nw ? bits[w] : 0ull;
    int tsum = 0;
    for (u64 b = word; b; b &= b - 1) tsum += area[(long long)w * 64 + (__ffsll((long long)b) - 1)];
    int inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        bbase = tot ? atomicAdd(total, tot) : 0;
    }
    __syncthreads();
    int run = bbase + inc - tsum;
    for (int k = 0; k < wave; ++k) run += wsum[k];
    for (u64 b = word; b; b &= b - 1) {
        const long long j = (long long)w * 64 + (__ffsll((long long)b) - 1);
        hoff[j] = run;
        run += area[j];
        hcnt[j] = 0;
        unl[j] = 0;
        lmin[j] = 0x7f7f7f7f;
        lmax[j] = 0;
        bb[j] = CBox{H, -1, W, -1};
    }
}
__global__ void ws_worklist_bits_kernel(const u64* __restrict__ bits, int nw, const int* __restrict__ hcnt, const int* __restrict__ area,
                                        const int* __restrict__ unl, const CBox* __restrict__ bb, const int* __restrict__ lmin, const int* __restrict__ lmax,
                                        int* __restrict__ wl, int* __restrict__ wl2, int* __restrict__ wl3, int* __restrict__ wl4, int* __restrict__ counts, int n) {
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += gridDim.x * blockDim.x)
        for (u64 bw = bits[w]; bw; bw &= bw - 1) {
            const int p = w * 64 + (__ffsll((long long)bw) - 1);
            if (!(hcnt[p] > 0 && lmin[p] != lmax[p])) continue;
            const CBox b = bb[p];
            const long long win = (long long)(b.y2 - b.y1 + 3) * (b.x2 - b.x1 + 3);
            const int need = hcnt[p] + unl[p];
            if (need <= WS_TINY_CAP && win <= WS_TINY_WIN) wl4[atomicAdd(counts + 4, 1)] = p;
            else if (need <= WS_LDS_CAP && win <= WS_WIN_CAP) wl[atomicAdd(counts + 0, 1)] = p;
            else if (need <= WS_BIGHEAP_CAP && win <= WS_BIGWIN_CAP) wl3[atomicAdd(counts + 3, 1)] = p;
            else if (area[p] <= WS_LDS_CAP) wl2[atomicAdd(counts + 1, 1)] = p;
            else wl[n - 1 - atomicAdd(counts + 2, 1)] = p;
        }
}

__global__ void ws_fill_single_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ L, const int* __restrict__ lmin,
                                      const int* __restrict__ lmax, int* __restrict__ out, int n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (!mask[p] || out[p]) continue;
        const int r = L[p];
        const int a = lmin[r];
        if (a == lmax[r]) out[p] = a;
    }
}

// four pixels per thread (W % 4 == 0, 16-byte aligned maps): one word of the mask decides whether the quad's labels are looked at at all
__global__ void ws_fill_single4_kernel(const uint32_t* __restrict__ mask4, const int4* __restrict__ L4, const int* __restrict__ lmin, const int* __restrict__ lmax,
                                       int4* __restrict__ out4, long long nq) {
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        const uint32_t mw = mask4[q];
        if (!mw) continue;
        int4 o = out4[q];
        int v[4] = {o.x, o.y, o.z, o.w};
        bool need = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) need = need || (((mw >> (8 * i)) & 0xffu) && !v[i]);
        if (!need) continue;
        const int4 l = L4[q];
        const int r[4] = {l.x, l.y, l.z, l.w};
        int pr = -1, pa = 0;
        bool single = false;
        int* out = reinterpret_cast<int*>(out4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!((mw >> (8 * i)) & 0xffu) || v[i]) continue;
            if (r[i] != pr) {
                pr = r[i];
                pa = lmin[pr];
                single = pa == lmax[pr];
            }
            // one store per pixel, never the whole quad: the flood tiers run beside this kernel on other streams and may be writing the quad's other pixels
            if (single) out[q * 4 + i] = pa;
        }
    }
}

__device__ __forceinline__ bool hless(u64 k1, u32 i1, u64 k2, u32 i2) { return k1 < k2 || (k1 == k2 && i1 < i2); }

// wave-wide min of a u32: 4 DPP row rotations (min inside each 16-lane row), then the 4 row results via readlane.
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
    const u32 r0 = (u32)__builtin_amdgcn_readlane((int)v, 0), r1 = (u32)__builtin_amdgcn_readlane((int)v, 16);
    const u32 r2 = (u32)__builtin_amdgcn_readlane((int)v, 32), r3 = (u32)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}
// wave-wide argmin of (key, idx), lexicographic (priority value, age, pixel index); every lane returns the winning lane id
__device__ __forceinline__ int wave_argmin(u64 k, u32 i, bool valid, u64* kout, u32* iout) {
    const u32 hi = valid ? (u32)(k >> 32) : ~0u;
    const u32 mh = wave_min_u32(hi);
    const bool c1 = valid && hi == mh;
    const u32 ml = wave_min_u32(c1 ? (u32)k : ~0u);
    const bool c2 = c1 && (u32)k == ml;
    const u32 mi = wave_min_u32(c2 ? i : ~0u);
    const u64 win = __ballot(c2 && i == mi);
    *kout = ((u64)mh << 32) | ml;
    *iout = mi;
    return win ? __ffsll((long long)win) - 1 : 0;
}

// 64-ary heap, node i has children 64 i + 1 .. 64 i + 64.  All lanes call these with uniform arguments.
__device__ __forceinline__ void heap_sift_down(volatile u64* hk, volatile u32* hi, int n, int i, u64 k, u32 x) {
    const int lane = threadIdx.x & 63;
    for (;;) {
        const long long c0 = 64ll * i + 1;
        if (c0 >= n) break;
        const long long c = c0 + lane;
        const bool valid = c < n;
        const u64 ck = valid ? hk[c] : 0;
        const u32 ci = valid ? hi[c] : 0;
        u64 mk;
        u32 mi;
        const int ml = wave_argmin(ck, ci, valid, &mk, &mi);
        if (!hless(mk, mi, k, x)) break;
        if (lane == 0) {
            hk[i] = mk;
            hi[i] = mi;
        }
        i = (int)(c0 + ml);
    }
    if (lane == 0) {
        hk[i] = k;
        hi[i] = x;
    }
}
__device__ __forceinline__ void heap_push(volatile u64* hk, volatile u32* hi, int* n, u64 k, u32 x) {
    int j = (*n)++;
    while (j > 0) {
        const int par = (j - 1) >> 6;
        const u64 pk = hk[par];
        const u32 pi = hi[par];
        if (!hless(k, x, pk, pi)) break;
        if ((threadIdx.x & 63) == 0) {
            hk[j] = pk;
            hi[j] = pi;
        }
        j = par;
    }
    if ((threadIdx.x & 63) == 0) {
        hk[j] = k;
        hi[j] = x;
    }
}

// One wavefront per mask component.  Exact sequential priority flood (skimage _watershed_cy.watershed_raveled,
// compactness 0, no watershed line): pop min (value, age); for neighbours in order up, left, right, down: if in mask
// and unlabelled -> label it NOW with the popped pixel's label, age += 1, push.
//
// Tie rule (all flood kernels; validated on the CPU against the oracle's literal heap by tests/tools/dev_tie_rule_fuzz.py).
// (value, age) is a total order except between age-0 seeds of equal value, where skimage's global heap layout decides and this
// kernel takes raster order instead.  That choice reaches the label map only through
//   (a) a DIRECT conflict: the popped entry X labels a pixel q that also touches an equal-valued pixel of ANOTHER label
//       (whichever of the two pops first claims q), or
//   (b) the AGES of the pixels X pushes, which inherit X's position inside its tie and matter again only where such
//       children tie in value among themselves -- so pushed entries carry the "uncertain age" flag of WS_UNC32 and rule (a) is
//       applied to them in turn, and
//   (c) a child that is smaller than the tied value itself: it pops before the rest of the tie, which the bookkeeping of
//       (b) assumes cannot happen (conservatively counted as ambiguous).
// X is "inside a tie" when it carries the flag and the entry popped before it or the one popped after it has X's value.
// A component where none of (a)/(c) fires is labelled identically under every tie order, skimage's included.
__device__ __forceinline__ bool ws_conflict_global(const float* __restrict__ inst, long long row_stride, int pix_stride, const int* out, long long q,
                                                   long long p, int lab, u32 v, int H, int W) {
    const int qy = (int)(q / W), qx = (int)(q % W);
    bool hit = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wy = qy + (j == 0 ? -1 : j == 3 ? 1 : 0), wx = qx + (j == 1 ? -1 : j == 2 ? 1 : 0);
        if (wy < 0 || wy >= H || wx < 0 || wx >= W) continue;
        const long long w = (long long)wy * W + wx;
        if (w == p) continue;
        const int ow = __hip_atomic_load(&out[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ow != 0 && ow != lab && order_key(-inst[wy * row_stride + (long long)wx * pix_stride]) == v) hit = true;
    }
    return hit;
}

// Work items are handed out one at a time (a counter per tier, zeroed with the call's scratch words): the floods of a tier differ several-fold in
// length, and a static stride over the list left a tier waiting for the wave that happened to draw the long ones.
__device__ __forceinline__ int ws_next_item(int* next) {
    int v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(next, 1);
    return __shfl(v, 0);
}
__global__ __launch_bounds__(256) void ws_flood_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride,
                                                       const uint8_t* __restrict__ mask, int* out, const int* __restrict__ wl,
                                                       const int* __restrict__ wl_n, const int* __restrict__ hoff, const int* __restrict__ hcnt,
                                                       u64* hkey, u32* hidx, int H, int W, int* __restrict__ n_ambiguous, int wl_len, int* next) {
    const int lane = threadIdx.x & 63;
    for (int w = ws_next_item(next); w < *wl_n; w = ws_next_item(next)) {
        const int root = wl[wl_len - 1 - w];  // large-component list grows from the back of wl
        volatile u64* hk = hkey + hoff[root];
        volatile u32* hi = hidx + hoff[root];
        int n = hcnt[root];
        // heapify (Floyd): sift down every internal node, last to first
        for (int i = (n - 2) >> 6; i >= 0 && n > 1; --i) {
            const u64 k = hk[i];
            const u32 x = hi[i];
            heap_sift_down(hk, hi, n, i, k, x);
        }
        u32 age = 0;
        bool have_prev = false, ambiguous = false;
        u32 prev_val = 0;
        while (n > 0) {
            const u64 k = hk[0];
            const u32 pu = hi[0];
            const u32 p = pu & ~WS_UNC32;
            --n;
            if (n > 0) {
                const u64 lk = hk[n];
                const u32 li = hi[n];
                heap_sift_down(hk, hi, n, 0, lk, li);
            }
            const int lab = __hip_atomic_load(&out[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 v = (u32)(k >> 32);
            bool tied = have_prev && prev_val == v;
            if (!tied && n > 0) tied = (u32)(hk[0] >> 32) == v;  // the new minimum
            const bool unc = (pu & WS_UNC32) && tied;
            have_prev = true;
            prev_val = v;
            const int y = (int)(p / (u32)W), x = (int)(p % (u32)W);
            // lanes 0..3 look at the four neighbours in skimage's order: up, left, right, down
            long long q = -1;
            if (lane == 0 && y > 0) q = (long long)p - W;
            if (lane == 1 && x > 0) q = (long long)p - 1;
            if (lane == 2 && x < W - 1) q = (long long)p + 1;
            if (lane == 3 && y < H - 1) q = (long long)p + W;
            bool elig = false;
            u32 vq = 0;
            if (q >= 0 && mask[q]) {
                elig = __hip_atomic_load(&out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
                if (elig) {
                    __hip_atomic_store(&out[q], lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int qy = (int)(q / W), qx = (int)(q % W);
                    vq = order_key(-inst[qy * row_stride + (long long)qx * pix_stride]);
                }
            }
            if (unc && elig && (vq < v || ws_conflict_global(inst, row_stride, pix_stride, out, q, p, lab, v, H, W))) ambiguous = true;
            const u64 em = __ballot(elig);
            for (int t = 0; t < 4; ++t) {
                if (!((em >> t) & 1)) continue;
                ++age;
                const u32 qk = (u32)__shfl((int)vq, t);
                const u32 qi = (u32)__shfl((int)q, t) | (unc ? WS_UNC32 : 0u);
                heap_push(hk, hi, &n, ((u64)qk << 32) | age, qi);
            }
        }
        if (__any(ambiguous) && lane == 0) atomicAdd(n_ambiguous, 1);
    }
}

// Window tier: the whole flood of a small component runs out of LDS.  Per wave: a (bbox + 1 px ring) window with the
// per-pixel state (-1 outside this component's mask, 0 unlabelled, > 0 label) and the monotone-mapped priority, plus the
// 64-ary heap (key = priority << 32 | age, index = window index, whose order equals raster order inside the component).
// Seeds are found by the wave itself; no global memory is touched between the window load and the final write-back.
template <int WIN_CAP, int HEAP_CAP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ws_flood_window_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride,
                                                                     const uint8_t* __restrict__ mask, const int* __restrict__ L, int* out,
                                                                     const int* __restrict__ wl, const int* __restrict__ wl_n,
                                                                     const CBox* __restrict__ bb, int H, int W, int* __restrict__ n_ambiguous, int* next) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr size_t PER_WAVE = (size_t)HEAP_CAP * 8 + (size_t)WIN_CAP * 8 + (size_t)HEAP_CAP * 2;
    unsigned char* mine = s_raw + (size_t)wv * PER_WAVE;
    volatile u64* hk = reinterpret_cast<u64*>(mine);
    volatile int* st = reinterpret_cast<int*>(mine + (size_t)HEAP_CAP * 8);
    volatile u32* vl = reinterpret_cast<u32*>(mine + (size_t)HEAP_CAP * 8 + (size_t)WIN_CAP * 4);
    volatile unsigned short* hi = reinterpret_cast<unsigned short*>(mine + (size_t)HEAP_CAP * 8 + (size_t)WIN_CAP * 8);
    for (int w = ws_next_item(next); w < *wl_n; w = ws_next_item(next)) {
        const int root = wl[w];
        const CBox b = bb[root];
        const int wh = b.y2 - b.y1 + 3, ww = b.x2 - b.x1 + 3, wn = wh * ww;
        // ---- load the window -----------------------------------------------------------------------------------------
        for (int i = lane; i < wn; i += 64) {
            const int y = b.y1 - 1 + i / ww, x = b.x1 - 1 + i % ww;
            int s = -1;
            u32 v = 0;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const long long p = (long long)y * W + x;
                if (mask[p] && L[p] == root) {
                    s = out[p];
                    v = order_key(-inst[y * row_stride + (long long)x * pix_stride]);
                }
            }
            st[i] = s;
            vl[i] = v;
        }
        // ---- seeds: labelled pixels with an unlabelled in-mask neighbour (window order == raster order) -----------------
        int n = 0;
        for (int i0 = 0; i0 < wn; i0 += 64) {
            const int i = i0 + lane;
            bool seed = false;
            if (i < wn && st[i] > 0) seed = st[i - ww] == 0 || st[i - 1] == 0 || st[i + 1] == 0 || st[i + ww] == 0;
            const u64 m = __ballot(seed);
            if (seed) {
                const int slot = n + __popcll(m & ((1ull << lane) - 1));
                hk[slot] = (u64)vl[i] << 32;  // age 0
                hi[slot] = (unsigned short)(i | WS_UNC16);
            }
            n += __popcll(m);
        }
        // ---- flood with an UNORDERED queue: the frontier of a nucleus cluster is a few dozen to a few hundred entries, so the
        // minimum is found by one strided scan (each lane keeps the best of its entries) + one wave-wide argmin, the popped slot is
        // refilled with the last entry and the (up to four) pushes are plain appends done by lanes 0..3 in parallel.  Same total
        // order as the heap ((value, age), then window index), hence the same pops.  Tie rule: see ws_flood_kernel.
        u32 age = 0;
        bool have_prev = false, ambiguous = false;
        u32 prev_val = 0;
        while (n > 0) {
            u64 bk = ~0ull;
            u32 bi = ~0u;
            int bpos = -1, bc = 0;  // bc: how many of this lane's entries carry the lane's smallest VALUE
            {
                // four strided entries per lane and trip: the LDS reads of a trip are independent, so their latencies overlap
                // (plain pointers for that; the empty asm keeps the compiler from carrying queue contents across pops)
                asm volatile("" ::: "memory");
                const u64* hkq = const_cast<const u64*>(hk);
                const unsigned short* hiq = const_cast<const unsigned short*>(hi);
                for (int c = lane; c < n; c += 256) {
                    u64 ck[4];
                    u32 ci[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cc = c + 64 * u;
                        const bool in = cc < n;
                        ck[u] = in ? hkq[cc] : ~0ull;
                        ci[u] = in ? (u32)hiq[cc] : ~0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (c + 64 * u < n) {
                            const u32 cv = (u32)(ck[u] >> 32), bv = (u32)(bk >> 32);
                            bc = (bpos < 0 || cv < bv) ? 1 : bc + (cv == bv);
                            if (bpos < 0 || hless(ck[u], ci[u], bk, bi)) {
                                bk = ck[u];
                                bi = ci[u];
                                bpos = c + 64 * u;
                            }
                        }
                }
                asm volatile("" ::: "memory");
            }
            u64 k;
            u32 wi_u;
            const int wl_ = wave_argmin(bk, bi, bpos >= 0, &k, &wi_u);
            const int pos = __shfl(bpos, wl_);
            const int wi = (int)(wi_u & ~WS_UNC16);
            const int lab = st[wi];
            const u32 v = (u32)(k >> 32);
            // another queue entry with the popped value?  (two lanes hold one, or one lane holds two)
            const bool mine = bpos >= 0 && (u32)(bk >> 32) == v;
            const u64 holders = __ballot(mine);
            const bool tied = (have_prev && prev_val == v) || __popcll(holders) >= 2 || __ballot(mine && bc >= 2) != 0;
            const bool unc = (wi_u & WS_UNC16) && tied;
            have_prev = true;
            prev_val = v;
            --n;
            if (lane == 0 && pos != n) {
                hk[pos] = hk[n];
                hi[pos] = hi[n];
            }
            // lanes 0..3: up, left, right, down (skimage's neighbour order); the ring of -1 makes bounds checks unnecessary
            const int nq = wi + (lane == 0 ? -ww : lane == 1 ? -1 : lane == 2 ? 1 : ww);
            bool elig = false;
            if (lane < 4) elig = st[nq] == 0;
            const u64 em = __ballot(elig);
            if (elig) {
                const int before = __popcll(em & ((1ull << lane) - 1));
                const u32 vq = vl[nq];
                st[nq] = lab;
                hk[n + before] = ((u64)vq << 32) | (u64)(age + 1 + before);
                hi[n + before] = (unsigned short)(nq | (unc ? WS_UNC16 : 0u));
                if (unc) {
                    if (vq < v) ambiguous = true;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int wq = nq + (j == 0 ? -ww : j == 1 ? -1 : j == 2 ? 1 : ww);
                        if (wq == wi) continue;
                        const int sw = st[wq];
                        if (sw > 0 && sw != lab && vl[wq] == v) ambiguous = true;
                    }
                }
            }
            const int np = __popcll(em);
            n += np;
            age += np;
        }
        // ---- write back -----------------------------------------------------------------------------------------------------
        for (int i = lane; i < wn; i += 64) {
            const int s = st[i];
            if (s > 0) out[(long long)(b.y1 - 1 + i / ww) * W + (b.x1 - 1 + i % ww)] = s;
        }
        if (__any(ambiguous) && lane == 0) atomicAdd(n_ambiguous, 1);
    }
}

// Same flood, heap in LDS (one 1024-entry heap per wave; the heap never exceeds the component's area) with the label kept
// beside the pixel index, and the neighbour probe fused into ONE global round trip per pop (mask, label and priority of the
// four neighbours are fetched together by lanes 0..3).
__global__ __launch_bounds__(256) void ws_flood_lds_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride,
                                                           const uint8_t* __restrict__ mask, int* out, const int* __restrict__ wl,
                                                           const int* __restrict__ wl_n, const int* __restrict__ hoff,
                                                           const int* __restrict__ hcnt, const u64* __restrict__ hkey,
                                                           const u32* __restrict__ hidx, int H, int W, int* __restrict__ n_ambiguous, int* next) {
    __shared__ u64 s_key[4][WS_LDS_CAP];
    __shared__ u32 s_idx[4][WS_LDS_CAP];
    __shared__ int s_lab[4][WS_LDS_CAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    volatile u64* hk = s_key[wv];
    volatile u32* hi = s_idx[wv];
    volatile int* hl = s_lab[wv];
    for (int w = ws_next_item(next); w < *wl_n; w = ws_next_item(next)) {
        const int root = wl[w];
        int n = hcnt[root];
        const int base = hoff[root];
        for (int i = lane; i < n; i += 64) {
            const u32 px = hidx[base + i];
            hk[i] = hkey[base + i];
            hi[i] = px;
            hl[i] = out[px & ~WS_UNC32];
        }
        // heapify (Floyd) -- (key, idx) order; the label rides along
        for (int i = (n - 2) >> 6; i >= 0 && n > 1; --i) {
            u64 k = hk[i];
            u32 x = hi[i];
            int lb = hl[i], pos = i;
            for (;;) {
                const int c0 = 64 * pos + 1;
                if (c0 >= n) break;
                const int c = c0 + lane;
                const bool valid = c < n;
                const u64 ck = valid ? hk[c] : 0;
                const u32 ci = valid ? hi[c] : 0;
                u64 mk;
                u32 mi;
                const int ml = wave_argmin(ck, ci, valid, &mk, &mi);
                if (!hless(mk, mi, k, x)) break;
                const int ml_lab = hl[c0 + ml];
                if (lane == 0) {
                    hk[pos] = mk;
                    hi[pos] = mi;
                    hl[pos] = ml_lab;
                }
                pos = c0 + ml;
            }
            if (lane == 0) {
                hk[pos] = k;
                hi[pos] = x;
                hl[pos] = lb;
            }
        }
        u32 age = 0;
        bool have_prev = false, ambiguous = false;
        u32 prev_val = 0;
        while (n > 0) {
            const u64 k = hk[0];
            const u32 pu = hi[0];
            const u32 p = pu & ~WS_UNC32;
            const int lab = hl[0];
            --n;
            if (n > 0) {  // move the last entry to the root and sift it down
                const u64 lk = hk[n];
                const u32 li = hi[n];
                const int ll = hl[n];
                int pos = 0;
                for (;;) {
                    const int c0 = 64 * pos + 1;
                    if (c0 >= n) break;
                    const int c = c0 + lane;
                    const bool valid = c < n;
                    const u64 ck = valid ? hk[c] : 0;
                    const u32 ci = valid ? hi[c] : 0;
                    u64 mk;
                    u32 mi;
                    const int ml = wave_argmin(ck, ci, valid, &mk, &mi);
                    if (!hless(mk, mi, lk, li)) break;
                    const int ml_lab = hl[c0 + ml];
                    if (lane == 0) {
                        hk[pos] = mk;
                        hi[pos] = mi;
                        hl[pos] = ml_lab;
                    }
                    pos = c0 + ml;
                }
                if (lane == 0) {
                    hk[pos] = lk;
                    hi[pos] = li;
                    hl[pos] = ll;
                }
            }
            // tie rule: see ws_flood_kernel
            const u32 v = (u32)(k >> 32);
            bool tied = have_prev && prev_val == v;
            if (!tied && n > 0) tied = (u32)(hk[0] >> 32) == v;
            const bool unc = (pu & WS_UNC32) && tied;
            have_prev = true;
            prev_val = v;
            const int y = (int)(p / (u32)W), x = (int)(p % (u32)W);
            long long q = -1;
            if (lane == 0 && y > 0) q = (long long)p - W;
            if (lane == 1 && x > 0) q = (long long)p - 1;
            if (lane == 2 && x < W - 1) q = (long long)p + 1;
            if (lane == 3 && y < H - 1) q = (long long)p + W;
            bool elig = false;
            u32 vq = 0;
            if (q >= 0) {
                // one round trip: mask, current label and priority of the neighbour
                const uint8_t mq = mask[q];
                const int oq = __hip_atomic_load(&out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int qy = (int)(q / W), qx = (int)(q % W);
                vq = order_key(-inst[qy * row_stride + (long long)qx * pix_stride]);
                elig = mq && oq == 0;
                if (elig) __hip_atomic_store(&out[q], lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (unc && elig && (vq < v || ws_conflict_global(inst, row_stride, pix_stride, out, q, p, lab, v, H, W))) ambiguous = true;
            const u64 em = __ballot(elig);
            for (int t = 0; t < 4; ++t) {
                if (!((em >> t) & 1)) continue;
                ++age;
                const u64 nk = ((u64)(u32)__shfl((int)vq, t) << 32) | age;
                const u32 nx = (u32)__shfl((int)q, t) | (unc ? WS_UNC32 : 0u);
                int jn = n++;
                while (jn > 0) {
                    const int par = (jn - 1) >> 6;
                    const u64 pk = hk[par];
                    const u32 pi = hi[par];
                    if (!hless(nk, nx, pk, pi)) break;
                    const int pl = hl[par];
                    if (lane == 0) {
                        hk[jn] = pk;
                        hi[jn] = pi;
                        hl[jn] = pl;
                    }
                    jn = par;
                }
                if (lane == 0) {
                    hk[jn] = nk;
                    hi[jn] = nx;
                    hl[jn] = lab;
                }
            }
        }
        if (__any(ambiguous) && lane == 0) atomicAdd(n_ambiguous, 1);
    }
}

// =================================================================================================================
// Exact tie order: literal emulation of skimage's global heap (runs only when a flood flagged an ambiguous component)
// =================================================================================================================
// skimage.segmentation.watershed keeps ONE binary heap for the whole image (_watershed_cy.pyx + _shared/heap_general.pxi;
// the checker's C restatement of it is pinned against the real library): every marker pixel is pushed in
// raster order with age 0, `smaller` is strict on (value, age), heappush sifts up while the new entry is smaller than its
// parent, heappop moves the LAST entry to the root and sifts it down preferring the left child on equal keys.  The order in
// which equal-valued markers leave that heap is a function of its whole push / pop history, so it cannot be evaluated per
// component: one wavefront replays the history.  The replay is literal (same array layout, same comparisons) but each heap
// operation is wave-cooperative: a pop gathers the six levels below the hole with one load per lane pair (126 descendants),
// walks them in registers (readlane) and repeats; a push gathers all <= 31 ancestors of the new leaf at once, and -- the path
// being sorted -- shifts the ancestors that are larger down by one in a single scatter.  Heap positions [0, 8191) (13 levels)
// live in LDS, deeper ones in global memory; the four neighbour probes of a pop are issued before its sift-down.
#define EX_LDS 8191
struct ExHeap {
    volatile u64* sk;
    volatile u32* si;
    u64* gk;
    u32* gi;
};
// Only this one wave ever touches the heap, so no agent-scope coherence is needed: workgroup-scope accesses stay in the CU's own
// L1 / the XCD's L2 (an agent-scope load has to miss both L2-non-coherent levels on a multi-XCD part: ~2 us per dependent access).
__device__ __forceinline__ void ex_load(const ExHeap& h, long long pos, u64& k, u32& i) {
    if (pos < EX_LDS) {
        k = h.sk[pos];
        i = h.si[pos];
    } else {
        k = __hip_atomic_load(&h.gk[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        i = __hip_atomic_load(&h.gi[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ void ex_store(const ExHeap& h, long long pos, u64 k, u32 i) {
    if (pos < EX_LDS) {
        h.sk[pos] = k;
        h.si[pos] = i;
    } else {
        __hip_atomic_store(&h.gk[pos], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&h.gi[pos], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ u32 ex_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 ex_uni64(u64 v) { return ((u64)ex_uni((u32)(v >> 32)) << 32) | ex_uni((u32)v); }
__device__ __forceinline__ u64 ex_lane64(u64 v, int l) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
}
// position of local node j (0 = the hole, children 2j+1 / 2j+2) of the subtree rooted at heap position i
__device__ __forceinline__ long long ex_pos(long long i, int j) {
    const int d = 31 - __clz(j + 1);
    return ((i + 1) << d) - 1 + (j + 1 - (1 << d));
}
// heappop's tail: entry x (the former last entry) enters at the root of a heap of n entries and sifts down
__device__ __forceinline__ void ex_sift_down(const ExHeap& h, int n, u64 xk, u32 xi) {
    const int lane = threadIdx.x & 63;
    long long i = 0;
    for (;;) {
        u64 ka = 0, kb = 0;
        u32 ia = 0, ib = 0;
        if (lane >= 1) {
            const long long pa = ex_pos(i, lane);
            if (pa < n) ex_load(h, pa, ka, ia);
        }
        if (lane <= 62) {
            const long long pb = ex_pos(i, lane + 64);
            if (pb < n) ex_load(h, pb, kb, ib);
        }
        int cur = 0;
        bool placed = false;
#pragma unroll 1
        for (int step = 0; step < 6; ++step) {
            const int l = 2 * cur + 1, r = l + 1;
            const long long pl = ex_pos(i, l);
            if (pl >= n) {
                placed = true;
                break;
            }
            const u64 kl = l < 64 ? ex_lane64(ka, l) : ex_lane64(kb, l - 64);
            int s = cur;
            u64 ks = xk;
            if (kl < xk) {  // smaller(l, i)
                s = l;
                ks = kl;
            }
            if (pl + 1 < n) {
                const u64 kr = r < 64 ? ex_lane64(ka, r) : ex_lane64(kb, r - 64);
                if (kr < ks) s = r;  // smaller(r, smallest)
            }
            if (s == cur) {
                placed = true;
                break;
            }
            if (lane == (s & 63)) ex_store(h, ex_pos(i, cur), s < 64 ? ka : kb, s < 64 ? ia : ib);  // child moves up into the hole
            cur = s;
        }
        const long long pc = ex_pos(i, cur);
        if (placed) {
            if (lane == 0) ex_store(h, pc, xk, xi);
            return;
        }
        i = pc;
    }
}
// heappush: new entry y at leaf position n (n = heap size before the push)
__device__ __forceinline__ void ex_push(const ExHeap& h, int n, u64 yk, u32 yi) {
    const int lane = threadIdx.x & 63;
    const long long np1 = (long long)n + 1;
    const long long qa = lane < 32 ? (np1 >> (lane + 1)) : 0;  // lane t: the ancestor t+1 levels above the leaf
    u64 ak = 0;
    u32 ai = 0;
    if (qa >= 1) ex_load(h, qa - 1, ak, ai);
    const u64 up = __ballot(qa >= 1 && yk < ak);                // smaller(child, parent): strict
    const int m = __ffsll((long long)~up) - 1;                  // the path is sorted: the larger ancestors are lanes 0 .. m-1
    if (lane < m) ex_store(h, (np1 >> lane) - 1, ak, ai);       // each moves one step down the path (lane 0's lands on the leaf)
    if (lane == 0) ex_store(h, (np1 >> m) - 1, yk, yi);
}

__global__ void ws_exact_reset_kernel(const uint8_t* __restrict__ mrk, const uint8_t* __restrict__ mask, int* __restrict__ out, int n,
                                      const int* __restrict__ flag) {
    if (*flag == 0) return;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        if (!(mrk[p] && mask[p])) out[p] = 0;  // back to markers * mask: a marker pixel keeps its label through every flood
}

__global__ __launch_bounds__(64) void ws_exact_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride,
                                                      const uint8_t* __restrict__ mask, int* out, u64* hkey, u32* hidx, int H, int W,
                                                      const int* __restrict__ flag) {
    if (*flag == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_ex[];
    ExHeap h;
    h.sk = reinterpret_cast<volatile u64*>(s_ex);
    h.si = reinterpret_cast<volatile u32*>(s_ex + (size_t)EX_LDS * 8 + 8);
    h.gk = hkey;
    h.gi = hidx;
    const int lane = threadIdx.x & 63;
    const long long N = (long long)H * W;
    const double invW = 1.0 / (double)W;
    int n = 0;
    // ---- all marker pixels, raster order, age 0 (8 chunks of 64 pixels in flight) ----------------------------------------
    for (long long base = 0; base < N; base += 512) {
        int o[8];
        u32 kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long p = base + u * 64 + lane;
            o[u] = p < N ? out[p] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long p = base + u * 64 + lane;
            kv[u] = 0;
            if (o[u]) {
                int y, x;
                pix_yx(p, W, invW, y, x);
                kv[u] = order_key(-inst[y * row_stride + (long long)x * pix_stride]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u64 m = __ballot(o[u] != 0);
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const u32 key = (u32)__builtin_amdgcn_readlane((int)kv[u], j);
                ex_push(h, n, (u64)key << 32, (u32)(base + u * 64 + j));
                ++n;
            }
        }
    }
    // ---- the flood -----------------------------------------------------------------------------------------------------------
    u32 age = 0;
    while (n > 0) {
        u64 rk;
        u32 ri;
        ex_load(h, 0, rk, ri);
        const u32 p = ex_uni(ri);
        const int y = (int)(p / (u32)W), x = (int)(p % (u32)W);
        // everything that only depends on the popped pixel / the heap size is requested first: the label, the four neighbour
        // probes and the last entry (which heappop moves to the root)
        const int lab = __hip_atomic_load(&out[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        --n;
        u64 xk = 0;
        u32 xi = 0;
        if (n > 0) ex_load(h, n, xk, xi);
        long long q = -1;
        if (lane == 0 && y > 0) q = (long long)p - W;
        if (lane == 1 && x > 0) q = (long long)p - 1;
        if (lane == 2 && x < W - 1) q = (long long)p + 1;
        if (lane == 3 && y < H - 1) q = (long long)p + W;
        uint8_t mq = 0;
        int oq = 1;
        float vq = 0.f;
        if (q >= 0) {
            mq = mask[q];
            oq = __hip_atomic_load(&out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int qy = (int)(q / W), qx = (int)(q % W);
            vq = -inst[qy * row_stride + (long long)qx * pix_stride];
        }
        if (n > 0) ex_sift_down(h, n, ex_uni64(xk), ex_uni(xi));
        const bool elig = q >= 0 && mq && oq == 0;
        if (elig) __hip_atomic_store(&out[q], lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 kq = order_key(vq);
        const u64 em = __ballot(elig);
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            if (!((em >> t) & 1)) continue;
            ++age;
            const u32 qk = (u32)__builtin_amdgcn_readlane((int)kq, t);
            const u32 qi = (u32)__builtin_amdgcn_readlane((int)(u32)q, t);
            ex_push(h, n, ((u64)qk << 32) | age, qi);
            ++n;
        }
    }
}

// =================================================================================================================
// Gland / lumen: threshold, per-instance crop -> dilate -> fill holes -> paste
// =================================================================================================================
__global__ void gl_threshold_kernel(const float* __restrict__ inst, long long row_stride, int pix_stride, int H, int W, float thr,
                                    uint8_t* __restrict__ fg) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y, x;
        pix_yx(p, W, invW, y, x);
        const float* s = inst + y * row_stride + (long long)x * pix_stride;
        const float c = s[1] > 0.5f ? 1.f : 0.f;  // inst_cnt binarised (postproc.py:280-282)
        fg[p] = (s[0] - c) > thr;
    }
}
struct Box {
    int y1, y2, x1, x2;
};
__global__ void box_init_kernel(Box* b, int n, int H, int W) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        b[i].y1 = H;
        b[i].x1 = W;
        b[i].y2 = 0;
        b[i].x2 = 0;
    }
}
__global__ void box_accum_kernel(const int* __restrict__ lab, Box* b, int H, int W) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int l = lab[p];
        if (!l) continue;
        int y, x;
        pix_yx(p, W, invW, y, x);
        // only pixels on the instance outline can be bounding-box extremes (keeps the atomics off the interior)
        if (y > 0 && y < H - 1 && x > 0 && x < W - 1 && lab[p - W] == l && lab[p + W] == l && lab[p - 1] == l && lab[p + 1] == l) continue;
        // the extremes only ever move outwards, so a (possibly stale) plain read that already covers this pixel makes the atomic
        // unnecessary -- a slide-sized ragged instance would otherwise serialise millions of atomics on one box
        const volatile Box* vb = b + l;
        if (y < vb->y1) atomicMin(&b[l].y1, y);
        if (y + 1 > vb->y2) atomicMax(&b[l].y2, y + 1);
        if (x < vb->x1) atomicMin(&b[l].x1, x);
        if (x + 1 > vb->x2) atomicMax(&b[l].x2, x + 1);
    }
}
// crop = bounding box padded by 2*ksize on each side only where the padded edge stays inside (postproc.py:296-300)
__global__ void box_pad_kernel(Box* b, int* __restrict__ area, int n_inst, int H, int W, int pad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_inst; i += gridDim.x * blockDim.x) {
        if (i == 0) {
            area[0] = 0;
            continue;
        }
        Box c = b[i];
        c.y1 = c.y1 - pad >= 0 ? c.y1 - pad : c.y1;
        c.x1 = c.x1 - pad >= 0 ? c.x1 - pad : c.x1;
        c.x2 = c.x2 + pad <= W - 1 ? c.x2 + pad : c.x2;
        c.y2 = c.y2 + pad <= H - 1 ? c.y2 + pad : c.y2;
        b[i] = c;
        area[i] = (c.y2 - c.y1) * (c.x2 - c.x1);
    }
}
struct Spans {
    int k, a;
    int j1[32], j2[32];
};
// The crops of a batch are laid out back to back (crop-local row-major); the kernels below run over that flat element
// space so large and small instances share the machine.  coff[id] = start of instance id's crop; element e belongs to the
// last instance whose start is <= e.
__device__ __forceinline__ int find_inst(const int* __restrict__ coff, int first, int cnt, int e_abs) {
    int lo = first, hi = first + cnt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (coff[mid] <= e_abs) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// dilation of (lab == id) by the ellipse, clipped to the crop (cv2.dilate on the crop; the border never wins the max)
__global__ __launch_bounds__(256) void gl_dilate_kernel(const int* __restrict__ lab, const Box* __restrict__ box, const int* __restrict__ coff,
                                                        int first, int cnt, int tot, int W, Spans se, uint8_t* __restrict__ dil) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        const int id = find_inst(coff, first, cnt, e + coff[first]);
        const Box c = box[id];
        const int ch = c.y2 - c.y1, cw = c.x2 - c.x1;
        const int i = e - (coff[id] - coff[first]);
        const int y = i / cw, x = i % cw;
        uint8_t v = 0;
        for (int r = 0; r < se.k && !v; ++r) {
            const int yy = y + r - se.a;
            if (yy < 0 || yy >= ch) continue;
            const int xa = max(x + se.j1[r] - se.a, 0), xb = min(x + se.j2[r] - se.a, cw);
            const int* row = lab + (long long)(c.y1 + yy) * W + c.x1;
            for (int xx = xa; xx < xb; ++xx)
                if (row[xx] == id) {
                    v = 1;
                    break;
                }
        }
        dil[e] = v;
    }
}
// background CCL inside every crop of the batch (crop-local 4-connectivity, shared union-find array)
__global__ __launch_bounds__(256) void gl_bg_init_kernel(const uint8_t* __restrict__ dil, int tot, int* __restrict__ L, int* __restrict__ border) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        L[e] = dil[e] ? -1 : e;
        border[e] = 0;
    }
}
__global__ __launch_bounds__(256) void gl_bg_merge_kernel(const uint8_t* __restrict__ dil, const Box* __restrict__ box, const int* __restrict__ coff,
                                                          int first, int cnt, int tot, int* L) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        if (dil[e]) continue;
        const int id = find_inst(coff, first, cnt, e + coff[first]);
        const Box c = box[id];
        const int cw = c.x2 - c.x1;
        const int i = e - (coff[id] - coff[first]);
        const int x = i % cw;
        const bool left = x > 0 && !dil[e - 1];
        if (i >= cw && !dil[e - cw]) {
            // one union per pair of vertically adjacent background runs (see ccl_merge_kernel)
            const bool up_starts = !(x > 0 && !dil[e - cw - 1]);
            if (!left || up_starts) uf_union(L, e, e - cw);
        }
        if (left) uf_union(L, e, e - 1);
    }
}
__global__ __launch_bounds__(256) void gl_bg_border_kernel(const Box* __restrict__ box, const int* __restrict__ coff, int first, int cnt, int tot,
                                                           int* L, int* __restrict__ border) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        if (L[e] < 0) continue;
        const int r = uf_find(L, e);
        L[e] = r;
        const int id = find_inst(coff, first, cnt, e + coff[first]);
        const Box c = box[id];
        const int ch = c.y2 - c.y1, cw = c.x2 - c.x1;
        const int i = e - (coff[id] - coff[first]);
        const int y = i / cw, x = i % cw;
        if (y == 0 || y == ch - 1 || x == 0 || x == cw - 1) border[r] = 1;
    }
}
__global__ __launch_bounds__(256) void gl_paste_kernel(const uint8_t* __restrict__ dil, const Box* __restrict__ box, const int* __restrict__ coff,
                                                       int first, int cnt, int tot, const int* __restrict__ L, const int* __restrict__ border, int W,
                                                       int* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        bool in = dil[e];
        if (!in) in = !border[L[e]];  // hole of the dilated instance inside its crop
        if (!in) continue;
        const int id = find_inst(coff, first, cnt, e + coff[first]);
        const Box c = box[id];
        const int cw = c.x2 - c.x1;
        const int i = e - (coff[id] - coff[first]);
        atomicMax(&out[(long long)(c.y1 + i / cw) * W + c.x1 + i % cw], id);  // ids ascend: later id overwrites
    }
}
__global__ void mask_lumen_kernel(int* __restrict__ lumen, const int* __restrict__ gland, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        if (gland[p] <= 0) lumen[p] = 0;
}

// =================================================================================================================
// Host orchestration
// =================================================================================================================
struct Carve {
    char* p;
    size_t left;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) return nullptr;
        void* r = p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

extern "C" size_t cerb_pp_workspace_bytes(int h, int w) {
    const size_t n = (size_t)h * (size_t)w;
    return n * 96 + (4u << 20);
}

static void ellipse_spans(int k, Spans* s) {  // cv2.getStructuringElement(MORPH_ELLIPSE, (k,k)) row spans
    s->k = k;
    s->a = k / 2;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; ++i) {
        const int dy = i - r;
        s->j1[i] = s->j2[i] = 0;
        if (abs(dy) <= r) {
            const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            s->j1[i] = c - dx > 0 ? c - dx : 0;
            s->j2[i] = c + dx + 1 < k ? c + dx + 1 : k;
        }
    }
}

// hipFuncSetAttribute is per device: remember it per device, not per process (a host process may drive several handles)
static inline bool cerb_attr_needed(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

struct SideStreams {
    hipStream_t s[3];
    hipEvent_t fork, join[3];
};
static SideStreams* side_streams() {  // one set per device, created on first use (handles are used from one host thread, SURVEY par.8b)
    static SideStreams* per_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!per_dev[dev]) {
        SideStreams* ss = new SideStreams();
        bool ok = hipEventCreateWithFlags(&ss->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 3 && ok; ++i)
            ok = hipStreamCreateWithFlags(&ss->s[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&ss->join[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            delete ss;
            return nullptr;
        }
        per_dev[dev] = ss;
    }
    return per_dev[dev];
}

// exact_ties: literal skimage tie order for maps whose floods flag an ambiguous component -- a per-call argument (round 2 kept it in a
// process-wide flag that two host threads could race on).  The WSI band driver passes 0: the reference floods 4096^2 tiles there
// (infer/wsi.py:143-149), so the tie order of its heap is a property of ITS tiling.
extern "C" int cerb_postproc_nuclei(const float* inst, int H, int W, long long row_stride, int pix_stride, int32_t* labels_out,
                                    int32_t* n_inst_out, int32_t* n_ambiguous_out, int exact_ties, void* ws, size_t ws_bytes,
                                    void* hip_stream) {
    hipStream_t st = (hipStream_t)hip_stream;
    if (!inst || !labels_out || !ws || H <= 0 || W <= 0) return cerb_set_error("cerb_postproc_nuclei: bad arguments");
    if (ws_bytes < cerb_pp_workspace_bytes(H, W)) return cerb_set_error("cerb_postproc_nuclei: workspace too small");
    if ((long long)H * W >= (1ll << 31)) return cerb_set_error("cerb_postproc_nuclei: map too large (H*W must be < 2^31)");
    const int n = H * W;
    Carve cv{(char*)ws, ws_bytes};
    int* LA = (int*)cv.take((size_t)n * 4);     // mask components
    int* areaA = (int*)cv.take((size_t)n * 4);
    int* LB = (int*)cv.take((size_t)n * 4);     // scratch CCL (markers, background, filled markers)
    int* areaB = (int*)cv.take((size_t)n * 4);  // also: border flags, root flags
    int* rank = (int*)cv.take((size_t)n * 4);
    int* marker = (int*)cv.take((size_t)n * 4);
    int* hoff = (int*)cv.take((size_t)n * 4);
    int* hcnt = (int*)cv.take((size_t)n * 4);
    int* wl = (int*)cv.take((size_t)n * 4);
    u64* hkey = (u64*)cv.take((size_t)n * 8);
    u32* hidx = (u32*)cv.take((size_t)n * 4);
    CBox* cbox = (CBox*)cv.take((size_t)n * sizeof(CBox));
    uint8_t* msk0 = (uint8_t*)cv.take(n);
    uint8_t* msk = (uint8_t*)cv.take(n);
    uint8_t* mrk = (uint8_t*)cv.take(n);
    int* scantmp = (int*)cv.take((size_t)(n / SCAN_ITEMS + 4096) * 4 * 2);
    int* wl4 = (int*)cv.take((size_t)n * 4);    // tiny-window tier list
    int* lmax = (int*)cv.take((size_t)n * 4);   // per mask component: largest seed label (smallest one lives in LB once the markers are final)
    const int nw = (n + 63) / 64;               // root bitmaps: one bit per pixel
    u64* bitsA = (u64*)cv.take((size_t)nw * 8); // kept mask components
    u64* bitsB = (u64*)cv.take((size_t)nw * 8); // final markers
    int* wpre = (int*)cv.take((size_t)nw * 4);  // roots before a bitmap word
    int* lminbuf = (int*)cv.take((size_t)n * 4); // per mask component: smallest marker label (written while LB is still read)
    int* tcnt = (int*)cv.take((size_t)(((W + CT_W - 1) / CT_W) * ((H + CT_H - 1) / CT_H)) * 4);  // roots per labelling tile
    int* small = (int*)cv.take(256);  // [0]=any [1]=worklist count [2]=n_inst scratch [3]=ambiguous scratch [4]=two-colour root list [8..12]=tier counts [16]=heap total
    if (!small) return cerb_set_error("cerb_postproc_nuclei: workspace carve failed");
    const unsigned g = grid_for(n);
    static const bool pixel_scans = cerb_dev_getenv("CERB_PP_PIXEL_SCANS") != nullptr;  // developer A/B: round 4's whole-map scans instead of the root bitmaps

    PP_OK(hipMemsetAsync(small, 0, 256, st));
    // (A) mask: erode -> label -> drop components < 8 px   (postproc.py:365-368)
    // (round 5 tried threshold + erosion + tile labelling as ONE kernel over a 66 x 34 window per tile: bit-exact, 0.68 ms against 0.51 ms for the three
    // launches below -- the 8-byte loads, the divergent halo columns and three barriers per tile cost more than the two byte planes it saved)
    if (W % 4 == 0 && ((uintptr_t)msk0 | (uintptr_t)msk | (uintptr_t)mrk) % 4 == 0) {
        const unsigned g4 = grid_for(n / 4);
        const bool packed = pix_stride == 2 && row_stride % 4 == 0 && (uintptr_t)inst % 16 == 0;
        hipLaunchKernelGGL(packed ? nuc_threshold4_kernel<true> : nuc_threshold4_kernel<false>, dim3(g4), dim3(256), 0, st, inst, row_stride, pix_stride, H, W,
                           (uint32_t*)msk0, (uint32_t*)mrk, small);
        hipLaunchKernelGGL(erode_cross4_kernel, dim3(g4), dim3(256), 0, st, (const uint32_t*)msk0, (uint32_t*)msk, H, W);
    } else {
        hipLaunchKernelGGL(nuc_threshold_kernel, dim3(g), dim3(256), 0, st, inst, row_stride, pix_stride, H, W, msk0, mrk, small);
        hipLaunchKernelGGL(erode_cross_kernel, dim3(g), dim3(256), 0, st, msk0, msk, H, W);
    }
    // round 6: the tile labellings leave their border columns in a compact array (2 x (tiles_x - 1) x H ints: wl4 is free until the flood work lists) and the
    // seam passes read their vertical seams from it -- CERB_PP_SEAM_COLUMNS=0 (developers' build) keeps the strided reads of the maps
    static const bool seam_compact = cerb_dev_getenv("CERB_PP_SEAM_COLUMNS") == nullptr || atoi(cerb_dev_getenv("CERB_PP_SEAM_COLUMNS")) != 0;
    int* seam_cols = seam_compact ? wl4 : nullptr;
    static const bool narrow = cerb_dev_getenv("CERB_PP_ONE_PIXEL_THREADS") != nullptr;  // developer A/B: the one-pixel-per-thread passes
    const bool wide = !pixel_scans && !narrow && W % 4 == 0 && (uintptr_t)labels_out % 16 == 0;  // (the workspace arrays are 256-byte aligned)
    if (wide) {  // tile labelling with the root list + per-set counts, flatten and areas over the LIST, then one pass: min-area, root of every pixel, bitmap
        int* rootsA = hoff;       // free until roots_setup_kernel
        int* n_rootsA = tcnt;     // roots per tile
        const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H, n_tiles = tiles_x * tiles_y;
        hipLaunchKernelGGL(ccl_tile_kernel<true>, dim3(n_tiles < 256 * 16 ? n_tiles : 256 * 16), dim3(256), 0, st, (const uint8_t*)msk, (uint8_t)1, LA, H, W, tiles_x, n_tiles,
                           rootsA, n_rootsA, areaA, seam_cols);
        const long long seams = (long long)(tiles_x - 1) * H + (long long)(tiles_y - 1) * W;
        if (seams > 0) hipLaunchKernelGGL(ccl_seam_kernel, dim3(grid_for(seams)), dim3(256), 0, st, (const uint8_t*)msk, (uint8_t)1, LA, H, W, tiles_x, tiles_y, (const int*)seam_cols);
        const unsigned gl = nblk(n_tiles, 4) < 4096 ? nblk(n_tiles, 4) : 4096;
        hipLaunchKernelGGL(ccl2_flatten_roots_kernel, dim3(gl), dim3(256), 0, st, LA, (const int*)rootsA, (const int*)n_rootsA, n_tiles, tiles_x, H, W);
        hipLaunchKernelGGL(roots_area_merge_kernel, dim3(gl), dim3(256), 0, st, (const int*)LA, (const int*)rootsA, (const int*)n_rootsA, n_tiles, tiles_x, H, W, areaA);
        hipLaunchKernelGGL(apply_min_area_bits4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, (uint32_t*)msk, (int4*)LA, (const int*)areaA, 8, (long long)n / 4, bitsA);
        KCHECK();
    } else {
        PP_OK(hipMemsetAsync(areaA, 0, (size_t)n * 4, st));
        if (ccl_run(msk, 1, LA, H, W, st, areaA)) return 1;
        if (pixel_scans) hipLaunchKernelGGL(apply_min_area_kernel, dim3(g), dim3(256), 0, st, msk, LA, areaA, 8, n);
        else hipLaunchKernelGGL(apply_min_area_bits_kernel, dim3(g), dim3(256), 0, st, msk, LA, areaA, 8, n, bitsA);
    }
    // (B) markers: inner > 0.5 -> label -> drop < 4 px -> fill holes -> label (postproc.py:370-377)
    static const bool three_pass = cerb_dev_getenv("CERB_PP_THREE_LABELLINGS") != nullptr;  // developer A/B: round 4's three separate labellings
    if (!three_pass) {
        // (rank: free until the scan below, serves as the border flags; marker: free until the flood work lists, holds the root list)
        if (markers_two_colour(mrk, LB, areaB, rank, marker, tcnt, 4, H, W, st, pixel_scans ? nullptr : bitsB, wide, seam_cols)) return 1;
    } else {
        PP_OK(hipMemsetAsync(areaB, 0, (size_t)n * 4, st));
        if (ccl_run(mrk, 1, LB, H, W, st, areaB)) return 1;
        hipLaunchKernelGGL(apply_min_area_kernel, dim3(g), dim3(256), 0, st, mrk, LB, areaB, 4, n);
        if (ccl_run(mrk, 0, LB, H, W, st)) return 1;  // background of the marker image
        PP_OK(hipMemsetAsync(areaB, 0, (size_t)n * 4, st));
        hipLaunchKernelGGL(mark_border_kernel, dim3(nblk(2 * (H + W), 256)), dim3(256), 0, st, LB, areaB, H, W);
        hipLaunchKernelGGL(fill_holes_apply_kernel, dim3(g), dim3(256), 0, st, mrk, LB, areaB, n);
        if (ccl_run(mrk, 1, LB, H, W, st)) return 1;
    }
    // marker ids = 1 + rank of the component's root among all roots (scipy's label order), written straight into the watershed's start map
    const int* boff = nullptr;
    const bool bitmaps = !pixel_scans && !three_pass;
    int* unl = areaB;  // free again: per-root count of unlabelled mask pixels
    int* wl3 = marker; // big-window tier list
    int* lmin = bitmaps ? lminbuf : LB;  // (LB: free after nuc_marker_out_kernel)
    if (!pixel_scans && bitmaps) {  // per-component state and heap ranges from the root bitmap, before the pass that fills in the label extremes
        hipLaunchKernelGGL(roots_setup_kernel, dim3(nblk(nw, 256)), dim3(256), 0, st, bitsA, nw, areaA, hoff, hcnt, unl, lmin, lmax, cbox, H, W, small + 16);
        KCHECK();
    }
    if (bitmaps) {
        if (scan_exclusive_popc(bitsB, wpre, nw, scantmp, st, &boff)) return 1;
        if (n_inst_out) hipLaunchKernelGGL(count_roots_from_bits_kernel, dim3(1), dim3(1), 0, st, bitsB, wpre, boff, nw, n_inst_out, small);
        // (C) watershed(-inner, marker, mask)   (postproc.py:378)
        if (wide && bitmaps) hipLaunchKernelGGL(nuc_marker_out4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, (const int4*)LB, bitsB, wpre, boff, (const uint32_t*)msk, (int4*)labels_out,
                                     (long long)n / 4, (const int4*)LA, lmin, lmax, (const uint32_t*)mrk);
        else hipLaunchKernelGGL(nuc_marker_out_bits_kernel, dim3(g), dim3(256), 0, st, LB, bitsB, wpre, boff, msk, labels_out, n, LA, lmin, lmax);
    } else {
        if (scan_exclusive(LB, rank, n, scantmp, st, true, &boff)) return 1;
        if (n_inst_out) hipLaunchKernelGGL(count_roots_from_scan_kernel, dim3(1), dim3(1), 0, st, LB, rank, boff, n, n_inst_out, small);
        hipLaunchKernelGGL(nuc_marker_out_kernel, dim3(g), dim3(256), 0, st, LB, rank, boff, msk, labels_out, n);
    }
    if (!pixel_scans && !bitmaps) {  // (three separate labellings: the markers' L doubles as lmin, so the setup has to follow the start-map pass)
        hipLaunchKernelGGL(roots_setup_kernel, dim3(nblk(nw, 256)), dim3(256), 0, st, bitsA, nw, areaA, hoff, hcnt, unl, lmin, lmax, cbox, H, W, small + 16);
        KCHECK();
    } else if (pixel_scans) {   // heap offsets = exclusive scan of the kept components' areas at their roots; the same pass initialises the per-root state
        const int nbk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
        hipLaunchKernelGGL(scan_block_cap_kernel, dim3(nbk), dim3(256), 0, st, LA, areaA, msk, hoff, nbk > 1 ? scantmp : (int*)nullptr, n, hcnt, unl, lmin, lmax, cbox, H, W);
        KCHECK();
        if (nbk > 1) {
            if (scan_exclusive(scantmp, scantmp, nbk, scantmp + ((nbk + 3) & ~3), st)) return 1;
            hipLaunchKernelGGL(scan_add_kernel, dim3(nbk), dim3(256), 0, st, hoff, scantmp, n);
            KCHECK();
        }
    }
    if (wide && bitmaps) {
        hipLaunchKernelGGL(ws_seed_bbox4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, inst, row_stride, pix_stride, (const uint32_t*)msk, (const int4*)labels_out,
                           (const int4*)LA, hoff, hcnt, hkey, hidx, H, W, unl, lmin, lmax, cbox);
    } else {
        hipLaunchKernelGGL(ws_seed_kernel, dim3(g), dim3(256), 0, st, inst, row_stride, pix_stride, msk, labels_out, LA, hoff, hcnt, hkey, hidx, H, W, unl,
                           lmin, lmax, bitmaps ? 1 : 0);
        hipLaunchKernelGGL(ws_bbox_kernel, dim3(g), dim3(256), 0, st, LA, msk, cbox, H, W, lmin, lmax);
    }
    int* counts = small + 8;  // [0] window tier, [1] LDS-heap tier, [2] global tier
    if (pixel_scans) hipLaunchKernelGGL(ws_worklist_kernel, dim3(g), dim3(256), 0, st, hcnt, areaA, unl, cbox, lmin, lmax, wl, rank, wl3, wl4, counts, n, LA, msk);
    else hipLaunchKernelGGL(ws_worklist_bits_kernel, dim3(nblk(nw, 256) < 4096 ? nblk(nw, 256) : 4096), dim3(256), 0, st, bitsA, nw, hcnt, areaA, unl, cbox, lmin, lmax, wl, rank, wl3, wl4, counts, n);
    {
        auto k_tiny = ws_flood_window_kernel<WS_TINY_WIN, WS_TINY_CAP, 4>;
        auto k_small = ws_flood_window_kernel<WS_WIN_CAP, WS_LDS_CAP, 2>;
        auto k_big = ws_flood_window_kernel<WS_BIGWIN_CAP, WS_BIGHEAP_CAP, 1>;
        constexpr int lds_small = 2 * (WS_LDS_CAP * 10 + WS_WIN_CAP * 8), lds_big = WS_BIGHEAP_CAP * 10 + WS_BIGWIN_CAP * 8;
        constexpr int lds_tiny = 4 * (WS_TINY_CAP * 10 + WS_TINY_WIN * 8);
        static bool attr_done[64] = {};
        if (cerb_attr_needed(attr_done)) {
            PP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_small), hipFuncAttributeMaxDynamicSharedMemorySize, lds_small));
            PP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tiny), hipFuncAttributeMaxDynamicSharedMemorySize, lds_tiny));
            PP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_big), hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
        }
        // The tiers touch disjoint components and each one ends with the tail of its longest flood: fork them onto side streams
        // (created once per device) so that the tails overlap instead of adding up; `st` resumes when all of them are done.
        SideStreams* ss = side_streams();
        if (!ss) return cerb_set_error("cerb_postproc_nuclei: side stream creation failed");
        if (cerb_dev_getenv("CERB_PP_DEBUG_COUNTS")) {  // developer probe: components per flood tier
            int hc[8] = {};
            PP_OK(hipStreamSynchronize(st));
            PP_OK(hipMemcpy(hc, counts, sizeof(hc), hipMemcpyDeviceToHost));
            fprintf(stderr, "flood tiers: small-window %d, lds-heap %d, global-heap %d, big-window %d, tiny-window %d\n", hc[0], hc[1], hc[2], hc[3], hc[4]);
        }
        static const bool serial_floods = cerb_dev_getenv("CERB_PP_SERIAL_FLOODS") != nullptr;  // developer probe: the tiers one after the other on `st`
        SideStreams serial_ss;
        if (serial_floods) {
            serial_ss = *ss;
            serial_ss.s[0] = serial_ss.s[1] = serial_ss.s[2] = st;
            ss = &serial_ss;
        }
        PP_OK(hipEventRecord(ss->fork, st));
        for (int i = 0; i < 3; ++i) PP_OK(hipStreamWaitEvent(ss->s[i], ss->fork, 0));
        hipLaunchKernelGGL(k_tiny, dim3(256 * 3), dim3(256), lds_tiny, ss->s[0], inst, row_stride, pix_stride, msk, LA, labels_out, wl4, counts + 4, cbox,
                           H, W, small + 3, small + 24);
        hipLaunchKernelGGL(k_small, dim3(256 * 2), dim3(128), lds_small, ss->s[1], inst, row_stride, pix_stride, msk, LA, labels_out, wl, counts + 0, cbox,
                           H, W, small + 3, small + 25);
        // (launching this tier FIRST -- its longest flood is the critical path -- measured worse: its 151-KB workgroups then keep the other tiers off
        // every CU that holds one; behind them it starts where they have left)
        hipLaunchKernelGGL(k_big, dim3(256), dim3(64), lds_big, st, inst, row_stride, pix_stride, msk, LA, labels_out, wl3, counts + 3, cbox, H, W,
                           small + 3, small + 26);
        // the fill of the single-label components (isolated nuclei: no flood) touches other pixels than any flood: beside them, not before them
        if (wide) hipLaunchKernelGGL(ws_fill_single4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ss->s[2], (const uint32_t*)msk, (const int4*)LA, lmin, lmax, (int4*)labels_out, (long long)n / 4);
        else hipLaunchKernelGGL(ws_fill_single_kernel, dim3(g), dim3(256), 0, ss->s[2], msk, LA, lmin, lmax, labels_out, n);
        hipLaunchKernelGGL(ws_flood_lds_kernel, dim3(256 * 2), dim3(256), 0, ss->s[2], inst, row_stride, pix_stride, msk, labels_out, rank, counts + 1,
                           hoff, hcnt, hkey, hidx, H, W, small + 3, small + 27);
        hipLaunchKernelGGL(ws_flood_kernel, dim3(256 * 4), dim3(256), 0, ss->s[2], inst, row_stride, pix_stride, msk, labels_out, wl, counts + 2, hoff,
                           hcnt, hkey, hidx, H, W, small + 3, n, small + 28);
        for (int i = 0; i < 3; ++i) {
            PP_OK(hipEventRecord(ss->join[i], ss->s[i]));
            PP_OK(hipStreamWaitEvent(st, ss->join[i], 0));
        }
    }
    KCHECK();
    if (exact_ties) {
        // Components whose result depends on skimage's heap-layout order between equal-valued markers were counted in small[3]:
        // when there is one, the whole map is re-flooded through the literal emulation of that heap (both kernels return at once
        // when the count is zero -- no host round trip decides this).
        constexpr int lds_exact = EX_LDS * 8 + 8 + EX_LDS * 4 + 4;
        static bool attr_done2[64] = {};
        if (cerb_attr_needed(attr_done2))
            PP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(ws_exact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_exact));
        hipLaunchKernelGGL(ws_exact_reset_kernel, dim3(g), dim3(256), 0, st, mrk, msk, labels_out, n, small + 3);
        hipLaunchKernelGGL(ws_exact_kernel, dim3(1), dim3(64), lds_exact, st, inst, row_stride, pix_stride, msk, labels_out, hkey, hidx, H, W, small + 3);
        KCHECK();
    }
    if (n_ambiguous_out) PP_OK(hipMemcpyAsync(n_ambiguous_out, small + 3, 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int gland_lumen(const float* inst, int H, int W, long long row_stride, int pix_stride, float thr, int min_size, int ksize,
                       int32_t* labels_out, int32_t* n_inst_out, void* ws, size_t ws_bytes, hipStream_t st, const char* who) {
    if (!inst || !labels_out || !ws || H <= 0 || W <= 0) return cerb_set_error(std::string(who) + ": bad arguments");
    if (ws_bytes < cerb_pp_workspace_bytes(H, W)) return cerb_set_error(std::string(who) + ": workspace too small");
    if ((long long)H * W >= (1ll << 31)) return cerb_set_error(std::string(who) + ": map too large (H*W must be < 2^31)");
    if (ksize < 1 || ksize > 32) return cerb_set_error(std::string(who) + ": structuring element size out of range (ds_factor too small/large)");
    const int n = H * W;
    Carve cv{(char*)ws, ws_bytes};
    int* L = (int*)cv.take((size_t)n * 4);
    int* area = (int*)cv.take((size_t)n * 4);
    int* flag = (int*)cv.take((size_t)n * 4);
    int* rank = (int*)cv.take((size_t)n * 4);
    int* lab = (int*)cv.take((size_t)n * 4);
    uint8_t* fg = (uint8_t*)cv.take(n);
    int* scantmp = (int*)cv.take((size_t)(n / SCAN_ITEMS + 4096) * 4 * 2);
    int* small = (int*)cv.take(256);
    // per-instance tables: at most n / max(min_size,1) + 1 instances
    const int max_inst = n / (min_size > 0 ? min_size : 1) + 2;
    Box* box = (Box*)cv.take((size_t)max_inst * sizeof(Box));
    int* carea = (int*)cv.take((size_t)max_inst * 4);
    int* coff = (int*)cv.take((size_t)max_inst * 4);
    int* scantmp2 = (int*)cv.take((size_t)(max_inst / SCAN_ITEMS + 4096) * 4 * 2);
    if (!scantmp2) return cerb_set_error(std::string(who) + ": workspace carve failed");
    // the rest of the workspace holds the crops of one batch: 1 (dil) + 4 (L) + 4 (border) bytes per crop pixel
    const size_t crop_cap = cv.left / 10 > 256 ? (cv.left - 4096) / 10 : 0;
    uint8_t* dil = (uint8_t*)cv.take(crop_cap);
    int* cL = (int*)cv.take(crop_cap * 4);
    int* cB = (int*)cv.take(crop_cap * 4);
    if (!cB) return cerb_set_error(std::string(who) + ": workspace carve failed");
    const unsigned g = grid_for(n);

    hipLaunchKernelGGL(gl_threshold_kernel, dim3(g), dim3(256), 0, st, inst, row_stride, pix_stride, H, W, thr, fg);
    PP_OK(hipMemsetAsync(area, 0, (size_t)n * 4, st));
    if (ccl_run(fg, 1, L, H, W, st, area)) return 1;
    hipLaunchKernelGGL(ccl_keep_roots_kernel, dim3(g), dim3(256), 0, st, L, area, min_size, flag, n);
    if (scan_exclusive(flag, rank, n, scantmp, st)) return 1;
    hipLaunchKernelGGL(ccl_relabel_kernel, dim3(g), dim3(256), 0, st, L, flag, rank, lab, n);
    hipLaunchKernelGGL(count_from_scan_kernel, dim3(1), dim3(1), 0, st, flag, rank, n, small, (const int*)nullptr);
    PP_OK(hipMemsetAsync(labels_out, 0, (size_t)n * 4, st));
    int n_inst = 0;  // the one host round trip of this path: 4 bytes of metadata (number of instances)
    PP_OK(hipMemcpyAsync(&n_inst, small, 4, hipMemcpyDeviceToHost, st));
    PP_OK(hipStreamSynchronize(st));
    if (n_inst_out) PP_OK(hipMemcpyAsync(n_inst_out, small, 4, hipMemcpyDeviceToDevice, st));
    if (n_inst == 0) return 0;
    if (n_inst + 1 > max_inst) return cerb_set_error(std::string(who) + ": internal: instance table overflow");
    hipLaunchKernelGGL(box_init_kernel, dim3(nblk(n_inst + 1, 256)), dim3(256), 0, st, box, n_inst + 1, H, W);
    hipLaunchKernelGGL(box_accum_kernel, dim3(g), dim3(256), 0, st, lab, box, H, W);
    hipLaunchKernelGGL(box_pad_kernel, dim3(nblk(n_inst + 1, 256)), dim3(256), 0, st, box, carea, n_inst, H, W, ksize * 2);
    if (scan_exclusive(carea, coff, n_inst + 1, scantmp2, st)) return 1;
    KCHECK();
    // batch instances so that the crops of a batch fit the crop workspace (needs the crop areas on the host)
    std::string err;
    int* h_off = (int*)malloc((size_t)(n_inst + 2) * 4);
    int* h_area = (int*)malloc((size_t)(n_inst + 2) * 4);
    PP_OK(hipMemcpyAsync(h_off, coff, (size_t)(n_inst + 1) * 4, hipMemcpyDeviceToHost, st));
    PP_OK(hipMemcpyAsync(h_area, carea, (size_t)(n_inst + 1) * 4, hipMemcpyDeviceToHost, st));
    PP_OK(hipStreamSynchronize(st));
    Spans se;
    ellipse_spans(ksize, &se);
    int first = 1, rc = 0;
    while (first <= n_inst) {
        int last = first;
        size_t tot = (size_t)h_area[first];
        if (tot > crop_cap) {
            rc = cerb_set_error(std::string(who) + ": one instance crop exceeds the workspace");
            break;
        }
        while (last + 1 <= n_inst && tot + (size_t)h_area[last + 1] <= crop_cap) tot += (size_t)h_area[++last];
        const int cnt = last - first + 1;
        const int tot_i = (int)tot;
        const unsigned gb = grid_for(tot_i);
        hipLaunchKernelGGL(gl_dilate_kernel, dim3(gb), dim3(256), 0, st, lab, box, coff, first, cnt, tot_i, W, se, dil);
        hipLaunchKernelGGL(gl_bg_init_kernel, dim3(gb), dim3(256), 0, st, dil, tot_i, cL, cB);
        hipLaunchKernelGGL(gl_bg_merge_kernel, dim3(gb), dim3(256), 0, st, dil, box, coff, first, cnt, tot_i, cL);
        hipLaunchKernelGGL(gl_bg_border_kernel, dim3(gb), dim3(256), 0, st, box, coff, first, cnt, tot_i, cL, cB);
        hipLaunchKernelGGL(gl_paste_kernel, dim3(gb), dim3(256), 0, st, dil, box, coff, first, cnt, tot_i, cL, cB, W, labels_out);
        first = last + 1;
    }
    free(h_off);
    free(h_area);
    if (rc) return rc;
    KCHECK();
    return 0;
}

extern "C" int cerb_postproc_gland(const float* inst, int H, int W, long long row_stride, int pix_stride, float ds, int32_t* labels_out,
                                   int32_t* n_inst_out, void* ws, size_t ws_bytes, void* hip_stream) {
    // python: ksize = int((11-1)*ds); min_size = int(1000*(ds**2))   (postproc.py:272-274,287)
    return gland_lumen(inst, H, W, row_stride, pix_stride, 0.55f, (int)(1000.0 * ((double)ds * (double)ds)), (int)(10.0 * (double)ds), labels_out,
                       n_inst_out, ws, ws_bytes, (hipStream_t)hip_stream, "cerb_postproc_gland");
}
extern "C" int cerb_postproc_lumen(const float* inst, int H, int W, long long row_stride, int pix_stride, float ds, int32_t* labels_out,
                                   int32_t* n_inst_out, void* ws, size_t ws_bytes, void* hip_stream) {
    return gland_lumen(inst, H, W, row_stride, pix_stride, 0.5f, (int)(150.0 * ((double)ds * (double)ds)), (int)(2.0 * (double)ds), labels_out,
                       n_inst_out, ws, ws_bytes, (hipStream_t)hip_stream, "cerb_postproc_lumen");
}
extern "C" int cerb_mask_lumen_by_gland(int32_t* lumen, const int32_t* gland, long long n_pix, void* hip_stream) {
    if (!lumen || !gland || n_pix < 0) return cerb_set_error("cerb_mask_lumen_by_gland: bad arguments");
    hipLaunchKernelGGL(mask_lumen_kernel, dim3(grid_for(n_pix)), dim3(256), 0, (hipStream_t)hip_stream, lumen, gland, n_pix);
    KCHECK();
    return 0;
}

// =================================================================================================================
// Tissue-mask regions (infer/wsi.py:724 `measurements.label(wsi_mask)`): 4-connected components of mask != 0,
// ids in raster order of each component's first pixel (scipy.ndimage.label's order).
// =================================================================================================================
__global__ void binarize_kernel(const uint8_t* __restrict__ m, long long row_stride, int H, int W, uint8_t* __restrict__ fg) {
    const long long n = (long long)H * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        fg[p] = m[(p / W) * row_stride + (p % W)] != 0 ? 1 : 0;
}
extern "C" int cerb_label_mask(const uint8_t* mask, long long row_stride, int H, int W, int32_t* labels_out, int32_t* n_out, void* ws,
                               size_t ws_bytes, void* hip_stream) {
    if (!mask || !labels_out || !n_out || !ws || H <= 0 || W <= 0) return cerb_set_error("cerb_label_mask: bad arguments");
    if (ws_bytes < cerb_pp_workspace_bytes(H, W)) return cerb_set_error("cerb_label_mask: workspace too small");
    if ((long long)H * W >= (1ll << 31)) return cerb_set_error("cerb_label_mask: map too large (H*W must be < 2^31)");
    hipStream_t st = (hipStream_t)hip_stream;
    const int n = H * W;
    Carve cv{(char*)ws, ws_bytes};
    int* L = (int*)cv.take((size_t)n * 4);
    int* flag = (int*)cv.take((size_t)n * 4);
    int* rank = (int*)cv.take((size_t)n * 4);
    uint8_t* fg = (uint8_t*)cv.take(n);
    int* scantmp = (int*)cv.take((size_t)(n / SCAN_ITEMS + 4096) * 4 * 2);
    if (!scantmp) return cerb_set_error("cerb_label_mask: workspace carve failed");
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(binarize_kernel, dim3(g), dim3(256), 0, st, mask, row_stride, H, W, fg);
    if (ccl_run(fg, 1, L, H, W, st)) return 1;
    hipLaunchKernelGGL(ccl_keep_roots_kernel, dim3(g), dim3(256), 0, st, L, (const int*)L, INT_MIN, flag, n);  // every root: area test always true
    if (scan_exclusive(flag, rank, n, scantmp, st)) return 1;
    hipLaunchKernelGGL(ccl_relabel_kernel, dim3(g), dim3(256), 0, st, L, flag, rank, labels_out, n);
    hipLaunchKernelGGL(count_from_scan_kernel, dim3(1), dim3(1), 0, st, flag, rank, n, n_out, (const int*)nullptr);
    KCHECK();
    return 0;
}

// =================================================================================================================
// Which border cv2.findContours(...)[0][0] is, for an instance made of several 8-connected pieces (a lumen cut by its gland's
// edge, a gland partly overwritten by a later one): OpenCV returns top-level contours most-recently-found first, so element
// [0] is the outer border of the piece whose first pixel comes LAST in raster order (loader/postproc.py:29-33 takes [0][0]).
// One union-find pass over "same id, 8-neighbour" links; start[id-1] = max over the pieces of their first pixel.
// =================================================================================================================
// IDX = int for maps below 2^31 pixels (4 bytes of workspace per pixel), long long above (8): a 0.5-mpp scan of a large section is 60000 x 50000.
template <typename IDX>
__device__ __forceinline__ IDX uf_find_t(const IDX* L, IDX x) {
    IDX p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}
template <typename IDX>
__device__ __forceinline__ void uf_union_t(IDX* L, IDX a, IDX b) {
    bool done;
    do {
        a = uf_find_t<IDX>(L, a);
        b = uf_find_t<IDX>(L, b);
        if (a < b) {
            const IDX old = atomicMin(&L[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const IDX old = atomicMin(&L[a], b);
            done = (old == a);
            a = old;
        } else
            done = true;
    } while (!done);
}
template <typename IDX>
__global__ void cc8_init_kernel(const int* __restrict__ lab, long long ls, int H, int W, IDX* __restrict__ L) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y, x;
        pix_yx(p, W, invW, y, x);
        L[p] = lab[y * ls + x] > 0 ? (IDX)p : (IDX)-1;
    }
}
template <typename IDX>
__global__ void cc8_merge_kernel(const int* __restrict__ lab, long long ls, int H, int W, IDX* L) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y, x;
        pix_yx(p, W, invW, y, x);
        const int id = lab[y * ls + x];
        if (id <= 0) continue;
        if (x > 0 && lab[y * ls + x - 1] == id) uf_union_t<IDX>(L, (IDX)p, (IDX)(p - 1));
        if (y > 0) {
            const int* up = lab + (y - 1) * ls;
            if (up[x] == id) uf_union_t<IDX>(L, (IDX)p, (IDX)(p - W));
            else {  // the diagonal links only matter when the pixel above does not already join all three
                if (x > 0 && up[x - 1] == id) uf_union_t<IDX>(L, (IDX)p, (IDX)(p - W - 1));
                if (x + 1 < W && up[x + 1] == id) uf_union_t<IDX>(L, (IDX)p, (IDX)(p - W + 1));
            }
        }
    }
}
template <typename IDX>
__global__ void cc8_flatten_kernel(IDX* L, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (L[p] < 0) continue;
        L[p] = uf_find_t<IDX>(L, (IDX)p);
    }
}
template <typename IDX>
__global__ void cc8_last_root_kernel(const int* __restrict__ lab, long long ls, int H, int W, const IDX* __restrict__ L, int n_inst,
                                     long long* __restrict__ start) {
    const long long n = (long long)H * W;
    const double invW = 1.0 / (double)W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        if (L[p] != (IDX)p) continue;
        int y, x;
        pix_yx(p, W, invW, y, x);
        const int id = lab[y * ls + x];
        if (id >= 1 && id <= n_inst) atomicMax((unsigned long long*)&start[id - 1], (unsigned long long)p);
    }
}
template <typename IDX>
static void contour_start_launch(const int32_t* labels, long long ls, int H, int W, int n_inst, long long* start, void* ws, hipStream_t st) {
    const long long n = (long long)H * W;
    IDX* L = (IDX*)ws;
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(cc8_init_kernel<IDX>, dim3(g), dim3(256), 0, st, labels, ls, H, W, L);
    hipLaunchKernelGGL(cc8_merge_kernel<IDX>, dim3(g), dim3(256), 0, st, labels, ls, H, W, L);
    hipLaunchKernelGGL(cc8_flatten_kernel<IDX>, dim3(g), dim3(256), 0, st, L, n);
    hipLaunchKernelGGL(cc8_last_root_kernel<IDX>, dim3(g), dim3(256), 0, st, labels, ls, H, W, (const IDX*)L, n_inst, start);
}
extern "C" size_t cerb_inst_contour_start_workspace_bytes(int H, int W) {
    const long long n = (long long)(H > 0 ? H : 0) * (W > 0 ? W : 0);
    return (size_t)n * (n >= (1ll << 31) ? 8 : 4);
}
extern "C" int cerb_inst_contour_start(const int32_t* labels, long long lab_row_stride, int H, int W, int n_inst, long long* start, void* ws,
                                       size_t ws_bytes, void* hip_stream) {
    if (!labels || !start || !ws || H <= 0 || W <= 0 || n_inst < 0) return cerb_set_error("cerb_inst_contour_start: bad arguments");
    const bool wide = (long long)H * W >= (1ll << 31);
    if (ws_bytes < cerb_inst_contour_start_workspace_bytes(H, W))
        return cerb_set_error("cerb_inst_contour_start: workspace too small (4 bytes per pixel, 8 for maps of 2^31 pixels and more)");
    if (n_inst == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    PP_OK(hipMemsetAsync(start, 0, (size_t)n_inst * 8, st));
    if (wide) contour_start_launch<long long>(labels, lab_row_stride, H, W, n_inst, start, ws, st);
    else contour_start_launch<int>(labels, lab_row_stride, H, W, n_inst, start, ws, st);
    KCHECK();
    return 0;
}
