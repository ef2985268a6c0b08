// Slide-level data movement of the WSI path on gfx950 (HBM-bound byte work):
//   cerb_synth_slide     : counter-based synthetic RGB slide (value depends only on seed and absolute (y,x,c), so any
//                          sharding of the slide over GPUs sees the same pixels)
//   cerb_gather_patches  : patch extraction with mirror padding -- replaces the padded-image slicing of
//                          reference loader/infer_loader.py:54-69 + np.pad(..., "reflect") of infer/tile.py:69 and the
//                          WSIStreamDataset reads of infer/wsi.py:936-950 for a device-resident slide
//   cerb_downsample2_inst: x0.5 bilinear resize of an INST probability map (cv2.resize(fx=0.5, INTER_LINEAR),
//                          reference infer/wsi.py:786-788) == exact 2x2 box average at this scale
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <string>

#include "../../include/cerberus_hip.h"

int cerb_set_error(const std::string& m);

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__global__ void synth_slide_kernel(uint8_t* __restrict__ out, long long h, long long w, long long y0, long long x0, uint32_t seed) {
    const long long n = h * w;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const long long y = y0 + p / w, x = x0 + p % w;
        const uint32_t hsh = mix32((uint32_t)x * 0x9E3779B1u ^ mix32((uint32_t)y + seed * 0x85EBCA77u));
        out[p * 3 + 0] = (uint8_t)(hsh);
        out[p * 3 + 1] = (uint8_t)(hsh >> 8);
        out[p * 3 + 2] = (uint8_t)(hsh >> 16);
    }
}

__device__ __forceinline__ long long reflect_idx(long long i, long long n) {  // numpy "reflect": no edge repeat, period 2(n-1)
    if (n == 1) return 0;
    const long long period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

// tiles[k][y][x][c] = slide[reflect(tl_y[k] + y - slide_y0)][reflect(tl_x[k] + x)][c]; reflection is about the FULL slide
// extent (full_h rows starting at absolute row 0); rows outside [slide_y0, slide_y0 + h) after reflection are an error the
// host prevents by giving each rank its band plus halo.
__global__ void gather_patches_kernel(const uint8_t* __restrict__ slide, long long h, long long w, long long slide_y0, long long full_h,
                                      const long long* __restrict__ tl_y, const long long* __restrict__ tl_x, int win,
                                      uint8_t* __restrict__ tiles) {
    const int k = blockIdx.y;
    const long long ty = tl_y[k], tx = tl_x[k];
    const int per = win * win;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const int y = i / win, x = i % win;
        const long long sy = reflect_idx(ty + y, full_h) - slide_y0;
        const long long sx = reflect_idx(tx + x, w);
        uint8_t r = 0, g = 0, b = 0;
        if (sy >= 0 && sy < h) {
            const uint8_t* s = slide + (sy * w + sx) * 3;
            r = s[0];
            g = s[1];
            b = s[2];
        }
        uint8_t* d = tiles + ((long long)k * per + i) * 3;
        d[0] = r;
        d[1] = g;
        d[2] = b;
    }
}

// cv2.resize(src, (0, 0), fx=0.5, fy=0.5, INTER_LINEAR): dst size cvRound(n * 0.5) (half to even), source coordinate 2 d + 0.5 ->
// samples 2d and 2d+1 with weights (.5, .5); where 2d >= n - 1 (odd n, last output) the weights are (1, 0).  With a region
// (infer/wsi.py:742-776) every source sample is first multiplied by [region_lab[nearest(sy, sx)] == region_id], the nearest
// mapping being cv2.resize(INTER_NEAREST)'s min(floor(s * mh / h), mh - 1).
__global__ void downsample2_inst_kernel(const float* __restrict__ src, long long row_stride, int pix_stride, int h, int w, int ho, int wo,
                                        const int* __restrict__ region_lab, long long lab_row_stride, int mh, int mw, int region_id,
                                        float* __restrict__ dst) {
    const long long n = (long long)ho * wo;
    const double fy = 1.0 / ((double)h / (double)mh), fx = 1.0 / ((double)w / (double)mw);  // cv2: 1 / inv_scale
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / wo), x = (int)(p % wo);
        const int y0 = min(2 * y, h - 1), y1 = min(2 * y + 1, h - 1), x0 = min(2 * x, w - 1), x1 = min(2 * x + 1, w - 1);
        const float wy1 = (2 * y < h - 1) ? 0.5f : 0.f, wx1 = (2 * x < w - 1) ? 0.5f : 0.f;
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        float m00 = 1.f, m01 = 1.f, m10 = 1.f, m11 = 1.f;
        if (region_lab) {
            const int my0 = min((int)floor(y0 * fy), mh - 1), my1 = min((int)floor(y1 * fy), mh - 1);
            const int mx0 = min((int)floor(x0 * fx), mw - 1), mx1 = min((int)floor(x1 * fx), mw - 1);
            m00 = region_lab[my0 * lab_row_stride + mx0] == region_id ? 1.f : 0.f;
            m01 = region_lab[my0 * lab_row_stride + mx1] == region_id ? 1.f : 0.f;
            m10 = region_lab[my1 * lab_row_stride + mx0] == region_id ? 1.f : 0.f;
            m11 = region_lab[my1 * lab_row_stride + mx1] == region_id ? 1.f : 0.f;
        }
        const float* a = src + (long long)y0 * row_stride;
        const float* b = src + (long long)y1 * row_stride;
        const long long o0 = (long long)x0 * pix_stride, o1 = (long long)x1 * pix_stride;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float top = (a[o0 + c] * m00) * wx0 + (a[o1 + c] * m01) * wx1;
            const float bot = (b[o0 + c] * m10) * wx0 + (b[o1 + c] * m11) * wx1;
            dst[p * 2 + c] = top * wy0 + bot * wy1;
        }
    }
}
// Patch-Class tissue map (infer/wsi.py:688-716): x0.25 INTER_NEAREST of the class canvas times the INTER_NEAREST-resized mask
__global__ void pclass_tissue_kernel(const float* __restrict__ pclass, long long row_stride, int h, int w, const uint8_t* __restrict__ mask,
                                     long long mask_row_stride, int mh, int mw, int ph, int pw, float* __restrict__ dst) {
    const long long n = (long long)ph * pw;
    const double fy = 1.0 / ((double)ph / (double)mh), fx = 1.0 / ((double)pw / (double)mw);
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / pw), x = (int)(p % pw);
        const int sy = min(4 * y, h - 1), sx = min(4 * x, w - 1);  // floor(d / 0.25)
        float v = pclass[sy * row_stride + sx];
        if (mask) {
            const int my = min((int)floor(y * fy), mh - 1), mx = min((int)floor(x * fx), mw - 1);
            v *= mask[my * mask_row_stride + mx] != 0 ? 1.f : 0.f;
        }
        dst[p] = v;
    }
}

static unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}
#define SK_CHECK()                                                                                    \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) return cerb_set_error(std::string("kernel launch: ") + hipGetErrorString(e_)); \
    } while (0)

extern "C" int cerb_synth_slide(uint8_t* out, long long h, long long w, long long y0, long long x0, uint32_t seed, void* hip_stream) {
    if (!out || h <= 0 || w <= 0) return cerb_set_error("cerb_synth_slide: bad arguments");
    hipLaunchKernelGGL(synth_slide_kernel, dim3(grid_for(h * w)), dim3(256), 0, (hipStream_t)hip_stream, out, h, w, y0, x0, seed);
    SK_CHECK();
    return 0;
}
extern "C" int cerb_gather_patches(const uint8_t* slide, long long h, long long w, long long slide_y0, long long full_h, const long long* tl_y,
                                   const long long* tl_x, int n, int win, uint8_t* tiles, void* hip_stream) {
    if (!slide || !tl_y || !tl_x || !tiles || n <= 0 || win <= 0 || h <= 0 || w <= 0) return cerb_set_error("cerb_gather_patches: bad arguments");
    hipLaunchKernelGGL(gather_patches_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)hip_stream, slide, h, w, slide_y0, full_h, tl_y, tl_x, win, tiles);
    SK_CHECK();
    return 0;
}
static int half_size(int n) { return (int)lrint(n * 0.5); }  // cvRound
extern "C" int cerb_half_size(int n) { return half_size(n); }
extern "C" int cerb_downsample2_inst_region(const float* src, long long row_stride, int pix_stride, int h, int w, const int32_t* region_lab,
                                            long long lab_row_stride, int mh, int mw, int region_id, float* dst, void* hip_stream) {
    if (!src || !dst || h < 1 || w < 1 || (region_lab && (mh < 1 || mw < 1))) return cerb_set_error("cerb_downsample2_inst: bad arguments");
    const int ho = half_size(h), wo = half_size(w);
    if (ho < 1 || wo < 1) return cerb_set_error("cerb_downsample2_inst: map too small");
    hipLaunchKernelGGL(downsample2_inst_kernel, dim3(grid_for((long long)ho * wo)), dim3(256), 0, (hipStream_t)hip_stream, src, row_stride, pix_stride,
                       h, w, ho, wo, region_lab, lab_row_stride, region_lab ? mh : 1, region_lab ? mw : 1, region_id, dst);
    SK_CHECK();
    return 0;
}
extern "C" int cerb_downsample2_inst(const float* src, long long row_stride, int pix_stride, int h, int w, float* dst, void* hip_stream) {
    return cerb_downsample2_inst_region(src, row_stride, pix_stride, h, w, nullptr, 0, 0, 0, 0, dst, hip_stream);
}
extern "C" int cerb_pclass_tissue_map(const float* pclass, long long row_stride, int h, int w, const uint8_t* mask, long long mask_row_stride,
                                      int mh, int mw, float* dst, void* hip_stream) {
    if (!pclass || !dst || h < 1 || w < 1 || (mask && (mh < 1 || mw < 1))) return cerb_set_error("cerb_pclass_tissue_map: bad arguments");
    const int ph = (int)lrint(h * 0.25), pw = (int)lrint(w * 0.25);
    if (ph < 1 || pw < 1) return cerb_set_error("cerb_pclass_tissue_map: map too small");
    hipLaunchKernelGGL(pclass_tissue_kernel, dim3(grid_for((long long)ph * pw)), dim3(256), 0, (hipStream_t)hip_stream, pclass, row_stride, h, w, mask,
                       mask_row_stride, mask ? mh : 1, mask ? mw : 1, ph, pw, dst);
    SK_CHECK();
    return 0;
}

// =================================================================================================================
// Instance table: the segmented reductions of get_inst_info_dict (loader/postproc.py:12-75)
// -- bounding box, area, first moments (cv2.moments m10/m00, m01/m00 of the binary instance mask) and the histogram of
// the type map over the instance -- computed on the device from the label map; contour tracing stays a "next" row.
// table layout per instance id (1-based row id-1): int64[16] = {area, sum_x, sum_y, y1, y2(excl), x1, x2(excl), first,
//                                                              type_count[0..7]},  first = min(y*W + x) over its pixels
// =================================================================================================================
__global__ void inst_table_init_kernel(long long* __restrict__ t, int n_inst, int H, int W) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inst; i += gridDim.x * blockDim.x) {
        long long* r = t + 16ll * i;
        for (int k = 0; k < 16; ++k) r[k] = 0;
        r[3] = H;
        r[5] = W;
        r[7] = (long long)H * W;
    }
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
__global__ void inst_table_kernel(const int* __restrict__ lab, long long lab_row_stride, const uint8_t* __restrict__ type, long long type_row_stride,
                                  int H, int W, int n_inst, unsigned long long* __restrict__ t) {
    const long long n = (long long)H * W;
    const int lane = threadIdx.x & 63;
    for (long long p0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) - lane; p0 < n; p0 += (long long)gridDim.x * blockDim.x) {
        const long long p = p0 + lane;
        int y = 0, x = 0, l = 0, ty = 0;
        if (p < n) {
            y = (int)(p / W);
            x = (int)(p % W);
            l = lab[y * lab_row_stride + x];
            if (l < 0 || l > n_inst) l = 0;
            if (l && type) ty = type[y * type_row_stride + x] & 7;
        }
        const unsigned long long act = __ballot(l != 0);
        if (!act) continue;
        const int l0 = __shfl(l, __ffsll((long long)act) - 1);
        const bool uniform = __ballot(l != 0 && l != l0) == 0;  // interior of a large instance: one atomic set per wave
        if (uniform) {
            const long long sx = wave_sum_ll(l ? x : 0), sy = wave_sum_ll(l ? y : 0);
            const int y1 = wave_min_i(l ? y : H), y2 = wave_max_i(l ? y + 1 : 0), x1 = wave_min_i(l ? x : W), x2 = wave_max_i(l ? x + 1 : 0);
            unsigned long long* r = t + 16ll * (l0 - 1);
            if (type) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c = __popcll(__ballot(l != 0 && ty == k));
                    if (lane == 0 && c) atomicAdd(r + 8 + k, (unsigned long long)c);
                }
            }
            if (lane == 0) {
                atomicAdd(r + 0, (unsigned long long)__popcll(act));
                atomicAdd(r + 1, (unsigned long long)sx);
                atomicAdd(r + 2, (unsigned long long)sy);
                atomicMin((long long*)r + 3, (long long)y1);
                atomicMax((long long*)r + 4, (long long)y2);
                atomicMin((long long*)r + 5, (long long)x1);
                atomicMax((long long*)r + 6, (long long)x2);
                atomicMin((long long*)r + 7, p0 + (__ffsll((long long)act) - 1));
            }
        } else if (l) {
            unsigned long long* r = t + 16ll * (l - 1);
            atomicAdd(r + 0, 1ull);
            atomicAdd(r + 1, (unsigned long long)x);
            atomicAdd(r + 2, (unsigned long long)y);
            atomicMin((long long*)r + 3, (long long)y);
            atomicMax((long long*)r + 4, (long long)y + 1);
            atomicMin((long long*)r + 5, (long long)x);
            atomicMax((long long*)r + 6, (long long)x + 1);
            atomicMin((long long*)r + 7, p);
            if (type) atomicAdd(r + 8 + ty, 1ull);
        }
    }
}
extern "C" int cerb_inst_table(const int32_t* labels, long long lab_row_stride, const uint8_t* type_map, long long type_row_stride, int h, int w,
                               int n_inst, long long* table, void* hip_stream) {
    if (!labels || !table || h <= 0 || w <= 0 || n_inst < 0) return cerb_set_error("cerb_inst_table: bad arguments");
    if (n_inst == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(inst_table_init_kernel, dim3(grid_for(n_inst)), dim3(256), 0, st, table, n_inst, h, w);
    hipLaunchKernelGGL(inst_table_kernel, dim3(grid_for((long long)h * w)), dim3(256), 0, st, labels, lab_row_stride, type_map, type_row_stride, h, w,
                       n_inst, (unsigned long long*)table);
    SK_CHECK();
    return 0;
}

// =================================================================================================================
// cerb_relabel: out[y][x] = map[labels[y][x]] (map[0] must be 0; ids outside [0, n_map) become 0).  Used by the sharded
// post-processing (cerberus_amd/shard_postproc.py) to turn band-local instance ids into slide-global ones.
// =================================================================================================================
__global__ void relabel_kernel(const int* __restrict__ lab, long long lab_row_stride, const int* __restrict__ map, int n_map, int H, int W,
                               int* __restrict__ out, long long out_row_stride) {
    const long long n = (long long)H * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p % W);
        const int l = lab[y * lab_row_stride + x];
        out[y * out_row_stride + x] = (l > 0 && l < n_map) ? map[l] : 0;
    }
}
extern "C" int cerb_relabel(const int32_t* labels, long long lab_row_stride, const int32_t* map, int n_map, int h, int w, int32_t* out,
                            long long out_row_stride, void* hip_stream) {
    if (!labels || !map || !out || h <= 0 || w <= 0 || n_map < 1) return cerb_set_error("cerb_relabel: bad arguments");
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for((long long)h * w)), dim3(256), 0, (hipStream_t)hip_stream, labels, lab_row_stride, map, n_map,
                       h, w, out, out_row_stride);
    SK_CHECK();
    return 0;
}

// =================================================================================================================
// cerb_inst_contour_count / cerb_inst_contour_points: outer border of every instance, as the reference extracts it with
// cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0] on the instance's cropped binary mask (loader/postproc.py:29-41).
// Restated from Suzuki & Abe's border following (CVGIP 30, 1985) with OpenCV's conventions: 8-connected foreground, chain codes
// 0..7 = E, NE, N, NW, W, SW, S, SE, the trace starts at the instance's first pixel in raster order, looks clockwise from NW for
// its first neighbour and then always continues counter-clockwise from the direction it came from; CHAIN_APPROX_SIMPLE keeps a
// point exactly when the chain code changes.  One thread follows one instance (serial by nature; a slide has 10^5..10^6
// instances); `first` comes from cerb_inst_table column 7.  Two passes: count, (exclusive scan by the caller), write.
// An instance made of several 8-connected pieces yields the border of the piece holding its first pixel.
// =================================================================================================================
template <bool WRITE>
__global__ void inst_contour_kernel(const int* __restrict__ lab, long long stride, int H, int W, int n_inst, const long long* __restrict__ table,
                                    int* __restrict__ counts, const long long* __restrict__ offsets, int* __restrict__ points) {
    const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1}, DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inst; i += gridDim.x * blockDim.x) {
        const long long* r = table + 16ll * i;
        const int id = i + 1;
        int n = 0;
        int* out = WRITE ? points + 2 * offsets[i] : nullptr;
        if (r[0] > 0) {
            const long long first = r[7];
            const int x0 = (int)(first % W), y0 = (int)(first / W);
            auto fg = [&](int x, int y) { return x >= 0 && x < W && y >= 0 && y < H && lab[y * stride + x] == id; };
            auto emit = [&](int x, int y) {
                if (WRITE) {
                    out[2 * n] = x;
                    out[2 * n + 1] = y;
                }
                ++n;
            };
            int s = 4;
            const int s_stop = 4;
            int x1 = 0, y1 = 0;
            do {
                s = (s - 1) & 7;
                x1 = x0 + DX[s];
                y1 = y0 + DY[s];
            } while (!fg(x1, y1) && s != s_stop);
            if (s == s_stop) {
                emit(x0, y0);  // single-pixel domain
            } else {
                int x3 = x0, y3 = y0, px = x0, py = y0, prev_s = s ^ 4;
                const long long guard = 4ll * (r[0] + 4) + 16;  // a border visits each pixel at most 4 times
                for (long long it = 0; it < guard; ++it) {
                    int x4, y4;
                    do {  // counter-clockwise from the direction just after the one we came from
                        ++s;
                        x4 = x3 + DX[s & 7];
                        y4 = y3 + DY[s & 7];
                    } while (!fg(x4, y4));
                    s &= 7;
                    if (s != prev_s) {
                        emit(px, py);
                        prev_s = s;
                    }
                    px += DX[s];
                    py += DY[s];
                    if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
                    x3 = x4;
                    y3 = y4;
                    s = (s + 4) & 7;
                }
            }
        }
        if (!WRITE) counts[i] = n;
    }
}
extern "C" int cerb_inst_contour_count(const int32_t* labels, long long lab_row_stride, int h, int w, int n_inst, const long long* table,
                                       int32_t* counts, void* hip_stream) {
    if (!labels || !table || !counts || h <= 0 || w <= 0 || n_inst < 0) return cerb_set_error("cerb_inst_contour_count: bad arguments");
    if (n_inst == 0) return 0;
    hipLaunchKernelGGL(inst_contour_kernel<false>, dim3((n_inst + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, labels, lab_row_stride, h, w, n_inst,
                       table, counts, (const long long*)nullptr, (int*)nullptr);
    SK_CHECK();
    return 0;
}
extern "C" int cerb_inst_contour_points(const int32_t* labels, long long lab_row_stride, int h, int w, int n_inst, const long long* table,
                                        const long long* offsets, int32_t* points, void* hip_stream) {
    if (!labels || !table || !offsets || !points || h <= 0 || w <= 0 || n_inst < 0) return cerb_set_error("cerb_inst_contour_points: bad arguments");
    if (n_inst == 0) return 0;
    hipLaunchKernelGGL(inst_contour_kernel<true>, dim3((n_inst + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, labels, lab_row_stride, h, w, n_inst,
                       table, (int*)nullptr, offsets, points);
    SK_CHECK();
    return 0;
}

// =================================================================================================================
// cerb_resample_box / cerb_resample_area: the reader's reduction of a stored pyramid level to the processing resolution, on the device.
// A 40x scan (0.25 mpp; levels x1, x4, x16) is read at the 0.5 mpp the network runs on -- the reference's reader does that resize per patch on
// its 12 DataLoader workers (infer/wsi.py:936-950 through tiatoolbox's read_bounds); cerberus_amd/reader.py::read_bounds is the host statement of
// it (exact k x k box means for an integer factor, area means on one global grid otherwise) and these kernels return ITS bytes: the box kernel
// rounds the integer sum half to even (= rint of the exact mean), the area kernel applies the host's float32 tables with the host's operation
// order (multiply, add, one division per axis, rows first; no fused multiply-add), so a slab filled here equals a slab filled from read_bounds.
// src: uint8 RGB rows of the level's window (row stride in BYTES), edges replicate (indices are clamped / come clamped in the tables).
// =================================================================================================================
__global__ void resample_box_kernel(const uint8_t* __restrict__ src, long long ss, int src_rows, int src_cols, int k, uint8_t* __restrict__ dst,
                                    long long ds, int out_rows, int out_cols) {
    const long long n = (long long)out_rows * out_cols;
    const unsigned kk = (unsigned)(k * k);
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / out_cols), x = (int)(p % out_cols);
        unsigned s[3] = {0u, 0u, 0u};
        for (int i = 0; i < k; ++i) {
            const int sy = min(y * k + i, src_rows - 1);
            const uint8_t* row = src + sy * ss;
            for (int j = 0; j < k; ++j) {
                const uint8_t* q = row + 3ll * min(x * k + j, src_cols - 1);
                s[0] += q[0];
                s[1] += q[1];
                s[2] += q[2];
            }
        }
        uint8_t* o = dst + y * ds + 3ll * x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned q = s[c] / kk, r = s[c] % kk;
            o[c] = (uint8_t)(q + ((2u * r > kk || (2u * r == kk && (q & 1u))) ? 1u : 0u));
        }
    }
}
// HIP's __fmul_rn / __fadd_rn are the plain operators and device code is compiled with floating-point contraction on: a + s * w becomes one fused
// multiply-add -- one rounding instead of numpy's two, a different byte in ~1 of 10^6 pixels.  The product goes through an empty asm statement:
// the compiler has to materialise it (rounded) before the sum.
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__global__ void resample_area_kernel(const uint8_t* __restrict__ src, long long ss, uint8_t* __restrict__ dst, long long ds, int out_rows, int out_cols,
                                     const int* __restrict__ ridx, const float* __restrict__ rw, const float* __restrict__ rws,
                                     const int* __restrict__ rrep, int rt, const int* __restrict__ cidx, const float* __restrict__ cw,
                                     const float* __restrict__ cws, const int* __restrict__ crep, int ct) {
    const long long n = (long long)out_rows * out_cols;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / out_cols), x = (int)(p % out_cols);
        const int yrep = rrep[y], xrep = crep[x];
        const float ysum = rws[y], xsum = cws[x];
        // the row pass of source column c for this output row: ((0 + s0 * w0) + s1 * w1 + ...) / ws, or the repeated row past the level's end
        auto col = [&](int c, float* v) {
            if (yrep >= 0) {
                const uint8_t* q = src + yrep * ss + 3ll * c;
                v[0] = (float)q[0];
                v[1] = (float)q[1];
                v[2] = (float)q[2];
                return;
            }
            float a[3] = {0.f, 0.f, 0.f};
            for (int t = 0; t < rt; ++t) {
                const uint8_t* q = src + ridx[(long long)y * rt + t] * ss + 3ll * c;
                const float w = rw[(long long)y * rt + t];
                a[0] = a[0] + mul_rounded((float)q[0], w);
                a[1] = a[1] + mul_rounded((float)q[1], w);
                a[2] = a[2] + mul_rounded((float)q[2], w);
            }
            v[0] = a[0] / ysum;
            v[1] = a[1] / ysum;
            v[2] = a[2] / ysum;
        };
        float r[3];
        if (xrep >= 0) {
            col(xrep, r);
        } else {
            float a[3] = {0.f, 0.f, 0.f};
            for (int t = 0; t < ct; ++t) {
                float v[3];
                col(cidx[(long long)x * ct + t], v);
                const float w = cw[(long long)x * ct + t];
                a[0] = a[0] + mul_rounded(v[0], w);
                a[1] = a[1] + mul_rounded(v[1], w);
                a[2] = a[2] + mul_rounded(v[2], w);
            }
            r[0] = a[0] / xsum;
            r[1] = a[1] / xsum;
            r[2] = a[2] / xsum;
        }
        uint8_t* o = dst + y * ds + 3ll * x;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)fminf(fmaxf(rintf(r[c]), 0.f), 255.f);
    }
}
static unsigned resample_grid(long long n) {
    const long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}
extern "C" int cerb_resample_box(const uint8_t* src, long long src_row_stride, int src_rows, int src_cols, int k, uint8_t* dst, long long dst_row_stride,
                                 int out_rows, int out_cols, void* hip_stream) {
    if (!src || !dst || src_rows <= 0 || src_cols <= 0 || k < 1 || k > 64 || out_rows < 0 || out_cols < 0) return cerb_set_error("cerb_resample_box: bad arguments");
    if (out_rows == 0 || out_cols == 0) return 0;
    hipLaunchKernelGGL(resample_box_kernel, dim3(resample_grid((long long)out_rows * out_cols)), dim3(256), 0, (hipStream_t)hip_stream, src, src_row_stride,
                       src_rows, src_cols, k, dst, dst_row_stride, out_rows, out_cols);
    SK_CHECK();
    return 0;
}
extern "C" int cerb_resample_area(const uint8_t* src, long long src_row_stride, int src_rows, int src_cols, uint8_t* dst, long long dst_row_stride, int out_rows,
                                  int out_cols, const int32_t* row_idx, const float* row_w, const float* row_wsum, const int32_t* row_rep, int row_taps,
                                  const int32_t* col_idx, const float* col_w, const float* col_wsum, const int32_t* col_rep, int col_taps, void* hip_stream) {
    if (!src || !dst || src_rows <= 0 || src_cols <= 0 || out_rows < 0 || out_cols < 0 || !row_idx || !row_w || !row_wsum || !row_rep || row_taps < 1 ||
        !col_idx || !col_w || !col_wsum || !col_rep || col_taps < 1)
        return cerb_set_error("cerb_resample_area: bad arguments");
    if (out_rows == 0 || out_cols == 0) return 0;
    hipLaunchKernelGGL(resample_area_kernel, dim3(resample_grid((long long)out_rows * out_cols)), dim3(256), 0, (hipStream_t)hip_stream, src, src_row_stride, dst,
                       dst_row_stride, out_rows, out_cols, row_idx, row_w, row_wsum, row_rep, row_taps, col_idx, col_w, col_wsum, col_rep, col_taps);
    SK_CHECK();
    return 0;
}
