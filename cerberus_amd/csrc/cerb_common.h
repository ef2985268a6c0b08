// Shared declarations for the gfx950 kernels of the Cerberus tiled-inference path.
// Everything here is CDNA4-only (wave64, v_mfma_f32_32x32x2_f32); there is no other target.
#pragma once
#include "cerb_dev.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CERB_WAVE 64

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_igemm.hip).  "Swapped" GEMM: D[cout][pixel] = W[cout][k] X[k][pixel]
// so that one lane owns one pixel and 4 consecutive couts per accumulator quad -> float4 NHWC stores,
// and the accumulators are directly the B operand of a following 1x1 GEMM (head fusion).
// ---------------------------------------------------------------------------------------------
// hipFuncSetAttribute is per device: remember it per device, not per process (a host process may drive several handles)
static inline bool cerb_attr_needed(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

struct ConvParams {
    const float* in;     // NHWC fp32 [G][N][H][W][Cin]  (MODE 1: the skip tensor)
    const float* prev;   // MODE 1 only: [G][N][H/2][W/2][Cin], bilinearly upsampled x2 and added to `in`
    const float* wpack;  // packed, BN-folded weights (see pack_conv_weights)
    const float* bias;   // [G][Cout] BN-folded bias
    const float* resid;  // optional residual [G][N][Ho][Wo][Cout] added before ReLU
    float* out;          // [G][N][Ho][Wo][Cout]
    int N, H, W, Cin, Ho, Wo, Cout;
    int relu;
    int tiles_x, tiles_y;  // output tiles per image (set by the launcher; with a region of interest: tiles of the region)
    // Region of interest in OUTPUT pixels, [roi_y0, roi_y1) x [roi_x0, roi_x1); all zero = the whole map.  Only the work items that
    // overlap it are computed (conv_wino: the decoder convs whose output is centre-cropped afterwards, cerb_api.hip: crop_rois);
    // ty_off / tx_off are the launcher's item offsets of the region.
    int roi_y0, roi_y1, roi_x0, roi_x1;
    int ty_off, tx_off;
    int groups;
    long long in_gs, prev_gs, w_gs, bias_gs, resid_gs, out_gs;  // per-group strides in elements
    // conv_wino4p.hip only: `in` / `out` are tile-planar tensors (planar_elems below); blocks per image column / row incl. the guard ring
    int pl_byp, pl_bxp;
    int level_tag;        // conv_wino4p.hip: 1 = last decoder level, 0 = the level below it -- picks the kernel SYMBOL only (profiler statistics per launch size)
    // conv_wino4.hip / conv_wino4b.hip, training forward only: per 16 x 16 output block the (sum, sum of squares) of every output channel over the
    // block's pixels inside the image, [group][bn_bpg blocks][Cout][2] doubles -- the BatchNorm behind the convolution finalises its batch
    // statistics from these instead of reading the output again (train_kernels.hip: bn_finalize_kernel).  nullptr: not produced.
    double* bn_part;
    int bn_bpg;           // blocks per group = N * ceil(Ho / 16) * ceil(Wo / 16) (conv_wino4b.hip's packed items: ceil(tiles / 16), cerb_wino4b_bn_blocks)
    int pk_ty, pk_tx, pk_ntile;  // conv_wino4b.hip, packed items (set by the launcher): 4x4 tiles per image column / row, tiles per group
    // conv_wino4.hip / conv_wino4b.hip, training backward (bn_part set as well): this launch is the data gradient whose output IS the gradient behind a
    // BatchNorm + ReLU and its only writer -- bst_y = that BatchNorm's input [G][N][Ho][Wo][Cout] (group stride bst_y_gs), bst_* = its batch mean / rstd /
    // gamma / beta [G][Cout]: the output stage leaves (sum dz', sum dz' xhat) per block in bn_part instead of (sum, sum of squares).  nullptr: not asked.
    const float *bst_y, *bst_mean, *bst_rstd, *bst_gamma, *bst_beta;
    long long bst_y_gs;
    int pk_off;           // 1: keep the block form on maps that would take packed items (cerb_net_set_packed_items(net, 0), A/B)
};

// Tile-planar layout of the decoder's private tensors (conv_wino4p.hip; producer upsample2_add_planar, consumers conv_wino4p and the heads):
//   [group][image n][block row by + 1][block column bx + 1][16-channel plane cc][i = y & 3][j = x & 3][tile m = 4 ((y >> 2) & 3) + ((x >> 2) & 3)][16 ch]
// a block is 16 x 16 pixels; one 16-channel plane of a block is 16 KiB whose 1-KiB rows hold ONE pixel position (i, j) of all 16 4x4 tiles:
// exactly what one store instruction of a Winograd wave produces (lane = (tile, channel quad)) and what one patch load of the consumer reads.
// Every image carries a ring of guard blocks (row / column 0 and BY + 1 / BX + 1) that stay zero, and pixels of edge blocks beyond the image
// stay zero as well (nobody writes them): the 3x3 convolution's zero padding is DATA, no kernel masks an edge.
__host__ __device__ static inline int cerb_planar_blocks(int px) { return (px + 15) / 16 + 2; }
static inline long long cerb_planar_elems(int n, int h, int w, int c) {
    return (long long)n * cerb_planar_blocks(h) * cerb_planar_blocks(w) * (c / 16) * 4096;
}
__host__ __device__ __forceinline__ long long cerb_planar_offset(int n, int y, int x, int cc, int byp, int bxp, int ncc) {  // floats; channel 0 of plane cc
    return ((((long long)n * byp + (y >> 4) + 1) * bxp + (x >> 4) + 1) * ncc + cc) * 4096 + ((((y & 3) << 2) + (x & 3)) * 16 + (((y >> 2) & 3) << 2) + ((x >> 2) & 3)) * 16;
}

// Fused output head (head.hip): 1x1 64->96 (+BN+ReLU) -> 1x1 96->out_ch -> softmax -> INST probs / TYPE argmax
struct HeadParams {
    const float* feat;   // [N][H][W][64] NHWC (decoder output)
    const float* w1p;    // packed W1 (BN folded): [6 blk][4 g][64 lane][4]  (cerb_api.hip: head packing)
    const float* b1;     // [96]
    const float* w2p;    // packed W2: [6 blk][64 lane][4]  (rows >= out_ch are zero)
    const float* w2q;    // W2 for head_group_kernel<true> (4x4x1 matrix instructions): [set 2][blk 6][64 lane][r 4] = W2[4 set + (l & 3)][16 blk + 4 (l >> 4) + r]
    const float* b2;     // [32] (padded)
    int N, H, W;
    int out_ch;          // 3 or 7 (<= 8)
    int kind;            // 0 = INST (write softmax ch 1..2 as float2), 1 = TYPE (write argmax)
    int crop_y0, crop_x0, out_h, out_w;   // centre crop window in tile coordinates
    int roi;             // 1: only the pixel blocks that overlap the crop window are computed (logits must be NULL)
    int rows, row0, xa0, nxb;  // set by the launcher: the kernel walks 16-pixel blocks (n, row0 + r, xa0 + 16 xb), r < rows, xb < nxb
    float* logits;       // optional [N][H][W][out_ch] NHWC (tests)
    float* out_inst;     // kind 0: float [..][2]
    unsigned char* out_type_u8;  // kind 1, optional
    long long* out_type_i64;     // kind 1, optional
    const long long* tile_off;   // optional per-tile element offset (in pixels) into the destination canvas
    long long tile_stride;       // pixels between consecutive tiles when tile_off == nullptr
    long long row_stride;        // pixels between consecutive output rows
    unsigned int* absmax_bits;   // optional device uint32: atomicMax of the float bits of |logit| over every pixel this launch evaluates (cerb_forward_io.logit_absmax)
    int feat_planar;             // head_group_kernel only: `feat` is a tile-planar tensor (cerb_planar_offset) with pl_byp x pl_bxp blocks per image
    int pl_byp, pl_bxp;
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Bijective blockIdx -> logical id so that each XCD (bid % 8) owns a contiguous chunk of logical ids
    // (neighbouring tiles / cout-blocks share halo + weights in that XCD's L2).  Speed only.
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
