/* cerberus_hip.h -- C ABI of libcerberus_hip.so: the MI355X (gfx950) tiled-inference hot path of Cerberus.
 *
 * This is the drop-in boundary.  The reference is pure Python; the objects this ABI
 * replaces are cited per entry point as <reference file>:<line>.  No torch types cross the boundary: plain
 * pointers (device pointers are raw hipMalloc/torch addresses), sizes, and a hipStream_t passed as void*.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure (CERB_ERR_ALLOC = 2 when a workspace allocation did not fit the
 *     device -- the one failure a caller can react to, e.g. with a smaller batch --, 1 otherwise); cerb_last_error() returns a thread-local
 *     message (the reference's only error convention is Python exceptions / assert, e.g.
 *     loader/postproc.py:390,395 and load_state_dict(strict=True) at infer/base.py:45).
 *   - one cerb_net per GPU / per process; calls on one handle must be serialised by the caller (the reference
 *     calls run_step from the main thread only, infer/tile.py:349-359).
 *   - the library owns packed weights and its workspace; the caller owns every input/output buffer.
 *   - activations are NHWC fp32; image tiles are NHWC uint8 (what infer_step receives, models/run_desc.py:439-441).
 */
#ifndef CERBERUS_HIP_H
#define CERBERUS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cerb_net cerb_net;
#define CERB_ERR_ALLOC 2

int cerb_version(void);
const char* cerb_last_error(void);

/* ---- network construction: replaces create_model / NetDesc.__init__ (models/net_desc.py:23-103,203) ----------
 * decoder_names[i]  e.g. "Lumen","Gland","Nuclei","Nuclei#TYPE","Gland#TYPE","Patch-Class"  (decoder_kwargs order,
 *                   already filtered by considered_tasks, models/net_desc.py:61-63)
 * head_names[i]     "INST" | "TYPE" | "OUT"        (an output head of that decoder)
 * out_ch[i]         number of output channels of that head (3,3,3,7,3,9 in models/paramset.yml:46-60)
 * One entry per OUTPUT HEAD.  A decoder with several heads (models/net_desc.py:81-87: `{"Gland": {"INST": 3, "TYPE": 3}}` builds one decoder
 * trunk and a ModuleDict of heads over it, :196-198) is listed once per head, the decoder name repeated: the trunk's convolutions are loaded and
 * run ONCE, every head reads its features; out[] / logits[] of cerb_forward_io stay per entry.  (Inference only: a train-packed handle refuses it.)
 * Only encoder_backbone_name == "resnet34" exists in this library. */
int cerb_net_create(const char* const* decoder_names, const char* const* head_names, const int* out_ch,
                    int n_decoders, cerb_net** out_net);
void cerb_net_destroy(cerb_net* net);

/* ---- weights: replaces net.load_state_dict(saved_state_dict, strict=True) (infer/base.py:28-45) ----------------
 * One call per state-dict entry, using the reference's key names (backbone.layer1.0.conv1.weight, ...).  `data`
 * is a HOST pointer to contiguous fp32 in PyTorch's layout (conv: [Cout][Cin][kh][kw]); the library copies it.
 * Keys with integer payloads (num_batches_tracked) and the unused backbone.fc.* are accepted and ignored.
 * cerb_net_finalize folds eval-mode BatchNorm into the convolutions, packs for the MFMA kernels, uploads, and
 * fails (strict) if any required key is missing or has the wrong shape. */
int cerb_net_load_tensor(cerb_net* net, const char* key, const float* data, const int64_t* shape, int ndim);
int cerb_net_finalize(cerb_net* net);

/* ---- forward + output wrapper: replaces infer_step (models/run_desc.py:439-502) incl. NetDesc.forward
 * (models/net_desc.py:144-200), softmax / channel slice / centre crop / argmax, and -- through tile_off /
 * row_stride -- the stitching of infer/tile.py:141-163 (outputs can be written straight into a slide canvas).
 *
 * For decoder i (same order as cerb_net_create):
 *   INST head : out[i] -> float  [..][2]   softmax channels 1..2        (run_desc.py:452-455)
 *   TYPE head : out[i] -> int64 or uint8 class id (argmax of softmax)   (run_desc.py:490-491), see type_is_u8
 *   OUT  head : out[i] -> float class id broadcast over out_h x out_w   (Patch-Class, run_desc.py:480-487)
 * Destination addressing (in pixels): tile n, row y, col x  ->  (tile_off ? tile_off[n] : n*tile_stride)
 *                                                               + y*row_stride + x
 * logits[i] (optional, may be NULL) receives the raw head logits NHWC [N][H][W][out_ch] ([N][out_ch] for OUT). */
typedef struct cerb_forward_io {
    const uint8_t* tiles;      /* device, [N][H][W][3] uint8 RGB */
    int n, h, w;               /* batch, tile height/width (multiples of 16) */
    int out_h, out_w;          /* centre-crop size (cropping_center, misc/utils.py:94-104) */
    void* const* out;          /* [n_decoders] device pointers (NULL = head not wanted) */
    float* const* logits;      /* optional [n_decoders] device pointers or NULL */
    const long long* tile_off; /* optional device array [N] of pixel offsets into the destination */
    long long tile_stride;     /* used when tile_off == NULL; 0 means out_h*out_w */
    long long row_stride;      /* 0 means out_w */
    int type_is_u8;            /* 0: TYPE heads write int64 (reference dtype); 1: uint8 (device-resident canvas) */
    float* const* feats;       /* optional [6]: x0,x1,x2,x3,conv_map(x4),x4 NHWC dumps for tests, or NULL */
    const float* tiles_f32;    /* used when tiles == NULL: device [N][H][W][3] float pixel values (any floats; divided by 255 in fp32 like
                                * `imgs / 255.0`, models/net_desc.py:147) -- NetDesc.forward on inputs that are not whole numbers in 0..255 */
    unsigned int* logit_absmax; /* optional device uint32 [n_decoders] or NULL: the data-aware precision guard.  The head kernels raise word i
                                * (atomic max) to the IEEE-754 bit pattern of the largest |logit| of dense head i over every pixel this forward
                                * evaluates (non-negative floats order like their bits) -- the caller zeroes it when it wants a fresh maximum and
                                * reads it back as float.  Words of heads that were not requested, and of the OUT head, are left alone.  The
                                * F(4x4,3x3) default of cerb_net_set_conv_algo holds the 1e-4 contract while a model's logits stay in the range
                                * the parity fixtures cover; this is how a caller watches REAL data for batches that leave it
                                * (cerberus_amd/wsi.py counts them and can re-run them on conv_algo 1). */
} cerb_forward_io;

int cerb_net_forward(cerb_net* net, const cerb_forward_io* io, void* hip_stream);

/* Algorithm of the 3x3 stride-1 convolutions (90 % of the FLOPs), all on the exact-fp32 matrix instructions unless stated:
 *   6 (default) = Winograd F(4x4,3x3) for maps of 16 x 16 pixels and more -- conv_wino4b.hip up to 64 x 64, conv_wino4.hip above --
 *                 and F(2x2,3x3) below; the choice looks at the layer's geometry only, never at the batch size or the region of
 *                 interest, so a tile's values do not depend on what it is batched with;
 *   5, 7        = F(4x4,3x3) everywhere, with conv_wino4.hip / conv_wino4b.hip (36 products per 4x4 outputs instead of 144; transform
 *                 points 0, +-1, +-2, inf; probability maps within 4e-6 of the fp64 evaluation, tests/tools/dev_wino4_numerics.py);
 *   1           = Winograd F(2x2,3x3) (conv_wino.hip, round 1's default; same fp32 products, 2.25x fewer multiplies than direct);
 *   0           = direct implicit GEMM (conv_igemm.hip).
 * cuDNN makes the same kind of choice per layer for the reference (torch.backends.cudnn, models/run_desc.py:447).  All meet the 1e-4
 * bar; the switch exists for A/B measurement and for the parity tests of each path.
 * (Three F(2x2) variants that lost their A/B -- bf16x3-split products and two other work decompositions -- live in scripts/experiments/,
 * outside the product library.)
 * The training step follows the same rule for its forward and data-gradient convolutions (filter transform on the device). */
int cerb_net_set_conv_algo(cerb_net* net, int algo);
/* Layout of the private tensors (skip + upsample, first conv output, level output / head features; models/net_desc.py:182-198) of the LAST
 * decoder level and of the level below it (both 64 channels; a level takes part when its maps are above 64 x 64 pixels):
 *   1 (default) = tile-planar (cerb_common.h: cerb_planar_offset) through upsample2_add_planar -> conv_wino4p.hip x2 -> heads: a Winograd
 *                 wave's stores are contiguous 1-KiB rows and its patch loads whole lines, zero padding is data;
 *   0           = NHWC through conv_wino4.hip (round 2's path).
 * Same arithmetic in the same order: the outputs are bit-identical (tests/test_net_gpu.py).  Applies with conv_algo 6 and head_algo 1. */
int cerb_net_set_planar(cerb_net* net, int enable);
/* Work items of the F(4x4,3x3) kernel for maps up to 64 x 64 pixels (conv_wino4b.hip) on maps whose sides are multiples of 4 but not of 16 -- the
 * 28 x 28 / 56 x 56 maps of the reference's default 448-pixel patch (infer/tile.py:43-106, models/backbone/resnet.py:273-286), in inference and in
 * the training step's forward and data-gradient convolutions:
 *   1 (default) = an item is 16 CONSECUTIVE 4x4 tiles of the batch (image-major, row-major): no padding tiles but in the launch's last item;
 *   0           = an item is a 16 x 16-pixel block (28 x 28 -> 2 x 2 blocks per image, 23 % of them padding).
 * A tile's arithmetic does not depend on the item it rides in: the outputs are bit-identical (tests/test_net_gpu.py).  For A/B. */
int cerb_net_set_packed_items(cerb_net* net, int enable);
/* Output heads (models/utils/net_layers.py:31-38): 1 (default) = every dense head of the batch in ONE grouped launch with the head's
 * weights resident in LDS (head_group_kernel) and the 96 -> 3 / 7 logits on 4x4x1 matrix instructions (no zero-padded rows); 2 = round 3's
 * grouped launch (logits on a 16-row instruction, 13 / 9 rows of zeros); 0 = one launch per head (round-1 head_kernel).  0 and 2 are
 * bit-identical to each other; 1 sums the 96 products of a logit in another order (a few 1e-7 on the logits).  For A/B. */
int cerb_net_set_head_algo(cerb_net* net, int algo);
/* Centre-crop regions of interest (default 1 = on).  infer_step keeps only the centre out_h x out_w window of every head
 * (models/run_desc.py:452-491 cropping_center; the reference's default geometry 448 -> 144 keeps 10 % of the pixels it computes).
 * With the switch on (any Winograd algorithm), every decoder level computes only the part of its maps that the kept window depends on
 * (3x3 conv: +1 pixel per layer, then out to whole 4x4 Winograd tiles; bilinear x2: +1 source pixel) and the heads only the window; the encoder still sees the whole
 * tile.  Results inside the window are bit-identical to the full computation (tests/test_net_gpu.py).  Ignored (full
 * computation) when full-size logits are requested or out == in. */
int cerb_net_set_crop_roi(cerb_net* net, int enable);

/* FLOPs (2*MAC) of one forward for the given geometry -- used by bench.py for the roofline figure. */
double cerb_net_flops(const cerb_net* net, int n, int h, int w);

/* Bytes of activation workspace the handles of this process hold right now (they grow with the largest batch run and stay until the handle is
 * destroyed).  A caller that prices its next job against the FREE device memory -- cerberus_amd/stream_bands.py::plan_slide, the counterpart of the
 * reference's chunk / tile sizing in infer/wsi.py:551-556 -- adds them back: they are allocated already and part of what a forward needs. */
size_t cerb_device_bytes_held(void);

/* Per-launch timing of the NEXT forwards with HIP events on the caller's stream (bench.py roofline leg; adds two
 * event records per kernel, so never enabled inside a timed region).  After a forward + stream sync,
 * cerb_net_profile_get(i) returns layer name, kernel family, algorithmic FLOPs and elapsed ms of launch i. */
int cerb_net_profile_enable(cerb_net* net, int enable);
int cerb_net_profile_count(cerb_net* net);
int cerb_net_profile_get(cerb_net* net, int idx, char* name, int name_cap, char* kernel, int kernel_cap, double* flops,
                         float* ms);

/* ---- post-processing: replaces PostProcInstErodedContourMap.post_process (loader/postproc.py:268-407) -----------
 * inst : device float [H][W][2] (ch0 = inner, ch1 = contour), `pix_stride` floats between consecutive pixels
 *        (2 for a packed INST map) and `row_stride` floats between rows -- lets it read a canvas window in place.
 * labels_out : device int32 [H][W]; ids as the reference assigns them (raster order of first pixel).
 * n_inst_out : device int32[1]; number of instances (nuclei: number of watershed markers; -1 when the map has no
 *        foreground at all, the branch in which the reference returns an all-zero float64 map, postproc.py:379-380).
 *        n_ambiguous_out : device int32[1] (nuclei only): number of watershed regions whose result depends on the order
 *        in which skimage's global binary heap releases marker pixels of bit-identical priority
 *        (skimage.segmentation.watershed, loader/postproc.py:378; DESIGN.md "watershed ties").  0: the parallel floods'
 *        label map is provably what skimage produces.  > 0: with exact_ties != 0 the map has been
 *        re-flooded on the device by a literal replay of that heap and is skimage's result as well; with exact_ties == 0 those regions
 *        keep the raster-order tie break.
 * ws / ws_bytes : caller-allocated device workspace of at least cerb_pp_workspace_bytes(H, W).
 * Streams: everything is ordered on `hip_stream`.  cerb_postproc_nuclei forks its independent flood tiers onto three internal
 * side streams per device (created on first use, joined back into `hip_stream` with events before it returns) -- the only
 * process-level state of the library; like the reference's run_step / post_process it is meant to be driven from one host
 * thread per GPU (SURVEY par.8b). */
size_t cerb_pp_workspace_bytes(int h, int w);
int cerb_postproc_nuclei(const float* inst, int h, int w, long long row_stride, int pix_stride, int32_t* labels_out,
                         int32_t* n_inst_out, int32_t* n_ambiguous_out, int exact_ties, void* ws, size_t ws_bytes, void* hip_stream);
int cerb_postproc_gland(const float* inst, int h, int w, long long row_stride, int pix_stride, float ds_factor,
                        int32_t* labels_out, int32_t* n_inst_out, void* ws, size_t ws_bytes, void* hip_stream);
int cerb_postproc_lumen(const float* inst, int h, int w, long long row_stride, int pix_stride, float ds_factor,
                        int32_t* labels_out, int32_t* n_inst_out, void* ws, size_t ws_bytes, void* hip_stream);
/* Lumen *= (Gland > 0)   (infer/tile.py:187-191, infer/wsi.py:799-804) */
int cerb_mask_lumen_by_gland(int32_t* lumen_labels, const int32_t* gland_labels, long long n_pix, void* hip_stream);

/* ---- the slide reader's reduction to the processing resolution, on the device ------------------------------------------
 * The reference reads every patch through tiatoolbox's read_bounds at `wsi_proc_mag` (infer/wsi.py:521-527, 936-950): for a 40x scan that is a
 * x2 reduction of level 0 done on its DataLoader workers.  Here the host only decodes the stored level's rows; these two apply the reduction
 * (cerberus_amd/reader.py::read_bounds is the host statement, and they return its bytes).
 *   src : uint8 RGB rows of the level's window, src_row_stride BYTES between rows, src_rows x src_cols pixels; dst likewise.
 *   cerb_resample_box : integer factor k, exact k x k means rounded half to even, source indices clamped to the window (edge replication).
 *   cerb_resample_area: area means on a global grid through per-axis tables (reader.area_tables): idx int32 [n_out][taps] source positions,
 *                       w float32 [n_out][taps], wsum float32 [n_out] (the divisor), rep int32 [n_out] (>= 0: repeat that source position);
 *                       rows first, then columns, float32 multiply / add / divide in the host's order, rint, clip. */
int cerb_resample_box(const uint8_t* src, long long src_row_stride, int src_rows, int src_cols, int k, uint8_t* dst,
                      long long dst_row_stride, int out_rows, int out_cols, void* hip_stream);
int cerb_resample_area(const uint8_t* src, long long src_row_stride, int src_rows, int src_cols, uint8_t* dst,
                       long long dst_row_stride, int out_rows, int out_cols, const int32_t* row_idx, const float* row_w,
                       const float* row_wsum, const int32_t* row_rep, int row_taps, const int32_t* col_idx, const float* col_w,
                       const float* col_wsum, const int32_t* col_rep, int col_taps, void* hip_stream);

/* ---- slide-level data movement (device-resident slide; replaces the DataLoader side of the hot loop) ------------
 * cerb_synth_slide    : synthetic RGB slab [h][w][3]; pixel value depends only on (seed, y0+y, x0+x) so every sharding
 *                       of a slide sees the same pixels (BASELINE.json configs[2..3] "synthetic WSI").
 * cerb_gather_patches : tiles[k] = win x win window with top-left (tl_y[k], tl_x[k]) in ABSOLUTE slide coordinates,
 *                       mirror-padded outside the slide like np.pad(..., "reflect") (infer/tile.py:69); `slide` holds
 *                       rows [slide_y0, slide_y0+h) of a slide that is full_h rows tall (a rank's band + halo).
 *                       Replaces loader/infer_loader.py:54-69 / infer/wsi.py:936-950 for a resident slide.
 * cerb_downsample2_inst: dst[cerb_half_size(h)][cerb_half_size(w)][2] = cv2.resize(src, (0,0), fx=0.5, fy=0.5, INTER_LINEAR)
 *                       of an INST map (infer/wsi.py:786-788): the 2x2 box average, and for an odd side whose half rounds up
 *                       (cvRound, half to even) the last row / column replicated.
 * cerb_downsample2_inst_region: the same after multiplying the map by one tissue region of the slide mask
 *                       (infer/wsi.py:742-776): region_lab = int32 label map of the mask (cerb_label_mask) cropped to the
 *                       region's bounding box [mh][mw], resized to [h][w] like cv2.resize(INTER_NEAREST); samples whose
 *                       mask label != region_id count as 0.  region_lab NULL = no mask.
 * cerb_pclass_tissue_map: dst[cvRound(h/4)][cvRound(w/4)] = cv2.resize(pclass, fx=fy=0.25, INTER_NEAREST) times the
 *                       INTER_NEAREST-resized slide mask (infer/wsi.py:688-716); mask NULL = all tissue.
 * cerb_label_mask     : 4-connected components of mask != 0, ids in raster order of first pixel (scipy.ndimage.label,
 *                       infer/wsi.py:724); ws as for cerb_postproc_*; n_out = device int32[1] number of regions. */
int cerb_synth_slide(uint8_t* out, long long h, long long w, long long y0, long long x0, uint32_t seed, void* hip_stream);
int cerb_gather_patches(const uint8_t* slide, long long h, long long w, long long slide_y0, long long full_h,
                        const long long* tl_y, const long long* tl_x, int n, int win, uint8_t* tiles, void* hip_stream);
int cerb_downsample2_inst(const float* src, long long row_stride, int pix_stride, int h, int w, float* dst,
                          void* hip_stream);
int cerb_half_size(int n);
int cerb_downsample2_inst_region(const float* src, long long row_stride, int pix_stride, int h, int w,
                                 const int32_t* region_lab, long long lab_row_stride, int mh, int mw, int region_id,
                                 float* dst, void* hip_stream);
int cerb_pclass_tissue_map(const float* pclass, long long row_stride, int h, int w, const uint8_t* mask,
                           long long mask_row_stride, int mh, int mw, float* dst, void* hip_stream);
int cerb_label_mask(const uint8_t* mask, long long row_stride, int h, int w, int32_t* labels_out, int32_t* n_out,
                    void* ws, size_t ws_bytes, void* hip_stream);

/* ---- instance table: the segmented reductions of get_inst_info_dict (loader/postproc.py:12-75) on the device --------
 * labels: int32 label map (row stride in elements); type_map: optional uint8 class map (NULL = none).
 * table : device int64 [n_inst][16], row id-1 = {area, sum_x, sum_y, y1, y2 (exclusive), x1, x2 (exclusive), first,
 *         type_count[0..7]}, first = min(y*w + x) over the instance's pixels (its first pixel in raster order)  ->  box [[y1,x1],[y2,x2]], centroid (sum_x/area, sum_y/area) == cv2.moments m10/m00, m01/m00,
 *         majority type / type_prob.  Contour tracing (cv2.findContours) is not part of this entry point. */
int cerb_inst_table(const int32_t* labels, long long lab_row_stride, const uint8_t* type_map, long long type_row_stride,
                    int h, int w, int n_inst, long long* table, void* hip_stream);

/* Outer border of every instance = cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0] of loader/postproc.py:29-41
 * (Suzuki-Abe border following, 8-connected foreground, start at pixel table[i][7] -- `table` is cerb_inst_table's output,
 * whose column 7 is the instance's first pixel: right for an instance that is ONE 8-connected piece; for several pieces
 * overwrite the column with cerb_inst_contour_start's result -- first move downwards/counter-clockwise, a point is kept where
 * the chain code changes).  Two passes:
 *   cerb_inst_contour_count  -> counts[i]  = number of points of instance i+1 (0 for absent ids)
 *   cerb_inst_contour_points -> points[2*(offsets[i] + k)] = x, [.. + 1] = y of point k; offsets = exclusive scan of counts (int64).
 * cerb_inst_contour_start -> start[i] = y*w + x of the pixel whose border findContours returns FIRST: OpenCV lists top-level
 *   contours most-recently-found first, so [0][0] belongs to the 8-connected piece whose first pixel comes last in raster
 *   order (one union-find pass over the label map; ws >= cerb_inst_contour_start_workspace_bytes(h, w) = 4 bytes per pixel,
 *   8 for maps of 2^31 pixels and more -- a 0.5-mpp scan of 60000 x 50000 is one).  Not modelled: a piece lying inside a HOLE of another
 *   piece of the same id is not a top-level contour in RETR_TREE and would be skipped by OpenCV.
 * OpenCV is not installed in this image: restated from the published algorithm, not pinned against the library. */
size_t cerb_inst_contour_start_workspace_bytes(int h, int w);
int cerb_inst_contour_start(const int32_t* labels, long long lab_row_stride, int h, int w, int n_inst, long long* start,
                            void* ws, size_t ws_bytes, void* hip_stream);
int cerb_inst_contour_count(const int32_t* labels, long long lab_row_stride, int h, int w, int n_inst, const long long* table,
                            int32_t* counts, void* hip_stream);
int cerb_inst_contour_points(const int32_t* labels, long long lab_row_stride, int h, int w, int n_inst, const long long* table,
                             const long long* offsets, int32_t* points, void* hip_stream);

/* out[y][x] = map[labels[y][x]] with map[0] == 0; ids outside [0, n_map) become 0.  Turns band-local instance ids into
 * slide-global ones after the count exchange of the sharded post-processing (the reference only needs ids to be unique:
 * uuid4 at infer/wsi.py:265,831). */
int cerb_relabel(const int32_t* labels, long long lab_row_stride, const int32_t* map, int n_map, int h, int w, int32_t* out,
                 long long out_row_stride, void* hip_stream);

/* ---- device timing helper: HIP events on the given stream (bench.py; torch.cuda.Event only sees torch's stream) */
/* ---- train-mode forward (BASELINE.json configs[4], forward half; models/run_desc.py:79-86) ---------------------------------------
 * cerb_net_set_fold_bn(net, 0), called BEFORE cerb_net_finalize, packs the network for training: raw convolution weights, the
 * BatchNorm affine parameters kept apart.  Such a network only serves cerb_net_forward_train (and an inference-packed one only
 * cerb_net_forward).  cerb_net_forward_train runs the reference's forward in `model.train()` mode: every BatchNorm normalises with
 * the statistics of the batch (biased variance, eps 1e-5), dropout of the Patch-Class branch takes its keep mask from the caller.
 *   tiles         : device uint8 [n][h][w][3]
 *   dropout_scale : device float [n][512] = keep / (1 - 0.3) of nn.Dropout(p=0.3) (models/net_desc.py:70), or NULL (no dropout)
 *   logits        : per decoder (cerb_net_create order) a device float buffer or NULL: dense heads [n][h][w][out_ch] (NHWC, full
 *                   resolution), Patch-Class [n][out_ch]
 * cerb_net_forward_train is the forward half alone; cerb_net_train_grads (below) records the activations on a tape and runs the backward
 * pass and returns the batch statistics from which the Python train_step updates the running ones. */
typedef struct cerb_train_io {
    const uint8_t* tiles;
    int n, h, w;
    const float* dropout_scale;
    float* const* logits;
} cerb_train_io;
int cerb_net_set_fold_bn(cerb_net* net, int fold);
/* Frozen modules of the sub-typing fine-tune (models/net_desc.py:105-142 `_freeze_weight`, called by train_step, models/run_desc.py:83-84): the
 * BatchNorm with state-dict prefix `bn_prefix` (e.g. "backbone.layer1.0.bn1") of a train-packed, finalized network runs in EVAL mode from now on --
 * cerb_net_forward_train / cerb_net_train_grads normalise it with these running statistics (host float[channels] each, eps 1e-5) instead of the
 * batch's and publish no batch statistics for it.  NULL statistics put it back in training mode.  Which parameters an optimiser then skips is the
 * caller's business (cerberus_amd/train.py drops the gradients of the frozen modules). */
int cerb_net_set_bn_eval(cerb_net* net, const char* bn_prefix, const float* running_mean, const float* running_var, int channels);
/* After an optimiser step: drop the packed weights of a finalized handle (activation workspaces, the training tape and the mode stay), so
 * that cerb_net_load_tensor of EVERY tensor + cerb_net_finalize install the updated parameters (models/run_desc.py:165 optimizer.step()). */
int cerb_net_begin_reload(cerb_net* net);
/* The optimiser's step without a host round trip (handles packed for training): copies each listed state-dict tensor from `dev_src[i]`
 * (device, float32, state-dict layout) into the handle's own copies and re-packs the conv weights with device kernels on `hip_stream`.
 * Keys the train-mode device path does not read (running statistics, num_batches_tracked, backbone.fc.*) are accepted and ignored. */
int cerb_net_update_params(cerb_net* net, int count, const char* const* keys, const float* const* dev_src, void* hip_stream);
int cerb_net_forward_train(cerb_net* net, const cerb_train_io* io, void* hip_stream);

/* cerb_net_train_grads: train-mode forward + the head losses + the backward pass of one step (models/run_desc.py:79-170 up to
 * all_loss.backward()), for a network packed with cerb_net_set_fold_bn(net, 0).  Weight gradients, the 3x3 data gradients and
 * the heads' pointwise layers run on the matrix cores, every reduction adds its partials in a fixed order (bitwise reproducible, no float
 * atomics).  Arrays are per decoder in
 * cerb_net_create order; a decoder whose target pointer is NULL contributes no loss.
 *   target[d]       : device float [n][h][w] class ids (Patch-Class: [n])          has_target[d] : device float [n]
 *   class_weight[d] : device float [out_ch] or NULL (see cerb_head_loss)           ce_w / dice_w / head_w : host floats
 *   loss_out        : device float [n_decoders], the value train_step reports per head (0 where no target was given)
 *   logits          : optional, as in cerb_train_io
 * Gradients are then read per state-dict key with cerb_net_grad_lookup (device pointer valid until the next call). */
typedef struct cerb_train_step_io {
    const uint8_t* tiles;
    int n, h, w;
    const float* dropout_scale;
    const float* const* target;
    const float* const* has_target;
    const float* const* class_weight;
    const float* ce_w;
    const float* dice_w;
    const float* head_w;
    float* loss_out;
    float* const* logits;
    const int* decoder_trained;  /* host int per decoder or NULL (= all): 0 reproduces a decoder outside train_decoder_list, whose
                                    gradients stay inside each of its blocks (models/net_desc.py:182 + conv_layers.py:44-53) */
    const float* const* pixel_weight; /* NULL, or per decoder NULL / device float [n][h][w]: the head's "#WEIGHT-MAP" target
                                         (models/run_desc.py:111-117), see cerb_head_loss_wmap */
} cerb_train_step_io;
/* Streams: everything is ordered on `hip_stream` as the caller sees it.  Inside, the weight gradients of the convolutions are queued on a second stream that
 * the handle owns -- forked from `hip_stream` by an event when a layer's output gradient is final, joined back into `hip_stream` before the call returns -- so work
 * the caller queues on `hip_stream` afterwards (optimiser, all-reduce, the next step) sees every gradient complete.  Not capturable into a hipGraph with the side
 * stream on (the developers' build of the library, libcerberus_hip_dev.so, reads CERB_WGRAD_SIDE=0 to keep the call on `hip_stream` alone: same bits). */
int cerb_net_train_grads(cerb_net* net, const cerb_train_step_io* io, void* hip_stream);
int cerb_net_grad_lookup(cerb_net* net, const char* key, float** dev_ptr, long long* numel);
/* The same lookup also serves the batch statistics of every BatchNorm of the step under "<bn prefix>.batch_mean" and
 * "<bn prefix>.batch_var" (unbiased), from which the caller updates running_mean / running_var (momentum 0.1).
 * cerb_adam_step: torch.optim.Adam (no weight decay / amsgrad; models/opt.py:47-58) on one parameter tensor, in place; `step` counts from 1. */
int cerb_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
                   float beta2, float eps, int step, void* hip_stream);
/* The same update over `count` tensors in ONE launch (host arrays of device pointers and element counts): the 307 parameter tensors of
 * the six-head network per step. */
int cerb_adam_step_multi(int count, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                         const long long* numel, float lr, float beta1, float beta2, float eps, int step, void* hip_stream);
/* device-to-device copy on a stream (lets a host language without a HIP binding move a looked-up gradient into its own buffer) */
int cerb_copy_d2d(void* dst, const void* src, size_t bytes, void* hip_stream);

/* ---- training step: per-head losses (BASELINE.json configs[4]; the whole step is cerberus_amd/train.py::train_step) --------------
 * cerb_head_loss: the per-head loss of the reference's train_step (models/run_desc.py:88-170) and its gradient on the logits.
 *   logits / dlogits : device float, element strides (n, c, y, x) -- NCHW as the reference's forward returns them, or NHWC
 *   target           : device float [N][H][W] class ids (the reference keeps targets in float32, :57-62)
 *   has_target       : device float [N], 1 where the sample carries this head's annotation (dummy_target protocol, :96-97)
 *   class_weight     : device float [C] or NULL.  Given (TYPE heads): pixel weight = class_weight[target], 0 on background
 *                      (get_class_wmap, :18-22,118-124) and the Dice term is masked by target > 0
 *   ce_weight, dice_weight, head_weight : paramset.yml loss_info ("ce" / "dice" weights inside the head, the head's own weight)
 *   patch_class_mode : 1 for the [N][C][1][1] Patch-Class head, which inherits the reference's [N] x [N,1,1] broadcast (:126-154)
 *   loss_out         : device float[1], the value train_step reports for the head; dlogits may be NULL (value only)
 *   ws               : device workspace of cerb_head_loss_workspace_bytes(N, H, W)
 * Sums are taken in a fixed order (per-block partials, one-block double-precision finalise): bitwise reproducible. */
size_t cerb_head_loss_workspace_bytes(int n, int h, int w);
int cerb_head_loss(const float* logits, long long stride_n, long long stride_c, long long stride_y, long long stride_x,
                   const float* target, const float* has_target, int n, int h, int w, int c, const float* class_weight,
                   float ce_weight, float dice_weight, float head_weight, int patch_class_mode, float* loss_out,
                   float* dlogits, void* ws, size_t ws_bytes, void* hip_stream);

/* The same with the head's per-pixel weight map (loader/targets.py "#WEIGHT-MAP" channel; models/run_desc.py:111-117,150): the pixel's
 * cross-entropy is multiplied by pixel_weight[n][y][x] (device float [N][H][W] or NULL = ones).  As in the reference, class weights
 * (TYPE heads) REPLACE the map, and the Patch-Class head has none. */
int cerb_head_loss_wmap(const float* logits, long long stride_n, long long stride_c, long long stride_y, long long stride_x,
                        const float* target, const float* has_target, int n, int h, int w, int c, const float* class_weight,
                        const float* pixel_weight, float ce_weight, float dice_weight, float head_weight, int patch_class_mode,
                        float* loss_out, float* dlogits, void* ws, size_t ws_bytes, void* hip_stream);

int cerb_event_create(void** ev);
int cerb_event_record(void* ev, void* hip_stream);
int cerb_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */
int cerb_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* CERBERUS_HIP_H */
