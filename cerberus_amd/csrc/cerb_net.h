// Internal header of the C-ABI implementation: what cerb_api.hip (weight intake, packing, the inference forward) and cerb_train.hip (train-mode
// forward, the backward tape, optimiser entry points) share -- the launcher prototypes of the kernel translation units, the handle (struct
// cerb_net) and the host helpers both schedules call.  Not part of the boundary: include/cerberus_hip.h is.
#ifndef CERB_NET_H
#define CERB_NET_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/cerberus_hip.h"
#include "cerb_common.h"

// launchers implemented in the kernel translation units
hipError_t cerb_launch_conv(const ConvParams& p, int ks, int stride, int mode, hipStream_t st);
hipError_t cerb_launch_wino(ConvParams p, hipStream_t st);
hipError_t cerb_launch_wino4(ConvParams p, hipStream_t st);
hipError_t cerb_launch_wino4b(ConvParams p, hipStream_t st);
int cerb_wino4b_bn_blocks(const ConvParams& p);
bool cerb_wino4b_packed(const ConvParams& p);     // this launch takes packed items (16 consecutive tiles instead of a 16 x 16 block)  // BatchNorm partial rows per group the kernel leaves (packed items on 28^2 / 56^2 maps: fewer)
hipError_t cerb_launch_wino4p(ConvParams p, hipStream_t st);
hipError_t cerb_launch_upsample2_add_planar(const float* skip, const float* prev, float* out, int groups, int N, int H, int W, int C, long long prev_gs,
                                            long long out_gs, const int* roi, int prev_planar, hipStream_t st);
hipError_t cerb_launch_upsample2_add(const float* skip, const float* prev, float* out, int groups, int N, int H, int W, int C,
                                     long long prev_gs, const int* roi, hipStream_t st);
extern "C" int cerb_conv_chunk(int ks, int stride);
struct StemParams {
    const unsigned char* tiles;
    const float* tiles_f32;
    const float* wpack;
    const float* bias;
    float* out;
    int N, H, W, tiles_x, tiles_y;
    int relu;
};
hipError_t cerb_launch_stem(StemParams p, hipStream_t st);
// train-mode pieces (train_kernels.hip)
size_t cerb_bn_workspace_bytes(int groups, long long rows, int C);
hipError_t cerb_launch_bn_stats(const float* x, long long group_stride, long long rows, int C, int groups, float eps, float* mean, float* rstd,
                                float* var_unbiased, void* ws, hipStream_t st);
hipError_t cerb_launch_bn_apply(float* x, const float* src, const float* resid, long long group_stride, long long rows, int C, int groups, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, int relu, hipStream_t st);
hipError_t cerb_launch_bn_finalize(const double* partial, int blocks, long long rows, int C, float eps, float* mean, float* rstd, float* var_unbiased, hipStream_t st,
                                   int groups = 1, void* fold_ws = nullptr);
size_t cerb_bn_fold_workspace_bytes(int groups, int C);
hipError_t cerb_launch_pointwise(const float* in, const float* w, const float* bias, float* out, long long rows, int cin, int cout, const float* in_scale,
                                 hipStream_t st, double* bn_part = nullptr, int* bn_blocks = nullptr);
hipError_t cerb_launch_crop_gap(const float* x, int N, int H, int W, int C, int y0, int ch, int x0, int cw, float* out, hipStream_t st);
hipError_t cerb_launch_copy_multi(int count, float* const* dst, const float* const* src, const long long* n, void** dev_tab, size_t* dev_bytes,
                                  std::vector<char>* host_prev, hipStream_t st);
hipError_t cerb_launch_bn_bwd(const float* dz, const float* z, const float* y, float* dy, float* dresid, long long group_stride, long long rows, int C, int groups,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, int relu, int dy_assign, void* ws, hipStream_t st, unsigned long long eval_mask = 0, int dresid_assign = 0,
                              const double* pre_part = nullptr, int pre_bpg = 0);
int cerb_head_bwd2_blocks();
hipError_t cerb_launch_conv_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int G, int N, int H, int W, int Cin, int Cout,
                                int ks, int stride, long long x_gs, hipStream_t st);
hipError_t cerb_launch_stem_wgrad(const unsigned char* tiles, const float* dy, float* dw, int N, int H, int W, hipStream_t st);
hipError_t cerb_launch_maxpool_bwd(const float* x, const float* ypool, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t st);
hipError_t cerb_launch_maxpool_idx(const float* in, float* out, unsigned* idx, int N, int H, int W, int C, hipStream_t st);       // training forward: pooled map + window positions
hipError_t cerb_launch_maxpool_bwd_idx(const unsigned* idx, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t st);  // backward by the recorded positions
bool cerb_upadd_bwd_fused_ok(int H, int W, int C, int G);
hipError_t cerb_launch_upadd_bwd(const float* dout, float* dskip, float* dprev, int G, int N, int H, int W, int C, long long prev_gs, int shared_prev, hipStream_t st,
                                 unsigned group_mask = 0xffffffffu, int skip_assign = 0, int prev_assign = 0);
hipError_t cerb_launch_pointwise_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long rows, int cin, int cout,
                                     const float* in_scale, int dx_assign, hipStream_t st);
hipError_t cerb_launch_crop_gap_bwd(const float* dg, float* dx, int N, int H, int W, int C, int y0, int ch, int x0, int cw, hipStream_t st);
hipError_t cerb_launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step, hipStream_t st);
size_t cerb_wgrad_workspace_bytes(int G, int N, int Ho, int Wo, int Cin, int Cout, int ks, int* slices_out);
size_t cerb_stem_wgrad_workspace_bytes();
hipError_t cerb_launch_pack_stem(const float* w_raw, float* out, hipStream_t st);
struct PackJob {  // pack_kernels.hip
    const float* w;
    float* out;
    long long total;
    int cout, cin, kind, a, b, pad;
};
hipError_t cerb_launch_pack_multi(const PackJob* jobs, int count, void** dev_tab, size_t* dev_bytes, std::vector<char>* host_prev, hipStream_t st);
hipError_t cerb_launch_adam_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* n, float lr, float b1,
                                  float b2, float eps, int step, hipStream_t st);
hipError_t cerb_launch_pack_conv(const float* w_raw, float* out, int cout, int cin, int ks, int chunk, hipStream_t st);
hipError_t cerb_launch_pack_wino(const float* w_raw, float* out, int cout, int cin, int dgrad, hipStream_t st);
hipError_t cerb_launch_pack_wino4(const float* w_raw, float* out, int cout, int cin, int dgrad, int chunk32, int groups, hipStream_t st);
hipError_t cerb_launch_dilate2(const float* dy, float* d, long long n, int H, int W, int C, hipStream_t st);
size_t cerb_pw_wgrad_small_workspace_bytes(long long rows, int cin, int cout);
hipError_t cerb_launch_pw_wgrad_small(const float* x, const float* dy, float* dw, long long rows, int cin, int cout, void* ws, hipStream_t st);
size_t cerb_pw_bwd_small_workspace_bytes(long long rows, int cin, int cout);
hipError_t cerb_launch_pw_bwd_small(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long long rows, int cin, int cout,
                                    int dx_assign, void* ws, hipStream_t st);
hipError_t cerb_launch_stem_wgrad_mfma(const unsigned char* tiles, const float* dy, float* dw, int N, int H, int W, void* ws, hipStream_t st);
hipError_t cerb_launch_colsum(const float* d, long long group_stride, long long rows, int C, int G, float* out, void* ws, hipStream_t st);
bool cerb_wgrad_wino_supported(int H, int W, int Cin, int Cout);
size_t cerb_wgrad_wino_workspace_bytes(int G, int N, int H, int W, int Cin, int Cout);
hipError_t cerb_launch_wgrad_wino(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, long long x_gs, void* ws, hipStream_t st,
                                  float* db);
hipError_t cerb_launch_wgrad(const float* x, const float* dy, float* dw, int G, int N, int H, int W, int Cin, int Cout, int ks, int stride, long long x_gs, void* ws,
                             hipStream_t st, float* db = nullptr);
bool cerb_head_train_supported(long long rows, int cin, int chid, int out);
hipError_t cerb_launch_head_fwd1(const float* prev, const float* w1, const float* b1, float* hid, long long rows, double* bn_part, int* bn_blocks, hipStream_t st,
                                 const float* const* in_bn = nullptr);
hipError_t cerb_launch_head_fwd2(const float* hid, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2, const float* b2,
                                 float* logits, long long rows, int out, hipStream_t st);
size_t cerb_head_bwd_workspace_bytes(long long rows, int out);
hipError_t cerb_launch_head_bwd1(const float* hid, const float* dlog, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2,
                                 float* dw2, float* db2, float* dgamma, float* dbeta, long long rows, int out, void* ws, hipStream_t st);
hipError_t cerb_launch_head_bwd2(const float* hid, const float* dlog, const float* prev, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 const float* dgamma, const float* dbeta, const float* w1, const float* w2, float* dprev, float* dw1, float* db1, long long rows, int out,
                                 int eval_mode, int assign, void* ws, hipStream_t st, const float* const* in_bn = nullptr, double* in_part = nullptr);
hipError_t cerb_launch_maxpool(const float* in, float* out, int N, int H, int W, int C, hipStream_t st);
hipError_t cerb_launch_head(const HeadParams& p, hipStream_t st);
hipError_t cerb_launch_head_group(const HeadParams* heads, int n_heads, hipStream_t st, int w2_44);
struct PatchClassParams {
    const float* x4;
    const float* bn1_s;
    const float* bn1_b;
    const float* w1t;
    const float* b1;
    const float* w2t;
    const float* b2;
    int N, Hf, Wf, out_ch;
    int out_h, out_w;
    float* logits;
    float* out;
    const long long* tile_off;
    long long tile_stride, row_stride;
};
hipError_t cerb_launch_patch_class(const PatchClassParams& p, hipStream_t st);

int cerb_set_error(const std::string& m);  // thread-local message of cerb_last_error(); returns 1
static inline int fail(const std::string& m) { return cerb_set_error(m); }
// Return code 2 (include/cerberus_hip.h: CERB_ERR_ALLOC): a workspace / tape allocation did not fit -- the one failure a caller can act on (run on a
// smaller batch, drop its second handle: cerberus_amd/wsi.py) without parsing the message.
static inline int fail_alloc() { return cerb_set_error("workspace allocation failed") + 1; }
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

extern "C" size_t cerb_conv_guard_bytes(int tile_w);
// Activation buffer with a zero-filled guard band in front of and behind the payload: conv_igemm reads halo tiles with
// unclamped addresses (row wrap / out-of-image elements are masked later), so every byte it can touch must exist and hold
// a finite value.  The whole allocation is zeroed once; kernels only ever write payload bytes.
// The stream of the API call that is running on this thread (set at every entry point that may allocate): a fresh buffer is zero-filled ON it.
extern thread_local hipStream_t g_call_stream;
// bytes of activation workspace all handles of this process hold (cerb_device_bytes_held: a caller that prices its next job against the free HBM
// must not count them twice -- they are allocated already AND part of what a forward needs)
extern std::atomic<long long> g_devbuf_bytes;
struct DevBuf {
    float* p = nullptr;  // payload
    char* raw = nullptr;
    size_t bytes = 0, guard = 0, held = 0;
    int ensure(size_t need, size_t g) {
        if (need <= bytes && g <= guard) return 0;
        release();
        if (hipMalloc(&raw, need + 2 * g) != hipSuccess) {
            raw = nullptr;
            return 1;
        }
        g_devbuf_bytes += (long long)(need + 2 * g);
        held = need + 2 * g;
        // The fill is queued on the CALLER's stream: round 4 used hipMemset + hipDeviceSynchronize here because the NULL-stream fill
        // raced the first kernels of a non-blocking side stream (two handles on two streams, cerberus_amd/wsi.py) -- on the stream that will use the
        // buffer it is ordered by construction, stalls nothing else on the device and does not break a stream capture.  (The old buffer's hipFree
        // in release() waits for the work that may still read it.)
        if (hipMemsetAsync(raw, 0, need + 2 * g, g_call_stream) != hipSuccess) return 1;
        p = reinterpret_cast<float*>(raw + g);
        bytes = need;
        guard = g;
        return 0;
    }
    void release() {
        if (raw) {
            (void)hipFree(raw);
            g_devbuf_bytes -= (long long)held;
        }
        held = 0;
        raw = nullptr;
        p = nullptr;
        bytes = guard = 0;
    }
};

// A tile-planar tensor (cerb_common.h: cerb_planar_offset) of `groups` x up to cap_n images.  Its guard ring and the pixels of edge blocks
// beyond the image must read as zero and no kernel ever writes them, so the buffer is zeroed when it is made and again whenever the
// map geometry (and with it the position of those bytes) changes; a smaller batch keeps the image slots where they are.
struct PlanarBuf {
    DevBuf b;
    int h = 0, w = 0, c = 0, groups = 0;
    long long cap_n = 0;
    long long per_image() const { return cerb_planar_elems(1, h, w, c); }
    long long gs() const { return cap_n * per_image(); }  // elements between groups
    int ensure(int G, int N, int H, int W, int C, hipStream_t st) {
        if (H == h && W == w && C == c && G == groups && N <= cap_n) return 0;
        const size_t need = (size_t)G * (size_t)N * (size_t)cerb_planar_elems(1, H, W, C) * 4;
        if (need > b.bytes) {
            if (b.ensure(need, 0)) return 1;  // zeroed by DevBuf
        } else if (hipMemsetAsync(b.raw, 0, b.bytes + 2 * b.guard, st) != hipSuccess) {
            return 1;
        }
        h = H; w = W; c = C; groups = G;
        cap_n = (long long)(b.bytes / ((size_t)G * (size_t)cerb_planar_elems(1, H, W, C) * 4));
        return 0;
    }
    void release() { b.release(); h = w = c = groups = 0; cap_n = 0; }
};

struct PackedConv {
    int cin = 0, cout = 0, ks = 0, stride = 1, groups = 1;
    float* w = nullptr;     // device
    float* wino = nullptr;  // device, 3x3 stride-1 only: Winograd F(2x2,3x3) transformed weights (conv_wino.hip)
    float* wino_dgrad = nullptr;  // train packing only: the same for the DATA GRADIENT -- the conv with rotated, transposed weights
    // train packing: the F(2x2) copies are re-packed after an optimiser step only if a kernel has read them since the handle was made (a
    // network whose maps all take the F(4x4) kernels never does); a copy that was skipped is stale and is re-packed on first use
    bool wino_used = false, wino_dgrad_used = false, wino_stale = false, wino_dgrad_stale = false;
    float* wino4_t[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // train packing only: [layout 4 / 4b][forward / data gradient], packed on the
                                                                      // device at first use and again after every optimiser step
    float* wino4b = nullptr;  // device, the same transform in conv_wino4b.hip's layout (32-channel chunks, conv_algo 7), packed lazily
    float* wino4 = nullptr;   // device, F(4x4,3x3) transformed weights in conv_wino4.hip's layout (conv_algo 5), packed lazily from host_w
    std::vector<float> host_w;  // BN-folded 3x3 weights [G][cout][cin][9] kept on the host for the lazily packed Winograd variants
    float* b = nullptr;     // device
};

struct DecoderCfg {
    std::string name, head;
    int out_ch = 0;
    int kind = 0;  // 0 INST, 1 TYPE, 2 OUT (Patch-Class)
};

struct cerb_net {
    std::vector<DecoderCfg> dec;
    std::vector<int> dense_idx;  // indices into dec of the dense (non Patch-Class) entries: one per OUTPUT HEAD
    // models/net_desc.py:81-87, 196-198: a decoder may carry several output heads (a ModuleDict of heads over ONE decoder trunk).  Entries of
    // cerb_net_create that repeat a decoder name are further heads of that decoder: the trunk (its eight 3x3 convolutions) exists and runs once.
    std::vector<int> trunk_idx;  // per decoder trunk: index into dec of its first entry
    std::vector<int> trunk_of;   // per position in dense_idx: the trunk (group of the grouped decoder launches) whose features that head reads
    int pc_idx = -1;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    // packed device weights
    float *stem_w = nullptr, *stem_b = nullptr;
    std::map<std::string, PackedConv> conv;  // backbone convs + conv_map + grouped decoder convs ("dec.<u>.<j>")
    std::vector<float*> head_w1, head_b1, head_w2, head_b2, head_w2q;  // per dense decoder
    float *pc_bn1s = nullptr, *pc_bn1b = nullptr, *pc_w1t = nullptr, *pc_b1 = nullptr, *pc_w2t = nullptr, *pc_b2 = nullptr;
    std::vector<void*> dev_allocs;
    std::vector<size_t> dev_alloc_bytes;  // sizes of dev_allocs: a reload (cerb_net_begin_reload) hands the same buffers out again, in order
    size_t n_finalize_allocs = 0, reuse_cursor = 0;
    bool reusing = false;
    // train-mode packing (cerb_net_set_fold_bn(net, 0) before finalize): raw conv weights, BatchNorm affine parameters kept apart
    int fold_bn = 1;
    struct BnDev {
        float *gamma = nullptr, *beta = nullptr;
        int C = 0, groups = 1;
        // cerb_net_set_bn_eval: groups of a train-packed network whose BatchNorm runs in EVAL mode (the reference's frozen sub-typing modules,
        // models/net_desc.py:105-121): device copies of running_mean and 1 / sqrt(running_var + eps), [groups][C]; eval[g] != 0 where set
        float *run_mean = nullptr, *run_rstd = nullptr;
        std::vector<char> eval;
    };
    std::map<std::string, BnDev> bn;  // by conv name ("stem", "backbone.layer1.0.conv1", "dec.<u>.<j>", "head.<k>", "pc.bn1", "pc.bn2")
    std::vector<float*> head_rw1, head_rb1, head_rw2, head_rb2;  // raw head weights, row-major [cout][cin]
    float *pc_rw1 = nullptr, *pc_rb1 = nullptr, *pc_rw2 = nullptr, *pc_rb2 = nullptr;
    DevBuf t_mean, t_rstd, t_ws, t_hid, t_gap, t_pc1, t_idn, t_dil;
    // backward pass (cerb_net_train_grads): raw weights in state-dict layout, per conv name, groups concatenated; the tape's buffers
    struct RawW { float* w = nullptr; float* b = nullptr; std::vector<std::string> wkeys, bkeys, bnkeys; };
    std::map<std::string, RawW> raw;
    // handles packed for training: where each state-dict tensor lives verbatim on the device (cerb_net_update_params copies into these)
    struct ParamSlot { float* dst; long long n; };
    std::map<std::string, std::vector<ParamSlot>> param_slots;
    float* stem_raw = nullptr;  // [64][3][7][7]
    std::vector<DevBuf> tape;
    size_t tape_pos = 0;
    void* copy_tab = nullptr;    // cerb_net_update_params: device table of the parameter copies (cerb_launch_copy_multi)
    size_t copy_tab_bytes = 0;
    std::vector<char> copy_tab_host;
    void* pack_tab = nullptr;    // ... and of the re-pack jobs (cerb_launch_pack_multi)
    size_t pack_tab_bytes = 0;
    std::vector<char> pack_tab_host;
    float* zero_bias = nullptr;  // 512 zeros: the bias operand of the data-gradient convs
    std::map<std::string, std::pair<float*, long long>> grads;  // state-dict key -> (device gradient, numel) of the last cerb_net_train_grads
    std::map<std::string, std::vector<std::string>> bn_keys;   // conv / bn name -> state-dict prefixes of its BatchNorm, one per group
    // workspace
    DevBuf x0, pool, x[5], ta, tb, cm, dmid, dsum, dout[4];
    PlanarBuf psum, pmid, pout;  // the last decoder level's private tensors in the tile-planar layout (conv_wino4p.hip), cerb_net_set_planar; psum also
                                 // receives the level's OUTPUT (it is dead once the first conv has read it): pout / pout2 are never allocated any more
    bool planar_half = false;       // set by the decoder loop around the half-resolution level's run_conv calls (names the kernel symbol)
    PlanarBuf psum2, pmid2, pout2;  // the same for the level below it (64 channels at half the resolution) when its maps are large enough
    int packed_items = 1;        // cerb_net_set_packed_items: conv_wino4b.hip packs 16 consecutive tiles per item on maps that are not whole 16 x 16 blocks (28^2, 56^2)
    int planar = 1;              // cerb_net_set_planar: 1 (default) = that level runs upsample2_add_planar -> conv_wino4p x2 -> heads reading planar features
    // optional per-launch timing (HIP events on the caller's stream)
    bool profiling = false;
    int crop_roi = 1;   // cerb_net_set_crop_roi: decoders / heads only compute what the centre crop keeps (conv_algo 1)
    int head_algo = 1;  // cerb_net_set_head_algo: 1 = all dense heads in one grouped launch, logits on 4x4x1 matrix instructions (default); 2 = round 3's grouped
                        // launch (logits on a zero-padded 16-row instruction); 0 = one launch per head
    int conv_algo = 6;  // cerb_net_set_conv_algo: 6 = Winograd F(4x4,3x3) / F(2x2,3x3) by launch size (default), 1 = F(2x2,3x3), 0 = direct implicit GEMM
    struct ProfRec { std::string name, kernel; double flops; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof;
    size_t prof_n = 0;
    bool prof_open = false;  // a record is open (prof_begin without its prof_end yet)
    // training forward: where the NEXT run_conv may leave BatchNorm statistics partials (ConvParams::bn_part); run_conv clears the request and
    // reports in conv_bn_bpg how many blocks per group it wrote (0: this convolution's kernel does not produce them)
    double* conv_bn_part = nullptr;
    int conv_bn_bpg = 0;
    // training backward: the weight gradients of the 3x3 / 1x1 convolutions run on a side stream of the handle's own (forked from the caller's stream when the
    // layer's output gradient is final, joined at the end of cerb_net_train_grads): matrix-core work that overlaps the BatchNorm backward passes (HBM-bound, no
    // LDS) and fills the last-round tails of the data-gradient launches.  Same kernels, same arithmetic.  CERB_WGRAD_SIDE=0 / profiling: everything on one stream.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf t_ws2;  // the side stream's split-K workspace
    ~cerb_net() {
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        t_ws2.release();
        for (auto& r : prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        for (void* p : dev_allocs) (void)hipFree(p);
        if (copy_tab) (void)hipFree(copy_tab);
        if (pack_tab) (void)hipFree(pack_tab);
        x0.release(); pool.release(); ta.release(); tb.release(); cm.release(); dmid.release(); dsum.release(); psum.release(); pmid.release(); pout.release(); psum2.release(); pmid2.release(); pout2.release();
        t_mean.release(); t_rstd.release(); t_ws.release(); t_hid.release(); t_gap.release(); t_pc1.release(); t_idn.release(); t_dil.release();
        for (auto& b : tape) b.release();
        for (auto& b : x) b.release();
        for (auto& b : dout) b.release();
    }
};

// ---- host helpers shared by the two schedules (defined in cerb_api.hip) ---------------------------------------------------------------

static const int kLayers[4] = {3, 4, 6, 3};            // ResNet34 (models/backbone/resnet.py:273-286)
static const int kFilters[5] = {64, 64, 128, 256, 512};
int prof_begin(cerb_net* net, const std::string& name, const std::string& kernel, double flops, hipStream_t st);  // per-launch records (cerb_net_profile_*)
int prof_end(cerb_net* net, hipStream_t st);
int train_wino2_fresh(cerb_net* net, const std::string& name, PackedConv& cm, int dgrad, hipStream_t st);
int train_wino4_slot(cerb_net* net, const std::string& name, PackedConv& cm, int w4b, int dgrad, hipStream_t st, float** out);
// one convolution of the schedule by its packed name (algorithm by cerb_net_set_conv_algo and the layer's geometry); planar_out_gs > 0: in / out are tile-planar
int run_conv(cerb_net* net, const std::string& name, const float* in, const float* prev, const float* resid, float* out, int N, int H, int W, int relu, int mode,
             long long in_gs, long long prev_gs, hipStream_t st, double* macs, const int* roi = nullptr, long long planar_out_gs = 0);
#endif  // CERB_NET_H
