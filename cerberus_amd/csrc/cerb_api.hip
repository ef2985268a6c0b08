// C-ABI implementation (include/cerberus_hip.h) for the network part of the Cerberus tile path:
// weight intake (BN folding + MFMA packing), workspace, and the forward schedule
//   stem -> maxpool -> 16 BasicBlocks -> conv_map -> Patch-Class branch -> 5 dense decoders (grouped launches) -> heads.
// Host code only; kernels live in conv_igemm.hip / net_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return 1;
}
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

int cerb_set_error(const std::string& m) { return fail(m); }  // shared with postproc.hip
extern "C" int cerb_version(void) { return 1; }
extern "C" const char* cerb_last_error(void) { return g_err.c_str(); }

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
static thread_local hipStream_t g_call_stream = nullptr;
struct DevBuf {
    float* p = nullptr;  // payload
    char* raw = nullptr;
    size_t bytes = 0, guard = 0;
    int ensure(size_t need, size_t g) {
        if (need <= bytes && g <= guard) return 0;
        release();
        if (hipMalloc(&raw, need + 2 * g) != hipSuccess) return 1;
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
        if (raw) (void)hipFree(raw);
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
    std::vector<int> dense_idx;  // indices into dec of the dense (non Patch-Class) decoders
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

static int alloc_dev(cerb_net* net, size_t bytes, void** out) {
    if (net->reusing) {
        if (net->reuse_cursor >= net->n_finalize_allocs || net->dev_alloc_bytes[net->reuse_cursor] != bytes)
            return fail("reload: the tensors do not have the shapes the handle was finalized with");
        *out = net->dev_allocs[net->reuse_cursor++];
        return 0;
    }
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, bytes));
    net->dev_allocs.push_back(d);
    net->dev_alloc_bytes.push_back(bytes);
    *out = d;
    return 0;
}
static int upload(cerb_net* net, const std::vector<float>& v, float** out) {
    void* d = nullptr;
    if (alloc_dev(net, v.size() * sizeof(float), &d)) return 1;
    HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = reinterpret_cast<float*>(d);
    return 0;
}

extern "C" int cerb_net_create(const char* const* decoder_names, const char* const* head_names, const int* out_ch,
                               int n_decoders, cerb_net** out_net) {
    if (!decoder_names || !head_names || !out_ch || !out_net || n_decoders <= 0) return fail("cerb_net_create: bad arguments");
    cerb_net* net = new cerb_net();
    for (int i = 0; i < n_decoders; ++i) {
        DecoderCfg d;
        d.name = decoder_names[i];
        d.head = head_names[i];
        d.out_ch = out_ch[i];
        if (d.name == "Patch-Class") {
            d.kind = 2;
            if (d.out_ch < 1 || d.out_ch > 16) { delete net; return fail("Patch-Class: out_ch must be 1..16"); }
            net->pc_idx = i;
        } else {
            if (d.head == "INST") d.kind = 0;
            else if (d.head == "TYPE") d.kind = 1;
            else { delete net; return fail("decoder " + d.name + ": head must be INST or TYPE, got " + d.head); }
            if (d.kind == 0 && d.out_ch != 3) { delete net; return fail("INST heads must have 3 channels (infer_step keeps channels 1..2)"); }
            if (d.out_ch < 2 || d.out_ch > 8) { delete net; return fail("head out_ch must be 2..8"); }
            net->dense_idx.push_back(i);
        }
        net->dec.push_back(d);
    }
    *out_net = net;
    return 0;
}

extern "C" void cerb_net_destroy(cerb_net* net) { delete net; }

extern "C" int cerb_net_load_tensor(cerb_net* net, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!net || !key) return fail("cerb_net_load_tensor: bad arguments");
    if (net->finalized) return fail("cerb_net_load_tensor: network already finalized");
    std::string k(key);
    if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return 0;  // integer bookkeeping, unused in eval
    if (k.rfind("backbone.fc.", 0) == 0) return 0;  // resnet.py:212-213: fc exists but is never called
    if (!data && ndim > 0) return fail("cerb_net_load_tensor: null data for " + k);
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    net->host[k] = std::move(t);
    return 0;
}

// ---- BN folding helpers ----------------------------------------------------------------------------------------
struct Fold {
    std::vector<float> scale, shift;
};
static int get(cerb_net* net, const std::string& k, std::vector<int64_t> shape, const HostTensor** out) {
    auto it = net->host.find(k);
    if (it == net->host.end()) return fail("missing key in state dict: " + k);
    if (it->second.shape != shape) {
        std::string s = "shape mismatch for " + k + ": got [";
        for (auto v : it->second.shape) s += std::to_string(v) + ",";
        s += "] expected [";
        for (auto v : shape) s += std::to_string(v) + ",";
        return fail(s + "]");
    }
    *out = &it->second;
    return 0;
}
static int bn_fold(cerb_net* net, const std::string& p, int ch, Fold* f) {
    const HostTensor *w, *b, *m, *v;
    if (get(net, p + ".weight", {ch}, &w) || get(net, p + ".bias", {ch}, &b) || get(net, p + ".running_mean", {ch}, &m) ||
        get(net, p + ".running_var", {ch}, &v))
        return 1;
    f->scale.resize(ch);
    f->shift.resize(ch);
    for (int c = 0; c < ch; ++c) {
        const float s = w->data[c] / std::sqrt(v->data[c] + 1e-5f);
        f->scale[c] = s;
        f->shift[c] = b->data[c] - m->data[c] * s;
    }
    return 0;
}

// Pack one [Cout][Cin][ks][ks] conv (scaled per cout) into the layout conv_igemm.hip streams:
//   [cb][chunk][tap][G][s][lane][t]  ->  W[cb*64 + s*32 + (lane&31)][chunk*CB + G*8 + 4*(lane>>5) + t][tap]
static void pack_conv(const float* w, const float* scale, int cout, int cin, int ks, int CB, std::vector<float>* out) {
    const int T = ks * ks, NG = CB / 8, nchunk = cin / CB, ncb = cout / 64;
    const size_t base = out->size();
    out->resize(base + (size_t)cout * cin * T);
    float* o = out->data() + base;
    size_t idx = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int tap = 0; tap < T; ++tap)
                for (int G = 0; G < NG; ++G)
                    for (int s = 0; s < 2; ++s)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int t = 0; t < 4; ++t) {
                                const int co = cb * 64 + s * 32 + (lane & 31);
                                const int ci = ch * CB + G * 8 + 4 * (lane >> 5) + t;
                                o[idx++] = w[((size_t)co * cin + ci) * T + tap] * (scale ? scale[co] : 1.f);
                            }
}

// Winograd F(2x2,3x3) filter transform U = G g G^T (in double, rounded once) in the layout conv_wino.hip streams:
//   [cb][chunk][a][b][G][s][lane][t]  ->  U[a][b] of W[cb*64 + s*32 + (lane&31)][chunk*32 + G*8 + 4*(lane>>5) + t]
static void pack_wino(const float* w, const float* scale, int cout, int cin, std::vector<float>* out) {
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nchunk = cin / 32, ncb = cout / 64;
    const size_t base = out->size();
    out->resize(base + (size_t)cout * cin * 16);
    float* o = out->data() + base;
    std::vector<float> U((size_t)cout * cin * 16);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], t[4][3];
            for (int y = 0; y < 3; ++y)
                for (int x = 0; x < 3; ++x) g[y][x] = (double)(w[((size_t)co * cin + ci) * 9 + y * 3 + x] * (scale ? scale[co] : 1.f));
            for (int a = 0; a < 4; ++a)
                for (int x = 0; x < 3; ++x) t[a][x] = Gm[a][0] * g[0][x] + Gm[a][1] * g[1][x] + Gm[a][2] * g[2][x];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b)
                    U[((size_t)co * cin + ci) * 16 + a * 4 + b] = (float)(t[a][0] * Gm[b][0] + t[a][1] * Gm[b][1] + t[a][2] * Gm[b][2]);
        }
    size_t idx = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b)
                    for (int G = 0; G < 4; ++G)
                        for (int s = 0; s < 2; ++s)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int t = 0; t < 4; ++t) {
                                    const int co = cb * 64 + s * 32 + (lane & 31);
                                    const int ci = ch * 32 + G * 8 + 4 * (lane >> 5) + t;
                                    o[idx++] = U[((size_t)co * cin + ci) * 16 + a * 4 + b];
                                }
}

// Winograd F(4x4,3x3) filter transform U = G g G^T for the points (0, 1, -1, 2, -2, inf), in double, rounded once, in the layout
// conv_wino4.hip streams: [cb][16-channel chunk][wave a][position xi = 6 ya + xb][lane][t]
//   ->  U[xi] of W[cb*64 + 16 a + (lane & 15)][chunk*16 + 4 (lane >> 4) + t]
//   conv_wino4b.hip (chunk32 = true): [cb][32-channel chunk][wave a][xi][channel group G][lane][t]
//   ->  U[xi] of W[cb*64 + 16 a + (lane & 15)][chunk*32 + 16 G + 4 (lane >> 4) + t]
static void pack_wino4(const float* w, int cout, int cin, std::vector<float>* out, int layout = 0) {  // w: BN-folded [cout][cin][3][3]; layout 0 = conv_wino4 / 4p, 1 = conv_wino4b
    const bool chunk32 = layout == 1;
    static const double Gm[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int nchunk = cin / 16, ncb = cout / 64;
    std::vector<float> U((size_t)cout * cin * 36);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int x = 0; x < 3; ++x) t[a][x] = Gm[a][0] * (double)g[x] + Gm[a][1] * (double)g[3 + x] + Gm[a][2] * (double)g[6 + x];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b)
                    U[((size_t)co * cin + ci) * 36 + a * 6 + b] = (float)(t[a][0] * Gm[b][0] + t[a][1] * Gm[b][1] + t[a][2] * Gm[b][2]);
        }
    const size_t base = out->size();
    out->resize(base + (size_t)cout * cin * 36);
    float* o = out->data() + base;
    size_t idx = 0;
    if (chunk32) {
        for (int cb = 0; cb < ncb; ++cb)
            for (int ch = 0; ch < cin / 32; ++ch)
                for (int a = 0; a < 4; ++a)
                    for (int xi = 0; xi < 36; ++xi)
                        for (int G = 0; G < 2; ++G)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int t = 0; t < 4; ++t) {
                                    const int co = cb * 64 + 16 * a + (lane & 15);
                                    const int ci = ch * 32 + 16 * G + 4 * (lane >> 4) + t;
                                    o[idx++] = U[((size_t)co * cin + ci) * 36 + xi];
                                }
        return;
    }
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int a = 0; a < 4; ++a)
                for (int xi = 0; xi < 36; ++xi)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int t = 0; t < 4; ++t) {
                            const int co = cb * 64 + 16 * a + (lane & 15);
                            const int ci = ch * 16 + 4 * (lane >> 4) + t;
                            o[idx++] = U[((size_t)co * cin + ci) * 36 + xi];
                        }
}

static int store_bn(cerb_net* net, const std::string& name, const std::vector<std::string>& bnkeys, int ch) {
    std::vector<float> ga, be;
    for (const std::string& k : bnkeys) {
        const HostTensor *w, *b;
        if (get(net, k + ".weight", {ch}, &w) || get(net, k + ".bias", {ch}, &b)) return 1;
        ga.insert(ga.end(), w->data.begin(), w->data.end());
        be.insert(be.end(), b->data.begin(), b->data.end());
    }
    cerb_net::BnDev d;
    d.C = ch;
    d.groups = (int)bnkeys.size();
    if (upload(net, ga, &d.gamma) || upload(net, be, &d.beta)) return 1;
    for (size_t g = 0; g < bnkeys.size(); ++g) {
        net->param_slots[bnkeys[g] + ".weight"].push_back({d.gamma + g * ch, ch});
        net->param_slots[bnkeys[g] + ".bias"].push_back({d.beta + g * ch, ch});
    }
    net->bn[name] = d;
    net->bn_keys[name] = bnkeys;
    return 0;
}

static int make_conv(cerb_net* net, const std::string& name, const std::vector<std::string>& wkeys,
                     const std::vector<std::string>& bkeys, const std::vector<std::string>& bnkeys, int cout, int cin, int ks,
                     int stride) {
    // one entry per group
    const int CB = cerb_conv_chunk(ks, stride);
    if (cout % 64 || cin % CB) return fail("conv " + name + ": unsupported channel counts");
    std::vector<float> wp, bp, wwino, hw;
    const bool wino = (ks == 3 && stride == 1 && cin % 32 == 0);
    for (size_t g = 0; g < wkeys.size(); ++g) {
        const HostTensor* w;
        if (get(net, wkeys[g], {cout, cin, ks, ks}, &w)) return 1;
        Fold f;
        bool have_bn = !bnkeys.empty() && net->fold_bn;
        if (have_bn && bn_fold(net, bnkeys[g], cout, &f)) return 1;
        if (net->fold_bn) {  // handles packed for training lay their weights out on the device (below)
            pack_conv(w->data.data(), have_bn ? f.scale.data() : nullptr, cout, cin, ks, CB, &wp);
            if (wino) {
                pack_wino(w->data.data(), have_bn ? f.scale.data() : nullptr, cout, cin, &wwino);
                const size_t b0 = hw.size();
                hw.insert(hw.end(), w->data.begin(), w->data.end());
                if (have_bn)
                    for (int co = 0; co < cout; ++co)
                        for (size_t e = 0; e < (size_t)cin * 9; ++e) hw[b0 + (size_t)co * cin * 9 + e] *= f.scale[co];
            }
        }
        const HostTensor* b = nullptr;
        if (!bkeys.empty() && get(net, bkeys[g], {cout}, &b)) return 1;
        for (int c = 0; c < cout; ++c) {
            float v = b ? b->data[c] : 0.f;
            if (have_bn) v = v * f.scale[c] + f.shift[c];
            bp.push_back(v);
        }
    }
    if (!net->fold_bn && !bnkeys.empty() && store_bn(net, name, bnkeys, cout)) return 1;
    if (!net->fold_bn) {  // raw copies for the backward kernels (state-dict layout [G][cout][cin][ks][ks]) and the key names of the gradients
        std::vector<float> rw, rb;
        for (size_t g = 0; g < wkeys.size(); ++g) {
            const HostTensor* w;
            if (get(net, wkeys[g], {cout, cin, ks, ks}, &w)) return 1;
            rw.insert(rw.end(), w->data.begin(), w->data.end());
            if (!bkeys.empty()) {
                const HostTensor* b;
                if (get(net, bkeys[g], {cout}, &b)) return 1;
                rb.insert(rb.end(), b->data.begin(), b->data.end());
            }
        }
        cerb_net::RawW r;
        r.wkeys = wkeys; r.bkeys = bkeys; r.bnkeys = bnkeys;
        if (upload(net, rw, &r.w) || (!rb.empty() && upload(net, rb, &r.b))) return 1;
        for (size_t g = 0; g < wkeys.size(); ++g) {
            const long long nwg = (long long)cout * cin * ks * ks;
            net->param_slots[wkeys[g]].push_back({r.w + g * nwg, nwg});
            if (r.b) net->param_slots[bkeys[g]].push_back({r.b + g * cout, cout});
        }
        net->raw[name] = r;
    }
    PackedConv pc;
    pc.cin = cin; pc.cout = cout; pc.ks = ks; pc.stride = stride; pc.groups = (int)wkeys.size();
    if (upload(net, bp, &pc.b)) return 1;
    if (!net->fold_bn)
        for (size_t g = 0; g < bkeys.size(); ++g) net->param_slots[bkeys[g]].push_back({pc.b + g * cout, cout});
    if (net->fold_bn) {
        if (upload(net, wp, &pc.w)) return 1;
        if (wino && upload(net, wwino, &pc.wino)) return 1;
        pc.host_w.swap(hw);
    } else {
        // packed on the device from the raw copy (pack_kernels.hip): nothing but the state-dict tensors crosses PCIe after an optimiser step.
        // Data gradient of a 3x3 pad-1 conv: dx = conv(dy, W') with W'[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx] -- the same Winograd conv
        // (stride 2: over dy spread onto the even positions of a zero map)
        const float* rawd = net->raw[name].w;
        const size_t G = wkeys.size(), nw = (size_t)cout * cin * ks * ks, nu = (size_t)cout * cin * 16;
        const bool dgrad = ks == 3 && cin % 64 == 0 && cout % 32 == 0;
        void *dwp = nullptr, *dwi = nullptr, *ddg = nullptr;
        if (alloc_dev(net, G * nw * 4, &dwp) || (wino && alloc_dev(net, G * nu * 4, &dwi)) || (dgrad && alloc_dev(net, G * nu * 4, &ddg))) return 1;
        pc.w = reinterpret_cast<float*>(dwp);
        pc.wino = reinterpret_cast<float*>(dwi);
        pc.wino_dgrad = reinterpret_cast<float*>(ddg);
        for (size_t g = 0; g < G; ++g) {
            HIP_OK(cerb_launch_pack_conv(rawd + g * nw, pc.w + g * nw, cout, cin, ks, CB, 0));
            if (wino) HIP_OK(cerb_launch_pack_wino(rawd + g * nw, pc.wino + g * nu, cout, cin, 0, 0));
            if (dgrad) HIP_OK(cerb_launch_pack_wino(rawd + g * nw, pc.wino_dgrad + g * nu, cin, cout, 1, 0));
        }
    }
    net->conv[name] = pc;
    return 0;
}

static const int kLayers[4] = {3, 4, 6, 3};
static const int kFilters[5] = {64, 64, 128, 256, 512};

extern "C" int cerb_net_finalize(cerb_net* net) {
    if (!net) return fail("cerb_net_finalize: null handle");
    if (net->finalized) return 0;
    // ---- stem (7x7, Cin=3) ------------------------------------------------------------------------------------
    {
        const HostTensor* w;
        if (get(net, "backbone.conv1.weight", {64, 3, 7, 7}, &w)) return 1;
        Fold f;
        if (bn_fold(net, "backbone.bn1", 64, &f)) return 1;
        if (!net->fold_bn) {
            f.scale.assign(64, 1.f);
            f.shift.assign(64, 0.f);
            if (store_bn(net, "stem", {"backbone.bn1"}, 64)) return 1;
        }
        std::vector<float> wp(7 * 12 * 2 * 64, 0.f);
        for (int ky = 0; ky < 7; ++ky)
            for (int t = 0; t < 12; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = s * 32 + (lane & 31), kk = 2 * t + (lane >> 5);
                        float v = 0.f;
                        if (kk < 21) {
                            const int kx = kk / 3, c = kk % 3;
                            v = w->data[(((size_t)co * 3 + c) * 7 + ky) * 7 + kx] * f.scale[co];
                        }
                        wp[((ky * 12 + t) * 2 + s) * 64 + lane] = v;
                    }
        if (upload(net, wp, &net->stem_w) || upload(net, f.shift, &net->stem_b)) return 1;
        if (!net->fold_bn) {
            if (upload(net, w->data, &net->stem_raw)) return 1;
            net->param_slots["backbone.conv1.weight"].push_back({net->stem_raw, 64 * 147});
        }
    }
    // ---- residual trunk ---------------------------------------------------------------------------------------
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = kFilters[li + 1];
        for (int b = 0; b < kLayers[li]; ++b) {
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            if (make_conv(net, p + ".conv1", {p + ".conv1.weight"}, {}, {p + ".bn1"}, planes, inpl, 3, stride)) return 1;
            if (make_conv(net, p + ".conv2", {p + ".conv2.weight"}, {}, {p + ".bn2"}, planes, planes, 3, 1)) return 1;
            if (stride != 1 || inpl != planes)
                if (make_conv(net, p + ".downsample", {p + ".downsample.0.weight"}, {}, {p + ".downsample.1"}, planes, inpl, 1, stride)) return 1;
            inpl = planes;
        }
    }
    if (make_conv(net, "conv_map", {"conv_map.weight"}, {}, {}, 256, 512, 1, 1)) return 1;
    // ---- dense decoders: grouped over decoders ----------------------------------------------------------------
    const int dec_in[4] = {256, 128, 64, 64};
    const int dec_u[4][2] = {{256, 128}, {128, 64}, {64, 64}, {64, 64}};
    if (!net->dense_idx.empty()) {
        for (int u = 0; u < 4; ++u) {
            int c = dec_in[u];
            for (int j = 0; j < 2; ++j) {
                std::vector<std::string> wk, bk, bnk;
                for (int di : net->dense_idx) {
                    const std::string p = "decoder_head." + net->dec[di].name + "." + std::to_string(u) + ".block." + std::to_string(j);
                    wk.push_back(p + ".conv.weight");
                    bk.push_back(p + ".conv.bias");
                    bnk.push_back(p + ".bn");
                }
                if (make_conv(net, "dec." + std::to_string(u) + "." + std::to_string(j), wk, bk, bnk, dec_u[u][j], c, 3, 1)) return 1;
                c = dec_u[u][j];
            }
        }
        for (int di : net->dense_idx) {
            const DecoderCfg& d = net->dec[di];
            const std::string p = "output_head." + d.name + "." + d.head + ".x";
            const HostTensor *w1, *b1, *w2, *b2;
            Fold f;
            if (get(net, p + ".0.block.0.conv.weight", {96, 64, 1, 1}, &w1) || get(net, p + ".0.block.0.conv.bias", {96}, &b1) ||
                bn_fold(net, p + ".0.block.0.bn", 96, &f) || get(net, p + ".1.conv.weight", {d.out_ch, 96, 1, 1}, &w2) ||
                get(net, p + ".1.conv.bias", {d.out_ch}, &b2))
                return 1;
            // layouts of head_kernel (v_mfma_f32_16x16x4_f32, lane = (row/col l & 15, k-slot l >> 4)):
            //   w1p[blk 6][g 4][lane][t]  = W1[16 blk + (l & 15)][16 g + 4 (l >> 4) + t]  (BN folded)
            //   w2p[blk 6][lane][r]       = W2[l & 15][16 blk + 4 (l >> 4) + r]           (rows >= out_ch are zero)
            std::vector<float> w1p(6 * 4 * 64 * 4), b1p(96), w2p(6 * 64 * 4, 0.f), b2p(32, 0.f);
            for (int blk = 0; blk < 6; ++blk)
                for (int G = 0; G < 4; ++G)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int t = 0; t < 4; ++t) {
                            const int hid = blk * 16 + (lane & 15), ci = G * 16 + 4 * (lane >> 4) + t;
                            w1p[((blk * 4 + G) * 64 + lane) * 4 + t] = w1->data[(size_t)hid * 64 + ci] * f.scale[hid];
                        }
            for (int c = 0; c < 96; ++c) b1p[c] = b1->data[c] * f.scale[c] + f.shift[c];
            for (int blk = 0; blk < 6; ++blk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int o = lane & 15, hid = blk * 16 + 4 * (lane >> 4) + r;
                        w2p[(blk * 64 + lane) * 4 + r] = (o < d.out_ch) ? w2->data[(size_t)o * 96 + hid] : 0.f;
                    }
            for (int c = 0; c < d.out_ch; ++c) b2p[c] = b2->data[c];
            //   w2q[set 2][blk 6][lane][r] = W2[4 set + (l & 3)][16 blk + 4 (l >> 4) + r]   (head_group_kernel<true>: v_mfma_f32_4x4x1_16B_f32)
            std::vector<float> w2q(2 * 24 * 64, 0.f);
            for (int set = 0; set < 2; ++set)
                for (int blk = 0; blk < 6; ++blk)
                    for (int r = 0; r < 4; ++r)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = 4 * set + (lane & 3), hid = blk * 16 + 4 * (lane >> 4) + r;
                            if (o < d.out_ch) w2q[((set * 6 + blk) * 64 + lane) * 4 + r] = w2->data[(size_t)o * 96 + hid];
                        }
            if (!net->fold_bn) {
                float *r1, *rb1, *r2, *rb2;
                if (upload(net, w1->data, &r1) || upload(net, b1->data, &rb1) || upload(net, w2->data, &r2) || upload(net, b2->data, &rb2)) return 1;
                net->head_rw1.push_back(r1); net->head_rb1.push_back(rb1); net->head_rw2.push_back(r2); net->head_rb2.push_back(rb2);
                net->param_slots[p + ".0.block.0.conv.weight"].push_back({r1, 96 * 64});
                net->param_slots[p + ".0.block.0.conv.bias"].push_back({rb1, 96});
                net->param_slots[p + ".1.conv.weight"].push_back({r2, (long long)d.out_ch * 96});
                net->param_slots[p + ".1.conv.bias"].push_back({rb2, d.out_ch});
                if (store_bn(net, "head." + std::to_string(net->head_rw1.size() - 1), {p + ".0.block.0.bn"}, 96)) return 1;
            }
            float *dw1, *db1, *dw2, *db2;
            float* dw2q;
            if (upload(net, w1p, &dw1) || upload(net, b1p, &db1) || upload(net, w2p, &dw2) || upload(net, b2p, &db2) || upload(net, w2q, &dw2q)) return 1;
            net->head_w1.push_back(dw1); net->head_b1.push_back(db1); net->head_w2.push_back(dw2); net->head_b2.push_back(db2); net->head_w2q.push_back(dw2q);
        }
    }
    // ---- Patch-Class branch -----------------------------------------------------------------------------------
    if (net->pc_idx >= 0) {
        const int oc = net->dec[net->pc_idx].out_ch;
        const std::string p = "decoder_head.Patch-Class";
        Fold f1, f2;
        const HostTensor *w1, *b1, *w2, *b2;
        if (bn_fold(net, p + ".bn1", 512, &f1) || bn_fold(net, p + ".bn2", 256, &f2) || get(net, p + ".conv1.weight", {256, 512, 1, 1}, &w1) ||
            get(net, p + ".conv1.bias", {256}, &b1) || get(net, p + ".conv2.weight", {oc, 256, 1, 1}, &w2) || get(net, p + ".conv2.bias", {oc}, &b2))
            return 1;
        if (!net->fold_bn) {
            if (upload(net, w1->data, &net->pc_rw1) || upload(net, b1->data, &net->pc_rb1) || upload(net, w2->data, &net->pc_rw2) ||
                upload(net, b2->data, &net->pc_rb2) || store_bn(net, "pc.bn1", {p + ".bn1"}, 512) || store_bn(net, "pc.bn2", {p + ".bn2"}, 256))
                return 1;
            net->param_slots[p + ".conv1.weight"].push_back({net->pc_rw1, 256 * 512});
            net->param_slots[p + ".conv1.bias"].push_back({net->pc_rb1, 256});
            net->param_slots[p + ".conv2.weight"].push_back({net->pc_rw2, (long long)oc * 256});
            net->param_slots[p + ".conv2.bias"].push_back({net->pc_rb2, oc});
        }
        std::vector<float> w1t(512 * 256), b1f(256), w2t(256 * 16, 0.f), b2f(16, 0.f);
        for (int o = 0; o < 256; ++o) {
            for (int c = 0; c < 512; ++c) w1t[(size_t)c * 256 + o] = w1->data[(size_t)o * 512 + c] * f2.scale[o];
            b1f[o] = b1->data[o] * f2.scale[o] + f2.shift[o];
        }
        for (int o = 0; o < oc; ++o) {
            for (int c = 0; c < 256; ++c) w2t[(size_t)c * 16 + o] = w2->data[(size_t)o * 256 + c];
            b2f[o] = b2->data[o];
        }
        if (upload(net, f1.scale, &net->pc_bn1s) || upload(net, f1.shift, &net->pc_bn1b) || upload(net, w1t, &net->pc_w1t) ||
            upload(net, b1f, &net->pc_b1) || upload(net, w2t, &net->pc_w2t) || upload(net, b2f, &net->pc_b2))
            return 1;
    }
    if (!net->fold_bn) {
        std::vector<float> z(512, 0.f);
        if (upload(net, z, &net->zero_bias)) return 1;
    }
    if (!net->fold_bn) HIP_OK(hipDeviceSynchronize());  // the packing kernels ran on the null stream
    if (net->reusing && net->reuse_cursor != net->n_finalize_allocs) return fail("reload: fewer tensors than the handle was finalized with");
    if (!net->reusing) net->n_finalize_allocs = net->dev_allocs.size();
    net->reusing = false;
    net->host.clear();
    net->finalized = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
static int prof_begin(cerb_net* net, const std::string& name, const std::string& kernel, double flops, hipStream_t st) {
    if (!net->profiling) return 0;
    if (net->prof_n == net->prof.size()) {
        cerb_net::ProfRec r;
        HIP_OK(hipEventCreate(&r.e0));
        HIP_OK(hipEventCreate(&r.e1));
        net->prof.push_back(r);
    }
    cerb_net::ProfRec& r = net->prof[net->prof_n];
    r.name = name; r.kernel = kernel; r.flops = flops;
    HIP_OK(hipEventRecord(r.e0, st));
    net->prof_open = true;
    return 0;
}
static int prof_end(cerb_net* net, hipStream_t st) {
    if (!net->profiling) return 0;
    HIP_OK(hipEventRecord(net->prof[net->prof_n].e1, st));
    net->prof_n++;
    net->prof_open = false;
    return 0;
}

static int train_wino2_fresh(cerb_net* net, const std::string& name, PackedConv& cm, int dgrad, hipStream_t st) {
    if (net->fold_bn) return 0;
    bool& used = dgrad ? cm.wino_dgrad_used : cm.wino_used;
    bool& stale = dgrad ? cm.wino_dgrad_stale : cm.wino_stale;
    used = true;
    if (!stale) return 0;
    auto rit = net->raw.find(name);
    if (rit == net->raw.end()) return fail("conv " + name + ": no raw weights to re-pack");
    const size_t nw = (size_t)cm.cout * cm.cin * 9, nu = (size_t)cm.cout * cm.cin * 16;
    for (int g = 0; g < cm.groups; ++g)
        HIP_OK(cerb_launch_pack_wino(rit->second.w + g * nw, (dgrad ? cm.wino_dgrad : cm.wino) + g * nu, dgrad ? cm.cin : cm.cout, dgrad ? cm.cout : cm.cin, dgrad, st));
    stale = false;
    return 0;
}

// Train packing: the F(4x4,3x3) weights of one conv ([layout 4 / 4b][forward / data gradient]) are transformed on the device from the raw
// state-dict copy at first use; cerb_net_update_params repeats it for the slots that exist.
static int train_wino4_slot(cerb_net* net, const std::string& name, PackedConv& cm, int w4b, int dgrad, hipStream_t st, float** out) {
    float*& slot = cm.wino4_t[w4b][dgrad];
    if (!slot) {
        auto rit = net->raw.find(name);
        if (rit == net->raw.end()) return fail("conv " + name + ": no raw weights for the F(4x4) transform");
        const size_t nw = (size_t)cm.cout * cm.cin * 9, nu = (size_t)cm.cout * cm.cin * 36;
        void* d = nullptr;
        HIP_OK(hipMalloc(&d, (size_t)cm.groups * nu * 4));
        net->dev_allocs.push_back(d);
        net->dev_alloc_bytes.push_back((size_t)cm.groups * nu * 4);
        slot = (float*)d;
        (void)nw;
        HIP_OK(cerb_launch_pack_wino4(rit->second.w, slot, dgrad ? cm.cin : cm.cout, dgrad ? cm.cout : cm.cin, dgrad, w4b, cm.groups, st));
    }
    *out = slot;
    return 0;
}

static int run_conv(cerb_net* net, const std::string& name, const float* in, const float* prev, const float* resid, float* out, int N,
                    int H, int W, int relu, int mode, long long in_gs, long long prev_gs, hipStream_t st, double* macs,
                    const int* roi = nullptr, long long planar_out_gs = 0) {  // planar_out_gs > 0: in / out are tile-planar (conv_wino4p.hip)
    auto it = net->conv.find(name);
    if (it == net->conv.end()) return fail("internal: conv " + name + " not packed");
    const PackedConv& c = it->second;
    struct BnReq {  // the request holds for this call only
        cerb_net* n;
        ~BnReq() { n->conv_bn_part = nullptr; }
    } bn_req{net};
    net->conv_bn_bpg = 0;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = in; p.prev = prev; p.wpack = c.w; p.bias = c.b; p.resid = resid; p.out = out;
    p.N = N; p.H = H; p.W = W; p.Cin = c.cin; p.Cout = c.cout;
    p.Ho = (c.stride == 2) ? H / 2 : H;
    p.Wo = (c.stride == 2) ? W / 2 : W;
    p.relu = relu;
    p.groups = c.groups;
    p.in_gs = in_gs; p.prev_gs = prev_gs;
    p.w_gs = (long long)c.cout * c.cin * c.ks * c.ks;
    p.bias_gs = c.cout;
    p.resid_gs = 0;
    p.out_gs = (long long)N * p.Ho * p.Wo * c.cout;
    p.pk_off = net->packed_items ? 0 : 1;
    if (macs) {
        *macs += (double)c.groups * N * p.Ho * p.Wo * (double)c.cout * c.cin * c.ks * c.ks;
        if (!out) return 0;
    }
    const double fl = 2.0 * (double)c.groups * N * p.Ho * p.Wo * (double)c.cout * c.cin * c.ks * c.ks;
    // conv_algo 6 (default): F(4x4,3x3) for maps of at least 16 x 16 pixels -- conv_wino4b.hip (one-block items, 32-channel chunks) up to
    // 64 x 64, conv_wino4.hip (two-block items, half the weight traffic) above; F(2x2,3x3) for smaller maps, where a 16 x 16 block would
    // be mostly padding.  Measured per layer on a batch of 32 256-pixel tiles (scripts/dev_conv_layers.py): 16^2 x 512 ch 0.115 / 0.21 /
    // 0.146 ms (4b / 4 / F(2x2)), 64^2 x 128 ch 0.131 / 0.138 / 0.158, 256^2 x 64 ch x 5 decoders 2.63 / 2.52 / 3.28.  The rule looks at
    // the layer's geometry only -- never at the batch size or the region of interest -- so that a tile's values do not depend on what it
    // is batched with (sharded == unsharded, cropped == full stay bitwise, tests/test_drivers_gpu.py, test_net_gpu.py).
    const long long map_px = (long long)p.Ho * p.Wo;
    const bool use_w4 = net->conv_algo == 5 || net->conv_algo == 7 || (net->conv_algo == 6 && map_px >= 256);
    const bool planar = planar_out_gs > 0;
    static const long long w4b_max_px = [] {  // developer A/B only (scripts/gpu_session_r04c.sh): where conv_wino4b hands over to conv_wino4
        const char* e = getenv("CERB_W4B_MAX_PX");
        return e ? atoll(e) : 4096ll;
    }();
    const bool w4b = !planar && (net->conv_algo == 7 || (net->conv_algo == 6 && map_px <= w4b_max_px)) && c.cin % 64 == 0;
    if (planar && !(use_w4 && c.wino && mode == 0 && net->fold_bn && !resid)) return fail("internal: conv " + name + " cannot take the planar path");
    if (use_w4 && c.wino && mode == 0 && (!it->second.host_w.empty() || !net->fold_bn)) {
        PackedConv& cm = it->second;
        float* train_slot = nullptr;
        if (!net->fold_bn && train_wino4_slot(net, name, cm, w4b ? 1 : 0, 0, st, &train_slot)) return 1;
        float*& slot4 = !net->fold_bn ? train_slot : (w4b ? cm.wino4b : cm.wino4);
        if (!slot4) {  // first use: F(4x4,3x3) filter transform on the host, the kernel's per-wave layout, upload
            std::vector<float> w4;
            for (int g = 0; g < cm.groups; ++g) pack_wino4(cm.host_w.data() + (size_t)g * cm.cout * cm.cin * 9, cm.cout, cm.cin, &w4, w4b ? 1 : 0);
            void* d = nullptr;
            HIP_OK(hipMalloc(&d, w4.size() * 4));
            net->dev_allocs.push_back(d);
            net->dev_alloc_bytes.push_back(w4.size() * 4);
            HIP_OK(hipMemcpy(d, w4.data(), w4.size() * 4, hipMemcpyHostToDevice));
            slot4 = (float*)d;
        }
        p.wpack = slot4;
        p.w_gs = (long long)c.cout * c.cin * 36;
        double fl_done = fl;
        if (roi && roi[1] > roi[0] && roi[3] > roi[2]) {  // region of interest: report the work of the 16 x 16 blocks that run
            p.roi_y0 = roi[0]; p.roi_y1 = roi[1]; p.roi_x0 = roi[2]; p.roi_x1 = roi[3];
            const double ty = (roi[1] + 15) / 16 - roi[0] / 16, tx = (roi[3] + 15) / 16 - roi[2] / 16;
            fl_done = fl * (ty * 16.0 * tx * 16.0) / ((double)p.Ho * p.Wo);
        }
        if (planar) {
            p.out_gs = planar_out_gs;
            p.level_tag = net->planar_half ? 0 : 1;
            p.pl_byp = cerb_planar_blocks(p.Ho);
            p.pl_bxp = cerb_planar_blocks(p.Wo);
        }
        if (prof_begin(net, name, planar ? (p.level_tag ? "conv_wino4p<f4x4,16x16x2,planar>" : "conv_wino4p<f4x4,16x16x2,planar,half-res>") : w4b ? (cerb_wino4b_packed(p) ? (resid ? "conv_wino4b<f4x4,16t,res>" : "conv_wino4b<f4x4,16t>") : (resid ? "conv_wino4b<f4x4,16x16,res>" : "conv_wino4b<f4x4,16x16>")) : (resid ? "conv_wino4<f4x4,16x16x2,res>" : "conv_wino4<f4x4,16x16x2>"), fl_done, st)) return 1;
        if (net->conv_bn_part && !planar && !resid && !(roi && roi[1] > roi[0] && roi[3] > roi[2])) {
            p.bn_part = net->conv_bn_part;
            net->conv_bn_bpg = w4b ? cerb_wino4b_bn_blocks(p) : N * ((p.Ho + 15) / 16) * ((p.Wo + 15) / 16);
        }
        HIP_OK(planar ? cerb_launch_wino4p(p, st) : w4b ? cerb_launch_wino4b(p, st) : cerb_launch_wino4(p, st));
        if (prof_end(net, st)) return 1;
        return 0;
    }
    if (net->conv_algo && c.wino && mode == 0) {
        if (train_wino2_fresh(net, name, it->second, 0, st)) return 1;
        p.wpack = c.wino;
        p.w_gs = (long long)c.cout * c.cin * 16;
        double fl_done = fl;
        if (roi && roi[1] > roi[0] && roi[3] > roi[2]) {  // region of interest: report the work of the items that run (8 x 16 px each)
            p.roi_y0 = roi[0]; p.roi_y1 = roi[1]; p.roi_x0 = roi[2]; p.roi_x1 = roi[3];
            const double ty = (roi[1] + 7) / 8 - roi[0] / 8, tx = (roi[3] + 15) / 16 - roi[2] / 16;
            fl_done = fl * (ty * 8.0 * tx * 16.0) / ((double)p.Ho * p.Wo);
        }
        if (prof_begin(net, name, resid ? "conv_wino<f2x2,8x16,res>" : "conv_wino<f2x2,8x16>", fl_done, st)) return 1;
        HIP_OK(cerb_launch_wino(p, st));
        if (prof_end(net, st)) return 1;
        return 0;
    }
    const std::string kn = "conv_igemm<ks" + std::to_string(c.ks) + ",s" + std::to_string(c.stride) + ",mode" + std::to_string(mode) +
                           (p.Wo < 32 ? ",16x16>" : ",8x32>");
    if (prof_begin(net, name, kn, fl, st)) return 1;
    HIP_OK(cerb_launch_conv(p, c.ks, c.stride, mode, st));
    if (prof_end(net, st)) return 1;
    return 0;
}

static int forward_impl(cerb_net* net, const cerb_forward_io* io, hipStream_t st, double* macs) {
    g_call_stream = st;
    const bool dry = (macs != nullptr) && (io->tiles == nullptr) && (io->tiles_f32 == nullptr);
    const int N = io->n, H = io->h, W = io->w;
    if (N <= 0 || H <= 0 || W <= 0 || (H % 16) || (W % 16)) return fail("cerb_net_forward: tile H,W must be positive multiples of 16");
    const int out_h = io->out_h > 0 ? io->out_h : H, out_w = io->out_w > 0 ? io->out_w : W;
    if (out_h > H || out_w > W) return fail("cerb_net_forward: crop larger than tile");
    const int hs[5] = {H, H / 2, H / 4, H / 8, H / 16}, ws[5] = {W, W / 2, W / 4, W / 8, W / 16};
    const size_t D = net->dense_idx.size();
    const size_t guard = cerb_conv_guard_bytes(W);
    // Which decoder levels keep their three private tensors in the tile-planar layout (conv_wino4p.hip): the two last levels (64 channels) when
    // their maps are above conv_wino4b's range, on the default algorithms with folded BatchNorm.  One predicate for the allocation and the loop.
    auto level_is_planar = [&](int u) {
        if (dry || u < 2 || !net->planar || !net->fold_bn || net->conv_algo != 6 || net->head_algo < 1) return false;
        const std::string n0 = "dec." + std::to_string(u) + ".0", n1 = "dec." + std::to_string(u) + ".1";
        auto c0 = net->conv.find(n0), c1 = net->conv.find(n1);
        if (c0 == net->conv.end() || c1 == net->conv.end() || !c0->second.wino) return false;
        return (long long)hs[3 - u] * ws[3 - u] > 4096 && c0->second.cin == 64 && c0->second.cout == 64 && c1->second.cout == 64;
    };
    if (!dry) {
        if (net->x0.ensure((size_t)N * H * W * 64 * 4, guard) || net->pool.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard)) return fail("workspace allocation failed");
        for (int i = 1; i < 5; ++i)
            if (net->x[i].ensure((size_t)N * hs[i] * ws[i] * kFilters[i] * 4, guard)) return fail("workspace allocation failed");
        if (net->ta.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) || net->tb.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) ||
            net->cm.ensure((size_t)N * hs[4] * ws[4] * 256 * 4, guard))
            return fail("workspace allocation failed");
        if (D) {
            // dsum / dmid hold the entry sum and the first conv's output of a level that runs NHWC (per decoder 32^2 x 256, 64^2 x 128, and -- when the
            // tile-planar levels are off -- 128^2 x 64, 256^2 x 64), dout[u] its second conv's output; the planar levels have their own buffers, so with
            // them on (the default) only the two coarse levels count here: 8 GB less per handle at 32 tiles of 256^2 (a first attempt in this
            // round faulted on a wrong channel count in its own formula, not on a kernel reaching past its tensor: sized from the packed convolutions'
            // channel counts below, every fixture and geometry of the GPU suite runs).  CERB_LEGACY_WS=1: everything at last-level size, as before.
            static const bool exact_ws = getenv("CERB_LEGACY_WS") == nullptr;
            size_t need_sum = 0, need_mid = 0;
            const int oc[4] = {128, 64, 64, 64};
            for (int u = 0; u < 4; ++u) {
                if (exact_ws && level_is_planar(u)) continue;
                const std::string n0 = "dec." + std::to_string(u) + ".0";
                auto c0 = net->conv.find(n0);
                const size_t px = (size_t)D * N * hs[3 - u] * ws[3 - u];
                need_sum = std::max(need_sum, px * (size_t)(c0 != net->conv.end() ? c0->second.cin : 256) * 4);
                need_mid = std::max(need_mid, px * (size_t)(c0 != net->conv.end() ? c0->second.cout : 256) * 4);
            }
            if (!exact_ws) need_sum = need_mid = D * (size_t)N * H * W * 64 * 4;
            if (net->dmid.ensure(need_mid, guard)) return fail("workspace allocation failed");
            if (net->conv_algo && net->dsum.ensure(need_sum, guard)) return fail("workspace allocation failed");
            for (int u = 0; u < 4; ++u) {
                if (exact_ws && level_is_planar(u)) continue;  // its outputs live in the planar buffers
                if (net->dout[u].ensure(D * (size_t)N * hs[3 - u] * ws[3 - u] * oc[u] * 4, guard)) return fail("workspace allocation failed");
            }
        }
    }
    // ---- encoder ----------------------------------------------------------------------------------------------
    if (macs) *macs += (double)N * H * W * 64.0 * 147.0;
    if (!dry) {
        StemParams sp;
        sp.tiles = io->tiles; sp.tiles_f32 = io->tiles ? nullptr : io->tiles_f32; sp.wpack = net->stem_w; sp.bias = net->stem_b; sp.out = net->x0.p; sp.N = N; sp.H = H; sp.W = W; sp.relu = 1;
        sp.tiles_x = sp.tiles_y = 0;
        if (prof_begin(net, "stem", "stem_conv7x7", 2.0 * N * H * W * 64.0 * 147.0, st)) return 1;
        HIP_OK(cerb_launch_stem(sp, st));
        if (prof_end(net, st)) return 1;
        if (prof_begin(net, "maxpool", "maxpool3x3s2", 0.0, st)) return 1;
        HIP_OK(cerb_launch_maxpool(net->x0.p, net->pool.p, N, H, W, 64, st));
        if (prof_end(net, st)) return 1;
    }
    float* cur = dry ? nullptr : net->pool.p;
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = kFilters[li + 1];
        const int Hi = (li == 0) ? hs[1] : hs[li], Wi = (li == 0) ? ws[1] : ws[li];  // input resolution of this layer
        for (int b = 0; b < kLayers[li]; ++b) {
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int hin = (b == 0) ? Hi : hs[li + 1], win = (b == 0) ? Wi : ws[li + 1];
            float* t1 = dry ? nullptr : net->ta.p;
            // block output: ping-pong between tb and x[li+1]; make the LAST block of the layer land in x[li+1]
            const bool last = (b == kLayers[li] - 1);
            float* outb = dry ? nullptr : (((kLayers[li] - 1 - b) % 2 == 0) ? net->x[li + 1].p : net->tb.p);
            (void)last;
            const float* idt = cur;
            if (run_conv(net, p + ".conv1", cur, nullptr, nullptr, t1, N, hin, win, 1, 0, 0, 0, st, macs)) return 1;
            if (stride != 1 || inpl != planes) {
                // downsample(x): 1x1 stride-2 conv + BN, written to dmid-free scratch: reuse pool buffer (dead after layer1.0)
                float* ds = dry ? nullptr : net->pool.p;
                if (run_conv(net, p + ".downsample", cur, nullptr, nullptr, ds, N, hin, win, 0, 0, 0, 0, st, macs)) return 1;
                idt = ds;
            }
            if (run_conv(net, p + ".conv2", t1, nullptr, idt, outb, N, hs[li + 1], ws[li + 1], 1, 0, 0, 0, st, macs)) return 1;
            cur = outb;
            inpl = planes;
        }
    }
    if (run_conv(net, "conv_map", dry ? nullptr : net->x[4].p, nullptr, nullptr, dry ? nullptr : net->cm.p, N, hs[4], ws[4], 0, 0, 0, 0, st, macs)) return 1;

    const long long tile_stride = io->tile_stride ? io->tile_stride : (long long)out_h * out_w;
    const long long row_stride = io->row_stride ? io->row_stride : out_w;
    // ---- Patch-Class ------------------------------------------------------------------------------------------
    if (net->pc_idx >= 0) {
        if (macs) *macs += (double)N * (512.0 * 256 + 256.0 * net->dec[net->pc_idx].out_ch);
        const bool want = io->out && io->out[net->pc_idx];
        const bool wantl = io->logits && io->logits[net->pc_idx];
        if (!dry && (want || wantl)) {
            PatchClassParams pp;
            memset(&pp, 0, sizeof(pp));
            pp.x4 = net->x[4].p; pp.bn1_s = net->pc_bn1s; pp.bn1_b = net->pc_bn1b; pp.w1t = net->pc_w1t; pp.b1 = net->pc_b1;
            pp.w2t = net->pc_w2t; pp.b2 = net->pc_b2;
            pp.N = N; pp.Hf = hs[4]; pp.Wf = ws[4]; pp.out_ch = net->dec[net->pc_idx].out_ch;
            pp.out_h = out_h; pp.out_w = out_w;
            pp.logits = wantl ? io->logits[net->pc_idx] : nullptr;
            pp.out = want ? (float*)io->out[net->pc_idx] : nullptr;
            pp.tile_off = io->tile_off; pp.tile_stride = tile_stride; pp.row_stride = row_stride;
            if (prof_begin(net, "patch_class", "patch_class", 2.0 * N * (512.0 * 256 + 256.0 * pp.out_ch), st)) return 1;
            HIP_OK(cerb_launch_patch_class(pp, st));
            if (prof_end(net, st)) return 1;
        }
    }
    // ---- dense decoders (all decoders of a level in ONE grouped launch) ---------------------------------------
    if (D) {
        const float* skips[4] = {dry ? nullptr : net->x[3].p, dry ? nullptr : net->x[2].p, dry ? nullptr : net->x[1].p, dry ? nullptr : net->x0.p};
        const float* prev = dry ? nullptr : net->cm.p;
        long long prev_gs = 0;  // conv_map output is shared by every decoder
        const int oc[4] = {128, 64, 64, 64};
        // Regions of interest (crop_rois below): with a centre crop smaller than the tile (the reference's default 448 -> 144 keeps
        // 10 % of the pixels, infer/wsi.py / run_desc.py:452-491) only the part of every decoder map that the kept window depends on is
        // computed: 3x3 convs widen the window by one pixel each, the bilinear x2 by one source pixel.  The encoder sees the whole tile
        // (its receptive field covers it); results inside the window are bit-identical to the full computation.
        int roi_out[4][4], roi_mid[4][4], roi_sum[4][4];
        const bool any_logits = [&] {
            if (!io->logits) return false;
            for (size_t k = 0; k < net->dec.size(); ++k)
                if (io->logits[k]) return true;
            return false;
        }();
        const bool use_roi = !dry && net->crop_roi && (net->conv_algo == 1 || net->conv_algo >= 5) && !any_logits && (out_h < H || out_w < W);
        if (use_roi) {
            int y0 = (int)((H - out_h) * 0.5), x0 = (int)((W - out_w) * 0.5), y1 = y0 + out_h, x1 = x0 + out_w;
            // A Winograd tile mixes its WHOLE input patch into every output (the contributions of the pixels a 3x3 filter does not touch
            // cancel only up to rounding), so each window is widened to whole 4x4 tiles (F(4x4); F(2x2)'s 2x2 tiles divide them) before
            // the next one is derived from it: every tile that overlaps a window then reads nothing but valid producer pixels, and a
            // tile's values do not depend on what the workspace held before (or on the batch it is computed with).
            for (int u = 3; u >= 0; --u) {
                const int hh = hs[3 - u], ww = ws[3 - u];
                auto widen = [&](int* r, const int* in, int d) {  // in grown by d pixels, then out to multiples of 4, clamped to the map
                    r[0] = std::max(0, (in[0] - d) & ~3); r[1] = std::min(hh, (in[1] + d + 3) & ~3);
                    r[2] = std::max(0, (in[2] - d) & ~3); r[3] = std::min(ww, (in[3] + d + 3) & ~3);
                };
                const int win[4] = {y0, y1, x0, x1};
                widen(roi_out[u], win, 0);
                widen(roi_mid[u], roi_out[u], 1);
                widen(roi_sum[u], roi_mid[u], 1);
                // bilinear x2, align_corners = False: output o reads sources floor(o / 2 - 0.25) and the next one
                y0 = std::max(0, roi_sum[u][0] / 2 - 1); y1 = std::min(hh / 2, (roi_sum[u][1] - 1) / 2 + 2);
                x0 = std::max(0, roi_sum[u][2] / 2 - 1); x1 = std::min(ww / 2, (roi_sum[u][3] - 1) / 2 + 2);
            }
        }
        bool feat_planar = false, prev_planar = false;
        for (int u = 0; u < 4; ++u) {
            const int hh = hs[3 - u], ww = ws[3 - u];
            const std::string n0 = "dec." + std::to_string(u) + ".0", n1 = "dec." + std::to_string(u) + ".1";
            const int cmid = net->conv[n0].cout;
            const int cin0 = net->conv[n0].cin;
            // The last level (40 % of the network's FLOPs) keeps its three private tensors -- skip + upsample, the first conv's output, the
            // features the heads read -- in the tile-planar layout: conv_wino4p.hip stores 1-KiB rows straight from its registers and reads
            // whole lines, nothing masks an edge.  Same arithmetic in the same order: bit-identical to the NHWC path (cerb_net_set_planar(0)).
            // The level below (same 64 channels at half the resolution) does the same when its maps are above conv_wino4b's range, and hands
            // its output to the last level's up-sampling in that layout.
            const bool lvl_planar = level_is_planar(u) && (u == 3 || prev_gs > 0);
            if (lvl_planar) {
                // two buffers per planar level: the entry sum is dead once the first conv has read it, so the second conv writes its output there
                // (a third buffer of 8.6 GB at 64 tiles of 256^2 held it before); same geometry, same zero ring
                PlanarBuf &bs = u == 3 ? net->psum : net->psum2, &bm = u == 3 ? net->pmid : net->pmid2, &bo = bs;
                if (bs.ensure((int)D, N, hh, ww, 64, st) || bm.ensure((int)D, N, hh, ww, 64, st)) return fail("workspace allocation failed");
                if (prof_begin(net, n0 + ".up", "upsample2_add_planar", 0.0, st)) return 1;
                HIP_OK(cerb_launch_upsample2_add_planar(skips[u], prev, bs.b.p, (int)D, N, hh, ww, cin0, prev_gs, bs.gs(), use_roi ? roi_sum[u] : nullptr, prev_planar ? 1 : 0, st));
                if (prof_end(net, st)) return 1;
                net->planar_half = (u != 3);
                if (run_conv(net, n0, bs.b.p, nullptr, nullptr, bm.b.p, N, hh, ww, 1, 0, bs.gs(), 0, st, macs, use_roi ? roi_mid[u] : nullptr, bm.gs()))
                    return 1;
                if (run_conv(net, n1, bm.b.p, nullptr, nullptr, bo.b.p, N, hh, ww, 1, 0, bm.gs(), 0, st, macs, use_roi ? roi_out[u] : nullptr, bo.gs()))
                    return 1;
                if (u == 3) feat_planar = true;
                prev = bo.b.p;
                prev_gs = bo.gs();
                prev_planar = true;
                continue;
            }
            if (prev_planar) return fail("internal: a tile-planar decoder level feeds an NHWC one");
            if (net->conv_algo && net->conv[n0].wino && !dry) {
                // skip + upsample2x(prev) as one HBM pass, then the Winograd conv over the materialised sum
                if (prof_begin(net, n0 + ".up", "upsample2_add", 0.0, st)) return 1;
                HIP_OK(cerb_launch_upsample2_add(skips[u], prev, net->dsum.p, (int)D, N, hh, ww, cin0, prev_gs, use_roi ? roi_sum[u] : nullptr, st));
                if (prof_end(net, st)) return 1;
                if (run_conv(net, n0, net->dsum.p, nullptr, nullptr, net->dmid.p, N, hh, ww, 1, 0, (long long)N * hh * ww * cin0, 0, st, macs,
                             use_roi ? roi_mid[u] : nullptr))
                    return 1;
            } else if (run_conv(net, n0, skips[u], prev, nullptr, dry ? nullptr : net->dmid.p, N, hh, ww, 1, 1, 0, prev_gs, st, macs)) return 1;
            if (run_conv(net, n1, dry ? nullptr : net->dmid.p, nullptr, nullptr, dry ? nullptr : net->dout[u].p, N, hh, ww, 1, 0,
                         (long long)N * hh * ww * cmid, 0, st, macs, use_roi ? roi_out[u] : nullptr))
                return 1;
            prev = dry ? nullptr : net->dout[u].p;
            prev_gs = (long long)N * hh * ww * oc[u];
        }
        HeadParams hps[8];
        int n_hp = 0;
        double head_flops = 0.0;
        for (size_t k = 0; k < D; ++k) {
            const int di = net->dense_idx[k];
            const DecoderCfg& d = net->dec[di];
            if (macs) *macs += (double)N * H * W * (64.0 * 96 + 96.0 * d.out_ch);
            const bool want = io->out && io->out[di];
            const bool wantl = io->logits && io->logits[di];
            if (dry || !(want || wantl)) continue;
            HeadParams hp;
            memset(&hp, 0, sizeof(hp));
            hp.feat = feat_planar ? net->psum.b.p + k * net->psum.gs() : net->dout[3].p + k * (size_t)N * H * W * 64;  // (planar: the last level's output lives in its sum buffer)
            hp.feat_planar = feat_planar ? 1 : 0;
            hp.pl_byp = cerb_planar_blocks(H);
            hp.pl_bxp = cerb_planar_blocks(W);
            hp.w1p = net->head_w1[k]; hp.b1 = net->head_b1[k]; hp.w2p = net->head_w2[k]; hp.b2 = net->head_b2[k]; hp.w2q = net->head_w2q[k];
            hp.N = N; hp.H = H; hp.W = W; hp.out_ch = d.out_ch; hp.kind = d.kind;
            hp.crop_y0 = (int)((H - out_h) * 0.5); hp.crop_x0 = (int)((W - out_w) * 0.5);  // cropping_center, misc/utils.py:94-104
            hp.out_h = want ? out_h : 0; hp.out_w = want ? out_w : 0;
            hp.roi = use_roi ? 1 : 0;
            hp.logits = wantl ? io->logits[di] : nullptr;
            hp.absmax_bits = io->logit_absmax ? io->logit_absmax + di : nullptr;
            if (want) {
                if (d.kind == 0) hp.out_inst = (float*)io->out[di];
                else if (io->type_is_u8) hp.out_type_u8 = (unsigned char*)io->out[di];
                else hp.out_type_i64 = (long long*)io->out[di];
            }
            hp.tile_off = io->tile_off; hp.tile_stride = tile_stride; hp.row_stride = row_stride;
            const double head_px = hp.roi ? (double)out_h * (((hp.crop_x0 + out_w + 15) / 16 - hp.crop_x0 / 16) * 16.0) : (double)H * W;
            const double fl = 2.0 * N * head_px * (64.0 * 96 + 96.0 * d.out_ch);
            if (net->head_algo == 0) {  // one launch per head (round-1 kernel, kept for A/B: cerb_net_set_head_algo)
                if (prof_begin(net, "head." + d.name, "head", fl, st)) return 1;
                HIP_OK(cerb_launch_head(hp, st));
                if (prof_end(net, st)) return 1;
            } else {
                if (n_hp == 8) {  // a grouped launch carries at most 8 heads: more dense decoders go out in chunks of 8
                    if (prof_begin(net, "heads", "head_group", head_flops, st)) return 1;
                    HIP_OK(cerb_launch_head_group(hps, n_hp, st, net->head_algo == 1 ? 1 : 0));
                    if (prof_end(net, st)) return 1;
                    n_hp = 0;
                    head_flops = 0.0;
                }
                hps[n_hp++] = hp;
                head_flops += fl;
            }
        }
        if (n_hp > 0) {  // all dense heads of the batch in ONE grouped launch (models/utils/net_layers.py:31-38 x5)
            if (prof_begin(net, "heads", "head_group", head_flops, st)) return 1;
            HIP_OK(cerb_launch_head_group(hps, n_hp, st, net->head_algo == 1 ? 1 : 0));
            if (prof_end(net, st)) return 1;
        }
    }
    if (!dry && io->feats) {
        const float* src[6] = {net->x0.p, net->x[1].p, net->x[2].p, net->x[3].p, net->cm.p, net->x[4].p};
        const size_t nb[6] = {(size_t)N * H * W * 64, (size_t)N * hs[1] * ws[1] * 64, (size_t)N * hs[2] * ws[2] * 128,
                              (size_t)N * hs[3] * ws[3] * 256, (size_t)N * hs[4] * ws[4] * 256, (size_t)N * hs[4] * ws[4] * 512};
        for (int i = 0; i < 6; ++i)
            if (io->feats[i]) HIP_OK(hipMemcpyAsync(io->feats[i], src[i], nb[i] * 4, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Train-mode forward (models/run_desc.py:79-86: model.train(); pred_dict = model(img_list, train_dec_list)): every BatchNorm uses the
// statistics of the batch, so the convolutions run with their raw weights (net packed with cerb_net_set_fold_bn(net, 0)) and each is
// followed by cerb_launch_bn_stats / cerb_launch_bn_apply.  Returns the full-resolution logits of every head.  First version of
// the forward half of BASELINE configs[4]: nothing is kept for a backward pass yet and the running statistics are not updated.
// Groups in eval mode normalise with their running statistics: the batch statistics just computed are replaced before the apply pass reads them.
static int bn_eval_override(const cerb_net::BnDev& b, float* mean, float* rstd, hipStream_t st) {
    for (int g = 0; g < (int)b.eval.size(); ++g)
        if (b.eval[g]) {
            HIP_OK(hipMemcpyAsync(mean + (size_t)g * b.C, b.run_mean + (size_t)g * b.C, (size_t)b.C * 4, hipMemcpyDeviceToDevice, st));
            HIP_OK(hipMemcpyAsync(rstd + (size_t)g * b.C, b.run_rstd + (size_t)g * b.C, (size_t)b.C * 4, hipMemcpyDeviceToDevice, st));
        }
    return 0;
}

static int bn_train(cerb_net* net, const std::string& name, float* x, const float* resid, long long group_stride, long long rows, int relu,
                    hipStream_t st) {
    auto it = net->bn.find(name);
    if (it == net->bn.end()) return fail("internal: no BatchNorm parameters for " + name);
    const cerb_net::BnDev& b = it->second;
    if (net->t_mean.ensure((size_t)b.groups * b.C * 4, 0) || net->t_rstd.ensure((size_t)b.groups * b.C * 4, 0) ||
        net->t_ws.ensure(cerb_bn_workspace_bytes(b.groups, rows, b.C), 0))
        return fail("workspace allocation failed");
    HIP_OK(cerb_launch_bn_stats(x, group_stride, rows, b.C, b.groups, 1e-5f, net->t_mean.p, net->t_rstd.p, nullptr, net->t_ws.p, st));
    if (bn_eval_override(b, net->t_mean.p, net->t_rstd.p, st)) return 1;
    HIP_OK(cerb_launch_bn_apply(x, nullptr, resid, group_stride, rows, b.C, b.groups, net->t_mean.p, net->t_rstd.p, b.gamma, b.beta, relu, st));
    return 0;
}

extern "C" int cerb_net_forward_train(cerb_net* net, const cerb_train_io* io, void* hip_stream) {
    if (!net || !io || !io->tiles || !io->logits) return fail("cerb_net_forward_train: null argument");
    if (!net->finalized) return fail("cerb_net_forward_train: call cerb_net_finalize first");
    if (net->fold_bn) return fail("cerb_net_forward_train: the network was packed for inference (BatchNorm folded); call cerb_net_set_fold_bn(net, 0) before cerb_net_finalize");
    hipStream_t st = (hipStream_t)hip_stream;
    g_call_stream = st;
    const int N = io->n, H = io->h, W = io->w;
    if (N <= 0 || H <= 0 || W <= 0 || (H % 16) || (W % 16)) return fail("cerb_net_forward_train: tile H,W must be positive multiples of 16");
    const int hs[5] = {H, H / 2, H / 4, H / 8, H / 16}, ws[5] = {W, W / 2, W / 4, W / 8, W / 16};
    const size_t D = net->dense_idx.size();
    const size_t guard = cerb_conv_guard_bytes(W);
    if (net->x0.ensure((size_t)N * H * W * 64 * 4, guard) || net->pool.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) ||
        net->ta.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) || net->tb.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) ||
        net->cm.ensure((size_t)N * hs[4] * ws[4] * 256 * 4, guard))
        return fail("workspace allocation failed");
    for (int i = 1; i < 5; ++i)
        if (net->x[i].ensure((size_t)N * hs[i] * ws[i] * kFilters[i] * 4, guard)) return fail("workspace allocation failed");
    const int oc[4] = {128, 64, 64, 64};
    if (D) {
        if (net->dmid.ensure(D * (size_t)N * H * W * 64 * 4, guard) || net->dsum.ensure(D * (size_t)N * H * W * 64 * 4, guard)) return fail("workspace allocation failed");
        for (int u = 0; u < 4; ++u)
            if (net->dout[u].ensure(D * (size_t)N * hs[3 - u] * ws[3 - u] * oc[u] * 4, guard)) return fail("workspace allocation failed");
    }
    const int saved_algo = net->conv_algo;
    // ---- encoder: conv -> BN(batch) -> ReLU -----------------------------------------------------------------------------------
    {
        StemParams sp;
        sp.tiles = io->tiles; sp.tiles_f32 = nullptr; sp.wpack = net->stem_w; sp.bias = net->stem_b; sp.out = net->x0.p; sp.N = N; sp.H = H; sp.W = W; sp.relu = 0;
        sp.tiles_x = sp.tiles_y = 0;
        HIP_OK(cerb_launch_stem(sp, st));
        if (bn_train(net, "stem", net->x0.p, nullptr, 0, (long long)N * H * W, 1, st)) return 1;
        HIP_OK(cerb_launch_maxpool(net->x0.p, net->pool.p, N, H, W, 64, st));
    }
    float* cur = net->pool.p;
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = kFilters[li + 1];
        const int Hi = (li == 0) ? hs[1] : hs[li], Wi = (li == 0) ? ws[1] : ws[li];
        for (int b = 0; b < kLayers[li]; ++b) {
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int hin = (b == 0) ? Hi : hs[li + 1], win = (b == 0) ? Wi : ws[li + 1];
            const long long rows_out = (long long)N * hs[li + 1] * ws[li + 1];
            float* t1 = net->ta.p;
            float* outb = ((kLayers[li] - 1 - b) % 2 == 0) ? net->x[li + 1].p : net->tb.p;
            const float* idt = cur;
            if (run_conv(net, p + ".conv1", cur, nullptr, nullptr, t1, N, hin, win, 0, 0, 0, 0, st, nullptr)) return 1;
            if (bn_train(net, p + ".conv1", t1, nullptr, 0, rows_out, 1, st)) return 1;
            if (stride != 1 || inpl != planes) {
                // the identity branch needs its own buffer here: in layer1.0 there is none, later the pool buffer is free but smaller maps fit
                if (net->t_idn.ensure((size_t)rows_out * planes * 4, guard)) return fail("workspace allocation failed");
                if (run_conv(net, p + ".downsample", cur, nullptr, nullptr, net->t_idn.p, N, hin, win, 0, 0, 0, 0, st, nullptr)) return 1;
                if (bn_train(net, p + ".downsample", net->t_idn.p, nullptr, 0, rows_out, 0, st)) return 1;
                idt = net->t_idn.p;
            }
            if (run_conv(net, p + ".conv2", t1, nullptr, nullptr, outb, N, hs[li + 1], ws[li + 1], 0, 0, 0, 0, st, nullptr)) return 1;
            if (bn_train(net, p + ".conv2", outb, idt, 0, rows_out, 1, st)) return 1;  // relu(bn2(conv2) + identity)
            cur = outb;
            inpl = planes;
        }
    }
    if (run_conv(net, "conv_map", net->x[4].p, nullptr, nullptr, net->cm.p, N, hs[4], ws[4], 0, 0, 0, 0, st, nullptr)) return 1;
    // ---- Patch-Class: crop -> GAP -> BN -> ReLU -> dropout -> 1x1 -> BN -> ReLU -> 1x1 (models/net_desc.py:64-76,169-180) -----------------
    if (net->pc_idx >= 0 && io->logits[net->pc_idx]) {
        const int ocp = net->dec[net->pc_idx].out_ch;
        int y0 = 0, x0 = 0, ch = hs[4], cw = ws[4];
        if (hs[4] != 9 && ws[4] != 9) {  // cropping_center as a Python slice (negative start wraps, stop clipped): see patch_class_kernel
            auto py_slice = [](int len, int& start, int& count) {
                const int h0 = (int)((len - 9) * 0.5);
                const int a0 = h0 < 0 ? std::max(len + h0, 0) : std::min(h0, len);
                const int a1 = std::min(h0 + 9, len);
                start = a0;
                count = std::max(a1 - a0, 0);
            };
            py_slice(hs[4], y0, ch);
            py_slice(ws[4], x0, cw);
        }
        if (ch <= 0 || cw <= 0) return fail("cerb_net_forward_train: empty Patch-Class crop");
        if (net->t_gap.ensure((size_t)N * 512 * 4, 0) || net->t_pc1.ensure((size_t)N * 256 * 4, 0)) return fail("workspace allocation failed");
        HIP_OK(cerb_launch_crop_gap(net->x[4].p, N, hs[4], ws[4], 512, y0, ch, x0, cw, net->t_gap.p, st));
        if (bn_train(net, "pc.bn1", net->t_gap.p, nullptr, 0, N, 1, st)) return 1;
        HIP_OK(cerb_launch_pointwise(net->t_gap.p, net->pc_rw1, net->pc_rb1, net->t_pc1.p, N, 512, 256, io->dropout_scale, st));
        if (bn_train(net, "pc.bn2", net->t_pc1.p, nullptr, 0, N, 1, st)) return 1;
        HIP_OK(cerb_launch_pointwise(net->t_pc1.p, net->pc_rw2, net->pc_rb2, io->logits[net->pc_idx], N, 256, ocp, nullptr, st));
    }
    // ---- dense decoders (grouped) and heads ---------------------------------------------------------------------------------------
    if (D) {
        const float* skips[4] = {net->x[3].p, net->x[2].p, net->x[1].p, net->x0.p};
        const float* prev = net->cm.p;
        long long prev_gs = 0;
        for (int u = 0; u < 4; ++u) {
            const int hh = hs[3 - u], ww = ws[3 - u];
            const std::string n0 = "dec." + std::to_string(u) + ".0", n1 = "dec." + std::to_string(u) + ".1";
            const int cmid = net->conv[n0].cout, cin0 = net->conv[n0].cin;
            const long long rows = (long long)N * hh * ww;
            HIP_OK(cerb_launch_upsample2_add(skips[u], prev, net->dsum.p, (int)D, N, hh, ww, cin0, prev_gs, nullptr, st));
            if (run_conv(net, n0, net->dsum.p, nullptr, nullptr, net->dmid.p, N, hh, ww, 0, 0, rows * cin0, 0, st, nullptr)) return 1;
            if (bn_train(net, n0, net->dmid.p, nullptr, rows * cmid, rows, 1, st)) return 1;
            if (run_conv(net, n1, net->dmid.p, nullptr, nullptr, net->dout[u].p, N, hh, ww, 0, 0, rows * cmid, 0, st, nullptr)) return 1;
            if (bn_train(net, n1, net->dout[u].p, nullptr, rows * oc[u], rows, 1, st)) return 1;
            prev = net->dout[u].p;
            prev_gs = rows * oc[u];
        }
        const long long rows = (long long)N * H * W;
        if (net->t_hid.ensure((size_t)rows * 96 * 4, 0)) return fail("workspace allocation failed");
        for (size_t k = 0; k < D; ++k) {
            const int di = net->dense_idx[k];
            if (!io->logits[di]) continue;
            HIP_OK(cerb_launch_pointwise(net->dout[3].p + k * (size_t)rows * 64, net->head_rw1[k], net->head_rb1[k], net->t_hid.p, rows, 64, 96, nullptr, st));
            if (bn_train(net, "head." + std::to_string(k), net->t_hid.p, nullptr, 0, rows, 1, st)) return 1;
            HIP_OK(cerb_launch_pointwise(net->t_hid.p, net->head_rw2[k], net->head_rb2[k], io->logits[di], rows, 96, net->dec[di].out_ch, nullptr, st));
        }
    }
    net->conv_algo = saved_algo;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One training step's gradients (models/run_desc.py:79-170: train-mode forward, the six head losses, all_loss.backward()).
// FIRST VERSION: the forward runs on the production kernels, the backward on plain gather kernels (train_kernels.hip) -- correct
// and reproducible, not fast.  The forward is recorded on a tape (every op keeps its input and output), the backward walks it in
// reverse; gradients accumulate with += into zeroed buffers, so tensors with several consumers (skips, residual identities, the
// shared conv_map output) need no special casing.  Gradients are published per state-dict key (cerb_net_grad_lookup).
struct TapeOp {
    int type = 0;  // 0 stem, 1 conv, 2 bn, 3 maxpool, 4 upadd, 5 pointwise, 6 crop+gap, 7 a whole output head (head_train.hip)
    int hid = -1, head_k = 0;          // type 7: the stored 96-channel hidden map (a = the grouped decoder tensor, o = the logits, stat = [mean | rstd])
    std::string wkey2, bkey2;          // type 7: keys of the second pointwise layer
    int in_stat = -1;                  // type 7: [mean | rstd] tensor of a BatchNorm applied on the head's LOAD of `a` (a = that BatchNorm's raw input); -1: a is normalised
    std::string in_bn;                 // ... and its name (gamma / beta, group = head_k)
    int deferred = 0;                  // type 2: the normalised output was never written (o aliases a): its consumers apply the BatchNorm themselves
    std::string name;
    int a = -1, b = -1, o = -1;        // tensor ids: input, second input (residual / prev), output
    int N = 0, H = 0, W = 0, Cin = 0, Cout = 0, ks = 0, stride = 1, G = 1, relu = 0;
    long long a_gs = 0, o_gs = 0, b_gs = 0, rows = 0;
    int stat = -1;                     // bn: tensor id holding [mean | rstd]
    const float *w = nullptr, *bias = nullptr, *scale = nullptr;
    int y0 = 0, ch = 0, x0 = 0, cw = 0;
    std::string wkey, bkey;            // pointwise: state-dict keys of its weight / bias
};

extern "C" int cerb_net_train_grads(cerb_net* net, const cerb_train_step_io* io, void* hip_stream) {
    if (!net || !io || !io->tiles || !io->target || !io->has_target || !io->loss_out) return fail("cerb_net_train_grads: null argument");
    if (!net->finalized || net->fold_bn) return fail("cerb_net_train_grads: needs a network packed with cerb_net_set_fold_bn(net, 0)");
    hipStream_t st = (hipStream_t)hip_stream;
    g_call_stream = st;
    const int N = io->n, H = io->h, W = io->w;
    if (N <= 0 || H <= 0 || W <= 0 || (H % 16) || (W % 16)) return fail("cerb_net_train_grads: tile H,W must be positive multiples of 16");
    const int hs[5] = {H, H / 2, H / 4, H / 8, H / 16}, ws[5] = {W, W / 2, W / 4, W / 8, W / 16};
    const size_t D = net->dense_idx.size();
    const size_t guard = cerb_conv_guard_bytes(W);
    net->tape_pos = 0;
    net->prof_n = 0;  // per-launch records of this step (cerb_net_profile_*): the forward convs through run_conv, plus the backward families below
    std::vector<float*> val, grd;
    std::vector<size_t> cnt;
    auto take = [&](size_t nfloat, bool zero) -> float* {  // next buffer of the tape arena (kept across steps)
        if (net->tape_pos == net->tape.size()) net->tape.emplace_back();
        DevBuf& b = net->tape[net->tape_pos++];
        if (b.ensure(nfloat * 4, guard)) return nullptr;
        if (zero) {  // its own profile record unless a family's record is open (then the fill is that family's)
            const bool own = net->profiling && !net->prof_open;
            if (own && prof_begin(net, "zero_fill", "zero_fill", nfloat * 4.0, st)) return nullptr;
            if (hipMemsetAsync(b.p, 0, nfloat * 4, st) != hipSuccess) return nullptr;
            if (own && prof_end(net, st)) return nullptr;
        }
        return b.p;
    };
    auto newT = [&](size_t nfloat) {
        val.push_back(take(nfloat, false));
        grd.push_back(nullptr);
        cnt.push_back(nfloat);
        return (int)val.size() - 1;
    };
    // A grouped tensor whose gradient arrives slice by slice (the decoders' last maps: one 1x1 head per group reads its own slice): the buffer is
    // made WITHOUT a zero fill, the first writer of a slice assigns, and the slices nobody wrote (decoders without a target) are zeroed just
    // before the tensor's producer reads the gradient -- instead of 4 GB of fill plus a read-modify-write per slice.
    std::map<int, unsigned long long> slice_written;  // tensor -> bit k = slice k holds a gradient
    std::map<int, std::pair<int, size_t>> slice_geom;    // tensor -> (slices, floats per slice)
    auto G_ = [&](int t) -> float* {  // gradient buffer of tensor t, created zeroed on first use
        if (!grd[t]) grd[t] = take(cnt[t], true);
        auto sw = slice_written.find(t);
        if (sw != slice_written.end()) {  // a whole-tensor writer arrives while slices are still unwritten: they must read as zero from here on
            const std::pair<int, size_t> ge = slice_geom[t];
            for (int k = 0; k < ge.first; ++k)
                if (!((sw->second >> k) & 1ull) && hipMemsetAsync(grd[t] + (size_t)k * ge.second, 0, ge.second * 4, st) != hipSuccess) return nullptr;
            slice_written.erase(sw);
        }
        return grd[t];
    };
    std::vector<TapeOp> tape;
    // BatchNorm backward sums out of the data gradient that produced the BatchNorm's output gradient (conv_wino4 / conv_wino4b STATS 2): [mean | rstd] tensor of
    // the BatchNorm -> (partials, blocks per group).  CERB_BN_BWD_PASS1=1 keeps the BatchNorm's own reduction pass (developer A/B).
    std::map<int, std::pair<double*, int>> bst_part;
    const bool bst_on = getenv("CERB_BN_BWD_PASS1") == nullptr;
    std::map<int, std::pair<double*, bool>> deferred_part;  // [mean | rstd] tensor of a deferred BatchNorm -> (its backward partials from the heads, still complete?)
    net->grads.clear();
    auto pub = [&](const std::string& key, size_t n) -> float* {  // a published parameter gradient
        float* p = take(n, true);
        net->grads[key] = std::make_pair(p, (long long)n);
        return p;
    };
    const int saved_algo = net->conv_algo;
    // side stream of the weight gradients (see cerb_net::side): off under per-launch profiling (the records time one stream) and with CERB_WGRAD_SIDE=0
    bool side_wgrad = !net->profiling;
    {
        const char* e = getenv("CERB_WGRAD_SIDE");
        if (e && e[0] == '0') side_wgrad = false;
    }
    bool side_used = false;
    if (side_wgrad && !net->side) {
        HIP_OK(hipStreamCreateWithFlags(&net->side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&net->ev_join, hipEventDisableTiming));
    }
    if (net->side) {  // whatever an earlier call left on the side stream (a call that failed half way) is complete before this call's tape is written
        HIP_OK(hipEventRecord(net->ev_join, net->side));
        HIP_OK(hipStreamWaitEvent(st, net->ev_join, 0));
    }
    std::map<int, std::pair<double*, int>> conv_stats;  // conv output tensor -> (statistics partials, blocks per group)
    const bool bn_stats_pass = getenv("CERB_BN_STATS_PASS") != nullptr;  // developer A/B: the separate statistics pass (read once per step)
    // ---------------------------------------------------------------- forward, recorded ----------------------------------------
    auto conv = [&](const std::string& name, int a, int n_, int h_, int w_, long long a_gs) -> int {
        const PackedConv& c = net->conv[name];
        const int ho = c.stride == 2 ? h_ / 2 : h_, wo = c.stride == 2 ? w_ / 2 : w_;
        const int o = newT((size_t)c.groups * n_ * ho * wo * c.cout);
        // BatchNorm statistics partials from the convolution's own output stage (3x3 stride-1 layers on the F(4x4) kernels): one (sum, sum of
        // squares) per 16 x 16 block, group and channel, in a buffer of the tape arena that the bn() behind this conv finalises
        double* part = nullptr;
        // only where run_conv will pick an F(4x4) kernel (its own rule: maps of at least 16 x 16 pixels under the default algorithm) -- the F(2x2) and
        // direct kernels have no statistics stage, a buffer taken for them would only be arena churn
        const long long map_px = (long long)ho * wo;
        const bool f4 = net->conv_algo == 5 || net->conv_algo == 7 || (net->conv_algo == 6 && map_px >= 256);
        if (c.ks == 3 && c.stride == 1 && f4 && c.wino && !bn_stats_pass) {
            const size_t nblk = (size_t)n_ * ((ho + 15) / 16) * ((wo + 15) / 16);
            part = (double*)take((size_t)c.groups * nblk * c.cout * 2 * 2, false);
            if (!part) return -1;
            net->conv_bn_part = part;
        }
        if (!val[o] || run_conv(net, name, val[a], nullptr, nullptr, val[o], n_, h_, w_, 0, 0, a_gs, 0, st, nullptr)) return -1;
        if (part && net->conv_bn_bpg > 0) conv_stats[o] = std::make_pair(part, net->conv_bn_bpg);
        TapeOp op;
        op.type = 1; op.name = name; op.a = a; op.o = o; op.N = n_; op.H = h_; op.W = w_; op.Cin = c.cin; op.Cout = c.cout; op.ks = c.ks; op.stride = c.stride;
        op.G = c.groups; op.a_gs = a_gs;
        tape.push_back(op);
        return o;
    };
    // pre_blocks > 0: the producer of y already left pre_blocks rows of statistics partials in net->t_ws (one group): no statistics pass over y
    // stat_only != nullptr: batch statistics only (published as usual) -- *stat_only = the [mean | rstd] tensor, no normalised copy of y is
    // made and no tape entry (the fused heads apply the normalisation inside their own kernels); returns y
    // defer_stat != nullptr: the statistics are taken and the tape entry is made as usual, but the normalised tensor is NOT written -- the returned tensor
    // id aliases y's values (its own gradient buffer), *defer_stat = the [mean | rstd] tensor, and every consumer applies relu(bn(.)) on its loads
    // (the fused heads behind the last decoder level: head_fwd1 / head_bwd2; the BatchNorm's backward reads y only -- relu mode 2 -- so it does not care)
    auto bn = [&](const std::string& name, int y, int resid, long long rows, int relu, int pre_blocks = 0, int* stat_only = nullptr, int* defer_stat = nullptr) -> int {
        const cerb_net::BnDev& b = net->bn[name];
        int z;
        if (stat_only) z = y;
        else if (defer_stat) {
            val.push_back(val[y]);
            grd.push_back(nullptr);
            cnt.push_back(cnt[y]);
            z = (int)val.size() - 1;
        } else z = newT(cnt[y]);
        const int stt = newT((size_t)2 * b.groups * b.C);
        if (!val[z] || !val[stt] || (!pre_blocks && net->t_ws.ensure(cerb_bn_workspace_bytes(b.groups, rows, b.C), 0))) return -1;
        float* mean = val[stt];
        float* rstd = val[stt] + (size_t)b.groups * b.C;
        const long long gs = b.groups > 1 ? rows * b.C : 0;
        float* var_u = take((size_t)b.groups * b.C, false);  // unbiased batch variance: what the running_var update uses
        // `flops` field = algorithmic bytes of the two forward BatchNorm passes (statistics: read y; apply: read y (+ residual), write z)
        if (prof_begin(net, name + ".bn_fwd", (stat_only || defer_stat) ? "bn_finalize" : "bn_fwd", (stat_only || defer_stat) ? (double)(pre_blocks > 0 ? pre_blocks : 256) * b.C * 16.0 : (double)b.groups * rows * b.C * 4.0 * (3.0 + (resid >= 0 ? 1.0 : 0.0)), st)) return -1;
        if (!var_u) return -1;
        auto cs = conv_stats.find(y);
        if (cs != conv_stats.end()) {  // the convolution that made y left the partials: [groups][blocks][C][2]
            if (net->t_ws.ensure(cerb_bn_fold_workspace_bytes(b.groups, b.C), 0)) return -1;
            if (cerb_launch_bn_finalize(cs->second.first, cs->second.second, rows, b.C, 1e-5f, mean, rstd, var_u, st, b.groups, net->t_ws.p) != hipSuccess) return -1;
        } else if (pre_blocks > 0 && b.groups == 1) {
            // (the producer's partial rows sit at the front of t_ws -- at most 2048 of them; rows beyond 4096 x C x 16 bytes serve as the fold area)
            if (cerb_launch_bn_finalize((const double*)net->t_ws.p, pre_blocks, rows, b.C, 1e-5f, mean, rstd, var_u, st, 1,
                                        net->t_ws.bytes >= (size_t)4352 * b.C * 16 ? (char*)net->t_ws.p + (size_t)4096 * b.C * 16 : nullptr) != hipSuccess) return -1;
        } else if (cerb_launch_bn_stats(val[y], gs, rows, b.C, b.groups, 1e-5f, mean, rstd, var_u, net->t_ws.p, st) != hipSuccess) return -1;
        if (bn_eval_override(b, mean, rstd, st)) return -1;
        {
            const std::vector<std::string>& keys = net->bn_keys[name];
            for (int g = 0; g < b.groups; ++g) {
                if (g < (int)b.eval.size() && b.eval[g]) continue;  // eval mode: running statistics are not updated (no batch statistics published)
                net->grads[keys[g] + ".batch_mean"] = std::make_pair(mean + (size_t)g * b.C, (long long)b.C);
                net->grads[keys[g] + ".batch_var"] = std::make_pair(var_u + (size_t)g * b.C, (long long)b.C);
            }
        }
        if (stat_only) {
            if (prof_end(net, st)) return -1;
            *stat_only = stt;
            return y;
        }
        if (defer_stat) {
            if (prof_end(net, st)) return -1;
            *defer_stat = stt;
            TapeOp op;
            op.type = 2; op.name = name; op.a = y; op.b = resid; op.o = z; op.stat = stt; op.rows = rows; op.Cout = b.C; op.G = b.groups; op.relu = relu; op.a_gs = gs; op.deferred = 1;
            tape.push_back(op);
            return z;
        }
        if (cerb_launch_bn_apply(val[z], val[y], resid >= 0 ? val[resid] : nullptr, gs, rows, b.C, b.groups, mean, rstd, b.gamma, b.beta, relu, st) != hipSuccess) return -1;
        if (prof_end(net, st)) return -1;
        TapeOp op;
        op.type = 2; op.name = name; op.a = y; op.b = resid; op.o = z; op.stat = stt; op.rows = rows; op.Cout = b.C; op.G = b.groups; op.relu = relu; op.a_gs = gs;
        tape.push_back(op);
        return z;
    };
#define TCHK(x) do { if ((x) < 0) { net->conv_algo = saved_algo; return fail(std::string("cerb_net_train_grads: ") + #x + " failed"); } } while (0)
// one per-launch profile record (cerb_net_profile_*) around a launch of the families that run_conv / bn / wgrad do not cover themselves, so that a
// profiled step attributes ALL of its device time; `work` = FLOPs of a matrix-core family, algorithmic bytes of an HBM-bound one
#define PROF(nm, kern, work, stmt) do { if (prof_begin(net, (nm), (kern), (work), st)) return 1; stmt; if (prof_end(net, st)) return 1; } while (0)
#define PROFN(nm, kern, work, stmt) do { if (prof_begin(net, (nm), (kern), (work), st)) return -1; stmt; if (prof_end(net, st)) return -1; } while (0)
    const int t_stem = newT((size_t)N * H * W * 64);
    {
        StemParams sp;
        sp.tiles = io->tiles; sp.tiles_f32 = nullptr; sp.wpack = net->stem_w; sp.bias = net->stem_b; sp.out = val[t_stem]; sp.N = N; sp.H = H; sp.W = W; sp.relu = 0;
        sp.tiles_x = sp.tiles_y = 0;
        PROF("stem", "stem_conv7x7", 2.0 * N * H * W * 64.0 * 147.0, HIP_OK(cerb_launch_stem(sp, st)));
        TapeOp op;
        op.type = 0; op.o = t_stem; op.N = N; op.H = H; op.W = W;
        tape.push_back(op);
    }
    const int x0 = bn("stem", t_stem, -1, (long long)N * H * W, 1);
    TCHK(x0);
    const int pool = newT((size_t)N * hs[1] * ws[1] * 64);
    // the pooling records the position of every window's first maximum (one byte per element) and the backward pass routes by it;
    // CERB_MAXPOOL_SCAN=1 keeps round 4's backward that re-finds the maxima from the input and the pooled map (developer A/B: identical bits)
    const bool pool_scan = getenv("CERB_MAXPOOL_SCAN") != nullptr;  // read once per step
    const int pool_idx = pool_scan ? -1 : newT(((size_t)N * hs[1] * ws[1] * 64 + 3) / 4);
    if (pool_idx >= 0) {
        if (!val[pool_idx]) { net->conv_algo = saved_algo; return fail("workspace allocation failed"); }
        PROF("maxpool", "maxpool3x3s2", (double)N * H * W * 64 * 4.0 * 1.25 + (double)N * hs[1] * ws[1] * 64.0,
             HIP_OK(cerb_launch_maxpool_idx(val[x0], val[pool], reinterpret_cast<unsigned*>(val[pool_idx]), N, H, W, 64, st)));
    } else {
        PROF("maxpool", "maxpool3x3s2", (double)N * H * W * 64 * 4.0 * 1.25, HIP_OK(cerb_launch_maxpool(val[x0], val[pool], N, H, W, 64, st)));
    }
    {
        TapeOp op;
        op.type = 3; op.a = x0; op.o = pool; op.b = pool_idx; op.N = N; op.H = H; op.W = W; op.Cout = 64;
        tape.push_back(op);
    }
    int cur = pool, inpl = 64, xs[5] = {x0, -1, -1, -1, -1};
    for (int li = 0; li < 4; ++li) {
        const int planes = kFilters[li + 1];
        const int Hi = (li == 0) ? hs[1] : hs[li], Wi = (li == 0) ? ws[1] : ws[li];
        for (int b = 0; b < kLayers[li]; ++b) {
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int hin = (b == 0) ? Hi : hs[li + 1], win = (b == 0) ? Wi : ws[li + 1];
            const long long rows_out = (long long)N * hs[li + 1] * ws[li + 1];
            int idt = cur;
            const int c1 = conv(p + ".conv1", cur, N, hin, win, 0);
            TCHK(c1);
            const int z1 = bn(p + ".conv1", c1, -1, rows_out, 1);
            TCHK(z1);
            if (stride != 1 || inpl != planes) {
                const int d = conv(p + ".downsample", cur, N, hin, win, 0);
                TCHK(d);
                idt = bn(p + ".downsample", d, -1, rows_out, 0);
                TCHK(idt);
            }
            const int c2 = conv(p + ".conv2", z1, N, hs[li + 1], ws[li + 1], 0);
            TCHK(c2);
            cur = bn(p + ".conv2", c2, idt, rows_out, 1);
            TCHK(cur);
            inpl = planes;
        }
        xs[li + 1] = cur;
    }
    const int cm = conv("conv_map", xs[4], N, hs[4], ws[4], 0);
    TCHK(cm);
    std::vector<int> logit_t(net->dec.size(), -1);
    // Patch-Class
    if (net->pc_idx >= 0) {
        const int ocp = net->dec[net->pc_idx].out_ch;
        int y0 = 0, x0c = 0, ch = hs[4], cw = ws[4];
        if (hs[4] != 9 && ws[4] != 9) {
            auto py_slice = [](int len, int& start, int& count) {
                const int h0 = (int)((len - 9) * 0.5);
                const int a0 = h0 < 0 ? std::max(len + h0, 0) : std::min(h0, len);
                const int a1 = std::min(h0 + 9, len);
                start = a0;
                count = std::max(a1 - a0, 0);
            };
            py_slice(hs[4], y0, ch);
            py_slice(ws[4], x0c, cw);
        }
        const int gap = newT((size_t)N * 512);
        PROF("pc.crop_gap", "crop_gap", (double)N * ch * cw * 512 * 4.0, HIP_OK(cerb_launch_crop_gap(val[xs[4]], N, hs[4], ws[4], 512, y0, ch, x0c, cw, val[gap], st)));
        {
            TapeOp op;
            op.type = 6; op.a = xs[4]; op.o = gap; op.N = N; op.H = hs[4]; op.W = ws[4]; op.Cout = 512; op.y0 = y0; op.ch = ch; op.x0 = x0c; op.cw = cw;
            tape.push_back(op);
        }
        const int g1 = bn("pc.bn1", gap, -1, N, 1);
        TCHK(g1);
        auto pw = [&](int a, const float* w, const float* bias, long long rows, int cin, int cout, const float* scale, const std::string& wk, const std::string& bk) {
            const int o = newT((size_t)rows * cout);
            if (!val[o]) return -1;
            PROFN(wk, "pointwise_fwd", 2.0 * rows * cin * cout, if (cerb_launch_pointwise(val[a], w, bias, val[o], rows, cin, cout, scale, st) != hipSuccess) return -1);
            TapeOp op;
            op.type = 5; op.a = a; op.o = o; op.rows = rows; op.Cin = cin; op.Cout = cout; op.w = w; op.bias = bias; op.scale = scale; op.wkey = wk; op.bkey = bk;
            tape.push_back(op);
            return o;
        };
        const std::string pp = "decoder_head.Patch-Class";
        const int h1 = pw(g1, net->pc_rw1, net->pc_rb1, N, 512, 256, io->dropout_scale, pp + ".conv1.weight", pp + ".conv1.bias");
        TCHK(h1);
        const int h2 = bn("pc.bn2", h1, -1, N, 1);
        TCHK(h2);
        logit_t[net->pc_idx] = pw(h2, net->pc_rw2, net->pc_rb2, N, 256, ocp, nullptr, pp + ".conv2.weight", pp + ".conv2.bias");
        TCHK(logit_t[net->pc_idx]);
    }
    if (D) {
        const int skips[4] = {xs[3], xs[2], xs[1], xs[0]};
        const int oc[4] = {128, 64, 64, 64};
        int prev = cm;
        long long prev_gs = 0;
        int last_stat = -1;     // >= 0: the last level's BatchNorm was deferred to the heads ([mean | rstd] tensor)
        std::string last_bn;
        for (int u = 0; u < 4; ++u) {
            const int hh = hs[3 - u], ww = ws[3 - u];
            const std::string n0 = "dec." + std::to_string(u) + ".0", n1 = "dec." + std::to_string(u) + ".1";
            const int cin0 = net->conv[n0].cin;
            const long long rows = (long long)N * hh * ww;
            const int dsum = newT((size_t)D * rows * cin0);
            PROF(n0 + ".up", "upsample2_add", (double)rows * cin0 * 4.0 * (1.0 + D * 1.25), HIP_OK(cerb_launch_upsample2_add(val[skips[u]], val[prev], val[dsum], (int)D, N, hh, ww, cin0, prev_gs, nullptr, st)));
            {
                TapeOp op;
                op.type = 4; op.a = skips[u]; op.b = prev; op.o = dsum; op.N = N; op.H = hh; op.W = ww; op.Cout = cin0; op.G = (int)D; op.b_gs = prev_gs;
                tape.push_back(op);
            }
            const int c0 = conv(n0, dsum, N, hh, ww, rows * cin0);
            TCHK(c0);
            const int z0 = bn(n0, c0, -1, rows, 1);
            TCHK(z0);
            const int c1 = conv(n1, z0, N, hh, ww, rows * net->conv[n0].cout);
            TCHK(c1);
            // last level: its output feeds the heads only -- when they all run fused (head_train.hip) they normalise on their loads
            bool defer = u == 3 && net->conv_algo && getenv("CERB_HEAD_UNFUSED") == nullptr && getenv("CERB_HEAD_BN_APPLY_PASS") == nullptr && net->bn[n1].eval.empty();
            for (size_t k = 0; defer && k < D; ++k) defer = cerb_head_train_supported(rows, 64, 96, net->dec[net->dense_idx[k]].out_ch);
            if (defer) {
                prev = bn(n1, c1, -1, rows, 1, 0, nullptr, &last_stat);
                last_bn = n1;
            } else prev = bn(n1, c1, -1, rows, 1);
            TCHK(prev);
            prev_gs = rows * oc[u];
        }
        const long long rows = (long long)N * H * W;
        for (size_t k = 0; k < D; ++k) {
            const int di = net->dense_idx[k];
            const DecoderCfg& d = net->dec[di];
            const std::string p = "output_head." + d.name + "." + d.head + ".x";
            // the head reads decoder k's slice of the grouped tensor: a view (tensor id with its own grad slice) is the slice itself
            const int hid = newT((size_t)rows * 96);
            // the hidden map's BatchNorm statistics come out of the layer itself (per-wave partials in t_ws, sized for either way before the launch)
            int pre_blocks = 0;
            if (net->t_ws.ensure(std::max(cerb_bn_workspace_bytes(1, rows, 96), (size_t)8192 * 96 * 16), 0)) return fail("workspace allocation failed");
            const bool heads_fused = net->conv_algo && getenv("CERB_HEAD_UNFUSED") == nullptr && cerb_head_train_supported(rows, 64, 96, d.out_ch);
            const float* in_bn[4] = {nullptr, nullptr, nullptr, nullptr};
            if (last_stat >= 0) {  // group k's statistics and affine parameters of the deferred BatchNorm
                const cerb_net::BnDev& lb = net->bn[last_bn];
                in_bn[0] = val[last_stat] + k * (size_t)lb.C;
                in_bn[1] = val[last_stat] + (size_t)lb.groups * lb.C + k * (size_t)lb.C;
                in_bn[2] = lb.gamma + k * (size_t)lb.C;
                in_bn[3] = lb.beta + k * (size_t)lb.C;
                if (!heads_fused) return fail("internal: deferred BatchNorm in front of an unfused head");
            }
            if (heads_fused)
                PROF(p + ".0", "head_fwd1", (double)rows * (64 + 96) * 4.0,
                     HIP_OK(cerb_launch_head_fwd1(val[prev] + k * (size_t)rows * 64, net->head_rw1[k], net->head_rb1[k], val[hid], rows, (double*)net->t_ws.p, &pre_blocks, st,
                                                  last_stat >= 0 ? in_bn : nullptr)));
            else
            PROF(p + ".0", "pointwise_fwd", 2.0 * rows * 64 * 96, HIP_OK(cerb_launch_pointwise(val[prev] + k * (size_t)rows * 64, net->head_rw1[k], net->head_rb1[k], val[hid], rows, 64, 96, nullptr, st,
                                                                                          net->conv_algo ? (double*)net->t_ws.p : nullptr, &pre_blocks)));
            // The head as ONE tape entry (head_train.hip): the hidden map is stored once and read three times (forward 2, backward 1, backward 2);
            // its normalised copy and both gradients of the hidden layer never exist.  CERB_HEAD_UNFUSED=1 keeps round 4's separate passes (A/B, tests).
            if (heads_fused) {  // (CERB_HEAD_UNFUSED is read per step: the A/B test flips it inside one process)
                const std::string bname = "head." + std::to_string(k);
                int stt = -1;
                TCHK(bn(bname, hid, -1, rows, 1, pre_blocks, &stt));
                const cerb_net::BnDev& hb = net->bn[bname];
                const int lg = newT((size_t)rows * d.out_ch);
                if (!val[lg]) return fail("workspace allocation failed");
                PROF(p + ".1", "head_fwd2", (double)rows * (96 + d.out_ch) * 4.0,
                     HIP_OK(cerb_launch_head_fwd2(val[hid], val[stt], val[stt] + 96, hb.gamma, hb.beta, net->head_rw2[k], net->head_rb2[k], val[lg], rows, d.out_ch, st)));
                TapeOp op;
                op.type = 7; op.name = bname; op.a = prev; op.o = lg; op.hid = hid; op.stat = stt; op.head_k = (int)k; op.rows = rows; op.Cin = 64; op.Cout = d.out_ch;
                op.a_gs = (long long)k * rows * 64;
                op.wkey = p + ".0.block.0.conv.weight"; op.bkey = p + ".0.block.0.conv.bias";
                op.wkey2 = p + ".1.conv.weight"; op.bkey2 = p + ".1.conv.bias";
                op.in_stat = last_stat; op.in_bn = last_bn;
                tape.push_back(op);
                logit_t[di] = lg;
                continue;
            }
            {
                TapeOp op;
                op.type = 5; op.a = prev; op.o = hid; op.rows = rows; op.Cin = 64; op.Cout = 96; op.w = net->head_rw1[k]; op.bias = net->head_rb1[k];
                op.a_gs = (long long)k * rows * 64;  // offset of the slice inside tensor a
                op.wkey = p + ".0.block.0.conv.weight"; op.bkey = p + ".0.block.0.conv.bias";
                tape.push_back(op);
            }
            const int hz = bn("head." + std::to_string(k), hid, -1, rows, 1, pre_blocks);
            TCHK(hz);
            const int lg = newT((size_t)rows * d.out_ch);
            PROF(p + ".1", "pointwise_fwd", 2.0 * rows * 96 * d.out_ch, HIP_OK(cerb_launch_pointwise(val[hz], net->head_rw2[k], net->head_rb2[k], val[lg], rows, 96, d.out_ch, nullptr, st)));
            {
                TapeOp op;
                op.type = 5; op.a = hz; op.o = lg; op.rows = rows; op.Cin = 96; op.Cout = d.out_ch; op.w = net->head_rw2[k]; op.bias = net->head_rb2[k];
                op.wkey = p + ".1.conv.weight"; op.bkey = p + ".1.conv.bias";
                tape.push_back(op);
            }
            logit_t[di] = lg;
        }
    }
    // ---------------------------------------------------------------- losses: d(overall) / d(logits) ------------------------------
    for (size_t di = 0; di < net->dec.size(); ++di) {
        const int lg = logit_t[di];
        if (lg < 0 || !io->target[di]) continue;
        const DecoderCfg& d = net->dec[di];
        const bool pc = (int)di == net->pc_idx;
        const int hh = pc ? 1 : H, ww = pc ? 1 : W, C = d.out_ch;
        if (net->t_hid.ensure(cerb_head_loss_workspace_bytes(N, hh, ww), 0)) return fail("workspace allocation failed");
        // NHWC logits: strides (n, c, y, x) = (h w C, 1, w C, C)
        float* glg = G_(lg);
        PROF("loss." + d.name, "head_loss", (double)N * hh * ww * (C * 8.0 + 8.0),
             if (cerb_head_loss_wmap(val[lg], (long long)hh * ww * C, 1, (long long)ww * C, C, io->target[di], io->has_target[di], N, hh, ww, C,
                                     io->class_weight ? io->class_weight[di] : nullptr, io->pixel_weight ? io->pixel_weight[di] : nullptr, io->ce_w[di], io->dice_w[di],
                                     io->head_w[di], pc ? 1 : 0, io->loss_out + di, glg, net->t_hid.p, cerb_head_loss_workspace_bytes(N, hh, ww), st)) return 1;
             if (io->logits && io->logits[di]) HIP_OK(hipMemcpyAsync(io->logits[di], val[lg], cnt[lg] * 4, hipMemcpyDeviceToDevice, st)));
    }
    // ---------------------------------------------------------------- backward ------------------------------------------------------
    for (int i = (int)tape.size() - 1; i >= 0; --i) {
        const TapeOp& op = tape[i];
        if (!grd[op.o]) continue;  // nothing flowed into this output (a head without target)
        float* go = grd[op.o];
        {
            auto sw = slice_written.find(op.o);
            if (sw != slice_written.end()) {  // the slices no head wrote read as zero
                const std::pair<int, size_t> ge = slice_geom[op.o];
                for (int k = 0; k < ge.first; ++k)
                    if (!((sw->second >> k) & 1ull)) HIP_OK(hipMemsetAsync(go + (size_t)k * ge.second, 0, ge.second * 4, st));
                slice_written.erase(sw);
            }
        }
        switch (op.type) {
            case 0: {  // stem: weight gradient only
                if (net->t_ws.ensure(cerb_stem_wgrad_workspace_bytes(), 0)) return fail("workspace allocation failed");
                float* dws = pub("backbone.conv1.weight", 64 * 147);
                PROF("stem.wgrad", "stem_wgrad", 2.0 * N * H * W * 64.0 * 147.0, HIP_OK(cerb_launch_stem_wgrad_mfma(io->tiles, go, dws, N, H, W, net->t_ws.p, st)));
                break;
            }
            case 1: {
                const cerb_net::RawW& r = net->raw[op.name];
                const size_t wn = (size_t)op.Cout * op.Cin * op.ks * op.ks;
                // which weight-gradient kernel this layer takes, decided up front: its workspace is sized and the side stream forked BEFORE the data gradient is queued
                const bool wg_wino = op.ks == 3 && op.stride == 1 && net->conv_algo >= 5 && cerb_wgrad_wino_supported(op.H, op.W, op.Cin, op.Cout) && !getenv("CERB_WGRAD_DIRECT");
                const bool wg_mfma = !wg_wino && (op.ks == 3 || op.ks == 1) && net->conv_algo;
                hipStream_t wst = st;        // the stream the weight gradient is queued on
                DevBuf* wws = &net->t_ws;    // ... and its workspace
                // the MFMA weight gradient (wgrad_reduce_kernel) and the bias column sums ASSIGN their outputs: no zero fill (a step issued
                // ~230 of these 18-us memsets: 4 ms).  Taken BEFORE the fork event: a new arena slot queues its zero fill on the
                // caller's stream, and the side stream -- which writes dw / db -- only waits for what that event covers.
                const bool dw_assigned = (op.ks == 3 || op.ks == 1) && net->conv_algo;
                float* dw = take(wn * op.G, !dw_assigned);
                float* db = r.b ? take((size_t)op.Cout * op.G, false) : nullptr;
                if (!dw) return fail("workspace allocation failed");
                if (side_wgrad && (wg_wino || wg_mfma)) {
                    const int ho_ = op.stride == 2 ? op.H / 2 : op.H, wo_ = op.stride == 2 ? op.W / 2 : op.W;
                    const size_t need = wg_wino ? cerb_wgrad_wino_workspace_bytes(op.G, op.N, op.H, op.W, op.Cin, op.Cout) : cerb_wgrad_workspace_bytes(op.G, op.N, ho_, wo_, op.Cin, op.Cout, op.ks, nullptr);
                    if (net->t_ws2.ensure(need, 0)) return fail("workspace allocation failed");
                    HIP_OK(hipEventRecord(net->ev_fork, st));  // the layer's output gradient (and a fresh workspace's fill) is complete on the caller's stream
                    HIP_OK(hipStreamWaitEvent(net->side, net->ev_fork, 0));
                    wst = net->side;
                    wws = &net->t_ws2;
                    side_used = true;
                }
                const PackedConv& pcv = net->conv[op.name];
                bool dx_done = false;
                if (pcv.wino_dgrad && net->conv_algo && (op.stride == 1 || (op.H % 2 == 0 && op.W % 2 == 0))) {
                    // data gradient on the forward Winograd kernel: in = dy, weights rotated + transposed, the gradient already held by the
                    // input (other consumers) rides in as the residual and is written back in place
                    const long long in_n = (long long)op.N * op.H * op.W * op.Cin;
                    const bool fresh = !grd[op.a] && cnt[op.a] == (size_t)op.G * in_n && (op.G == 1 || op.a_gs == in_n);  // first writer: no residual, no zero fill
                    if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                    float* dx = G_(op.a);
                    const long long map_px = (long long)op.H * op.W;
                    const bool d_w4 = saved_algo == 5 || saved_algo == 7 || (saved_algo == 6 && map_px >= 256);
                    static const long long d_w4b_max_px = [] {  // developer A/B only, as in run_conv
                        const char* e = getenv("CERB_W4B_MAX_PX");
                        return e ? atoll(e) : 4096ll;
                    }();
                    const bool d_w4b = (saved_algo == 7 || (saved_algo == 6 && map_px <= d_w4b_max_px)) && op.Cout % 64 == 0;
                    const bool d_f4 = d_w4 && op.Cout % 16 == 0 && op.Cin % 64 == 0;
                    // the data gradient as its own family: the forward Winograd kernels on rotated weights (+ the stride-2 dilation pass)
                    ConvParams q;  // the geometry conv_wino4b's launcher decides its item form by
                    memset(&q, 0, sizeof(q));
                    q.N = op.N; q.H = q.Ho = op.H; q.W = q.Wo = op.W; q.Cin = op.Cout; q.Cout = op.Cin; q.pk_off = net->packed_items ? 0 : 1;
                    const bool d_pk = d_f4 && d_w4b && cerb_wino4b_packed(q);
                    if (prof_begin(net, op.name + ".dgrad", std::string("dgrad:") + (d_f4 ? (d_w4b ? (d_pk ? "conv_wino4b<f4x4,16t>" : "conv_wino4b<f4x4,16x16>") : "conv_wino4<f4x4,16x16x2>") : "conv_wino<f2x2,8x16>"),
                                   2.0 * op.G * op.N * op.H * op.W * (double)op.Cin * op.Cout * 9.0, st)) return 1;
                    if (op.stride == 2) {  // y = 2 yo - 1 + ky  <=>  dx = conv_s1(D, W'), D[2 yo][2 xo] = dy[yo][xo], zero elsewhere
                        const long long dn = (long long)op.G * op.N * op.H * op.W * op.Cout;
                        if (net->t_dil.ensure((size_t)dn * 4, cerb_conv_guard_bytes(W))) return fail("workspace allocation failed");
                        HIP_OK(cerb_launch_dilate2(go, net->t_dil.p, (long long)op.G * op.N, op.H, op.W, op.Cout, st));
                        go = net->t_dil.p;
                    }
                    ConvParams p;
                    memset(&p, 0, sizeof(p));
                    p.in = go; p.wpack = pcv.wino_dgrad; p.bias = net->zero_bias; p.resid = fresh ? nullptr : dx; p.out = dx;
                    p.N = op.N; p.H = op.H; p.W = op.W; p.Cin = op.Cout; p.Cout = op.Cin; p.Ho = op.H; p.Wo = op.W; p.relu = 0; p.groups = op.G;
                    p.in_gs = (long long)op.N * op.H * op.W * op.Cout;
                    p.w_gs = (long long)op.Cout * op.Cin * 16;
                    p.bias_gs = 0;
                    p.resid_gs = op.a_gs;
                    p.out_gs = op.a_gs;
                    if (op.G == 1) p.resid_gs = p.out_gs = 0;
                    p.pk_off = net->packed_items ? 0 : 1;
                    // the same per-geometry choice as the forward convolutions (run_conv): F(4x4,3x3) for maps of 16 x 16 and more
                    if (d_f4) {
                        float* w4 = nullptr;
                        if (train_wino4_slot(net, op.name, net->conv[op.name], d_w4b ? 1 : 0, 1, st, &w4)) return 1;
                        p.wpack = w4;
                        p.w_gs = (long long)op.Cout * op.Cin * 36;
                        // Is this data gradient the ONLY writer of the gradient behind a train-mode BatchNorm + ReLU (no residual)?  Then its output stage leaves
                        // that BatchNorm's backward sums (it reads the BatchNorm's input at its own pixels) and the BatchNorm's reduction pass does not run.
                        if (bst_on && fresh && op.stride == 1) {
                            int n_read = 0, bi = -1;
                            for (size_t k = 0; k < tape.size(); ++k) {
                                const TapeOp& o2 = tape[k];
                                if (o2.a == op.a || o2.b == op.a) ++n_read;
                                if (o2.type == 2 && o2.o == op.a && o2.o != o2.a) bi = (int)k;
                            }
                            if (bi >= 0 && n_read == 1) {
                                const TapeOp& bo = tape[bi];
                                const cerb_net::BnDev& bb = net->bn[bo.name];
                                bool any_eval = false;
                                for (size_t g = 0; g < bb.eval.size(); ++g) any_eval = any_eval || bb.eval[g];
                                if (bo.relu && bo.b < 0 && !bo.deferred && !any_eval && bo.G == op.G && bo.Cout == op.Cin && bo.rows == (long long)op.N * op.H * op.W &&
                                    (op.G == 1 || bo.a_gs == in_n)) {
                                    const int bpg = d_w4b ? cerb_wino4b_bn_blocks(p) : op.N * ((op.H + 15) / 16) * ((op.W + 15) / 16);
                                    double* part = (double*)take((size_t)op.G * bpg * op.Cin * 2 * 2, false);
                                    if (!part) return fail("workspace allocation failed");
                                    p.bn_part = part;
                                    p.bst_y = val[bo.a];
                                    p.bst_y_gs = op.G == 1 ? 0 : bo.a_gs;
                                    p.bst_mean = val[bo.stat];
                                    p.bst_rstd = val[bo.stat] + (size_t)bo.G * bo.Cout;
                                    p.bst_gamma = bb.gamma;
                                    p.bst_beta = bb.beta;
                                    bst_part[bo.stat] = std::make_pair(part, bpg);
                                }
                            }
                        }
                        HIP_OK(d_w4b ? cerb_launch_wino4b(p, st) : cerb_launch_wino4(p, st));
                    } else {
                        if (train_wino2_fresh(net, op.name, net->conv[op.name], 1, st)) return 1;
                        HIP_OK(cerb_launch_wino(p, st));
                    }
                    if (prof_end(net, st)) return 1;
                    dx_done = true;
                    go = grd[op.o];
                }
                bool dw_done = false, db_done = false;
                // 3x3 stride 1 on whole 64-channel blocks: the weight gradient in the Winograd domain (conv_wgrad_wino.hip: a quarter of the matrix
                // instructions of the direct form); CERB_WGRAD_DIRECT=1 keeps round 4's direct kernel everywhere (A/B, tests)
                if (wg_wino) {
                    if (wws->ensure(cerb_wgrad_wino_workspace_bytes(op.G, op.N, op.H, op.W, op.Cin, op.Cout), 0)) return fail("workspace allocation failed");
                    if (prof_begin(net, op.name + ".wgrad", "wgrad_wino4<f4x4>", 2.0 * op.G * op.N * op.H * op.W * (double)op.Cin * op.Cout * 9.0, st)) return 1;
                    HIP_OK(cerb_launch_wgrad_wino(val[op.a], go, dw, op.G, op.N, op.H, op.W, op.Cin, op.Cout, op.a_gs, wws->p, wst, db));
                    if (prof_end(net, st)) return 1;
                    dw_done = true;
                    if (db) db_done = true;
                }
                if (!dw_done && wg_mfma) {  // weight gradient on the matrix cores
                    const int ho = op.stride == 2 ? op.H / 2 : op.H, wo = op.stride == 2 ? op.W / 2 : op.W;
                    if (wws->ensure(cerb_wgrad_workspace_bytes(op.G, op.N, ho, wo, op.Cin, op.Cout, op.ks, nullptr), 0)) return fail("workspace allocation failed");
                    // `flops` field: executed MFMA FLOPs of the weight gradient (2 x outputs x taps x Cin x Cout)
                    if (prof_begin(net, op.name + ".wgrad", "wgrad<ks" + std::to_string(op.ks) + ",s" + std::to_string(op.stride) + ">",
                                   2.0 * op.G * op.N * ho * wo * (double)op.Cin * op.Cout * op.ks * op.ks, st)) return 1;
                    // the bias gradient (sums of dy over the pixels) rides inside the same pass when the channel count allows
                    const bool db_in_wgrad = db && op.Cout % 64 == 0;
                    HIP_OK(cerb_launch_wgrad(val[op.a], go, dw, op.G, op.N, op.H, op.W, op.Cin, op.Cout, op.ks, op.stride, op.a_gs, wws->p, wst, db_in_wgrad ? db : nullptr));
                    if (prof_end(net, st)) return 1;
                    dw_done = true;
                    if (db_in_wgrad) db_done = true;
                }
                if (db && !db_done) {
                    const long long orow = (long long)op.N * (op.stride == 2 ? op.H / 2 : op.H) * (op.stride == 2 ? op.W / 2 : op.W);
                    if (net->t_ws.ensure((size_t)op.G * 2048 * op.Cout * 4 + 256, 0)) return fail("workspace allocation failed");
                    PROF(op.name + ".dbias", "bias_colsum", (double)op.G * orow * op.Cout * 4.0, HIP_OK(cerb_launch_colsum(go, orow * op.Cout, orow, op.Cout, op.G, db, net->t_ws.p, st)));
                }
                if (!dx_done || !dw_done) {
                    float* dxg = dx_done ? nullptr : G_(op.a);
                    PROF(op.name + ".bwd", "conv_bwd_direct", 2.0 * op.G * op.N * op.H * op.W * (double)op.Cin * op.Cout * op.ks * op.ks / (op.stride * op.stride) * ((dx_done ? 0 : 1) + (dw_done ? 0 : 1)),
                         HIP_OK(cerb_launch_conv_bwd(val[op.a], go, r.w, dxg, dw_done ? nullptr : dw, nullptr, op.G, op.N, op.H, op.W, op.Cin, op.Cout, op.ks, op.stride, op.a_gs, st)));
                }
                for (int g = 0; g < op.G; ++g) {
                    net->grads[r.wkeys[g]] = std::make_pair(dw + g * wn, (long long)wn);
                    if (db) net->grads[r.bkeys[g]] = std::make_pair(db + (size_t)g * op.Cout, (long long)op.Cout);
                }
                break;
            }
            case 2: {
                const cerb_net::BnDev& b = net->bn[op.name];
                float* dgb = take((size_t)2 * op.G * op.Cout, false);  // bn_bwd_finalize_kernel assigns both halves
                if (!dgb || net->t_ws.ensure(cerb_bn_workspace_bytes(op.G, op.rows, op.Cout), 0)) return fail("workspace allocation failed");
                float* dgamma = dgb;
                float* dbeta = dgb + (size_t)op.G * op.Cout;
                // the conv output's gradient has this BatchNorm as its first writer almost always: then the kernel assigns and the buffer needs no zero fill
                const bool fresh = !grd[op.a] && cnt[op.a] == (size_t)op.G * op.rows * op.Cout && (op.G == 1 || op.a_gs == op.rows * op.Cout);
                if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                // `flops` field of an HBM-bound family: its algorithmic BYTES (reads dz, z, y twice -- reduction pass + apply pass --, writes dy
                // (+ the residual branch's gradient)), fp32
                // groups whose BatchNorm ran in eval mode (cerb_net_set_bn_eval): the backward of a normalisation by CONSTANTS -- the data gradient
                // upstream of a frozen BatchNorm is then right for callers that keep the convolutions under it trainable
                unsigned long long eval_mask = 0;
                if (b.eval.size() > 64) return fail("cerb_net_train_grads: more than 64 groups under one eval-mode BatchNorm");
                for (size_t g = 0; g < b.eval.size(); ++g)
                    if (b.eval[g]) eval_mask |= 1ull << g;
                // the residual branch's gradient likewise: assigned when this BatchNorm is its first writer (the identity of a BasicBlock that is not a decoder skip)
                const bool fresh_r = op.b >= 0 && !grd[op.b] && cnt[op.b] == (size_t)op.G * op.rows * op.Cout;
                if (fresh_r && !(grd[op.b] = take(cnt[op.b], false))) return fail("workspace allocation failed");
                // a deferred BatchNorm whose gradient came from the fused heads alone: its reduction pass already happened in their epilogues
                const double* pre_part = nullptr;
                if (op.deferred && !getenv("CERB_HEAD_BN_BWD_PASS")) {
                    auto dp = deferred_part.find(op.stat);
                    if (dp != deferred_part.end() && dp->second.second && !slice_written.count(op.o)) pre_part = dp->second.first;
                }
                int pre_bpg = pre_part ? cerb_head_bwd2_blocks() : 0;
                if (!pre_part) {  // ... or in the output stage of the data gradient that wrote this BatchNorm's output gradient
                    auto bp = bst_part.find(op.stat);
                    if (bp != bst_part.end()) {
                        pre_part = bp->second.first;
                        pre_bpg = bp->second.second;
                    }
                }
                if (prof_begin(net, op.name + ".bn_bwd", "bn_bwd", (double)op.G * op.rows * op.Cout * 4.0 * ((pre_part ? 3.0 : 5.0) + (op.b >= 0 ? (fresh_r ? 1.0 : 2.0) : 0.0)), st)) return 1;
                HIP_OK(cerb_launch_bn_bwd(go, val[op.o], val[op.a], G_(op.a), op.b >= 0 ? G_(op.b) : nullptr, op.a_gs, op.rows, op.Cout, op.G, val[op.stat],
                                          val[op.stat] + (size_t)op.G * op.Cout, b.gamma, b.beta, dgamma, dbeta, op.relu, fresh ? 1 : 0, net->t_ws.p, st, eval_mask,
                                          fresh_r ? 1 : 0, pre_part, pre_bpg));
                if (prof_end(net, st)) return 1;
                const std::vector<std::string>& keys = net->bn_keys[op.name];
                for (int g = 0; g < op.G; ++g) {
                    net->grads[keys[g] + ".weight"] = std::make_pair(dgamma + (size_t)g * op.Cout, (long long)op.Cout);
                    net->grads[keys[g] + ".bias"] = std::make_pair(dbeta + (size_t)g * op.Cout, (long long)op.Cout);
                }
                break;
            }
            case 3: {
                float* dxp = G_(op.a);
                if (op.b >= 0) {  // by the recorded positions: dx read + written, dy and the position bytes read
                    PROF("maxpool.bwd", "maxpool_bwd", (double)op.N * op.H * op.W * op.Cout * 4.0 * 2.25 + (double)op.N * op.H * op.W * op.Cout / 4.0,
                         HIP_OK(cerb_launch_maxpool_bwd_idx(reinterpret_cast<const unsigned*>(val[op.b]), go, dxp, op.N, op.H, op.W, op.Cout, st)));
                } else {
                    PROF("maxpool.bwd", "maxpool_bwd", (double)op.N * op.H * op.W * op.Cout * 4.0 * 2.5, HIP_OK(cerb_launch_maxpool_bwd(val[op.a], val[op.o], go, dxp, op.N, op.H, op.W, op.Cout, st)));
                }
                break;
            }
            case 4: {
                // The reference runs a decoder that is not in train_decoder_list under torch.set_grad_enabled(False) (models/net_desc.py:182), but
                // its conv layers switch autograd back on inside themselves (models/utils/conv_layers.py:44-53): gradients then live only
                // INSIDE each block and stop at the skip + upsample sum.  With train_step's substring test (run_desc.py:70-74) that is the
                // fate of the "#TYPE" decoders ("Gland#TYPE" is not a substring of "Gland-TYPE"): cut their slices here.
                const long long per_group = (long long)op.N * op.H * op.W * op.Cout;
                // the fused kernel takes the untrained decoders as a group mask and ASSIGNS outputs it is the first writer of (the skip tensors'
                // gradients always: the decoders run their backward before the encoder; the level below's gradient too): round 4 zero-filled the
                // masked slices of `go` and both outputs first -- 5 GB of fills and as many extra reads per step
                const bool fused = cerb_upadd_bwd_fused_ok(op.H, op.W, op.Cout, op.G);
                unsigned mask = 0xffffffffu;
                if (io->decoder_trained)
                    for (int k = 0; k < op.G; ++k)
                        if (!io->decoder_trained[net->dense_idx[k]]) mask &= ~(1u << k);
                const bool skip_fresh = fused && !grd[op.a] && cnt[op.a] == (size_t)per_group;
                const size_t prev_n = (size_t)op.N * (op.H / 2) * (op.W / 2) * op.Cout;
                const bool prev_fresh = fused && !grd[op.b] && (op.b_gs == 0 ? cnt[op.b] == prev_n : (cnt[op.b] == (size_t)op.G * prev_n && op.b_gs == (long long)prev_n));
                if (skip_fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                if (prev_fresh && !(grd[op.b] = take(cnt[op.b], false))) return fail("workspace allocation failed");
                float* ga = G_(op.a);
                float* gb = G_(op.b);
                // algorithmic bytes: read the live groups' gradients, write the skip gradient (read it too when it already holds one) and the level below's
                int live_g = 0;
                for (int k = 0; k < op.G; ++k) live_g += (mask >> k) & 1u;
                PROF("upadd.bwd", "upadd_bwd", (double)per_group * 4.0 * (live_g + (skip_fresh ? 1.0 : 2.0) + (op.b_gs == 0 ? 0.25 : 0.25 * op.G) * (prev_fresh ? 1.0 : 2.0)),
                     if (!fused && io->decoder_trained)
                         for (int k = 0; k < op.G; ++k)
                             if (!io->decoder_trained[net->dense_idx[k]]) HIP_OK(hipMemsetAsync(go + k * per_group, 0, per_group * 4, st));
                     HIP_OK(cerb_launch_upadd_bwd(go, ga, gb, op.G, op.N, op.H, op.W, op.Cout, op.b_gs, op.b_gs == 0 ? 1 : 0, st, fused ? mask : 0xffffffffu,
                                                  skip_fresh ? 1 : 0, prev_fresh ? 1 : 0)));
                break;
            }
            case 5: {
                float* dw = pub(op.wkey, (size_t)op.Cin * op.Cout);
                float* db = pub(op.bkey, (size_t)op.Cout);
                if (!dw || !db) return fail("workspace allocation failed");
                bool pw_dw = false, pw_db = false;
                if (prof_begin(net, op.wkey + ".bwd", "pointwise_bwd", 4.0 * op.rows * (double)op.Cin * op.Cout, st)) return 1;  // weight + data gradient + bias sums
                if (op.Cin % 4 == 0 && op.Cout % 4 == 0 && !op.scale && op.rows >= 4096 && op.rows < (1ll << 31) && net->conv_algo) {
                    if (net->t_ws.ensure(cerb_wgrad_workspace_bytes(1, 1, 1, (int)op.rows, op.Cin, op.Cout, 1, nullptr), 0)) return fail("workspace allocation failed");
                    HIP_OK(cerb_launch_wgrad(val[op.a] + op.a_gs, go, dw, 1, 1, 1, (int)op.rows, op.Cin, op.Cout, 1, 1, 0, net->t_ws.p, st, db));  // (+ the bias sums)
                    pw_dw = true;
                    pw_db = true;
                }
                if (!pw_dw && op.Cout <= 8 && !op.scale && op.rows >= 4096 && net->conv_algo) {
                    // the heads' 96 -> 3 / 7: weight gradient, bias sums and data gradient in one pass over the rows (cerb_launch_pw_bwd_small)
                    const bool fresh1 = !grd[op.a] && op.a_gs == 0 && cnt[op.a] == (size_t)op.rows * op.Cin;
                    if (fresh1 && !(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                    float* dxs = G_(op.a) + op.a_gs;
                    if (net->t_ws.ensure(cerb_pw_bwd_small_workspace_bytes(op.rows, op.Cin, op.Cout), 0)) return fail("workspace allocation failed");
                    HIP_OK(cerb_launch_pw_bwd_small(val[op.a] + op.a_gs, go, op.w, dxs, dw, db, op.rows, op.Cin, op.Cout, fresh1 ? 1 : 0, net->t_ws.p, st));
                    if (prof_end(net, st)) return 1;
                    break;
                }
                if (!pw_dw && op.Cout <= 8 && !op.scale && op.rows >= 4096) {  // (conv_algo 0: the separate passes)
                    if (net->t_ws.ensure(cerb_pw_wgrad_small_workspace_bytes(op.rows, op.Cin, op.Cout), 0)) return fail("workspace allocation failed");
                    HIP_OK(cerb_launch_pw_wgrad_small(val[op.a] + op.a_gs, go, dw, op.rows, op.Cin, op.Cout, net->t_ws.p, st));
                    pw_dw = true;
                }
                if (!pw_db) {
                    if (net->t_ws.ensure((size_t)2048 * op.Cout * 4 + 256, 0)) return fail("workspace allocation failed");
                    HIP_OK(cerb_launch_colsum(go, 0, op.rows, op.Cout, 1, db, net->t_ws.p, st));
                }
                // a hidden map read by this layer alone gets its gradient assigned (no zero fill, no read-modify-write)
                bool fresh = !grd[op.a] && op.a_gs == 0 && cnt[op.a] == (size_t)op.rows * op.Cin;
                if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                const size_t slice = (size_t)op.rows * op.Cin;
                if (!fresh && cnt[op.a] > slice && cnt[op.a] % slice == 0 && cnt[op.a] / slice <= 64 && op.a_gs % (long long)slice == 0 &&
                    (!grd[op.a] || slice_written.count(op.a))) {  // one slice of a grouped tensor that only such layers have written so far
                    if (!grd[op.a]) {
                        if (!(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                        slice_written[op.a] = 0ull;
                        slice_geom[op.a] = std::make_pair((int)(cnt[op.a] / slice), slice);
                    }
                    const int k = (int)(op.a_gs / (long long)slice);
                    if (!((slice_written[op.a] >> k) & 1ull)) {
                        fresh = true;
                        slice_written[op.a] |= 1ull << k;
                    }
                }
                float* dxp = (fresh && grd[op.a]) ? grd[op.a] : G_(op.a);  // (a slice's first writer must not trigger G_'s zeroing of the unwritten slices)
                HIP_OK(cerb_launch_pointwise_bwd(val[op.a] + op.a_gs, go, op.w, dxp + op.a_gs, pw_dw ? nullptr : dw, nullptr, op.rows, op.Cin, op.Cout, op.scale,
                                                 fresh ? 1 : 0, st));
                if (prof_end(net, st)) return 1;
                break;
            }
            case 7: {  // a whole output head: two passes over the stored hidden map (head_train.hip)
                const int k = op.head_k, oc = op.Cout;
                const cerb_net::BnDev& b = net->bn[op.name];
                float* dw2 = pub(op.wkey2, (size_t)oc * 96);
                float* db2 = pub(op.bkey2, (size_t)oc);
                float* dw1 = pub(op.wkey, (size_t)96 * 64);
                float* db1 = pub(op.bkey, 96);
                float* dgb = take(2 * 96, false);
                if (!dw2 || !db2 || !dw1 || !db1 || !dgb || net->t_ws.ensure(cerb_head_bwd_workspace_bytes(op.rows, oc), 0)) return fail("workspace allocation failed");
                float* dgamma = dgb;
                float* dbeta = dgb + 96;
                const float* mean = val[op.stat];
                const float* rstd = val[op.stat] + 96;
                PROF(op.name + ".bwd1", "head_bwd1", (double)op.rows * (96 + oc) * 4.0,
                     HIP_OK(cerb_launch_head_bwd1(val[op.hid], go, mean, rstd, b.gamma, b.beta, net->head_rw2[k], dw2, db2, dgamma, dbeta, op.rows, oc, net->t_ws.p, st)));
                // the head's slice of the grouped decoder tensor: first writer assigns (see case 5)
                const size_t slice = (size_t)op.rows * 64;
                bool fresh = false;
                if (cnt[op.a] >= slice && cnt[op.a] % slice == 0 && cnt[op.a] / slice <= 64 && op.a_gs % (long long)slice == 0 && (!grd[op.a] || slice_written.count(op.a))) {
                    if (!grd[op.a]) {
                        if (!(grd[op.a] = take(cnt[op.a], false))) return fail("workspace allocation failed");
                        slice_written[op.a] = 0ull;
                        slice_geom[op.a] = std::make_pair((int)(cnt[op.a] / slice), slice);
                    }
                    const int sk = (int)(op.a_gs / (long long)slice);
                    if (!((slice_written[op.a] >> sk) & 1ull)) {
                        fresh = true;
                        slice_written[op.a] |= 1ull << sk;
                    }
                }
                float* dxp = (fresh && grd[op.a]) ? grd[op.a] : G_(op.a);
                const int eval_mode = (!b.eval.empty() && b.eval[0]) ? 1 : 0;
                const float* in_bn[4] = {nullptr, nullptr, nullptr, nullptr};
                if (op.in_stat >= 0) {
                    const cerb_net::BnDev& lb = net->bn[op.in_bn];
                    in_bn[0] = val[op.in_stat] + k * (size_t)lb.C;
                    in_bn[1] = val[op.in_stat] + (size_t)lb.groups * lb.C + k * (size_t)lb.C;
                    in_bn[2] = lb.gamma + k * (size_t)lb.C;
                    in_bn[3] = lb.beta + k * (size_t)lb.C;
                }
                double* in_part = nullptr;
                if (op.in_stat >= 0) {  // the deferred BatchNorm's backward sums come out of this launch's epilogue -- as long as every head is its slice's first writer
                    const cerb_net::BnDev& lb = net->bn[op.in_bn];
                    const size_t per = (size_t)cerb_head_bwd2_blocks() * lb.C * 2;  // doubles per group
                    auto dp = deferred_part.find(op.in_stat);
                    if (dp == deferred_part.end()) {
                        double* pb_ = (double*)take((size_t)lb.groups * per * 2, true);
                        if (!pb_) return fail("workspace allocation failed");
                        dp = deferred_part.insert(std::make_pair(op.in_stat, std::make_pair(pb_, true))).first;
                    }
                    if (fresh) in_part = dp->second.first + (size_t)k * per;
                    else dp->second.second = false;
                }
                PROF(op.name + ".bwd2", "head_bwd2", (double)op.rows * (96 + 64 + 64 + oc) * 4.0,
                     HIP_OK(cerb_launch_head_bwd2(val[op.hid], go, val[op.a] + op.a_gs, mean, rstd, b.gamma, b.beta, dgamma, dbeta, net->head_rw1[k], net->head_rw2[k],
                                                  dxp + op.a_gs, dw1, db1, op.rows, oc, eval_mode, fresh ? 1 : 0, net->t_ws.p, st, op.in_stat >= 0 ? in_bn : nullptr, in_part)));
                const std::vector<std::string>& keys = net->bn_keys[op.name];
                net->grads[keys[0] + ".weight"] = std::make_pair(dgamma, 96ll);
                net->grads[keys[0] + ".bias"] = std::make_pair(dbeta, 96ll);
                break;
            }
            case 6: {
                float* dxc = G_(op.a);
                PROF("pc.crop_gap.bwd", "crop_gap_bwd", (double)op.N * op.H * op.W * op.Cout * 4.0, HIP_OK(cerb_launch_crop_gap_bwd(go, dxc, op.N, op.H, op.W, op.Cout, op.y0, op.ch, op.x0, op.cw, st)));
                break;
            }
        }
    }
#undef TCHK
#undef PROF
#undef PROFN
    net->conv_algo = saved_algo;
    if (side_used) {  // the caller's stream continues (optimiser, all-reduce, the next step's tape) once the side stream's weight gradients are complete
        HIP_OK(hipEventRecord(net->ev_join, net->side));
        HIP_OK(hipStreamWaitEvent(st, net->ev_join, 0));
    }
    return 0;
}

extern "C" int cerb_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1, float beta2,
                              float eps, int step, void* hip_stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || numel < 0 || step < 1) return fail("cerb_adam_step: bad arguments");
    HIP_OK(cerb_launch_adam(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step, (hipStream_t)hip_stream));
    return 0;
}
extern "C" int cerb_adam_step_multi(int count, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                                    const long long* numel, float lr, float beta1, float beta2, float eps, int step, void* hip_stream) {
    if (count < 0 || (count && (!param || !grad || !exp_avg || !exp_avg_sq || !numel)) || step < 1) return fail("cerb_adam_step_multi: bad arguments");
    for (int i = 0; i < count; ++i)
        if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return fail("cerb_adam_step_multi: null tensor in the list");
    HIP_OK(cerb_launch_adam_multi(count, param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step, (hipStream_t)hip_stream));
    return 0;
}
extern "C" int cerb_copy_d2d(void* dst, const void* src, size_t bytes, void* hip_stream) {
    if (!dst || !src) return fail("cerb_copy_d2d: null pointer");
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    return 0;
}
extern "C" int cerb_net_grad_lookup(cerb_net* net, const char* key, float** dev_ptr, long long* numel) {
    if (!net || !key || !dev_ptr || !numel) return fail("cerb_net_grad_lookup: null argument");
    auto it = net->grads.find(key);
    if (it == net->grads.end()) return fail(std::string("cerb_net_grad_lookup: no gradient for ") + key);
    *dev_ptr = it->second.first;
    *numel = it->second.second;
    return 0;
}

extern "C" int cerb_net_begin_reload(cerb_net* net) {
    if (!net) return fail("cerb_net_begin_reload: null handle");
    HIP_OK(hipDeviceSynchronize());
    while (net->dev_allocs.size() > net->n_finalize_allocs) {  // buffers made after finalize (lazy packings); the others are handed out again
        (void)hipFree(net->dev_allocs.back());
        net->dev_allocs.pop_back();
        net->dev_alloc_bytes.pop_back();
    }
    net->reusing = true;
    net->reuse_cursor = 0;
    net->param_slots.clear();
    net->conv.clear();
    net->bn.clear();
    net->raw.clear();
    net->grads.clear();
    net->head_w1.clear(); net->head_b1.clear(); net->head_w2.clear(); net->head_b2.clear(); net->head_w2q.clear();
    net->head_rw1.clear(); net->head_rb1.clear(); net->head_rw2.clear(); net->head_rb2.clear();
    net->host.clear();
    net->finalized = false;
    return 0;
}
extern "C" int cerb_net_update_params(cerb_net* net, int count, const char* const* keys, const float* const* dev_src, void* hip_stream) {
    if (!net || count < 0 || (count && (!keys || !dev_src))) return fail("cerb_net_update_params: bad arguments");
    if (!net->finalized || net->fold_bn) return fail("cerb_net_update_params: needs a finalized handle packed for training (cerb_net_set_fold_bn(net, 0))");
    hipStream_t st = (hipStream_t)hip_stream;
    std::vector<float*> cd;
    std::vector<const float*> cs;
    std::vector<long long> cn;
    for (int i = 0; i < count; ++i) {
        auto it = net->param_slots.find(keys[i]);
        if (it == net->param_slots.end()) continue;  // running statistics, num_batches_tracked, backbone.fc.*: nothing on the device reads them
        if (!dev_src[i]) return fail(std::string("cerb_net_update_params: null source for ") + keys[i]);
        for (const cerb_net::ParamSlot& sl : it->second) {
            cd.push_back(sl.dst); cs.push_back(dev_src[i]); cn.push_back(sl.n);
        }
    }
    // every parameter tensor into its slot(s) in ONE launch (round 3: one hipMemcpyAsync per tensor, ~470 per optimiser step)
    HIP_OK(cerb_launch_copy_multi((int)cd.size(), cd.data(), cs.data(), cn.data(), &net->copy_tab, &net->copy_tab_bytes, &net->copy_tab_host, st));
    HIP_OK(cerb_launch_pack_stem(net->stem_raw, net->stem_w, st));
    // every conv's re-layouts / filter transforms as jobs of ONE launch (pack_kernels.hip: pack_multi_kernel); CERB_PACK_PER_CONV=1: round 4's launches
    static const bool per_conv = getenv("CERB_PACK_PER_CONV") != nullptr;
    std::vector<PackJob> jobs;
    auto job = [&](const float* w, float* out, long long total, int cout, int cin, int kind, int a, int b) { jobs.push_back(PackJob{w, out, total, cout, cin, kind, a, b, 0}); };
    for (auto& kv : net->conv) {
        PackedConv& pc = kv.second;
        const float* rawd = net->raw[kv.first].w;
        if (pc.wino && !pc.wino_used) pc.wino_stale = true;
        if (pc.wino_dgrad && !pc.wino_dgrad_used) pc.wino_dgrad_stale = true;
        const size_t nw = (size_t)pc.cout * pc.cin * pc.ks * pc.ks, nu = (size_t)pc.cout * pc.cin * 16, nu4 = (size_t)pc.cout * pc.cin * 36;
        for (int g = 0; g < pc.groups; ++g) {
            if (per_conv) {
                HIP_OK(cerb_launch_pack_conv(rawd + g * nw, pc.w + g * nw, pc.cout, pc.cin, pc.ks, cerb_conv_chunk(pc.ks, pc.stride), st));
                if (pc.wino && pc.wino_used) HIP_OK(cerb_launch_pack_wino(rawd + g * nw, pc.wino + g * nu, pc.cout, pc.cin, 0, st));
                if (pc.wino_dgrad && pc.wino_dgrad_used) HIP_OK(cerb_launch_pack_wino(rawd + g * nw, pc.wino_dgrad + g * nu, pc.cin, pc.cout, 1, st));
                continue;
            }
            job(rawd + g * nw, pc.w + g * nw, (long long)nw, pc.cout, pc.cin, 0, pc.ks * pc.ks, cerb_conv_chunk(pc.ks, pc.stride));
            if (pc.wino && pc.wino_used) job(rawd + g * nw, pc.wino + g * nu, (long long)nu, pc.cout, pc.cin, 1, 0, 0);
            if (pc.wino_dgrad && pc.wino_dgrad_used) job(rawd + g * nw, pc.wino_dgrad + g * nu, (long long)nu, pc.cin, pc.cout, 1, 1, 0);
        }
        for (int l = 0; l < 2; ++l)  // the F(4x4) layouts in use
            for (int dg = 0; dg < 2; ++dg)
                if (pc.wino4_t[l][dg]) {
                    if (per_conv) {
                        HIP_OK(cerb_launch_pack_wino4(rawd, pc.wino4_t[l][dg], dg ? pc.cin : pc.cout, dg ? pc.cout : pc.cin, dg, l, pc.groups, st));
                        continue;
                    }
                    for (int g = 0; g < pc.groups; ++g)
                        job(rawd + g * nw, pc.wino4_t[l][dg] + g * nu4, (long long)nu4, dg ? pc.cin : pc.cout, dg ? pc.cout : pc.cin, 2, dg, l);
                }
    }
    if (!jobs.empty()) HIP_OK(cerb_launch_pack_multi(jobs.data(), (int)jobs.size(), &net->pack_tab, &net->pack_tab_bytes, &net->pack_tab_host, st));
    return 0;
}
extern "C" int cerb_net_set_fold_bn(cerb_net* net, int fold) {
    if (!net) return fail("cerb_net_set_fold_bn: null handle");
    if (net->finalized) return fail("cerb_net_set_fold_bn: must be called before cerb_net_finalize");
    net->fold_bn = fold ? 1 : 0;
    return 0;
}

extern "C" int cerb_net_forward(cerb_net* net, const cerb_forward_io* io, void* hip_stream) {
    if (!net || !io) return fail("cerb_net_forward: null argument");
    if (!net->finalized) return fail("cerb_net_forward: call cerb_net_finalize first");
    if (!io->tiles && !io->tiles_f32) return fail("cerb_net_forward: null tiles pointer (neither tiles nor tiles_f32)");
    if (!net->fold_bn) return fail("cerb_net_forward: the network was packed for training (cerb_net_set_fold_bn(net, 0)); use cerb_net_forward_train");
    net->prof_n = 0;
    return forward_impl(net, io, (hipStream_t)hip_stream, nullptr);
}

extern "C" double cerb_net_flops(const cerb_net* net, int n, int h, int w) {
    if (!net || !net->finalized) return -1.0;
    cerb_forward_io io;
    memset(&io, 0, sizeof(io));
    io.n = n; io.h = h; io.w = w;
    double macs = 0.0;
    if (forward_impl(const_cast<cerb_net*>(net), &io, nullptr, &macs)) return -1.0;
    return 2.0 * macs;
}

// ---- per-launch profile (bench.py roofline leg) ----------------------------------------------------------------
extern "C" int cerb_net_set_crop_roi(cerb_net* net, int enable) {
    if (!net) return fail("cerb_net_set_crop_roi: null handle");
    net->crop_roi = enable ? 1 : 0;
    return 0;
}
extern "C" int cerb_net_set_conv_algo(cerb_net* net, int algo) {
    if (!net) return fail("cerb_net_set_conv_algo: null handle");
    if (algo < 0 || algo > 7 || (algo >= 2 && algo <= 4))
        return fail("cerb_net_set_conv_algo: algo must be 0 (direct), 1 (Winograd F(2x2) fp32), 5 (Winograd F(4x4) fp32), 7 (F(4x4), one-block items with 32-channel chunks) or 6 (F(4x4) for maps of 16 x 16 pixels and more -- 7's kernel up to 64 x 64, 5's above --, else F(2x2): the default); 2-4 were experiments (scripts/experiments/)");
    net->conv_algo = algo;
    return 0;
}

extern "C" int cerb_net_set_bn_eval(cerb_net* net, const char* bn_prefix, const float* running_mean, const float* running_var, int channels) {
    if (!net || !bn_prefix) return fail("cerb_net_set_bn_eval: null argument");
    if (!net->finalized || net->fold_bn) return fail("cerb_net_set_bn_eval: needs a finalized network packed with cerb_net_set_fold_bn(net, 0)");
    for (auto& kv : net->bn_keys) {
        const std::vector<std::string>& keys = kv.second;
        for (size_t g = 0; g < keys.size(); ++g) {
            if (keys[g] != bn_prefix) continue;
            auto it = net->bn.find(kv.first);
            if (it == net->bn.end()) return fail(std::string("cerb_net_set_bn_eval: internal: no BatchNorm ") + kv.first);
            cerb_net::BnDev& b = it->second;
            if (!running_mean || !running_var) {  // back to training mode
                if (g < b.eval.size()) b.eval[g] = 0;
                return 0;
            }
            if (channels != b.C) return fail(std::string("cerb_net_set_bn_eval: ") + bn_prefix + " has " + std::to_string(b.C) + " channels");
            if (!b.run_mean) {
                void* d = nullptr;
                HIP_OK(hipMalloc(&d, (size_t)2 * b.groups * b.C * 4));
                net->dev_allocs.push_back(d);
                net->dev_alloc_bytes.push_back((size_t)2 * b.groups * b.C * 4);
                b.run_mean = (float*)d;
                b.run_rstd = b.run_mean + (size_t)b.groups * b.C;
                b.eval.assign(b.groups, 0);
            }
            std::vector<float> rs(b.C);
            for (int c = 0; c < b.C; ++c) rs[c] = 1.0f / sqrtf(running_var[c] + 1e-5f);
            HIP_OK(hipMemcpy(b.run_mean + g * b.C, running_mean, (size_t)b.C * 4, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(b.run_rstd + g * b.C, rs.data(), (size_t)b.C * 4, hipMemcpyHostToDevice));
            b.eval[g] = 1;
            return 0;
        }
    }
    return fail(std::string("cerb_net_set_bn_eval: no BatchNorm with the state-dict prefix ") + bn_prefix);
}

extern "C" int cerb_net_set_planar(cerb_net* net, int enable) {
    if (!net) return fail("cerb_net_set_planar: null handle");
    if (enable < 0 || enable > 1) return fail("cerb_net_set_planar: 0 (NHWC) or 1 (tile-planar, conv_wino4p.hip)");
    net->planar = enable;
    return 0;
}

extern "C" int cerb_net_set_packed_items(cerb_net* net, int enable) {
    if (!net) return fail("cerb_net_set_packed_items: null handle");
    if (enable < 0 || enable > 1) return fail("cerb_net_set_packed_items: 0 (16 x 16 blocks everywhere) or 1 (16 consecutive tiles per item on maps that are not whole blocks)");
    net->packed_items = enable;
    return 0;
}

extern "C" int cerb_net_set_head_algo(cerb_net* net, int algo) {
    if (!net) return fail("cerb_net_set_head_algo: null handle");
    if (algo < 0 || algo > 2) return fail("cerb_net_set_head_algo: algo must be 0 (one launch per head), 1 (grouped launch) or 2 (round 3's grouped launch)");
    net->head_algo = algo;
    return 0;
}

extern "C" int cerb_net_profile_enable(cerb_net* net, int enable) {
    if (!net) return fail("cerb_net_profile_enable: null handle");
    net->profiling = enable != 0;
    net->prof_n = 0;
    return 0;
}
extern "C" int cerb_net_profile_count(cerb_net* net) { return net ? (int)net->prof_n : 0; }
extern "C" int cerb_net_profile_get(cerb_net* net, int idx, char* name, int name_cap, char* kernel, int kernel_cap, double* flops, float* ms) {
    if (!net || idx < 0 || (size_t)idx >= net->prof_n) return fail("cerb_net_profile_get: index out of range");
    cerb_net::ProfRec& r = net->prof[idx];
    HIP_OK(hipEventSynchronize(r.e1));
    HIP_OK(hipEventElapsedTime(ms, r.e0, r.e1));
    snprintf(name, name_cap, "%s", r.name.c_str());
    snprintf(kernel, kernel_cap, "%s", r.kernel.c_str());
    *flops = r.flops;
    return 0;
}

// ---- events ----------------------------------------------------------------------------------------------------
extern "C" int cerb_event_create(void** ev) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    *ev = (void*)e;
    return 0;
}
extern "C" int cerb_event_record(void* ev, void* hip_stream) {
    HIP_OK(hipEventRecord((hipEvent_t)ev, (hipStream_t)hip_stream));
    return 0;
}
extern "C" int cerb_event_elapsed_ms(void* a, void* b, float* ms) {
    HIP_OK(hipEventSynchronize((hipEvent_t)b));
    HIP_OK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}
extern "C" int cerb_event_destroy(void* ev) {
    HIP_OK(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
