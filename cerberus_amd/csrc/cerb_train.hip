// Training half of the C-ABI implementation (include/cerberus_hip.h): train-mode forward (models/run_desc.py:79-86), the backward tape of one step
// (cerb_net_train_grads: models/run_desc.py:79-170), the optimiser entry points and the gradient lookup.  Split out of cerb_api.hip in round 6;
// the handle and the helpers both halves share are in cerb_net.h.  Host code only; kernels live in train_kernels.hip / conv_wgrad*.hip / head_train.hip.
#include "cerb_net.h"

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
        return fail_alloc();
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
        return fail_alloc();
    for (int i = 1; i < 5; ++i)
        if (net->x[i].ensure((size_t)N * hs[i] * ws[i] * kFilters[i] * 4, guard)) return fail_alloc();
    const int oc[4] = {128, 64, 64, 64};
    if (D) {
        if (net->dmid.ensure(D * (size_t)N * H * W * 64 * 4, guard) || net->dsum.ensure(D * (size_t)N * H * W * 64 * 4, guard)) return fail_alloc();
        for (int u = 0; u < 4; ++u)
            if (net->dout[u].ensure(D * (size_t)N * hs[3 - u] * ws[3 - u] * oc[u] * 4, guard)) return fail_alloc();
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
                if (net->t_idn.ensure((size_t)rows_out * planes * 4, guard)) return fail_alloc();
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
        if (net->t_gap.ensure((size_t)N * 512 * 4, 0) || net->t_pc1.ensure((size_t)N * 256 * 4, 0)) return fail_alloc();
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
        if (net->t_hid.ensure((size_t)rows * 96 * 4, 0)) return fail_alloc();
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
    const bool bst_on = cerb_dev_getenv("CERB_BN_BWD_PASS1") == nullptr;
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
        const char* e = cerb_dev_getenv("CERB_WGRAD_SIDE");
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
    const bool bn_stats_pass = cerb_dev_getenv("CERB_BN_STATS_PASS") != nullptr;  // developer A/B: the separate statistics pass (read once per step)
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
    const bool pool_scan = cerb_dev_getenv("CERB_MAXPOOL_SCAN") != nullptr;  // read once per step
    const int pool_idx = pool_scan ? -1 : newT(((size_t)N * hs[1] * ws[1] * 64 + 3) / 4);
    if (pool_idx >= 0) {
        if (!val[pool_idx]) { net->conv_algo = saved_algo; return fail_alloc(); }
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
            bool defer = u == 3 && net->conv_algo && cerb_dev_getenv("CERB_HEAD_UNFUSED") == nullptr && cerb_dev_getenv("CERB_HEAD_BN_APPLY_PASS") == nullptr && net->bn[n1].eval.empty();
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
            if (net->t_ws.ensure(std::max(cerb_bn_workspace_bytes(1, rows, 96), (size_t)8192 * 96 * 16), 0)) return fail_alloc();
            const bool heads_fused = net->conv_algo && cerb_dev_getenv("CERB_HEAD_UNFUSED") == nullptr && cerb_head_train_supported(rows, 64, 96, d.out_ch);
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
                if (!val[lg]) return fail_alloc();
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
        if (net->t_hid.ensure(cerb_head_loss_workspace_bytes(N, hh, ww), 0)) return fail_alloc();
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
                if (net->t_ws.ensure(cerb_stem_wgrad_workspace_bytes(), 0)) return fail_alloc();
                float* dws = pub("backbone.conv1.weight", 64 * 147);
                PROF("stem.wgrad", "stem_wgrad", 2.0 * N * H * W * 64.0 * 147.0, HIP_OK(cerb_launch_stem_wgrad_mfma(io->tiles, go, dws, N, H, W, net->t_ws.p, st)));
                break;
            }
            case 1: {
                const cerb_net::RawW& r = net->raw[op.name];
                const size_t wn = (size_t)op.Cout * op.Cin * op.ks * op.ks;
                // which weight-gradient kernel this layer takes, decided up front: its workspace is sized and the side stream forked BEFORE the data gradient is queued
                const bool wg_wino = op.ks == 3 && op.stride == 1 && net->conv_algo >= 5 && cerb_wgrad_wino_supported(op.H, op.W, op.Cin, op.Cout) && !cerb_dev_getenv("CERB_WGRAD_DIRECT");
                const bool wg_mfma = !wg_wino && (op.ks == 3 || op.ks == 1) && net->conv_algo;
                hipStream_t wst = st;        // the stream the weight gradient is queued on
                DevBuf* wws = &net->t_ws;    // ... and its workspace
                // the MFMA weight gradient (wgrad_reduce_kernel) and the bias column sums ASSIGN their outputs: no zero fill (a step issued
                // ~230 of these 18-us memsets: 4 ms).  Taken BEFORE the fork event: a new arena slot queues its zero fill on the
                // caller's stream, and the side stream -- which writes dw / db -- only waits for what that event covers.
                const bool dw_assigned = (op.ks == 3 || op.ks == 1) && net->conv_algo;
                float* dw = take(wn * op.G, !dw_assigned);
                float* db = r.b ? take((size_t)op.Cout * op.G, false) : nullptr;
                if (!dw) return fail_alloc();
                if (side_wgrad && (wg_wino || wg_mfma)) {
                    const int ho_ = op.stride == 2 ? op.H / 2 : op.H, wo_ = op.stride == 2 ? op.W / 2 : op.W;
                    const size_t need = wg_wino ? cerb_wgrad_wino_workspace_bytes(op.G, op.N, op.H, op.W, op.Cin, op.Cout) : cerb_wgrad_workspace_bytes(op.G, op.N, ho_, wo_, op.Cin, op.Cout, op.ks, nullptr);
                    if (net->t_ws2.ensure(need, 0)) return fail_alloc();
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
                    if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
                    float* dx = G_(op.a);
                    const long long map_px = (long long)op.H * op.W;
                    const bool d_w4 = saved_algo == 5 || saved_algo == 7 || (saved_algo == 6 && map_px >= 256);
                    static const long long d_w4b_max_px = [] {  // developer A/B only, as in run_conv
                        const char* e = cerb_dev_getenv("CERB_W4B_MAX_PX");
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
                        if (net->t_dil.ensure((size_t)dn * 4, cerb_conv_guard_bytes(W))) return fail_alloc();
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
                                    if (!part) return fail_alloc();
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
                    if (wws->ensure(cerb_wgrad_wino_workspace_bytes(op.G, op.N, op.H, op.W, op.Cin, op.Cout), 0)) return fail_alloc();
                    if (prof_begin(net, op.name + ".wgrad", "wgrad_wino4<f4x4>", 2.0 * op.G * op.N * op.H * op.W * (double)op.Cin * op.Cout * 9.0, st)) return 1;
                    HIP_OK(cerb_launch_wgrad_wino(val[op.a], go, dw, op.G, op.N, op.H, op.W, op.Cin, op.Cout, op.a_gs, wws->p, wst, db));
                    if (prof_end(net, st)) return 1;
                    dw_done = true;
                    if (db) db_done = true;
                }
                if (!dw_done && wg_mfma) {  // weight gradient on the matrix cores
                    const int ho = op.stride == 2 ? op.H / 2 : op.H, wo = op.stride == 2 ? op.W / 2 : op.W;
                    if (wws->ensure(cerb_wgrad_workspace_bytes(op.G, op.N, ho, wo, op.Cin, op.Cout, op.ks, nullptr), 0)) return fail_alloc();
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
                    if (net->t_ws.ensure((size_t)op.G * 2048 * op.Cout * 4 + 256, 0)) return fail_alloc();
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
                if (!dgb || net->t_ws.ensure(cerb_bn_workspace_bytes(op.G, op.rows, op.Cout), 0)) return fail_alloc();
                float* dgamma = dgb;
                float* dbeta = dgb + (size_t)op.G * op.Cout;
                // the conv output's gradient has this BatchNorm as its first writer almost always: then the kernel assigns and the buffer needs no zero fill
                const bool fresh = !grd[op.a] && cnt[op.a] == (size_t)op.G * op.rows * op.Cout && (op.G == 1 || op.a_gs == op.rows * op.Cout);
                if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
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
                if (fresh_r && !(grd[op.b] = take(cnt[op.b], false))) return fail_alloc();
                // a deferred BatchNorm whose gradient came from the fused heads alone: its reduction pass already happened in their epilogues
                const double* pre_part = nullptr;
                if (op.deferred && !cerb_dev_getenv("CERB_HEAD_BN_BWD_PASS")) {
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
                if (skip_fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
                if (prev_fresh && !(grd[op.b] = take(cnt[op.b], false))) return fail_alloc();
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
                if (!dw || !db) return fail_alloc();
                bool pw_dw = false, pw_db = false;
                if (prof_begin(net, op.wkey + ".bwd", "pointwise_bwd", 4.0 * op.rows * (double)op.Cin * op.Cout, st)) return 1;  // weight + data gradient + bias sums
                if (op.Cin % 4 == 0 && op.Cout % 4 == 0 && !op.scale && op.rows >= 4096 && op.rows < (1ll << 31) && net->conv_algo) {
                    if (net->t_ws.ensure(cerb_wgrad_workspace_bytes(1, 1, 1, (int)op.rows, op.Cin, op.Cout, 1, nullptr), 0)) return fail_alloc();
                    HIP_OK(cerb_launch_wgrad(val[op.a] + op.a_gs, go, dw, 1, 1, 1, (int)op.rows, op.Cin, op.Cout, 1, 1, 0, net->t_ws.p, st, db));  // (+ the bias sums)
                    pw_dw = true;
                    pw_db = true;
                }
                if (!pw_dw && op.Cout <= 8 && !op.scale && op.rows >= 4096 && net->conv_algo) {
                    // the heads' 96 -> 3 / 7: weight gradient, bias sums and data gradient in one pass over the rows (cerb_launch_pw_bwd_small)
                    const bool fresh1 = !grd[op.a] && op.a_gs == 0 && cnt[op.a] == (size_t)op.rows * op.Cin;
                    if (fresh1 && !(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
                    float* dxs = G_(op.a) + op.a_gs;
                    if (net->t_ws.ensure(cerb_pw_bwd_small_workspace_bytes(op.rows, op.Cin, op.Cout), 0)) return fail_alloc();
                    HIP_OK(cerb_launch_pw_bwd_small(val[op.a] + op.a_gs, go, op.w, dxs, dw, db, op.rows, op.Cin, op.Cout, fresh1 ? 1 : 0, net->t_ws.p, st));
                    if (prof_end(net, st)) return 1;
                    break;
                }
                if (!pw_dw && op.Cout <= 8 && !op.scale && op.rows >= 4096) {  // (conv_algo 0: the separate passes)
                    if (net->t_ws.ensure(cerb_pw_wgrad_small_workspace_bytes(op.rows, op.Cin, op.Cout), 0)) return fail_alloc();
                    HIP_OK(cerb_launch_pw_wgrad_small(val[op.a] + op.a_gs, go, dw, op.rows, op.Cin, op.Cout, net->t_ws.p, st));
                    pw_dw = true;
                }
                if (!pw_db) {
                    if (net->t_ws.ensure((size_t)2048 * op.Cout * 4 + 256, 0)) return fail_alloc();
                    HIP_OK(cerb_launch_colsum(go, 0, op.rows, op.Cout, 1, db, net->t_ws.p, st));
                }
                // a hidden map read by this layer alone gets its gradient assigned (no zero fill, no read-modify-write)
                bool fresh = !grd[op.a] && op.a_gs == 0 && cnt[op.a] == (size_t)op.rows * op.Cin;
                if (fresh && !(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
                const size_t slice = (size_t)op.rows * op.Cin;
                if (!fresh && cnt[op.a] > slice && cnt[op.a] % slice == 0 && cnt[op.a] / slice <= 64 && op.a_gs % (long long)slice == 0 &&
                    (!grd[op.a] || slice_written.count(op.a))) {  // one slice of a grouped tensor that only such layers have written so far
                    if (!grd[op.a]) {
                        if (!(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
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
                if (!dw2 || !db2 || !dw1 || !db1 || !dgb || net->t_ws.ensure(cerb_head_bwd_workspace_bytes(op.rows, oc), 0)) return fail_alloc();
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
                        if (!(grd[op.a] = take(cnt[op.a], false))) return fail_alloc();
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
                        if (!pb_) return fail_alloc();
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

