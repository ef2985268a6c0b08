// C-ABI implementation (include/cerberus_hip.h) for the network part of the Cerberus tile path:
// weight intake (BN folding + MFMA packing), workspace, and the forward schedule
//   stem -> maxpool -> 16 BasicBlocks -> conv_map -> Patch-Class branch -> 5 dense decoders (grouped launches) -> heads.
// Host code only; kernels live in conv_igemm.hip / net_kernels.hip.
#include "cerb_net.h"

static thread_local std::string g_err;
int cerb_set_error(const std::string& m) {
    g_err = m;
    return 1;
}
extern "C" int cerb_version(void) { return 1; }
extern "C" const char* cerb_last_error(void) { return g_err.c_str(); }
std::atomic<long long> g_devbuf_bytes{0};
extern "C" size_t cerb_device_bytes_held(void) { const long long v = g_devbuf_bytes.load(); return v > 0 ? (size_t)v : 0; }
thread_local hipStream_t g_call_stream = nullptr;  // the stream of the API call running on this thread (DevBuf::ensure fills fresh buffers on it)

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
            int trunk = -1;
            for (size_t t = 0; t < net->trunk_idx.size(); ++t)
                if (net->dec[net->trunk_idx[t]].name == d.name) trunk = (int)t;
            if (trunk < 0) {
                trunk = (int)net->trunk_idx.size();
                net->trunk_idx.push_back(i);
            } else {
                for (int e : net->dense_idx)
                    if (net->dec[e].name == d.name && net->dec[e].head == d.head) { delete net; return fail("decoder " + d.name + ": head " + d.head + " listed twice"); }
            }
            net->dense_idx.push_back(i);
            net->trunk_of.push_back(trunk);
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


extern "C" int cerb_net_finalize(cerb_net* net) {
    if (!net) return fail("cerb_net_finalize: null handle");
    if (net->finalized) return 0;
    if (!net->fold_bn && net->trunk_idx.size() != net->dense_idx.size())
        return fail("cerb_net_finalize: a decoder with several output heads runs in inference only (the training tape sums one head's gradient per decoder)");
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
                for (int di : net->trunk_idx) {
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
int prof_begin(cerb_net* net, const std::string& name, const std::string& kernel, double flops, hipStream_t st) {
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
int prof_end(cerb_net* net, hipStream_t st) {
    if (!net->profiling) return 0;
    HIP_OK(hipEventRecord(net->prof[net->prof_n].e1, st));
    net->prof_n++;
    net->prof_open = false;
    return 0;
}

int train_wino2_fresh(cerb_net* net, const std::string& name, PackedConv& cm, int dgrad, hipStream_t st) {
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
int train_wino4_slot(cerb_net* net, const std::string& name, PackedConv& cm, int w4b, int dgrad, hipStream_t st, float** out) {
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

int run_conv(cerb_net* net, const std::string& name, const float* in, const float* prev, const float* resid, float* out, int N,
             int H, int W, int relu, int mode, long long in_gs, long long prev_gs, hipStream_t st, double* macs,
             const int* roi, long long planar_out_gs) {  // planar_out_gs > 0: in / out are tile-planar (conv_wino4p.hip)
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
        const char* e = cerb_dev_getenv("CERB_W4B_MAX_PX");
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
    const size_t D = net->trunk_idx.size();  // groups of the grouped decoder launches (one per decoder TRUNK; a trunk may feed several heads)
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
        if (net->x0.ensure((size_t)N * H * W * 64 * 4, guard) || net->pool.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard)) return fail_alloc();
        for (int i = 1; i < 5; ++i)
            if (net->x[i].ensure((size_t)N * hs[i] * ws[i] * kFilters[i] * 4, guard)) return fail_alloc();
        if (net->ta.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) || net->tb.ensure((size_t)N * hs[1] * ws[1] * 64 * 4, guard) ||
            net->cm.ensure((size_t)N * hs[4] * ws[4] * 256 * 4, guard))
            return fail_alloc();
        if (D) {
            // dsum / dmid hold the entry sum and the first conv's output of a level that runs NHWC (per decoder 32^2 x 256, 64^2 x 128, and -- when the
            // tile-planar levels are off -- 128^2 x 64, 256^2 x 64), dout[u] its second conv's output; the planar levels have their own buffers, so with
            // them on (the default) only the two coarse levels count here: 8 GB less per handle at 32 tiles of 256^2 (a first attempt in this
            // round faulted on a wrong channel count in its own formula, not on a kernel reaching past its tensor: sized from the packed convolutions'
            // channel counts below, every fixture and geometry of the GPU suite runs).  CERB_LEGACY_WS=1: everything at last-level size, as before.
            static const bool exact_ws = cerb_dev_getenv("CERB_LEGACY_WS") == nullptr;
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
            if (net->dmid.ensure(need_mid, guard)) return fail_alloc();
            if (net->conv_algo && net->dsum.ensure(need_sum, guard)) return fail_alloc();
            for (int u = 0; u < 4; ++u) {
                if (exact_ws && level_is_planar(u)) continue;  // its outputs live in the planar buffers
                if (net->dout[u].ensure(D * (size_t)N * hs[3 - u] * ws[3 - u] * oc[u] * 4, guard)) return fail_alloc();
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
                if (bs.ensure((int)D, N, hh, ww, 64, st) || bm.ensure((int)D, N, hh, ww, 64, st)) return fail_alloc();
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
        for (size_t k = 0; k < net->dense_idx.size(); ++k) {
            const int di = net->dense_idx[k];
            const size_t tk = (size_t)net->trunk_of[k];  // the decoder whose features this head reads
            const DecoderCfg& d = net->dec[di];
            if (macs) *macs += (double)N * H * W * (64.0 * 96 + 96.0 * d.out_ch);
            const bool want = io->out && io->out[di];
            const bool wantl = io->logits && io->logits[di];
            if (dry || !(want || wantl)) continue;
            HeadParams hp;
            memset(&hp, 0, sizeof(hp));
            hp.feat = feat_planar ? net->psum.b.p + tk * net->psum.gs() : net->dout[3].p + tk * (size_t)N * H * W * 64;  // (planar: the last level's output lives in its sum buffer)
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
    static const bool per_conv = cerb_dev_getenv("CERB_PACK_PER_CONV") != nullptr;
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
