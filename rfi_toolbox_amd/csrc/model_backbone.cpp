// ResNet-50-FPN backbone of the Mask R-CNN path (BASELINE.json configs[3]; arch 5).  NOT in the
// reference (no detector there) and torchvision is absent: builder-defined as the published networks (He et al. 2016,
// ResNet-50 with the stride on the 3x3 conv of a Bottleneck; Lin et al. 2017, FPN) with the layer names and the frozen
// BatchNorm of the usual detection backbone; oracle/backbone_ref.py holds it as plain torch.nn modules.
//
//   body.conv1 7x7/2 (bias=False) + bn1 + ReLU -> MaxPool(3, 2, 1)                                  H/4, w channels
//   body.layer1..4: [3, 4, 6, 3] Bottlenecks (1x1 -> 3x3(stride) -> 1x1, x4 expansion, projection shortcut in the first
//                   block of a stage), strides 1, 2, 2, 2                                            C2..C5: 4w, 8w, 16w, 32w
//   fpn.inner_blocks.i 1x1 (+bias), top-down nearest-2x merge, fpn.layer_blocks.i 3x3 (+bias)         P2..P5: F channels
//   P6 = P5[:, ::2, ::2]                                                                             (LastLevelMaxPool)
//
// Every BatchNorm is FROZEN (per-channel affine from its buffers, as detection fine-tuning does): a conv's raw output is
// stored and the affine (+ ReLU) is applied by its consumers' loads, exactly like the train-mode BatchNorm of the U-Net
// but with constant coefficients, so the backward pass needs no statistics: dY = dA * scale * [z > 0].  Trainable: every
// conv weight and the FPN biases (flat parameter buffer; clip + Adam as everywhere).  The 7x7 stem (3 input channels)
// runs as a GEMM on its K-packed (im2col) input: K = 147 -> 160; on the direct VALU kernels it took 3.8 of 18.5 ms of
// the forward + backward pass at batch 64 x 128^2.  Stride-2 3x3 convs and projections use the space-to-depth forms of
// resnet_kernels.hip; everything else is the MFMA conv / split weight-gradient kernels.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

namespace {
const int kBlocksPerStage[4] = {3, 4, 6, 3};
}

void rfi_model::build_backbone() {
    RFI_REQUIRE(in_ch > 0 && feat > 0 && feat % 4 == 0 && out_ch > 0 && out_ch % 4 == 0,
                "ResNet50FPN: base width and FPN channels must be positive multiples of 4");
    depth = 4;
    const int w0 = feat, F = out_ch;
    convs.clear();
    ups.clear();
    bb.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    auto add = [&](const std::string& cname, const std::string& bname, int cin, int cout, int R, int stride, int lvl, bool bn,
                   bool bias) {
        ConvBN c;
        c.conv_name = cname;
        c.bn_name = bname;
        c.cin = c.cin_p = cin;
        c.cout = cout;
        c.R = R; c.stride = stride; c.level = lvl;
        c.has_bn = bn; c.has_bias = bias;
        c.w_off = off; off = align4(off + (size_t)R * R * cin * cout);
        c.b_off = off; off = align4(off + cout);
        c.g_off = c.be_off = 0;
        chan_floats += align4((size_t)8 * cout);
        wd_floats += align4((size_t)R * R * cin * cout);
        convs.push_back(c);
        return (int)convs.size() - 1;
    };
    add("body.conv1", "body.bn1", in_ch, w0, 7, 2, 1, true, false);          // level = log2 of the OUTPUT stride
    int cin = w0;
    for (int s = 0; s < 4; ++s) {
        const int width = w0 << s, cout = 4 * width;
        for (int b = 0; b < kBlocksPerStage[s]; ++b) {
            BBlock k;
            const std::string p = "body.layer" + std::to_string(s + 1) + "." + std::to_string(b);
            k.stage = s;
            k.stride = (b == 0 && s > 0) ? 2 : 1;
            k.cin = cin; k.width = width; k.cout = cout;
            k.lvl_in = k.stride == 2 ? s + 1 : s + 2;      // resolution H >> lvl: layer1 at 2, layer2 at 3, ...
            k.lvl = s + 2;
            k.c1 = add(p + ".conv1", p + ".bn1", cin, width, 1, 1, k.lvl_in, true, false);
            k.c2 = add(p + ".conv2", p + ".bn2", width, width, 3, k.stride, k.lvl, true, false);
            k.c3 = add(p + ".conv3", p + ".bn3", width, cout, 1, 1, k.lvl, true, false);
            if (b == 0) k.cd = add(p + ".downsample.0", p + ".downsample.1", cin, cout, 1, k.stride, k.lvl, true, false);
            bb.push_back(k);
            cin = cout;
        }
    }
    for (int i = 0; i < 4; ++i)
        fpn_inner[i] = add("fpn.inner_blocks." + std::to_string(i) + ".0", "", 4 * (w0 << i), F, 1, 1, i + 2, false, true);
    for (int i = 0; i < 4; ++i)
        fpn_layer[i] = add("fpn.layer_blocks." + std::to_string(i) + ".0", "", F, F, 3, 1, i + 2, false, true);
    head_w_off = head_b_off = off;
    n_flat = off;

    entries.clear();
    entry_index.clear();
    n_params = 0;
    auto push = [&](Entry e, bool param) {
        entry_index[e.name] = (int)entries.size();
        if (param) n_params += e.numel();
        entries.push_back(e);
    };
    for (int ci = 0; ci < (int)convs.size(); ++ci) {
        const ConvBN& c = convs[ci];
        Entry e;
        e.layer = ci;
        e.name = c.conv_name + ".weight"; e.ndim = 4; e.dims[0] = c.cout; e.dims[1] = c.cin; e.dims[2] = c.R; e.dims[3] = c.R;
        e.kind = c.R == 1 ? 7 : 0;
        push(e, true);
        e = Entry(); e.layer = ci; e.ndim = 1; e.dims[0] = c.cout;
        if (c.has_bias) { e.name = c.conv_name + ".bias"; e.kind = 2; e.which = 0; push(e, true); }
        if (c.has_bn) {          // frozen: buffers, not parameters (kinds 8 / 9 / 3 / 4 live in the per-channel state)
            e.name = c.bn_name + ".weight"; e.kind = 8; push(e, false);
            e.name = c.bn_name + ".bias"; e.kind = 9; push(e, false);
            e.name = c.bn_name + ".running_mean"; e.kind = 3; push(e, false);
            e.name = c.bn_name + ".running_var"; e.kind = 4; push(e, false);
        }
    }

    ctx->activate();
    const size_t bytes = n_flat * sizeof(float);
    params = static_cast<float*>(ctx->alloc(bytes));
    grads = static_cast<float*>(ctx->alloc(bytes));
    adam_m = static_cast<float*>(ctx->alloc(bytes));
    adam_v = static_cast<float*>(ctx->alloc(bytes));
    chan_pool = static_cast<float*>(ctx->alloc(chan_floats * sizeof(float)));
    wd_pool = static_cast<float*>(ctx->alloc(wd_floats * sizeof(float)));
    d_sums = static_cast<double*>(ctx->alloc(8 * sizeof(double)));
    d_scalars = static_cast<float*>(ctx->alloc(8 * sizeof(float)));
    for (float* p : {params, grads, adam_m, adam_v}) RFI_CHECK_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(d_sums, 0, 8 * sizeof(double), ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(d_scalars, 0, 8 * sizeof(float), ctx->stream));
    size_t co = 0, wo = 0;
    for (auto& c : convs) {
        c.chan = chan_pool + co; co += align4((size_t)8 * c.cout);
        c.wd = wd_pool + wo; wo += align4((size_t)c.R * c.R * c.cin * c.cout);
    }
    // 2x2 forms of the stride-2 3x3 filters, 4x tiled affine coefficients of their inputs, identity vectors
    const size_t cmax = (size_t)32 * w0;
    size_t need = 2 * cmax + 16;
    // (+ the 3 x bf16 records of those forms and of the K-packed stem filters: a launch without them splits a temporary copy and
    // drains the stream to free it -- eight stalls per step, 0.2 to several ms each in the rocprofv3 timeline of the detector)
    for (auto& k : bb)
        if (k.stride == 2)
            need += 2 * align4((size_t)16 * k.width * k.width) + 2 * align4((size_t)4 * k.width) +
                    align4(weights_x3_floats(4, k.width, 4 * k.width)) + align4(weights_x3_floats(4, 4 * k.width, k.width));
    need += align4(weights_x3_floats(1, w0, stem_kp())) + 16;
    rs_wpool = static_cast<float*>(ctx->alloc(need * sizeof(float)));
    size_t o = 0;
    rs_ones = rs_wpool + o; o += cmax;
    rs_zeros = rs_wpool + o; o += cmax;
    for (auto& k : bb)
        if (k.stride == 2) {
            ConvBN& c = convs[k.c2];
            c.ws2d = rs_wpool + o; o += align4((size_t)16 * k.width * k.width);
            c.wds2d = rs_wpool + o; o += align4((size_t)16 * k.width * k.width);
            k.sc4 = rs_wpool + o; o += align4((size_t)4 * k.width);
            k.sh4 = rs_wpool + o; o += align4((size_t)4 * k.width);
            c.ws2d3 = rs_wpool + o; o += align4(weights_x3_floats(4, k.width, 4 * k.width));
            c.wds2d3 = rs_wpool + o; o += align4(weights_x3_floats(4, 4 * k.width, k.width));
        }
    bb_stem_w3 = rs_wpool + o; o += align4(weights_x3_floats(1, w0, stem_kp()));
    {
        std::vector<float> h(2 * cmax, 0.0f);
        for (size_t i = 0; i < cmax; ++i) h[i] = 1.0f;
        RFI_CHECK_HIP(hipMemcpyAsync(rs_ones, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    adam_step = 0;
    wd_dirty = true;
    x3_fresh = false;
    reset_channel_state();            // running stats 0 / 1; BN-less layers: scale 1, shift 0
    for (auto& c : convs)             // frozen gamma = 1, beta = 0 until loaded (slots 2 and 3 of the per-channel state)
        if (c.has_bn) {
            std::vector<float> g((size_t)2 * c.cout, 0.0f);
            for (int i = 0; i < c.cout; ++i) g[i] = 1.0f;
            RFI_CHECK_HIP(hipMemcpyAsync(c.mean(), g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        }
    frozen_dirty = true;
}

void rfi_model::prepare_backbone(int n, int h, int w) {
    RFI_REQUIRE(h % 64 == 0 && w % 64 == 0, "ResNet50FPN: H and W must be multiples of 64");
    if (n == pN && h == pH && w == pW && !bufs.empty()) return;
    ctx->activate();
    const int F = out_ch;
    if (bufs.empty()) {
        bY0 = new_buf(); bP0 = new_buf(); bArg = new_buf();
        for (auto& k : bb) {
            k.Y1 = new_buf(); k.Y2 = new_buf(); k.Y3 = new_buf(); k.A = new_buf();
            if (k.cd >= 0) k.Yd = new_buf();
            if (k.stride == 2) { k.xs1 = new_buf(); k.xsA = new_buf(); }
        }
        for (int i = 0; i < 4; ++i) { fL[i] = new_buf(); fM[i] = new_buf(); fP[i] = new_buf(); fdM[i] = new_buf(); fdP[i] = new_buf(); }
        fP6 = new_buf(); fdP6 = new_buf();
        for (int i = 0; i < 6; ++i) bG[i] = new_buf();
        for (int r = 0; r < 4; ++r) for (int p = 0; p < 3; ++p) bT[r][p] = new_buf();
        bdW = new_buf(); bS = new_buf(); bCol = new_buf(); bWp = new_buf();
        x_stage = new_buf(); x_stage2 = new_buf(); x_pad = new_buf(); out_stage = new_buf();
        ws_red = new_buf(); ws_slab = new_buf(); lab_stage = new_buf(); logits = new_buf(); dlogits = new_buf();
    }
    auto px = [&](int lvl) { return (size_t)n * (h >> lvl) * (w >> lvl); };
    bufs[bY0].ensure(ctx, px(1) * feat);
    bufs[bP0].ensure(ctx, px(2) * feat);
    bufs[bArg].ensure(ctx, px(2) * feat / 4 + 4);
    size_t gmax = px(1) * feat, wmax = 16, slab_need = 0;
    for (auto& k : bb) {
        bufs[k.Y1].ensure(ctx, px(k.lvl_in) * k.width);
        bufs[k.Y2].ensure(ctx, px(k.lvl) * k.width);
        bufs[k.Y3].ensure(ctx, px(k.lvl) * k.cout);
        bufs[k.A].ensure(ctx, px(k.lvl) * k.cout);
        if (k.cd >= 0) bufs[k.Yd].ensure(ctx, px(k.lvl) * k.cout);
        if (k.stride == 2) {
            bufs[k.xs1].ensure(ctx, px(k.lvl) * 4 * k.width);
            bufs[k.xsA].ensure(ctx, px(k.lvl) * 4 * k.cin);
            wmax = std::max(wmax, (size_t)16 * k.width * k.width);
        }
        gmax = std::max({gmax, px(k.lvl_in) * k.width, px(k.lvl_in) * k.cin, px(k.lvl) * k.cout, px(k.lvl) * 4 * k.width});
    }
    for (int i = 0; i < 4; ++i) {
        const size_t m = px(i + 2) * F;
        for (int b : {fL[i], fM[i], fP[i], fdM[i], fdP[i]}) bufs[b].ensure(ctx, m);
        gmax = std::max(gmax, m);
    }
    bufs[fP6].ensure(ctx, px(6) * F);
    bufs[fdP6].ensure(ctx, px(6) * F);
    for (int i = 0; i < 6; ++i) bufs[bG[i]].ensure(ctx, gmax);
    for (int r = 0; r < 4; ++r) for (int p = 0; p < 3; ++p) bufs[bT[r][p]].ensure(ctx, gmax);
    bufs[bS].ensure(ctx, gmax);
    bufs[bCol].ensure(ctx, px(1) * stem_kp());
    bufs[bWp].ensure(ctx, (size_t)feat * stem_kp() + 16);
    bufs[bdW].ensure(ctx, wmax + 16);
    bufs[x_stage].ensure(ctx, (size_t)n * h * w * in_ch);
    bufs[x_stage2].ensure(ctx, (size_t)n * h * w * in_ch);
    for (int b : {x_pad, out_stage, lab_stage, logits, dlogits}) bufs[b].ensure(ctx, 16);
    size_t red_need = sumsq_ws_doubles((int64_t)n_flat) * 2;
    const int cmax = 32 * feat;
    red_need = std::max({red_need, bn_bwd_ws_floats((int64_t)px(1), cmax), channel_sum_ws_floats((int64_t)px(2), std::max(cmax, F))});
    bufs[ws_red].ensure(ctx, red_need + 16);
    for (auto& c : convs) {
        WgradArgs a;
        a.N = n; a.H = h >> c.level; a.W = w >> c.level; a.Hx = a.H; a.Wx = a.W;
        a.Cx = c.cin; a.Cy = c.cout;
        a.R = c.R; a.S = 1; a.pad = c.R / 2;
        if (c.R == 3 && c.stride == 2) { a.Cx *= 4; a.R = 2; a.pad = 1; }
        if (c.R == 7) { a.R = 1; a.pad = 0; a.Cx = stem_kp(); }          // the K-packed stem
        a.xop.pstride = a.Cx; a.yop.pstride = a.Cy;
        a.tap_stride = (int64_t)a.Cx * a.Cy;
        a.bf16x3 = true;
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
        const int64_t M = (int64_t)a.N * a.H * a.W;
        if (a.R == 1 && a.W < 32 && M % 32 == 0) {      // the flattened layout wgrad() runs small 1x1 maps in
            a.N = 1; a.H = a.Hx = (int)(M / 32); a.W = a.Wx = 32;
            slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
        }
    }
    bufs[ws_slab].ensure(ctx, slab_need + 16);
    pN = n; pH = h; pW = w;
}

// frozen BatchNorm coefficients and the derived filter copies the batched relayout does not cover
void rfi_model::refresh_backbone() {
    if (frozen_dirty) {
        for (auto& c : convs)
            if (c.has_bn) launch_bn_eval_coeffs(ctx, c.cout, c.mean(), c.invstd(), c.running_mean(), c.running_var(), c.scale(), c.shift());
        for (auto& k : bb)
            if (k.stride == 2) {                  // the 2x2 conv reads the space-to-depth of conv1's RAW output: coefficients x 4
                const ConvBN& c1 = convs[k.c1];
                for (int r = 0; r < 4; ++r) {
                    RFI_CHECK_HIP(hipMemcpyAsync(k.sc4 + r * k.width, c1.scale(), k.width * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
                    RFI_CHECK_HIP(hipMemcpyAsync(k.sh4 + r * k.width, c1.shift(), k.width * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
        frozen_dirty = false;
    }
    for (auto& k : bb)
        if (k.stride == 2) {
            ConvBN& c = convs[k.c2];
            launch_w_s2d(ctx, params + c.w_off, c.cout, c.cin, c.ws2d, true);
            launch_weight_to_dgrad(ctx, c.ws2d, 4, c.cout, 4 * c.cin, 1, c.wds2d);
            if (use_w3()) {
                launch_weights_to_x3(ctx, c.ws2d, 4, c.cout, 4 * c.cin, c.ws2d3);
                launch_weights_to_x3(ctx, c.wds2d, 4, 4 * c.cin, c.cout, c.wds2d3);
            }
        }
}

namespace {

struct Sh { int N, H, W; };

void conv(rfi_model* m, View in, InXform xf, Sh s, int Hin, int Win, const float* w, const float* w3, const float* bias, int cin,
          int cout, int R, int S, int pad, float* Y) {
    // a 1x1 conv does not see the image structure: on small maps (W < 32) run it on the same pixels laid out as one
    // [M / 32] x 32 image, so that the 32-pixel-wide tiles of the kernels are full instead of mostly padding
    const int64_t M = (int64_t)s.N * s.H * s.W;
    if (R == 1 && S == 1 && Hin == s.H && Win == s.W && s.W < 32 && M % 32 == 0) {
        s = Sh{1, (int)(M / 32), 32};
        Hin = s.H; Win = s.W;
    }
    ConvArgs a;
    a.x = in;
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = Hin; a.Win = Win;
    a.Cin = cin; a.Cout = cout;
    a.w = w;
    a.w3 = m->use_w3() ? w3 : nullptr;
    m->ws_set(a);
    a.bias = bias;
    a.y = MutView{Y, cout};
    a.Hout = s.H; a.Wout = s.W;
    a.R = R; a.S = S; a.pad = pad;
    a.xf = xf;
    a.bf16 = m->compute_bf16;
    a.bf16x3 = m->compute_x3;
    launch_conv(m->ctx, a);
}
// RAII: launches between construction and end() go to the side stream (model.cpp, side_begin / side_end)
struct SideScopeB {
    rfi_model* m;
    bool ended = false;
    explicit SideScopeB(rfi_model* model) : m(model) { m->side_begin(); }
    void end() { m->side_end(); ended = true; }
    ~SideScopeB() { if (!ended) m->ctx->stream = m->ctx->main_stream; }
};

// (the caller decides the stream: backward_backbone puts the weight gradients on the side stream)
void wgrad(rfi_model* m, View x, InXform xf_x, const float* dY, int cy, int cx, Sh s, int Hx, int Wx, int R, int S, int pad, float* dw) {
    const int64_t M = (int64_t)s.N * s.H * s.W;
    if (R == 1 && S == 1 && Hx == s.H && Wx == s.W && s.W < 32 && M % 32 == 0) {       // (as in conv())
        s = Sh{1, (int)(M / 32), 32};
        Hx = s.H; Wx = s.W;
    }
    WgradArgs wa;
    wa.xop = x;
    wa.yop = View{dY, cy};
    wa.xf_x = xf_x;
    wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = Hx; wa.Wx = Wx;
    wa.Cx = cx; wa.Cy = cy;
    wa.R = R; wa.S = S; wa.pad = pad;
    wa.dw = dw;
    wa.tap_stride = (int64_t)cx * cy;
    wa.sy = cx; wa.sx = 1;
    wa.slab = m->buf(m->ws_slab);
    wa.slab_floats = m->bufs[m->ws_slab].n;
    wa.bf16 = m->compute_bf16;
    wa.bf16x3 = m->compute_x3;
    launch_wgrad(m->ctx, wa);
}
// dA (gradient w.r.t. act(Y * scale + shift), or w.r.t. the affine output when slope = 1) -> dY in place: dA * scale * act'
void affine_bwd(rfi_model* m, const ConvBN& c, float* dA, const float* Y, int64_t M, float slope) {
    launch_bn_bwd_apply(m->ctx, dA, Y, M, c.cout, c.scale(), c.shift(), m->rs_zeros, m->rs_ones, c.scale(), m->rs_zeros, m->rs_zeros,
                        m->buf(m->ws_red), nullptr, slope);
}
InXform act_of(const ConvBN& c) { return InXform{c.scale(), c.shift(), 1, 0.0f}; }

}  // namespace

void rfi_model::forward_backbone(const float* x_dev, int n, int h, int w) {
    refresh_dgrad_weights();
    refresh_backbone();
    {   // stem: 7x7 / 2 as a GEMM on its K-packed input (K = 49 C padded to a multiple of 16), then max-pool of the activated output
        ConvBN& c = convs[0];
        const int Kp = stem_kp();
        launch_im2col(ctx, x_dev, n, h, w, in_ch, 7, 2, 3, h / 2, w / 2, Kp, buf(bCol));
        launch_w_pack(ctx, params + c.w_off, 49, c.cout, in_ch, Kp, buf(bWp), true);
        if (use_w3()) launch_weights_to_x3(ctx, buf(bWp), 1, c.cout, Kp, bb_stem_w3);
        conv(this, View{buf(bCol), Kp}, InXform{}, Sh{n, h / 2, w / 2}, h / 2, w / 2, buf(bWp), bb_stem_w3, nullptr, Kp, c.cout, 1, 1, 0, buf(bY0));
        launch_maxpool3_fwd(ctx, buf(bY0), n, h / 2, w / 2, c.cout, c.scale(), c.shift(), buf(bP0), reinterpret_cast<unsigned*>(buf(bArg)));
    }
    const float* a_in = buf(bP0);
    for (auto& k : bb) {
        ConvBN &c1 = convs[k.c1], &c2 = convs[k.c2], &c3 = convs[k.c3];
        const Sh si{n, h >> k.lvl_in, w >> k.lvl_in}, so{n, h >> k.lvl, w >> k.lvl};
        conv(this, View{a_in, k.cin}, InXform{}, si, si.H, si.W, params + c1.w_off, c1.w3, nullptr, k.cin, k.width, 1, 1, 0, buf(k.Y1));
        if (k.stride == 2) {
            launch_s2d(ctx, buf(k.Y1), n, si.H, si.W, k.width, buf(k.xs1));
            conv(this, View{buf(k.xs1), 4 * k.width}, InXform{k.sc4, k.sh4, 1, 0.0f}, so, so.H, so.W, c2.ws2d, c2.ws2d3, nullptr, 4 * k.width,
                 k.width, 2, 1, 1, buf(k.Y2));
        } else {
            conv(this, View{buf(k.Y1), k.width}, act_of(c1), so, so.H, so.W, params + c2.w_off, c2.w3, nullptr, k.width, k.width, 3, 1, 1,
                 buf(k.Y2));
        }
        conv(this, View{buf(k.Y2), k.width}, act_of(c2), so, so.H, so.W, params + c3.w_off, c3.w3, nullptr, k.width, k.cout, 1, 1, 0, buf(k.Y3));
        const int64_t M = (int64_t)so.N * so.H * so.W;
        if (k.cd >= 0) {
            ConvBN& cd = convs[k.cd];
            if (k.stride == 2) {
                launch_s2d(ctx, a_in, n, si.H, si.W, k.cin, buf(k.xsA));
                conv(this, View{buf(k.xsA), 4 * k.cin}, InXform{}, so, so.H, so.W, params + cd.w_off, cd.w3, nullptr, k.cin, k.cout, 1, 1, 0,
                     buf(k.Yd));
            } else {
                conv(this, View{a_in, k.cin}, InXform{}, so, so.H, so.W, params + cd.w_off, cd.w3, nullptr, k.cin, k.cout, 1, 1, 0, buf(k.Yd));
            }
            launch_bn_add_relu(ctx, buf(k.Y3), c3.scale(), c3.shift(), buf(k.Yd), cd.scale(), cd.shift(), M, k.cout,
                               MutView{buf(k.A), k.cout}, MutView{});
        } else {
            launch_bn_add_relu(ctx, buf(k.Y3), c3.scale(), c3.shift(), a_in, nullptr, nullptr, M, k.cout, MutView{buf(k.A), k.cout}, MutView{});
        }
        a_in = buf(k.A);
    }
    // FPN: laterals, top-down merge, output convs, extra level
    const int F = out_ch;
    int last[4], bi = -1;
    for (int s = 0; s < 4; ++s) { bi += kBlocksPerStage[s]; last[s] = bi; }
    for (int i = 3; i >= 0; --i) {
        const BBlock& k = bb[last[i]];
        const Sh s{n, h >> (i + 2), w >> (i + 2)};
        ConvBN& ci = convs[fpn_inner[i]];
        conv(this, View{buf(k.A), k.cout}, InXform{}, s, s.H, s.W, params + ci.w_off, ci.w3, params + ci.b_off, k.cout, F, 1, 1, 0, buf(fL[i]));
        if (i == 3) launch_copy_d2d(ctx, buf(fM[i]), buf(fL[i]), (size_t)s.N * s.H * s.W * F * sizeof(float));
        else launch_fpn_merge_fwd(ctx, buf(fL[i]), buf(fM[i + 1]), s.N, s.H, s.W, F, buf(fM[i]));
        ConvBN& cl = convs[fpn_layer[i]];
        conv(this, View{buf(fM[i]), F}, InXform{}, s, s.H, s.W, params + cl.w_off, cl.w3, params + cl.b_off, F, F, 3, 1, 1, buf(fP[i]));
    }
    launch_subsample2(ctx, buf(fP[3]), n, h >> 5, w >> 5, F, buf(fP6));
}

// dP2..dP6 sit in fdP[0..3] / fdP6 (rfi_backbone_backward copies them there)
void rfi_model::backward_backbone(const float* x_dev, int n, int h, int w) {
    refresh_dgrad_weights();
    static const int bound_env = getenv("RFI_BB_SIDE_BOUND") ? atoi(getenv("RFI_BB_SIDE_BOUND")) : 2;     // (1..5: A/B runs; measured 31.8 / 32.2 / 32.9 ms per
                                                                                                    // step of the detector at bound 2 / bound 5 / no side stream)
    side_bound = std::min(5, std::max(1, bound_env));
    const int F = out_ch;
    int last[4], bi = -1;
    for (int s = 0; s < 4; ++s) { bi += kBlocksPerStage[s]; last[s] = bi; }
    float* ws = buf(ws_red);
    // ---- FPN.  dM_i = dgrad(layer_block_i)(dP_i) + nearest-2x adjoint of dM_{i-1}; dL_i = dM_i; dC_i = dgrad(inner_i)(dM_i)
    launch_subsample2_bwd_add(ctx, buf(fdP6), n, h >> 5, w >> 5, F, buf(fdP[3]));
    float* dC[4] = {buf(bG[2]), buf(bG[3]), buf(bG[4]), buf(bG[5])};
    for (int i = 0; i < 4; ++i) {                   // fine to coarse: dM_i needs dM_{i-1}
        const Sh s{n, h >> (i + 2), w >> (i + 2)};
        const int64_t M = (int64_t)s.N * s.H * s.W;
        ConvBN& cl = convs[fpn_layer[i]];
        launch_channel_sum(ctx, View{buf(fdP[i]), F}, M, F, ws, grads + cl.b_off);
        { SideScopeB side(this); wgrad(this, View{buf(fM[i]), F}, InXform{}, buf(fdP[i]), F, F, s, s.H, s.W, 3, 1, 1, grads + cl.w_off); side.end(); }
        conv(this, View{buf(fdP[i]), F}, InXform{}, s, s.H, s.W, cl.wd, cl.wd3, nullptr, F, F, 3, 1, 1, buf(fdM[i]));
        if (i > 0) {                                // + the share of M_{i-1} = L_{i-1} + up(M_i)
            launch_fpn_merge_bwd_top(ctx, buf(fdM[i - 1]), n, h >> (i + 1), w >> (i + 1), F, buf(bG[0]));
            launch_add_inplace(ctx, buf(fdM[i]), buf(bG[0]), M * F);
        }
        const BBlock& k = bb[last[i]];
        ConvBN& ci = convs[fpn_inner[i]];
        launch_channel_sum(ctx, View{buf(fdM[i]), F}, M, F, ws, grads + ci.b_off);
        { SideScopeB side(this); wgrad(this, View{buf(k.A), k.cout}, InXform{}, buf(fdM[i]), F, k.cout, s, s.H, s.W, 1, 1, 0, grads + ci.w_off); side.end(); }
        conv(this, View{buf(fdM[i]), F}, InXform{}, s, s.H, s.W, ci.wd, ci.wd3, nullptr, F, k.cout, 1, 1, 0, dC[i]);
    }
    // ---- body, last block first.  gout: gradient w.r.t. the block output A
    float* gout = buf(bG[0]);
    float* gother = buf(bG[1]);
    {
        const BBlock& k = bb.back();
        launch_copy_d2d(ctx, gout, dC[3], (size_t)n * (h >> 5) * (w >> 5) * k.cout * sizeof(float));
    }
    for (int b = (int)bb.size() - 1; b >= 0; --b) {
        BBlock& k = bb[b];
        ConvBN &c1 = convs[k.c1], &c2 = convs[k.c2], &c3 = convs[k.c3];
        const Sh si{n, h >> k.lvl_in, w >> k.lvl_in}, so{n, h >> k.lvl, w >> k.lvl};
        const int64_t Mo = (int64_t)so.N * so.H * so.W, Mi = (int64_t)si.N * si.H * si.W;
        const float* a_in = b == 0 ? buf(bP0) : buf(bb[b - 1].A);
        // The weight gradients run on the side stream next to this chain (its masks, affine backward passes, space-to-depth
        // forms and shortcut adds are memory-bound).  What they read -- dY3, dY2, dY1, dYd -- lives in buffers of its own by
        // block index mod 3: block b - 3 rewrites them, and with a run-ahead bound of up to 5 side launches the main stream has
        // waited for every weight gradient of block b by then (a block issues at least three: after block b - 2's last,
        // launch i + 8 counted from block b's first, it has waited for launch i + 3 or later)
        const int par = b % 3;
        float* t3 = buf(bT[0][par]);                // dz -> dY3            [Mo][cout]
        float* dA2 = buf(bT[1][par]);               // dA2 -> dY2           [Mo][width]
        float* dA1 = buf(bT[2][par]);               // dA1 -> dY1           [Mi][width]
        float* td = buf(bT[3][par]);                // dz -> dYd            [Mo][cout]
        // dz = gout * [A > 0] -> dY3 = dz * scale3
        launch_relu_mask(ctx, View{gout, k.cout}, View{}, View{buf(k.A), k.cout}, View{}, Mo, k.cout, t3);
        affine_bwd(this, c3, t3, buf(k.Y3), Mo, 1.0f);
        { SideScopeB side(this); wgrad(this, View{buf(k.Y2), k.width}, act_of(c2), t3, k.cout, k.width, so, so.H, so.W, 1, 1, 0, grads + c3.w_off); side.end(); }
        conv(this, View{t3, k.cout}, InXform{}, so, so.H, so.W, c3.wd, c3.wd3, nullptr, k.cout, k.width, 1, 1, 0, dA2);
        affine_bwd(this, c2, dA2, buf(k.Y2), Mo, 0.0f);                   // -> dY2
        if (k.stride == 2) {
            {
                SideScopeB side(this);              // (the 2x2-form gradient goes back to the 3x3 layout behind its slab reduction)
                wgrad(this, View{buf(k.xs1), 4 * k.width}, InXform{k.sc4, k.sh4, 1, 0.0f}, dA2, k.width, 4 * k.width, so, so.H, so.W, 2, 1, 1,
                      buf(bdW));
                launch_w_s2d(ctx, grads + c2.w_off, c2.cout, c2.cin, buf(bdW), false);
                side.end();
            }
            float* dXs = buf(bS);                   // scratch [Mo][4 width]
            conv(this, View{dA2, k.width}, InXform{}, so, so.H, so.W, c2.wds2d, c2.wds2d3, nullptr, k.width, 4 * k.width, 2, 1, 0, dXs);
            launch_d2s_add(ctx, dXs, nullptr, View{}, n, si.H, si.W, k.width, dA1);
        } else {
            { SideScopeB side(this); wgrad(this, View{buf(k.Y1), k.width}, act_of(c1), dA2, k.width, k.width, so, so.H, so.W, 3, 1, 1, grads + c2.w_off); side.end(); }
            conv(this, View{dA2, k.width}, InXform{}, so, so.H, so.W, c2.wd, c2.wd3, nullptr, k.width, k.width, 3, 1, 1, dA1);
        }
        affine_bwd(this, c1, dA1, buf(k.Y1), Mi, 0.0f);                   // -> dY1
        { SideScopeB side(this); wgrad(this, View{a_in, k.cin}, InXform{}, dA1, k.width, k.cin, si, si.H, si.W, 1, 1, 0, grads + c1.w_off); side.end(); }
        float* dX = gother;                                               // [Mi][cin]
        conv(this, View{dA1, k.width}, InXform{}, si, si.H, si.W, c1.wd, c1.wd3, nullptr, k.width, k.cin, 1, 1, 0, dX);
        if (k.cd >= 0) {
            ConvBN& cd = convs[k.cd];
            launch_relu_mask(ctx, View{gout, k.cout}, View{}, View{buf(k.A), k.cout}, View{}, Mo, k.cout, td);
            affine_bwd(this, cd, td, buf(k.Yd), Mo, 1.0f);                // -> dYd
            if (k.stride == 2) {
                { SideScopeB side(this); wgrad(this, View{buf(k.xsA), 4 * k.cin}, InXform{}, td, k.cout, k.cin, so, so.H, so.W, 1, 1, 0, grads + cd.w_off); side.end(); }
                float* dS = buf(bS);
                conv(this, View{td, k.cout}, InXform{}, so, so.H, so.W, cd.wd, cd.wd3, nullptr, k.cout, k.cin, 1, 1, 0, dS);
                launch_subsample2_bwd_add(ctx, dS, n, si.H, si.W, k.cin, dX);      // the projection saw x[:, ::2, ::2]
            } else {
                { SideScopeB side(this); wgrad(this, View{a_in, k.cin}, InXform{}, td, k.cout, k.cin, so, so.H, so.W, 1, 1, 0, grads + cd.w_off); side.end(); }
                float* dS = buf(bS);
                conv(this, View{td, k.cout}, InXform{}, so, so.H, so.W, cd.wd, cd.wd3, nullptr, k.cout, k.cin, 1, 1, 0, dS);
                launch_add_inplace(ctx, dX, dS, Mi * k.cin);
            }
        } else {                                    // identity shortcut: + gout * [A > 0]
            launch_relu_mask(ctx, View{gout, k.cout}, View{}, View{buf(k.A), k.cout}, View{dX, k.cin}, Mo, k.cout, dX);
        }
        // dX is the gradient w.r.t. this block's input = the previous block's output (or the pooled stem output); a stage
        // boundary also receives the lateral connection's gradient
        if (b > 0 && bb[b - 1].stage != k.stage) launch_add_inplace(ctx, dX, dC[bb[b - 1].stage], Mi * k.cin);
        std::swap(gout, gother);
    }
    {   // stem: gout = gradient w.r.t. the pooled tensor
        ConvBN& c = convs[0];
        const int H2 = h / 2, W2 = w / 2;
        float* dA0 = gother;
        launch_maxpool3_bwd(ctx, gout, reinterpret_cast<const unsigned*>(buf(bArg)), n, H2, W2, c.cout, dA0);
        affine_bwd(this, c, dA0, buf(bY0), (int64_t)n * H2 * W2, 0.0f);
        (void)x_dev;                                  // (its K-packed copy from the forward pass is the operand)
        const int Kp = stem_kp();
        {
            SideScopeB side(this);
            wgrad(this, View{buf(bCol), Kp}, InXform{}, dA0, c.cout, Kp, Sh{n, H2, W2}, H2, W2, 1, 1, 0, buf(bWp));
            launch_w_pack(ctx, grads + c.w_off, 49, c.cout, in_ch, Kp, buf(bWp), false);
            side.end();
        }
    }
    side_join();
}
