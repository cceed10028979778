// U-Net with a ResNet-18-style encoder (BASELINE.json configs[2]).  NOT in the reference (it
// ships plain U-Nets only, models/unet.py) and neither torchvision nor segmentation_models_pytorch exist in this
// image: builder-defined, oracle/resnet_unet_ref.py, parity unpinned by the reference.
//
//   stem      Conv3x3(in -> f, bias=False) + BN + ReLU                                  at H
//   layer l   BasicBlock(c_{l-1} -> c_l, stride s_l) + BasicBlock(c_l -> c_l),  c_l = f 2^(l-1), s_1 = 1, s_l = 2
//             BasicBlock: conv3x3(s) + BN + ReLU + conv3x3 + BN, + shortcut (identity, or Conv1x1(s=2) + BN), ReLU
//   then      the reference's own bottleneck (MaxPool2 + DoubleConv), decoders and 1x1 head (models/unet.py:30-77)
//             fed by the four stage outputs as skip connections (model.cpp).
// The ResNet-18 topology (four stages of two BasicBlocks, widths f..8f, stride-2 3x3 + projection shortcuts between
// stages) with a full-resolution stem instead of 7x7/2 + max-pool, as ResNets for small images use: the decoder of
// this path wants a full-resolution skip, and RFI patches are 128 x 128.
//
// Stride-2 3x3 convolutions run as 2x2 / stride-1 convolutions on the space-to-depth input, the projection as a 1x1
// convolution on a channel slice of it (resnet_kernels.hip).  The ReLU after the residual add is applied by ONE
// elementwise kernel that materialises the block output (the next block needs it as its identity shortcut anyway);
// its backward masks by the stored output and hands the same dz to both BatchNorm branches.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

void rfi_model::build_resnet() {
    RFI_REQUIRE(in_ch > 0 && out_ch > 0 && feat > 0 && feat % 4 == 0, "UNetResNet18: init_features must be a positive multiple of 4");
    depth = 4;
    const int D = depth;
    convs.clear();
    ups.clear();
    blocks.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    auto add = [&](const std::string& cname, const std::string& bname, int cin, int cout, int R, int stride, int lvl, bool bias) {
        ConvBN c;
        c.conv_name = cname;
        c.bn_name = bname;
        c.cin = cin;
        c.cin_p = convs.empty() ? (int)align4((size_t)cin) : cin;
        c.cout = cout;
        c.R = R; c.stride = stride; c.level = lvl; c.has_bias = bias;
        c.ema_repeats = 1;
        c.w_off = off; off = align4(off + (size_t)R * R * c.cin_p * cout);
        c.b_off = off; off = align4(off + cout);
        c.g_off = off; off = align4(off + cout);
        c.be_off = off; off = align4(off + cout);
        chan_floats += align4((size_t)8 * cout);
        wd_floats += align4((size_t)R * R * c.cin_p * cout);
        convs.push_back(c);
        return (int)convs.size() - 1;
    };
    add("stem.0", "stem.1", in_ch, feat, 3, 1, 1, false);
    int cin = feat;
    for (int l = 1; l <= D; ++l) {
        const int cout = feat << (l - 1);
        for (int b = 0; b < 2; ++b) {
            ResBlock rb;
            const std::string p = "layer" + std::to_string(l) + "." + std::to_string(b);
            rb.stride = (b == 0 && l > 1) ? 2 : 1;
            rb.cin = cin; rb.cout = cout; rb.level = l;
            rb.c1 = add(p + ".conv1", p + ".bn1", cin, cout, 3, rb.stride, l, false);
            rb.c2 = add(p + ".conv2", p + ".bn2", cout, cout, 3, 1, l, false);
            if (rb.stride == 2) rb.cd = add(p + ".downsample.0", p + ".downsample.1", cin, cout, 1, 2, l, false);
            blocks.push_back(rb);
            cin = cout;
        }
    }
    i_bott = (int)convs.size();
    add("bottleneck.conv.0", "bottleneck.conv.1", cin, cin * 2, 3, 1, D + 1, true);
    add("bottleneck.conv.3", "bottleneck.conv.4", cin * 2, cin * 2, 3, 1, D + 1, true);
    cin *= 2;
    for (int l = D; l >= 1; --l) {
        const int cout = feat << (l - 1);
        UpConv u;
        u.name = "decoder" + std::to_string(l) + ".up";
        u.cin = cin;
        u.cout = cout;
        u.w_off = off; off = align4(off + (size_t)4 * cin * cout);
        u.b_off = off; off = align4(off + cout);
        wd_floats += align4((size_t)4 * cin * cout);
        ups.push_back(u);
        const std::string p = "decoder" + std::to_string(l) + ".conv.conv";
        add(p + ".0", p + ".1", cin, cout, 3, 1, l, true);
        add(p + ".3", p + ".4", cout, cout, 3, 1, l, true);
        cin = cout;
    }
    head_w_off = off; off = align4(off + (size_t)out_ch * feat);
    head_b_off = off; off = align4(off + out_ch);
    n_flat = off;

    // ---- state_dict entries in the oracle module's order
    entries.clear();
    entry_index.clear();
    n_params = 0;
    auto push = [&](Entry e) {
        entry_index[e.name] = (int)entries.size();
        if (e.kind == 0 || e.kind == 1 || e.kind == 2 || e.kind == 6 || e.kind == 7) n_params += e.numel();
        entries.push_back(e);
    };
    auto push_conv = [&](int ci) {
        const ConvBN& c = convs[ci];
        Entry e;
        e.layer = ci;
        e.name = c.conv_name + ".weight"; e.ndim = 4; e.dims[0] = c.cout; e.dims[1] = c.cin; e.dims[2] = c.R; e.dims[3] = c.R;
        e.kind = c.R == 1 ? 7 : 0;
        push(e);
        e = Entry(); e.layer = ci; e.ndim = 1; e.dims[0] = c.cout; e.kind = 2;
        if (c.has_bias) { e.name = c.conv_name + ".bias"; e.which = 0; push(e); }
        e.name = c.bn_name + ".weight"; e.which = 1; push(e);
        e.name = c.bn_name + ".bias"; e.which = 2; push(e);
        e.name = c.bn_name + ".running_mean"; e.kind = 3; push(e);
        e.name = c.bn_name + ".running_var"; e.kind = 4; push(e);
        e.name = c.bn_name + ".num_batches_tracked"; e.kind = 5; e.ndim = 0; e.dims[0] = 0; push(e);
    };
    for (int ci = 0; ci < i_bott + 2; ++ci) push_conv(ci);
    for (int k = 0; k < D; ++k) {
        const UpConv& u = ups[k];
        Entry e;
        e.layer = k;
        e.name = u.name + ".weight"; e.ndim = 4; e.dims[0] = u.cin; e.dims[1] = u.cout; e.dims[2] = 2; e.dims[3] = 2; e.kind = 1; push(e);
        e = Entry(); e.layer = k;
        e.name = u.name + ".bias"; e.ndim = 1; e.dims[0] = u.cout; e.kind = 2; e.which = 3; push(e);
        push_conv(i_bott + 2 + 2 * k);
        push_conv(i_bott + 2 + 2 * k + 1);
    }
    {
        Entry e;
        e.name = "final_conv.weight"; e.ndim = 4; e.dims[0] = out_ch; e.dims[1] = feat; e.dims[2] = 1; e.dims[3] = 1; e.kind = 6; push(e);
        e = Entry();
        e.name = "final_conv.bias"; e.ndim = 1; e.dims[0] = out_ch; e.kind = 2; e.which = 4; push(e);
    }

    // ---- device state
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
    RFI_CHECK_HIP(hipMemsetAsync(chan_pool, 0, chan_floats * sizeof(float), ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(d_sums, 0, 8 * sizeof(double), ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(d_scalars, 0, 8 * sizeof(float), ctx->stream));
    size_t co = 0, wo = 0;
    for (int ci = 0; ci < i_bott + 2; ++ci) {
        ConvBN& c = convs[ci];
        c.chan = chan_pool + co; co += align4((size_t)8 * c.cout);
        c.wd = wd_pool + wo; wo += align4((size_t)c.R * c.R * c.cin_p * c.cout);
    }
    for (int k = 0; k < D; ++k) {
        ups[k].wd = wd_pool + wo; wo += align4((size_t)4 * ups[k].cin * ups[k].cout);
        for (int jj = 0; jj < 2; ++jj) {
            ConvBN& c = convs[i_bott + 2 + 2 * k + jj];
            c.chan = chan_pool + co; co += align4((size_t)8 * c.cout);
            c.wd = wd_pool + wo; wo += align4((size_t)9 * c.cin_p * c.cout);
        }
    }
    // derived filters of the stride-2 convs (2x2 form + its dgrad layout) and identity scale / shift vectors
    size_t wneed = 0;
    for (auto& c : convs)
        if (c.stride == 2 && c.R == 3)
            wneed += 2 * align4((size_t)16 * c.cin * c.cout) + align4(weights_x3_floats(4, c.cout, 4 * c.cin)) + align4(weights_x3_floats(4, 4 * c.cin, c.cout));
    const size_t cmax = (size_t)feat << D;
    rs_wpool = static_cast<float*>(ctx->alloc((wneed + 2 * cmax + 16) * sizeof(float)));
    size_t o = 0;
    for (auto& c : convs)
        if (c.stride == 2 && c.R == 3) {
            c.ws2d = rs_wpool + o; o += align4((size_t)16 * c.cin * c.cout);
            c.wds2d = rs_wpool + o; o += align4((size_t)16 * c.cin * c.cout);
            c.ws2d3 = rs_wpool + o; o += align4(weights_x3_floats(4, c.cout, 4 * c.cin));
            c.wds2d3 = rs_wpool + o; o += align4(weights_x3_floats(4, 4 * c.cin, c.cout));
        }
    rs_ones = rs_wpool + o; o += cmax;
    rs_zeros = rs_wpool + o;
    {
        std::vector<float> h(2 * cmax, 0.0f);
        for (size_t i = 0; i < cmax; ++i) h[i] = 1.0f;
        RFI_CHECK_HIP(hipMemcpyAsync(rs_ones, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    adam_step = 0;
    wd_dirty = true;
    x3_fresh = false;
    reset_channel_state();
}

void rfi_model::prepare_resnet(int n, int h, int w) {
    if (rs_stemY < 0) {
        rs_stemY = new_buf(); rs_a0 = new_buf(); rs_g0 = new_buf(); rs_g1 = new_buf();
        rs_dX = new_buf(); rs_dS = new_buf(); rs_dW = new_buf(); rs_dzd = new_buf();
        for (int i = 0; i < 2; ++i) { rs_dz[i] = new_buf(); rs_dA1[i] = new_buf(); }
        for (auto& b : blocks) {
            b.Y1 = new_buf(); b.Y2 = new_buf(); b.A = new_buf();
            if (b.stride == 2) { b.Yd = new_buf(); b.xs = new_buf(); }
        }
    }
    const size_t M1 = (size_t)n * h * w;
    bufs[rs_stemY].ensure(ctx, M1 * feat);
    bufs[rs_a0].ensure(ctx, M1 * feat);
    size_t gmax = M1 * feat, wmax = 0;
    for (auto& b : blocks) {
        const size_t M = (size_t)n * (h >> (b.level - 1)) * (w >> (b.level - 1));
        for (int i : {b.Y1, b.Y2, b.A}) bufs[i].ensure(ctx, M * b.cout);
        if (b.stride == 2) {
            bufs[b.Yd].ensure(ctx, M * b.cout);
            bufs[b.xs].ensure(ctx, M * 4 * b.cin);
            wmax = std::max(wmax, (size_t)16 * b.cin * b.cout);
        }
        gmax = std::max(gmax, std::max(M * b.cout, M * 4 * b.cin));
    }
    for (int i : {rs_g0, rs_g1, rs_dz[0], rs_dz[1], rs_dA1[0], rs_dA1[1], rs_dX, rs_dS, rs_dzd}) bufs[i].ensure(ctx, gmax);
    bufs[rs_dW].ensure(ctx, wmax + 16);
}

// the 2x2 forms of the stride-2 filters (forward and input-gradient layouts); every other derived copy comes from
// the batched relayout of model.cpp
void rfi_model::refresh_resnet_weights() {
    for (int ci = 0; ci < i_bott; ++ci) {
        ConvBN& c = convs[ci];
        if (c.stride != 2 || c.R != 3) continue;
        launch_w_s2d(ctx, params + c.w_off, c.cout, c.cin, c.ws2d, true);
        launch_weight_to_dgrad(ctx, c.ws2d, 4, c.cout, 4 * c.cin, 1, c.wds2d);
        if (use_w3()) {
            launch_weights_to_x3(ctx, c.ws2d, 4, c.cout, 4 * c.cin, c.ws2d3);
            launch_weights_to_x3(ctx, c.wds2d, 4, 4 * c.cin, c.cout, c.wds2d3);
        }
    }
}

namespace {

struct Shape { int N, H, W; };

// generic conv + BatchNorm statistics on float32 tensors (round-1 kernels): R x R, stride 1, `pad`
void conv_bn(rfi_model* m, ConvBN& c, View in, InXform xf, Shape s, const float* w, const float* w3, int taps_R, int pad,
             int cin, float* Y, bool train, double flops) {
    ConvArgs a;
    a.x = in;
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
    a.Cin = cin; a.Cout = c.cout;
    a.w = w;
    a.w3 = m->use_w3() ? w3 : nullptr;            // null: the kernel launcher splits a temporary copy
    m->ws_set(a);
    a.bias = c.has_bias ? m->params + c.b_off : nullptr;
    a.y = MutView{Y, c.cout};
    a.Hout = s.H; a.Wout = s.W;
    a.R = taps_R; a.S = 1; a.pad = pad;
    a.xf = xf;
    a.algo_flops = flops;
    float* ws = m->buf(m->ws_red);
    if (train) {
        a.stats = reinterpret_cast<double*>(ws);
        a.stats_max_records = (int)(bn_stats_ws_floats(c.cout) / ((size_t)c.cout * 4));
    }
    a.bf16 = m->compute_bf16;
    a.bf16x3 = m->compute_x3;
    launch_conv(m->ctx, a);
    const int64_t M = (int64_t)s.N * s.H * s.W;
    if (train) {
        if (a.stats_records == 0) launch_bn_stats(m->ctx, Y, M, c.cout, ws);
        launch_bn_finalize(m->ctx, ws, M, c.cout, m->params + c.g_off, m->params + c.be_off, c.running_mean(), c.running_var(),
                           c.ema_repeats, c.mean(), c.invstd(), c.scale(), c.shift(), nullptr, a.stats_records);
        c.nbt += c.ema_repeats;
    } else {
        launch_bn_eval_coeffs(m->ctx, c.cout, m->params + c.g_off, m->params + c.be_off, c.running_mean(), c.running_var(),
                              c.scale(), c.shift());
    }
}

struct SideScopeR {
    rfi_model* m;
    bool ended = false;
    explicit SideScopeR(rfi_model* model) : m(model) { m->side_begin(); }
    void end() { m->side_end(); ended = true; }
    ~SideScopeR() { if (!ended) m->ctx->stream = m->ctx->main_stream; }
};

// dW of a conv (R x R stride 1 `pad`) into `dw` ([taps][cout][cx]) on the side stream
void wgrad(rfi_model* m, View x, InXform xf_x, const float* dY, int cy, int cx, Shape s, int R, int pad, float* dw, double flops) {
    WgradArgs wa;
    wa.xop = x;
    wa.yop = View{dY, cy};
    wa.xf_x = xf_x;
    wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = s.H; wa.Wx = s.W;
    wa.Cx = cx; wa.Cy = cy;
    wa.R = R; wa.S = 1; wa.pad = pad;
    wa.dw = dw;
    wa.tap_stride = (int64_t)cx * cy;
    wa.sy = cx; wa.sx = 1;
    wa.algo_flops = flops;
    wa.slab = m->buf(m->ws_slab);
    wa.slab_floats = m->bufs[m->ws_slab].n;
    wa.bf16 = m->compute_bf16;
    wa.bf16x3 = m->compute_x3;
    SideScopeR side(m);
    launch_wgrad(m->ctx, wa);
    side.end();
}

// dX = conv(dY, dgrad-layout filters): R x R stride 1 `pad` (pad of the GRADIENT conv)
void dgrad(rfi_model* m, const float* dY, int cy, const float* wd, const float* wd3, int cx, Shape s, int R, int pad, float* dx,
           double flops) {
    ConvArgs a;
    a.x = View{dY, cy};
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
    a.Cin = cy; a.Cout = cx;
    a.w = wd;
    a.w3 = m->use_w3() ? wd3 : nullptr;
    m->ws_set(a);
    a.y = MutView{dx, cx};
    a.Hout = s.H; a.Wout = s.W;
    a.R = R; a.S = 1; a.pad = pad;
    a.algo_flops = flops;
    a.bf16 = m->compute_bf16;
    a.bf16x3 = m->compute_x3;
    launch_conv(m->ctx, a);
}

}  // namespace

// -> the pooled output of layer 4 (input of the bottleneck); skip l is written into concat[l][..., C:2C]
View rfi_model::forward_resnet_encoder(View x, int n, int h, int w, bool train) {
    const int D = depth;
    {                                             // stem: a0 = relu(BN(conv3x3(x)))
        ConvBN& c = convs[0];
        Shape s{n, h, w};
        conv_bn(this, c, x, InXform{}, s, params + c.w_off, c.w3, 3, 1, c.cin_p, buf(rs_stemY), train, 2.0 * n * h * w * 9.0 * c.cin * c.cout);
        launch_bn_add_relu(ctx, buf(rs_stemY), c.scale(), c.shift(), nullptr, nullptr, nullptr, (int64_t)n * h * w, c.cout,
                           MutView{buf(rs_a0), c.cout}, MutView{});
    }
    const float* a_in = buf(rs_a0);
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        ResBlock& b = blocks[bi];
        ConvBN& c1 = convs[b.c1];
        ConvBN& c2 = convs[b.c2];
        Shape s{n, h >> (b.level - 1), w >> (b.level - 1)};
        const int64_t M = (int64_t)s.N * s.H * s.W;
        const double f1 = 2.0 * M * 9.0 * b.cin * b.cout, f2 = 2.0 * M * 9.0 * b.cout * b.cout;
        if (b.stride == 2) {
            launch_s2d(ctx, a_in, n, s.H * 2, s.W * 2, b.cin, buf(b.xs));
            conv_bn(this, c1, View{buf(b.xs), 4 * b.cin}, InXform{}, s, c1.ws2d, c1.ws2d3, 2, 1, 4 * b.cin, buf(b.Y1), train, f1);
            ConvBN& cd = convs[b.cd];             // projection: 1x1 on the (0, 0) slice of the space-to-depth input
            conv_bn(this, cd, View{buf(b.xs), 4 * b.cin}, InXform{}, s, params + cd.w_off, cd.w3, 1, 0, b.cin, buf(b.Yd), train,
                    2.0 * M * b.cin * b.cout);
        } else {
            conv_bn(this, c1, View{a_in, b.cin}, InXform{}, s, params + c1.w_off, c1.w3, 3, 1, b.cin, buf(b.Y1), train, f1);
        }
        conv_bn(this, c2, View{buf(b.Y1), b.cout}, bn_xf(c1), s, params + c2.w_off, c2.w3, 3, 1, b.cout, buf(b.Y2), train, f2);
        // a_out = relu(BN2(Y2) + shortcut); the stage output also goes into the decoder's concat buffer (the skip)
        const bool last = (bi & 1) == 1;
        MutView skip = last ? MutView{buf(concat[b.level]) + b.cout, 2 * b.cout} : MutView{};
        if (b.stride == 2) {
            ConvBN& cd = convs[b.cd];
            launch_bn_add_relu(ctx, buf(b.Y2), c2.scale(), c2.shift(), buf(b.Yd), cd.scale(), cd.shift(), M, b.cout,
                               MutView{buf(b.A), b.cout}, skip);
        } else {
            launch_bn_add_relu(ctx, buf(b.Y2), c2.scale(), c2.shift(), a_in, nullptr, nullptr, M, b.cout,
                               MutView{buf(b.A), b.cout}, skip);
        }
        a_in = buf(b.A);
    }
    // MaxPool2d(2) of the last stage (identity "BatchNorm": scale 1, shift 0; the values are already >= 0)
    const ResBlock& lb = blocks.back();
    Shape s{n, h >> (D - 1), w >> (D - 1)};
    launch_bn_relu_pool(ctx, buf(lb.A), s.N, s.H, s.W, lb.cout, rs_ones, rs_zeros,
                        MutView{buf(concat[D]) + lb.cout, 2 * lb.cout}, buf(pool[D]), 0.0f);
    return View{buf(pool[D]), lb.cout};
}

void rfi_model::backward_resnet_encoder(const float* x_dev, int n, int h, int w) {
    const int D = depth;
    // gradient w.r.t. the last stage's output: skip gradient + max-pool routing of dpool
    float* gout = buf(rs_g0);
    float* gother = buf(rs_g1);
    {
        const ResBlock& lb = blocks.back();
        Shape s{n, h >> (D - 1), w >> (D - 1)};
        launch_pool_bwd_merge(ctx, buf(lb.A), s.N, s.H, s.W, lb.cout, rs_ones, rs_zeros,
                              View{buf(dconcat[D]) + lb.cout, 2 * lb.cout}, buf(dpool[D]), gout, 0.0f);
    }
    for (int bi = (int)blocks.size() - 1; bi >= 0; --bi) {
        ResBlock& b = blocks[bi];
        ConvBN& c1 = convs[b.c1];
        ConvBN& c2 = convs[b.c2];
        Shape s{n, h >> (b.level - 1), w >> (b.level - 1)};
        const int64_t M = (int64_t)s.N * s.H * s.W;
        const double f1 = 2.0 * M * 9.0 * b.cin * b.cout, f2 = 2.0 * M * 9.0 * b.cout * b.cout;
        float* ws = buf(ws_red);
        const float* a_in = bi == 0 ? buf(rs_a0) : buf(blocks[bi - 1].A);
        // gout = gradient w.r.t. this block's output.  dz = gout * (a_out > 0) enters BOTH BatchNorm branches
        float* dz = buf(rs_dz[bi & 1]);
        launch_relu_mask(ctx, View{gout, b.cout}, View{}, View{buf(b.A), b.cout}, View{}, M, b.cout, dz);
        // ---- main branch: BN2 (no activation of its own: slope 1 = identity), conv2, BN1 + ReLU, conv1
        launch_bn_bwd_reduce(ctx, dz, buf(b.Y2), M, b.cout, c2.scale(), c2.shift(), c2.mean(), c2.invstd(), ws, c2.c1(), c2.c2(),
                             grads + c2.g_off, grads + c2.be_off, 1.0f);
        launch_bn_bwd_apply(ctx, dz, buf(b.Y2), M, b.cout, c2.scale(), c2.shift(), c2.mean(), c2.invstd(), params + c2.g_off,
                            c2.c1(), c2.c2(), ws, nullptr, 1.0f);                 // dz <- dY2
        wgrad(this, View{buf(b.Y1), b.cout}, bn_xf(c1), dz, b.cout, b.cout, s, 3, 1, grads + c2.w_off, f2);
        float* dA1 = buf(rs_dA1[bi & 1]);
        dgrad(this, dz, b.cout, c2.wd, c2.wd3, b.cout, s, 3, 1, dA1, f2);
        launch_bn_bwd_reduce(ctx, dA1, buf(b.Y1), M, b.cout, c1.scale(), c1.shift(), c1.mean(), c1.invstd(), ws, c1.c1(), c1.c2(),
                             grads + c1.g_off, grads + c1.be_off, act_slope);
        launch_bn_bwd_apply(ctx, dA1, buf(b.Y1), M, b.cout, c1.scale(), c1.shift(), c1.mean(), c1.invstd(), params + c1.g_off,
                            c1.c1(), c1.c2(), ws, nullptr, act_slope);            // dA1 <- dY1
        float* dX = buf(rs_dX);
        if (b.stride == 2) {
            // conv1 in its 2x2 form on the space-to-depth input: weight gradient in that layout, then back to 3x3
            wgrad(this, View{buf(b.xs), 4 * b.cin}, InXform{}, dA1, b.cout, 4 * b.cin, s, 2, 1, buf(rs_dW), f1);
            {
                SideScopeR side(this);            // (after the slab reduction of that wgrad, same stream)
                launch_w_s2d(ctx, grads + c1.w_off, b.cout, b.cin, buf(rs_dW), false);
                side.end();
            }
            dgrad(this, dA1, b.cout, c1.wds2d, c1.wds2d3, 4 * b.cin, s, 2, 0, dX, f1);             // [M][4 cin]
            // ---- projection branch: BNd on the same dz (recomputed: the first copy now holds dY2), 1x1 conv
            ConvBN& cd = convs[b.cd];
            dz = buf(rs_dzd);
            launch_relu_mask(ctx, View{gout, b.cout}, View{}, View{buf(b.A), b.cout}, View{}, M, b.cout, dz);
            launch_bn_bwd_reduce(ctx, dz, buf(b.Yd), M, b.cout, cd.scale(), cd.shift(), cd.mean(), cd.invstd(), ws, cd.c1(), cd.c2(),
                                 grads + cd.g_off, grads + cd.be_off, 1.0f);
            launch_bn_bwd_apply(ctx, dz, buf(b.Yd), M, b.cout, cd.scale(), cd.shift(), cd.mean(), cd.invstd(), params + cd.g_off,
                                cd.c1(), cd.c2(), ws, nullptr, 1.0f);             // dz <- dYd
            wgrad(this, View{buf(b.xs), 4 * b.cin}, InXform{}, dz, b.cout, b.cin, s, 1, 0, grads + cd.w_off, 2.0 * M * b.cin * b.cout);
            float* dS = buf(rs_dS);
            dgrad(this, dz, b.cout, cd.wd, cd.wd3, b.cin, s, 1, 0, dS, 2.0 * M * b.cin * b.cout);   // [M][cin]
            // back to full resolution, + the skip gradient of the previous stage (its output is this block's input)
            launch_d2s_add(ctx, dX, dS, View{buf(dconcat[b.level - 1]) + b.cin, 2 * b.cin}, n, s.H * 2, s.W * 2, b.cin, gother);
        } else {
            wgrad(this, View{a_in, b.cin}, InXform{}, dA1, b.cout, b.cin, s, 3, 1, grads + c1.w_off, f1);
            dgrad(this, dA1, b.cout, c1.wd, c1.wd3, b.cin, s, 3, 1, dX, f1);
            // identity shortcut: gin = dX + gout * (a_out > 0)
            launch_relu_mask(ctx, View{gout, b.cout}, View{}, View{buf(b.A), b.cout}, View{dX, b.cin}, M, b.cout, gother);
        }
        std::swap(gout, gother);
        bucket_ready(c1.w_off, (size_t)(bi + 1 < (int)blocks.size() ? convs[blocks[bi + 1].c1].w_off : convs[i_bott].w_off));
    }
    {                                             // stem: gout = gradient w.r.t. a0 = relu(BN(stemY))
        ConvBN& c = convs[0];
        Shape s{n, h, w};
        const int64_t M = (int64_t)n * h * w;
        float* ws = buf(ws_red);
        launch_bn_bwd_reduce(ctx, gout, buf(rs_stemY), M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(), ws, c.c1(), c.c2(),
                             grads + c.g_off, grads + c.be_off, act_slope);
        launch_bn_bwd_apply(ctx, gout, buf(rs_stemY), M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(), params + c.g_off,
                            c.c1(), c.c2(), ws, nullptr, act_slope);
        View in = c.cin_p == in_ch ? View{x_dev, in_ch} : View{buf(x_pad), c.cin_p};
        wgrad(this, in, InXform{}, gout, c.cout, c.cin_p, s, 3, 1, grads + c.w_off, 2.0 * M * 9.0 * c.cin * c.cout);
        bucket_ready(0, convs[blocks[0].c1].w_off);
    }
}
