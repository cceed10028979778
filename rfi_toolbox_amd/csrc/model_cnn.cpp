// The "3-layer CNN segmenter" (BASELINE.json configs[0]/[1]) on the same HIP
// kernels as the U-Net:  Conv3x3(in->C)+bias -> ReLU -> Conv3x3(C->C)+bias -> ReLU -> Conv1x1(C->out).
// Not a reference class (nearest text: README.md:379-398); trained with the reference's step
// (scripts/train_model.py:120-151) through the shared loss / clip / Adam code in model.cpp.
//
// Only the raw conv outputs Y1, Y2 are kept in HBM; ReLU is applied by the consumer's loads
// (InXform with scale 1, shift 0) exactly as the U-Net's BN+ReLU is, so the forward pass is three
// launches and writes 2*M*C + M floats.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

void rfi_model::build_cnn3() {
    RFI_REQUIRE(in_ch > 0 && out_ch > 0 && feat > 0, "SimpleCNN: channel counts must be positive");
    RFI_REQUIRE(feat % 4 == 0, "SimpleCNN: width must be a multiple of 4 (16-byte NHWC pixels)");
    convs.clear();
    ups.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    const char* names[2] = {"encoder.0", "encoder.2"};
    for (int i = 0; i < 2; ++i) {
        ConvBN c;
        c.conv_name = names[i];
        c.has_bn = false;
        c.cin = i == 0 ? in_ch : feat;
        c.cin_p = i == 0 ? (int)align4((size_t)in_ch) : feat;
        c.cout = feat;
        c.w_off = off; off = align4(off + (size_t)9 * c.cin_p * c.cout);
        c.b_off = off; off = align4(off + c.cout);
        c.g_off = c.be_off = 0;
        chan_floats += align4((size_t)8 * c.cout);
        wd_floats += align4((size_t)9 * c.cin_p * c.cout);
        convs.push_back(c);
    }
    head_w_off = off; off = align4(off + (size_t)out_ch * feat);
    head_b_off = off; off = align4(off + out_ch);
    n_flat = off;

    entries.clear();
    entry_index.clear();
    n_params = 0;
    auto push = [&](Entry e) {
        entry_index[e.name] = (int)entries.size();
        n_params += e.numel();
        entries.push_back(e);
    };
    for (int i = 0; i < 2; ++i) {
        Entry e;
        e.layer = i;
        e.name = convs[i].conv_name + ".weight"; e.ndim = 4;
        e.dims[0] = convs[i].cout; e.dims[1] = convs[i].cin; e.dims[2] = 3; e.dims[3] = 3; e.kind = 0;
        push(e);
        e = Entry(); e.layer = i;
        e.name = convs[i].conv_name + ".bias"; e.ndim = 1; e.dims[0] = convs[i].cout; e.kind = 2; e.which = 0;
        push(e);
    }
    {
        Entry e;
        e.name = "decoder.0.weight"; e.ndim = 4; e.dims[0] = out_ch; e.dims[1] = feat; e.dims[2] = 1; e.dims[3] = 1; e.kind = 6;
        push(e);
        e = Entry();
        e.name = "decoder.0.bias"; e.ndim = 1; e.dims[0] = out_ch; e.kind = 2; e.which = 4;
        push(e);
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
        c.wd = wd_pool + wo; wo += align4((size_t)9 * c.cin_p * c.cout);
    }
    adam_step = 0;
    wd_dirty = true;
        x3_fresh = false;
    reset_channel_state();
}

void rfi_model::prepare_cnn3(int n, int h, int w) {
    if (n == pN && h == pH && w == pW && !bufs.empty()) return;
    ctx->activate();
    if (bufs.empty()) {
        cY1 = new_buf(); cY2 = new_buf(); cG1 = new_buf(); cG2 = new_buf();
        logits = new_buf(); dlogits = new_buf();
        x_stage = new_buf(); x_stage2 = new_buf(); x_pad = new_buf(); out_stage = new_buf();
        ws_red = new_buf(); ws_slab = new_buf(); lab_stage = new_buf();
    }
    const size_t M1 = (size_t)n * h * w;
    for (int i : {cY1, cY2, cG1, cG2}) bufs[i].ensure(ctx, M1 * feat);
    bufs[logits].ensure(ctx, M1 * out_ch);
    bufs[dlogits].ensure(ctx, M1 * out_ch);
    bufs[x_stage].ensure(ctx, M1 * in_ch);
    bufs[x_stage2].ensure(ctx, M1 * in_ch);
    bufs[x_pad].ensure(ctx, M1 * convs[0].cin_p);
    bufs[out_stage].ensure(ctx, M1 * out_ch);
    bufs[lab_stage].ensure(ctx, (M1 + 3) / 4 + 4);
    size_t red_need = 0, slab_need = 0;
    auto upd = [&](size_t f) { red_need = std::max(red_need, f); };
    upd(head_bwd_ws_floats((int64_t)M1, feat, out_ch));
    upd(channel_sum_ws_floats((int64_t)M1, feat));
    upd(loss_ws_doubles((int64_t)M1) * 2);
    upd(sumsq_ws_doubles((int64_t)n_flat) * 2);
    bufs[ws_red].ensure(ctx, red_need + 16);
    for (auto& c : convs) {
        WgradArgs a;
        a.N = n; a.H = h; a.W = w; a.Hx = h; a.Wx = w;
        a.Cx = c.cin_p; a.Cy = c.cout;
        a.xop.pstride = a.Cx; a.yop.pstride = a.Cy;
        a.R = 3; a.S = 1; a.pad = 1;
        a.tap_stride = (int64_t)a.Cx * a.Cy;
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
    }
    bufs[ws_slab].ensure(ctx, slab_need + 16);
    pN = n; pH = h; pW = w;
}

namespace {

ConvArgs conv3x3(const ConvBN& c, View in, InXform xf, const float* w, const float* bias, float* y, int cin_gemm,
                 int cout_gemm, int n, int h, int wd) {
    ConvArgs a;
    a.x = in;
    a.N = n; a.H = h; a.W = wd; a.Hin = h; a.Win = wd;
    a.Cin = cin_gemm; a.Cout = cout_gemm;
    a.w = w;
    a.bias = bias;
    a.y = MutView{y, cout_gemm};
    a.Hout = h; a.Wout = wd;
    a.R = 3; a.S = 1; a.pad = 1;
    a.xf = xf;
    a.algo_flops = 2.0 * n * h * wd * 9.0 * c.cin * c.cout;
    return a;
}

InXform relu_xf(const ConvBN& c) { return InXform{c.scale(), c.shift(), 1}; }

}  // namespace

void rfi_model::forward_cnn3(const float* x_dev, int n, int h, int w) {
    refresh_dgrad_weights();
    ConvBN& c1 = convs[0];
    ConvBN& c2 = convs[1];
    View x = network_input(x_dev, n, h, w);
    ConvArgs a1 = conv3x3(c1, x, InXform{}, params + c1.w_off, params + c1.b_off, buf(cY1), c1.cin_p, c1.cout,
                          n, h, w);
    a1.w3 = use_w3() ? c1.w3 : nullptr;
    ws_set(a1);
    a1.bf16 = compute_bf16;
    a1.bf16x3 = compute_x3;
    launch_conv(ctx, a1);
    ConvArgs a2 = conv3x3(c2, View{buf(cY1), c1.cout}, relu_xf(c1), params + c2.w_off, params + c2.b_off,
                          buf(cY2), c2.cin_p, c2.cout, n, h, w);
    a2.w3 = use_w3() ? c2.w3 : nullptr;
    ws_set(a2);
    a2.bf16 = compute_bf16;
    a2.bf16x3 = compute_x3;
    launch_conv(ctx, a2);
    launch_head_fwd(ctx, buf(cY2), (int64_t)n * h * w, feat, c2.scale(), c2.shift(), params + head_w_off,
                    params + head_b_off, out_ch, buf(logits));
}

void rfi_model::backward_cnn3(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w) {
    ConvBN& c1 = convs[0];
    ConvBN& c2 = convs[1];
    const int64_t M = (int64_t)n * h * w;
    refresh_dgrad_weights();
    if (loss_kind == 1) launch_focal_bwd(ctx, buf(logits), labels_dev, M, focal_alpha, focal_gamma, buf(dlogits));
    else launch_loss_bwd(ctx, buf(logits), labels_dev, M, d_sums, buf(dlogits));
    // head: dW, db and the gradient w.r.t. relu(Y2)
    launch_head_bwd(ctx, buf(cY2), M, feat, c2.scale(), c2.shift(), params + head_w_off, out_ch, buf(dlogits),
                    buf(cG2), buf(ws_red), grads + head_w_off, grads + head_b_off);
    auto conv_backward = [&](ConvBN& c, float* dA, const float* Y, View in, InXform in_xf, float* dx) {
        launch_relu_bwd(ctx, dA, Y, M * c.cout);                       // dA -> dY
        launch_channel_sum(ctx, View{dA, c.cout}, M, c.cout, buf(ws_red), grads + c.b_off);
        WgradArgs wa;
        wa.xop = in;
        wa.yop = View{dA, c.cout};
        wa.xf_x = in_xf;
        wa.N = n; wa.H = h; wa.W = w; wa.Hx = h; wa.Wx = w;
        wa.Cx = c.cin_p; wa.Cy = c.cout;
        wa.R = 3; wa.S = 1; wa.pad = 1;
        wa.dw = grads + c.w_off;
        wa.tap_stride = (int64_t)c.cin_p * c.cout;
        wa.sy = c.cin_p; wa.sx = 1;
        wa.algo_flops = 2.0 * M * 9.0 * c.cin * c.cout;
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        side_begin();                      // wgrad on the side stream, next to the dgrad / ReLU chain
        wa.bf16 = compute_bf16;
    wa.bf16x3 = compute_x3;
        launch_wgrad(ctx, wa);
        side_end();
        if (dx) {
            ConvArgs a = conv3x3(c, View{dA, c.cout}, InXform{}, c.wd, nullptr, dx, c.cout, c.cin, n, h, w);
            a.w3 = use_w3() ? c.wd3 : nullptr;
            ws_set(a);
            a.bf16 = compute_bf16;
    a.bf16x3 = compute_x3;
            launch_conv(ctx, a);
        }
    };
    conv_backward(c2, buf(cG2), buf(cY2), View{buf(cY1), c1.cout}, relu_xf(c1), buf(cG1));
    View x = c1.cin_p == in_ch ? View{x_dev, in_ch} : View{buf(x_pad), c1.cin_p};
    conv_backward(c1, buf(cG1), buf(cY1), x, InXform{}, nullptr);
    side_join();
}
