// The per-RoI mask branch of Mask R-CNN (BASELINE.json configs[3] / north_star "per-pixel mask head"; SURVEY.md 8a row
// A11) on the same HIP kernels as the segmentation models.  NOT in the reference (it contains no detector) and
// torchvision is absent: builder-defined as the published head (He et al. 2017, fig. 4 right):
//
//   x [R, 14, 14, C]  RoIAlign-ed features (rfi_op_roi_align)
//   mask_fcn1..L      Conv3x3(C -> C, pad 1) + bias -> ReLU            (L = 4)
//   conv5_mask        ConvTranspose2d(C -> C, k 2, s 2) + bias -> ReLU
//   mask_fcn_logits   Conv1x1(C -> K)                                  logits [R, 28, 28, K]
//   loss              mean binary cross-entropy with logits over every RoI pixel (K = 1: one foreground class, RFI)
//
// oracle/mask_head_ref.py holds the same layers as plain torch.nn modules (parity unpinned by the reference).  Only the
// raw conv outputs are kept; ReLU is applied by the consumers' loads, as in model_cnn.cpp.  rfi_model_input_grad returns
// the gradient w.r.t. x, which rfi_op_roi_align_backward scatters back into the feature map.
//
// arch 4, the RPN head (Ren et al. 2015), is the same stack without the transposed conv: Conv3x3(C -> C) + ReLU, then ONE
// 1x1 conv with 5 A outputs per pixel = A objectness logits followed by A x 4 box deltas (cls_logits and bbox_pred of the
// usual implementation stacked; models/rpn_head.py splits them at the state_dict boundary).  Its loss needs per-anchor
// targets, so it lives outside the model: rfi_op_rpn_loss produces d(loss)/d(head output) and
// rfi_model_backward_dlogits runs the backward pass from there.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

void rfi_model::build_mask() {
    RFI_REQUIRE(in_ch > 0 && in_ch % 4 == 0, "MaskHead: in_channels must be a positive multiple of 4 (16-byte NHWC pixels)");
    RFI_REQUIRE(out_ch > 0 && depth >= 1 && depth <= 8, "MaskHead: out_channels > 0, 1..8 conv layers");
    const bool up = arch == 3;
    feat = in_ch;
    out_scale = up ? 2 : 1;
    loss_kind = 1;                // sigmoid focal loss with gamma = 0 and no alpha = plain mean BCE-with-logits
    focal_alpha = -1.0f;
    focal_gamma = 0.0f;
    const int L = depth, C = in_ch;
    convs.clear();
    ups.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    for (int i = 0; i < L; ++i) {
        ConvBN c;
        c.conv_name = up ? "mask_fcn" + std::to_string(i + 1) : "conv." + std::to_string(i) + ".0";
        c.has_bn = false;
        c.cin = c.cin_p = c.cout = C;
        c.w_off = off; off = align4(off + (size_t)9 * C * C);
        c.b_off = off; off = align4(off + C);
        c.g_off = c.be_off = 0;
        chan_floats += align4((size_t)8 * C);
        wd_floats += align4((size_t)9 * C * C);
        convs.push_back(c);
    }
    if (up) {
        UpConv u;
        u.name = "conv5_mask";
        u.cin = u.cout = C;
        u.w_off = off; off = align4(off + (size_t)4 * C * C);
        u.b_off = off; off = align4(off + C);
        wd_floats += align4((size_t)4 * C * C);
        ups.push_back(u);
    }
    head_w_off = off; off = align4(off + (size_t)out_ch * C);
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
    for (int i = 0; i < L; ++i) {
        Entry e;
        e.layer = i;
        e.name = convs[i].conv_name + ".weight"; e.ndim = 4; e.dims[0] = C; e.dims[1] = C; e.dims[2] = 3; e.dims[3] = 3; e.kind = 0;
        push(e);
        e = Entry(); e.layer = i;
        e.name = convs[i].conv_name + ".bias"; e.ndim = 1; e.dims[0] = C; e.kind = 2; e.which = 0;
        push(e);
    }
    if (up) {
        Entry e;
        e.layer = 0;
        e.name = "conv5_mask.weight"; e.ndim = 4; e.dims[0] = C; e.dims[1] = C; e.dims[2] = 2; e.dims[3] = 2; e.kind = 1;
        push(e);
        e = Entry(); e.layer = 0;
        e.name = "conv5_mask.bias"; e.ndim = 1; e.dims[0] = C; e.kind = 2; e.which = 3;
        push(e);
    }
    {
        const std::string hn = up ? "mask_fcn_logits" : "head";
        Entry e;
        e.name = hn + ".weight"; e.ndim = 4; e.dims[0] = out_ch; e.dims[1] = C; e.dims[2] = 1; e.dims[3] = 1; e.kind = 6;
        push(e);
        e = Entry();
        e.name = hn + ".bias"; e.ndim = 1; e.dims[0] = out_ch; e.kind = 2; e.which = 4;
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
    if (up) ups[0].wd = wd_pool + wo;
    adam_step = 0;
    wd_dirty = true;
    x3_fresh = false;
    reset_channel_state();
}

void rfi_model::prepare_mask(int n, int h, int w) {
    if (n == pN && h == pH && w == pW && !bufs.empty()) return;
    ctx->activate();
    const int L = depth, C = in_ch;
    if (bufs.empty()) {
        mkY.assign(L, -1); mkG.assign(L, -1);
        for (int i = 0; i < L; ++i) { mkY[i] = new_buf(); mkG[i] = new_buf(); }
        mkU = new_buf(); mkGU = new_buf(); mkGx = new_buf();
        logits = new_buf(); dlogits = new_buf(); head_wd = new_buf(); head_w3 = new_buf(); head_wd3 = new_buf();
        x_stage = new_buf(); x_stage2 = new_buf(); x_pad = new_buf(); out_stage = new_buf();
        ws_red = new_buf(); ws_slab = new_buf(); lab_stage = new_buf();
    }
    const size_t M = (size_t)n * h * w, M4 = (size_t)out_scale * out_scale * M;
    for (int i = 0; i < L; ++i) { bufs[mkY[i]].ensure(ctx, M * C); bufs[mkG[i]].ensure(ctx, M * C); }
    bufs[mkU].ensure(ctx, arch == 3 ? M4 * C : 16);
    bufs[mkGU].ensure(ctx, arch == 3 ? M4 * C : 16);
    bufs[mkGx].ensure(ctx, M * C);
    bufs[logits].ensure(ctx, M4 * out_ch);
    bufs[dlogits].ensure(ctx, M4 * out_ch);
    bufs[x_stage].ensure(ctx, M * C);
    bufs[x_stage2].ensure(ctx, M * C);
    bufs[x_pad].ensure(ctx, 16);
    bufs[out_stage].ensure(ctx, M4 * out_ch);
    bufs[lab_stage].ensure(ctx, (M4 + 3) / 4 + 4);
    size_t red_need = 0, slab_need = 0;
    auto upd = [&](size_t f) { red_need = std::max(red_need, f); };
    upd(head_bwd_ws_floats((int64_t)M4, C, out_ch));
    upd(channel_sum_ws_floats((int64_t)M4, C));
    upd(loss_ws_doubles((int64_t)M4) * 2);
    upd(sumsq_ws_doubles((int64_t)n_flat) * 2);
    bufs[ws_red].ensure(ctx, red_need + 16);
    {
        WgradArgs a;                                  // the 3x3 layers
        a.N = n; a.H = h; a.W = w; a.Hx = h; a.Wx = w;
        a.Cx = C; a.Cy = C;
        a.xop.pstride = C; a.yop.pstride = C;
        a.R = 3; a.S = 1; a.pad = 1;
        a.tap_stride = (int64_t)C * C;
        a.bf16x3 = true;
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
        WgradArgs b;                                  // the transposed conv
        b.N = n; b.H = h; b.W = w; b.Hx = 2 * h; b.Wx = 2 * w;
        b.Cx = C; b.Cy = C;
        b.xop.pstride = C; b.yop.pstride = C;
        b.R = 2; b.S = 2; b.pad = 0;
        b.tap_stride = (int64_t)C * C;
        slab_need = std::max(slab_need, wgrad_slab_floats(b, IMPL_AUTO));
    }
    if (head_on_mfma()) {
        WgradArgs c;                                  // the 1x1 head's weight gradient
        c.N = n; c.H = out_scale * h; c.W = out_scale * w; c.Hx = c.H; c.Wx = c.W;
        c.Cx = C; c.Cy = out_ch;
        c.xop.pstride = C; c.yop.pstride = out_ch;
        c.R = 1; c.S = 1; c.pad = 0;
        c.tap_stride = (int64_t)C * out_ch;
        c.bf16x3 = true;
        slab_need = std::max(slab_need, wgrad_slab_floats(c, IMPL_AUTO));
    }
    bufs[head_wd].ensure(ctx, (size_t)out_ch * C + 16);
    bufs[head_w3].ensure(ctx, weights_x3_floats(1, out_ch, C) + 16);
    bufs[head_wd3].ensure(ctx, weights_x3_floats(1, C, out_ch) + 16);
    bufs[ws_slab].ensure(ctx, slab_need + 16);
    pN = n; pH = h; pW = w;
}

namespace {

InXform relu_of(const ConvBN& c) { return InXform{c.scale(), c.shift(), 1}; }     // scale 1, shift 0 (reset_channel_state)

ConvArgs conv3x3_args(rfi_model* m, View in, InXform xf, const float* w, const float* w3, const float* bias, float* y, int C,
                      int n, int h, int wd) {
    ConvArgs a;
    a.x = in;
    a.N = n; a.H = h; a.W = wd; a.Hin = h; a.Win = wd;
    a.Cin = C; a.Cout = C;
    a.w = w;
    a.w3 = m->use_w3() ? w3 : nullptr;
    m->ws_set(a);
    a.bias = bias;
    a.y = MutView{y, C};
    a.Hout = h; a.Wout = wd;
    a.R = 3; a.S = 1; a.pad = 1;
    a.xf = xf;
    a.bf16 = m->compute_bf16;
    a.bf16x3 = m->compute_x3;
    return a;
}

}  // namespace

void rfi_model::forward_mask(const float* x_dev, int n, int h, int w) {
    refresh_dgrad_weights();
    const int L = depth, C = in_ch;
    const int64_t M4 = (int64_t)out_scale * out_scale * n * h * w;
    for (int i = 0; i < L; ++i) {
        ConvBN& c = convs[i];
        ConvArgs a = conv3x3_args(this, i == 0 ? View{x_dev, C} : View{buf(mkY[i - 1]), C}, i == 0 ? InXform{} : relu_of(convs[i - 1]),
                                  params + c.w_off, c.w3, params + c.b_off, buf(mkY[i]), C, n, h, w);
        launch_conv(ctx, a);
    }
    if (arch == 3) {
        UpConv& u = ups[0];
        ConvArgs a;
        a.x = View{buf(mkY[L - 1]), C};
        a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w;
        a.Cin = C; a.Cout = C;
        a.w = params + u.w_off;
        a.w3 = use_w3() ? u.w3 : nullptr;
        ws_set(a);
        a.bias = params + u.b_off;
        a.y = MutView{buf(mkU), C};
        a.Hout = 2 * h; a.Wout = 2 * w;
        a.osy = 2; a.osx = 2;
        a.R = 1; a.S = 1; a.pad = 0;
        a.zgroups = 4;
        a.xf = relu_of(convs[L - 1]);
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
    }
    const ConvBN& cl = convs[L - 1];                  // (its scale = 1 / shift = 0 vectors serve the ReLU of U as well)
    if (head_on_mfma()) {             // a wide 1x1 head (the RPN's 5 A outputs) is a GEMM: the conv kernels, not the per-pixel VALU kernel
        ConvArgs a;
        a.x = View{arch == 3 ? buf(mkU) : buf(mkY[L - 1]), C};
        a.N = n; a.H = out_scale * h; a.W = out_scale * w; a.Hin = a.H; a.Win = a.W;
        a.Cin = C; a.Cout = out_ch;
        a.w = params + head_w_off;
        if (use_w3()) {               // the pre-split records of this small filter are rebuilt per pass (one 5-us launch, no allocation)
            launch_weights_to_x3(ctx, a.w, 1, out_ch, C, buf(head_w3));
            a.w3 = buf(head_w3);
        }
        a.bias = params + head_b_off;
        a.y = MutView{buf(logits), out_ch};
        a.Hout = a.H; a.Wout = a.W;
        a.R = 1; a.S = 1; a.pad = 0;
        a.xf = relu_of(cl);
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
        return;
    }
    launch_head_fwd(ctx, arch == 3 ? buf(mkU) : buf(mkY[L - 1]), M4, C, cl.scale(), cl.shift(), params + head_w_off, params + head_b_off,
                    out_ch, buf(logits));
}

void rfi_model::backward_mask(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w) {
    const int L = depth, C = in_ch;
    const int64_t M = (int64_t)n * h * w, M4 = (int64_t)out_scale * out_scale * M;
    refresh_dgrad_weights();
    if (!ext_dlogits) {               // (rfi_model_backward_dlogits: the caller's loss kernel has filled dlogits)
        if (loss_kind == 1) launch_focal_bwd(ctx, buf(logits), labels_dev, M4 * out_ch, focal_alpha, focal_gamma, buf(dlogits));
        else launch_loss_bwd(ctx, buf(logits), labels_dev, M4 * out_ch, d_sums, buf(dlogits));
    }
    const ConvBN& cl = convs[L - 1];
    if (head_on_mfma()) {
        // da = dlogits . W (a 1x1 conv with the transposed filter), dW = dlogits^T . act (the 1x1 weight gradient), db = column sums
        float* const hin = arch == 3 ? buf(mkU) : buf(mkY[L - 1]);
        float* const da = arch == 3 ? buf(mkGU) : buf(mkG[L - 1]);
        launch_weight_to_dgrad(ctx, params + head_w_off, 1, out_ch, C, 0, buf(head_wd));
        ConvArgs a;
        a.x = View{buf(dlogits), out_ch};
        a.N = n; a.H = out_scale * h; a.W = out_scale * w; a.Hin = a.H; a.Win = a.W;
        a.Cin = out_ch; a.Cout = C;
        a.w = buf(head_wd);
        if (use_w3()) {
            launch_weights_to_x3(ctx, a.w, 1, C, out_ch, buf(head_wd3));
            a.w3 = buf(head_wd3);
        }
        a.y = MutView{da, C};
        a.Hout = a.H; a.Wout = a.W;
        a.R = 1; a.S = 1; a.pad = 0;
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
        launch_channel_sum(ctx, View{buf(dlogits), out_ch}, M4, out_ch, buf(ws_red), grads + head_b_off);
        WgradArgs wa;
        wa.xop = View{hin, C};
        wa.xf_x = relu_of(cl);
        wa.yop = View{buf(dlogits), out_ch};
        wa.N = n; wa.H = out_scale * h; wa.W = out_scale * w; wa.Hx = wa.H; wa.Wx = wa.W;
        wa.Cx = C; wa.Cy = out_ch;
        wa.R = 1; wa.S = 1; wa.pad = 0;
        wa.dw = grads + head_w_off;
        wa.tap_stride = (int64_t)C * out_ch;
        wa.sy = C; wa.sx = 1;
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        side_begin();
        launch_wgrad(ctx, wa);
        side_end();
    } else
    launch_head_bwd(ctx, arch == 3 ? buf(mkU) : buf(mkY[L - 1]), M4, C, cl.scale(), cl.shift(), params + head_w_off, out_ch, buf(dlogits),
                    arch == 3 ? buf(mkGU) : buf(mkG[L - 1]), buf(ws_red), grads + head_w_off, grads + head_b_off);
    // transposed conv: dU = dUa * (U > 0); bias, weight and input gradients
    if (arch == 3) {
    UpConv& u = ups[0];
    launch_relu_bwd(ctx, buf(mkGU), buf(mkU), M4 * C);
    launch_channel_sum(ctx, View{buf(mkGU), C}, M4, C, buf(ws_red), grads + u.b_off);
    {
        WgradArgs wa;
        wa.xop = View{buf(mkGU), C};
        wa.yop = View{buf(mkY[L - 1]), C};
        wa.xf_y = relu_of(convs[L - 1]);
        wa.N = n; wa.H = h; wa.W = w; wa.Hx = 2 * h; wa.Wx = 2 * w;
        wa.Cx = C; wa.Cy = C;
        wa.R = 2; wa.S = 2; wa.pad = 0;
        wa.dw = grads + u.w_off;
        wa.tap_stride = (int64_t)C * C;
        wa.sy = 1; wa.sx = C;                         // -> [tap][cout][cin]
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        side_begin();
        launch_wgrad(ctx, wa);
        side_end();
        ConvArgs a;
        a.x = View{buf(mkGU), C};
        a.N = n; a.H = h; a.W = w; a.Hin = 2 * h; a.Win = 2 * w;
        a.Cin = C; a.Cout = C;
        a.w = u.wd;
        a.w3 = use_w3() ? u.wd3 : nullptr;
        ws_set(a);
        a.y = MutView{buf(mkG[L - 1]), C};
        a.Hout = h; a.Wout = w;
        a.R = 2; a.S = 2; a.pad = 0;
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
    }
    }
    for (int i = L - 1; i >= 0; --i) {
        ConvBN& c = convs[i];
        float* dA = buf(mkG[i]);
        launch_relu_bwd(ctx, dA, buf(mkY[i]), M * C);                       // dA -> dY
        launch_channel_sum(ctx, View{dA, C}, M, C, buf(ws_red), grads + c.b_off);
        WgradArgs wa;
        wa.xop = i == 0 ? View{x_dev, C} : View{buf(mkY[i - 1]), C};
        if (i > 0) wa.xf_x = relu_of(convs[i - 1]);
        wa.yop = View{dA, C};
        wa.N = n; wa.H = h; wa.W = w; wa.Hx = h; wa.Wx = w;
        wa.Cx = C; wa.Cy = C;
        wa.R = 3; wa.S = 1; wa.pad = 1;
        wa.dw = grads + c.w_off;
        wa.tap_stride = (int64_t)C * C;
        wa.sy = C; wa.sx = 1;
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        side_begin();
        launch_wgrad(ctx, wa);
        side_end();
        ConvArgs a = conv3x3_args(this, View{dA, C}, InXform{}, c.wd, c.wd3, nullptr, i == 0 ? buf(mkGx) : buf(mkG[i - 1]), C, n, h, w);
        launch_conv(ctx, a);
    }
    side_join_lazy();                 // (a head inside the detector's step: the caller goes on with the input gradient)
}
