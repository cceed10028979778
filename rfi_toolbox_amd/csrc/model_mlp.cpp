// The box head of Faster / Mask R-CNN (BASELINE.json configs[3]; arch 6) as a stack of fully
// connected layers on the matrix cores.  NOT in the reference (no detector there) and torchvision is absent:
// builder-defined as the published head (Girshick 2015 / Lin et al. 2017: two FC layers of 1024 units + ReLU on the
// flattened 7 x 7 x 256 RoI features, then the class scores and the per-class box deltas; torchvision's TwoMLPHead +
// FastRCNNPredictor), oracle/mask_head_ref.py.
//
//   x [R, D]  ->  fc6 (D -> H) + ReLU  ->  fc7 (H -> H) + ReLU  ->  head (H -> 5 (K + 1)):  K + 1 class logits, then
//                                                                    (K + 1) x 4 box deltas (cls_score and bbox_pred stacked)
//
// A fully connected layer is a 1x1 conv over the R "pixels"; they are laid out as one [R / 32] x 32 image so that the
// conv kernels' 32-pixel-wide tiles are full.  The loss (cross-entropy + smooth L1 of the ground-truth class's deltas)
// needs per-RoI targets and lives outside the model (rfi_op_fastrcnn_loss -> rfi_model_backward_dlogits), as for the RPN.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

void rfi_model::build_mlp() {
    RFI_REQUIRE(in_ch > 0 && in_ch % 4 == 0 && feat > 0 && feat % 4 == 0 && out_ch > 0 && depth >= 1 && depth <= 8,
                "BoxHead: in_features and hidden width must be positive multiples of 4, 1..8 layers");
    const int L = depth;
    convs.clear();
    ups.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    for (int i = 0; i < L; ++i) {
        ConvBN c;
        c.conv_name = "fc" + std::to_string(6 + i);
        c.has_bn = false;
        c.R = 1;
        c.cin = c.cin_p = i == 0 ? in_ch : feat;
        c.cout = feat;
        c.w_off = off; off = align4(off + (size_t)c.cin * c.cout);
        c.b_off = off; off = align4(off + c.cout);
        c.g_off = c.be_off = 0;
        chan_floats += align4((size_t)8 * c.cout);
        wd_floats += align4((size_t)c.cin * c.cout);
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
    for (int i = 0; i < L; ++i) {
        Entry e;
        e.layer = i;
        e.name = convs[i].conv_name + ".weight"; e.ndim = 4; e.dims[0] = convs[i].cout; e.dims[1] = convs[i].cin; e.dims[2] = 1; e.dims[3] = 1;
        e.kind = 7;
        push(e);
        e = Entry(); e.layer = i;
        e.name = convs[i].conv_name + ".bias"; e.ndim = 1; e.dims[0] = convs[i].cout; e.kind = 2; e.which = 0;
        push(e);
    }
    {
        Entry e;
        e.name = "head.weight"; e.ndim = 4; e.dims[0] = out_ch; e.dims[1] = feat; e.dims[2] = 1; e.dims[3] = 1; e.kind = 6;
        push(e);
        e = Entry();
        e.name = "head.bias"; e.ndim = 1; e.dims[0] = out_ch; e.kind = 2; e.which = 4;
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
        c.wd = wd_pool + wo; wo += align4((size_t)c.cin * c.cout);
    }
    adam_step = 0;
    wd_dirty = true;
    x3_fresh = false;
    reset_channel_state();
}

namespace {
struct Lay { int N, H, W; };
// R rows as an image: [R / 32] x 32 when possible (full 32-wide tiles), else R x 1 x 1
Lay layout_of(int rows) { return rows % 32 == 0 ? Lay{1, rows / 32, 32} : Lay{rows, 1, 1}; }
InXform relu_of(const ConvBN& c) { return InXform{c.scale(), c.shift(), 1}; }
}  // namespace

void rfi_model::prepare_mlp(int n, int h, int w) {
    RFI_REQUIRE(h == 1 && w == 1, "BoxHead: the input is [R, in_features] (n = R, h = w = 1)");
    if (n == pN && !bufs.empty()) return;
    ctx->activate();
    const int L = depth;
    if (bufs.empty()) {
        mkY.assign(L, -1); mkG.assign(L, -1);
        for (int i = 0; i < L; ++i) { mkY[i] = new_buf(); mkG[i] = new_buf(); }
        mkGx = new_buf();
        logits = new_buf(); dlogits = new_buf();
        x_stage = new_buf(); x_stage2 = new_buf(); x_pad = new_buf(); out_stage = new_buf();
        ws_red = new_buf(); ws_slab = new_buf(); lab_stage = new_buf();
    }
    const size_t M = (size_t)n;
    for (int i = 0; i < L; ++i) { bufs[mkY[i]].ensure(ctx, M * feat); bufs[mkG[i]].ensure(ctx, M * feat); }
    bufs[mkGx].ensure(ctx, M * in_ch);
    bufs[logits].ensure(ctx, M * out_ch);
    bufs[dlogits].ensure(ctx, M * out_ch);
    bufs[x_stage].ensure(ctx, M * in_ch);
    bufs[x_stage2].ensure(ctx, M * in_ch);
    bufs[x_pad].ensure(ctx, 16);
    bufs[out_stage].ensure(ctx, M * out_ch);
    bufs[lab_stage].ensure(ctx, (M + 3) / 4 + 4);
    size_t red_need = std::max({head_bwd_ws_floats((int64_t)M, feat, out_ch), channel_sum_ws_floats((int64_t)M, feat),
                                sumsq_ws_doubles((int64_t)n_flat) * 2});
    bufs[ws_red].ensure(ctx, red_need + 16);
    size_t slab_need = 0;
    const Lay s = layout_of(n);
    for (auto& c : convs) {
        WgradArgs a;
        a.N = s.N; a.H = s.H; a.W = s.W; a.Hx = s.H; a.Wx = s.W;
        a.Cx = c.cin; a.Cy = c.cout;
        a.xop.pstride = a.Cx; a.yop.pstride = a.Cy;
        a.R = 1; a.S = 1; a.pad = 0;
        a.tap_stride = (int64_t)a.Cx * a.Cy;
        a.bf16x3 = true;
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
    }
    bufs[ws_slab].ensure(ctx, slab_need + 16);
    pN = n; pH = 1; pW = 1;
}

void rfi_model::forward_mlp(const float* x_dev, int n) {
    refresh_dgrad_weights();
    const int L = depth;
    const Lay s = layout_of(n);
    for (int i = 0; i < L; ++i) {
        ConvBN& c = convs[i];
        ConvArgs a;
        a.x = i == 0 ? View{x_dev, in_ch} : View{buf(mkY[i - 1]), feat};
        a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
        a.Cin = c.cin; a.Cout = c.cout;
        a.w = params + c.w_off;
        a.w3 = use_w3() ? c.w3 : nullptr;
        ws_set(a);
        a.bias = params + c.b_off;
        a.y = MutView{buf(mkY[i]), c.cout};
        a.Hout = s.H; a.Wout = s.W;
        a.R = 1; a.S = 1; a.pad = 0;
        if (i > 0) a.xf = relu_of(convs[i - 1]);
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
    }
    const ConvBN& cl = convs[L - 1];
    launch_head_fwd(ctx, buf(mkY[L - 1]), n, feat, cl.scale(), cl.shift(), params + head_w_off, params + head_b_off, out_ch, buf(logits));
}

// dlogits are the caller's (rfi_model_backward_dlogits): this head has no loss of its own
void rfi_model::backward_mlp(const float* x_dev, int n) {
    RFI_REQUIRE(ext_dlogits, "BoxHead: the loss lives outside the model (rfi_op_fastrcnn_loss + rfi_model_backward_dlogits)");
    side_bound = 0;
    const int L = depth;
    const Lay s = layout_of(n);
    const int64_t M = n;
    refresh_dgrad_weights();
    const ConvBN& cl = convs[L - 1];
    launch_head_bwd(ctx, buf(mkY[L - 1]), M, feat, cl.scale(), cl.shift(), params + head_w_off, out_ch, buf(dlogits), buf(mkG[L - 1]),
                    buf(ws_red), grads + head_w_off, grads + head_b_off);
    for (int i = L - 1; i >= 0; --i) {
        ConvBN& c = convs[i];
        float* dA = buf(mkG[i]);
        launch_relu_bwd(ctx, dA, buf(mkY[i]), M * c.cout);
        launch_channel_sum(ctx, View{dA, c.cout}, M, c.cout, buf(ws_red), grads + c.b_off);
        WgradArgs wa;
        wa.xop = i == 0 ? View{x_dev, in_ch} : View{buf(mkY[i - 1]), feat};
        if (i > 0) wa.xf_x = relu_of(convs[i - 1]);
        wa.yop = View{dA, c.cout};
        wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = s.H; wa.Wx = s.W;
        wa.Cx = c.cin; wa.Cy = c.cout;
        wa.R = 1; wa.S = 1; wa.pad = 0;
        wa.dw = grads + c.w_off;
        wa.tap_stride = (int64_t)c.cin * c.cout;
        wa.sy = c.cin; wa.sx = 1;
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        {   // side stream: every layer owns its gradient tensor mkG[i], nothing the weight gradient reads is rewritten in this pass
            struct Back { rfi_ctx* c; ~Back() { c->stream = c->main_stream; } } back{ctx};
            side_begin();
            launch_wgrad(ctx, wa);
            side_end();
        }
        ConvArgs a;
        a.x = View{dA, c.cout};
        a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
        a.Cin = c.cout; a.Cout = c.cin;
        a.w = c.wd;
        a.w3 = use_w3() ? c.wd3 : nullptr;
        ws_set(a);
        a.y = MutView{i == 0 ? buf(mkGx) : buf(mkG[i - 1]), c.cin};
        a.Hout = s.H; a.Wout = s.W;
        a.R = 1; a.S = 1; a.pad = 0;
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
    }
    side_join_lazy();                 // (the caller goes on with the input gradient; the weight gradients are needed at the optimiser step)
}
