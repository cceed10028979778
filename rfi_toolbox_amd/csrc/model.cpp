// U-Net graph (rfi_toolbox/models/unet.py:41-118) and the optimisation step
// (rfi_toolbox/scripts/train_model.py:139-154) orchestrated over the HIP kernels.
//
// Data layout in HBM
//   activations  NHWC fp32.  Per 3x3 conv only the RAW conv output Y (pre-BatchNorm) is kept;
//                consumers apply scale/shift/ReLU when they load it (InXform), so the activated
//                tensor is never written except where two consumers need it materialised:
//                the encoder output goes once, activated, into channels [C,2C) of the decoder's
//                concat buffer (the skip) and, pooled, into the next level's input.  ConvTranspose
//                writes channels [0,C) of the same concat buffer: torch.cat never happens.
//   parameters   one flat fp32 buffer (+ identical flat grad / Adam m / Adam v buffers) in forward
//                layer order, every tensor 16-byte aligned: conv weights [tap][cout][cin],
//                convT weights [a*2+b][cout][cin]; a second "dgrad layout" copy [tap'][cin][cout]
//                is rebuilt after each optimiser step.
//   Encoder blocks are evaluated ONCE; the reference evaluates them twice (unet.py:28), which
//   only shows in the BatchNorm running statistics (EMA applied twice, num_batches_tracked += 2).
#include "model.hpp"

#include <cmath>
#include <cstdlib>
#include <random>

using namespace rfi;

namespace rfi {

void DevBuf::ensure(rfi_ctx* c, size_t floats) {
    if (floats <= n && p) return;
    if (p) c->release(p);
    ctx = c;
    p = static_cast<float*>(c->alloc(floats * sizeof(float)));
    n = floats;
}
void DevBuf::free() {
    if (p && ctx) ctx->release(p);
    p = nullptr;
    n = 0;
}

}  // namespace rfi

static size_t align4(size_t v) { return (v + 3) & ~size_t(3); }

rfi_model::~rfi_model() {
    if (!ctx) return;
    ctx->activate();
    for (auto& b : bufs) b.free();
    for (auto& b : pl) b.free();
    if (wb_pool) ctx->release(wb_pool);
    if (wb_descs) ctx->release(wb_descs);
    if (ws_pool) ctx->release(ws_pool);
    if (dbias_pool) ctx->release(dbias_pool);
    if (dbias_descs) ctx->release(dbias_descs);
    if (ws_descs) ctx->release(ws_descs);
    for (float* p : {params, grads, adam_m, adam_v, chan_pool, wd_pool, w3_pool, rs_wpool, rs_cls_pool, grad_acc})   // (rs_wpool: arch 2 and 5)
        if (p) ctx->release(p);
    if (relayout_descs) ctx->release(relayout_descs);
    if (x3_descs) ctx->release(x3_descs);
    if (d_sums) ctx->release(d_sums);
    if (d_scalars) ctx->release(d_scalars);
    if (wd_ready) (void)hipEventDestroy(wd_ready);
    if (lazy_ev) (void)hipEventDestroy(lazy_ev);
    for (hipEvent_t e : skip_done) if (e) (void)hipEventDestroy(e);
}

// ------------------------------------------------------------------------------------ build
void rfi_model::build() {
    if (const char* e = getenv("RFI_COMPUTE")) {      // arithmetic of new models: f32 (default) | f32mfma | bf16
        compute_bf16 = std::string(e) == "bf16" || std::string(e) == "bf16regs";
        compute_x3 = std::string(e) == "f32" || std::string(e) == "f32x3" || std::string(e) == "f32planes";
        planesP = std::string(e) == "bf16" ? 1 : (std::string(e) == "f32planes" ? 3 : 0);
    }
    if (const char* e = getenv("RFI_BN_FUSE")) fuse_bn_bwd = e[0] == '1';
    if (arch != 0 && !(arch == 2 && planesP == 1 && feat % 16 == 0)) planesP = 0;     // the plane data flow: the plain U-Net; the
                                                                                      // ResNet-encoder U-Net's bfloat16 flow
    if (arch == 1) return build_cnn3();
    if (arch == 3 || arch == 4) return build_mask();
    if (arch == 5) return build_backbone();
    if (arch == 6) return build_mlp();
    if (arch == 2) return build_resnet();
    RFI_REQUIRE(in_ch > 0 && out_ch > 0 && feat > 0, "UNet: channel counts must be positive");
    RFI_REQUIRE(depth >= 1 && depth <= 6, "UNet: depth must be in [1,6]");
    const int D = depth;
    convs.clear();
    ups.clear();
    size_t off = 0, chan_floats = 0, wd_floats = 0;
    i_bott = 2 * D;
    auto add_conv = [&](const std::string& prefix, int conv_idx, int bn_idx, int cin, int cout, int ema, int lvl) {
        ConvBN c;
        c.conv_name = prefix + "." + std::to_string(conv_idx);
        c.bn_name = prefix + "." + std::to_string(bn_idx);
        c.cin = cin;
        // only the network input can be padded (its staging buffer is ours); inner layers read
        // tensors whose pixel stride is the true channel count
        c.cin_p = convs.empty() ? (int)align4((size_t)cin) : cin;
        c.cout = cout;
        c.ema_repeats = ema;
        c.level = lvl;
        c.w_off = off; off = align4(off + (size_t)9 * c.cin_p * cout);
        c.b_off = off; off = align4(off + cout);
        c.g_off = off; off = align4(off + cout);
        c.be_off = off; off = align4(off + cout);
        chan_floats += align4((size_t)8 * cout);
        wd_floats += align4((size_t)9 * c.cin_p * cout);
        convs.push_back(c);
    };
    int cin = in_ch;
    for (int l = 1; l <= D; ++l) {
        const int cout = feat << (l - 1);
        const std::string p = "encoder" + std::to_string(l) + ".conv.conv";
        add_conv(p, 0, 1, cin, cout, 2, l);
        add_conv(p, 3, 4, cout, cout, 2, l);
        cin = cout;
    }
    add_conv("bottleneck.conv", 0, 1, cin, cin * 2, 1, D + 1);
    add_conv("bottleneck.conv", 3, 4, cin * 2, cin * 2, 1, D + 1);
    cin *= 2;
    // decoder parameters come in forward order: up, conv1, conv2 per level; to keep `convs`
    // contiguous the up-convs get their offsets here and the convs right after
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
        add_conv(p, 0, 1, cin, cout, 1, l);
        add_conv(p, 3, 4, cout, cout, 1, l);
        cin = cout;
    }
    head_w_off = off; off = align4(off + (size_t)out_ch * feat);
    head_b_off = off; off = align4(off + out_ch);
    n_flat = off;

    // ---- state_dict entry table in the reference's order
    entries.clear();
    entry_index.clear();
    n_params = 0;
    auto push = [&](Entry e) {
        entry_index[e.name] = (int)entries.size();
        if (e.kind == 0 || e.kind == 1 || e.kind == 2 || e.kind == 6) n_params += e.numel();
        entries.push_back(e);
    };
    auto push_conv_entries = [&](int ci) {
        const ConvBN& c = convs[ci];
        Entry e;
        e.layer = ci;
        e.name = c.conv_name + ".weight"; e.ndim = 4; e.dims[0] = c.cout; e.dims[1] = c.cin; e.dims[2] = 3; e.dims[3] = 3; e.kind = 0; push(e);
        e = Entry(); e.layer = ci;
        e.name = c.conv_name + ".bias"; e.ndim = 1; e.dims[0] = c.cout; e.kind = 2; e.which = 0; push(e);
        e.name = c.bn_name + ".weight"; e.which = 1; push(e);
        e.name = c.bn_name + ".bias"; e.which = 2; push(e);
        e.name = c.bn_name + ".running_mean"; e.kind = 3; push(e);
        e.name = c.bn_name + ".running_var"; e.kind = 4; push(e);
        e.name = c.bn_name + ".num_batches_tracked"; e.kind = 5; e.ndim = 0; e.dims[0] = 0; push(e);
    };
    for (int l = 1; l <= D; ++l) {
        push_conv_entries(2 * (l - 1));
        push_conv_entries(2 * (l - 1) + 1);
    }
    push_conv_entries(2 * D);
    push_conv_entries(2 * D + 1);
    for (int l = D; l >= 1; --l) {
        const int k = D - l;
        const UpConv& u = ups[k];
        Entry e;
        e.layer = k;
        e.name = u.name + ".weight"; e.ndim = 4; e.dims[0] = u.cin; e.dims[1] = u.cout; e.dims[2] = 2; e.dims[3] = 2; e.kind = 1; push(e);
        e = Entry(); e.layer = k;
        e.name = u.name + ".bias"; e.ndim = 1; e.dims[0] = u.cout; e.kind = 2; e.which = 3; push(e);
        push_conv_entries(2 * D + 2 + 2 * k);
        push_conv_entries(2 * D + 2 + 2 * k + 1);
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
    size_t conv_i = 0;
    // wd_pool sub-allocation must follow the same order as the sizing above
    for (int l = 1; l <= 2 * D + 2; ++l) {   // encoder + bottleneck convs
        ConvBN& c = convs[conv_i++];
        c.chan = chan_pool + co; co += align4((size_t)8 * c.cout);
        c.wd = wd_pool + wo; wo += align4((size_t)9 * c.cin_p * c.cout);
    }
    for (int k = 0; k < D; ++k) {
        ups[k].wd = wd_pool + wo; wo += align4((size_t)4 * ups[k].cin * ups[k].cout);
        for (int j = 0; j < 2; ++j) {
            ConvBN& c = convs[conv_i++];
            c.chan = chan_pool + co; co += align4((size_t)8 * c.cout);
            c.wd = wd_pool + wo; wo += align4((size_t)9 * c.cin_p * c.cout);
        }
    }
    adam_step = 0;
    wd_dirty = true;
    x3_fresh = false;
    reset_channel_state();
}

void rfi_model::set_planes(int P) {
    if (P == planesP) return;
    ctx->activate();
    RFI_CHECK_HIP(hipStreamSynchronize(ctx->main_stream));
    RFI_CHECK_HIP(hipStreamSynchronize(ctx->side_stream));
    for (auto& b : pl) b.free();                  // tensors and filter copies of the other P
    if (wb_pool) { ctx->release(wb_pool); wb_pool = nullptr; }
    if (wb_descs) { ctx->release(wb_descs); wb_descs = nullptr; }
    planesP = P;
    wd_dirty = true;
    pN = 0;                                       // (prepare() again: the two flows of the ResNet-encoder model own different tensors)
}

void rfi_model::reset_channel_state() {
    for (auto& c : convs) {
        std::vector<float> ch((size_t)8 * c.cout, 0.0f);
        for (int i = 0; i < c.cout; ++i) ch[c.cout + i] = 1.0f;                          // running_var = 1
        if (!c.has_bn)
            for (int i = 0; i < c.cout; ++i) ch[(size_t)4 * c.cout + i] = 1.0f;          // scale = 1, shift = 0
        RFI_CHECK_HIP(hipMemcpyAsync(c.chan, ch.data(), ch.size() * sizeof(float), hipMemcpyHostToDevice,
                                     ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        c.nbt = 0;
    }
}

// ------------------------------------------------------------------------------------ prepare
void rfi_model::prepare(int n, int h, int w) {
    RFI_REQUIRE(n > 0 && h > 0 && w > 0, "forward: empty batch or image");
    join_pending_side();                          // (the last pass's weight gradients still read this model's tensors)
    if (ctx->stream != ctx->main_stream) {      // a side-stream launch threw last time: rejoin first
        ctx->stream = ctx->main_stream;
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->side_stream));
        side_seq = 0;
    }
    if (side_seq != 0 || ctx->fork_ring_used != 0 || ctx->bucket_ev_used != 0 || pend_hi > pend_lo) {
        // the previous pass did not reach side_join / exchange_join (an exception mid-backward): drain every stream and
        // start from clean counters and event pools
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->main_stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->side_stream));
        if (ctx->comm_stream) RFI_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream));
        side_seq = 0;
        ctx->fork_ring_used = 0;
        ctx->bucket_ev_used = 0;
        pend_lo = pend_hi = 0;
    }
    if (arch == 1) return prepare_cnn3(n, h, w);
    if (arch == 3 || arch == 4) return prepare_mask(n, h, w);
    if (arch == 5) return prepare_backbone(n, h, w);
    if (arch == 6) return prepare_mlp(n, h, w);
    const int div = 1 << depth;
    RFI_REQUIRE(h % div == 0 && w % div == 0,
                "forward: H and W must be multiples of 2^depth (" + std::to_string(div) +
                    "), as the reference's pool/up-conv/cat chain requires; got " +
                    std::to_string(h) + "x" + std::to_string(w));
    if (n == pN && h == pH && w == pW && !bufs.empty()) return;     // (set_planes() clears pN: the table below depends on the flow)
    ctx->activate();
    const int D = depth;
    if (bufs.empty()) {
        auto mk = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) v[l] = new_buf(); };
        mk(encY1); mk(encY2); mk(concat); mk(pool); mk(decY1); mk(decY2);
        mk(gA); mk(gB); mk(dconcat); mk(dpool); mk(gAe); mk(gBe);
        bottY1 = new_buf(); bottY2 = new_buf(); gBottA = new_buf(); gBottB = new_buf();
        logits = new_buf(); dlogits = new_buf(); probs = new_buf();
        x_stage = new_buf(); x_stage2 = new_buf(); x_pad = new_buf(); out_stage = new_buf();
        ws_red = new_buf(); ws_slab = new_buf(); lab_stage = new_buf();
    }
    size_t slab_need = 0, red_need = 0;
    auto upd_red = [&](size_t f) { if (f > red_need) red_need = f; };
    for (int l = 1; l <= D; ++l) {
        const size_t M = (size_t)n * (h >> (l - 1)) * (w >> (l - 1));
        const size_t C = (size_t)feat << (l - 1);
        for (int i : {decY1[l], decY2[l], gA[l], gB[l]}) bufs[i].ensure(ctx, M * C);
        if (!planesP && arch == 0) for (int i : {gAe[l], gBe[l]}) bufs[i].ensure(ctx, M * C);   // (the encoder phase's own gradient tensors)
        if (arch != 2) for (int i : {encY1[l], encY2[l]}) bufs[i].ensure(ctx, M * C);     // (arch 2: its own tensors, prepare_resnet)
        bufs[concat[l]].ensure(ctx, M * 2 * C);
        bufs[dconcat[l]].ensure(ctx, M * 2 * C);
        if (arch != 2 || l == D) {
            bufs[pool[l]].ensure(ctx, M / 4 * C);
            bufs[dpool[l]].ensure(ctx, M / 4 * C);
        }
    }
    {
        const size_t M = (size_t)n * (h >> D) * (w >> D);
        const size_t C = (size_t)feat << D;
        for (int i : {bottY1, bottY2, gBottA, gBottB}) bufs[i].ensure(ctx, M * C);
    }
    const size_t M1 = (size_t)n * h * w;
    bufs[logits].ensure(ctx, M1 * out_ch);
    bufs[dlogits].ensure(ctx, M1 * out_ch);
    bufs[probs].ensure(ctx, M1 * out_ch);
    bufs[x_stage].ensure(ctx, M1 * in_ch);
    bufs[x_stage2].ensure(ctx, M1 * in_ch);
    bufs[x_pad].ensure(ctx, M1 * convs[0].cin_p);
    bufs[out_stage].ensure(ctx, M1 * out_ch);
    bufs[lab_stage].ensure(ctx, (M1 + 3) / 4 + 4);
    // workspaces
    size_t cmax = (size_t)feat << depth;
    upd_red(bn_stats_ws_floats((int)cmax));
    upd_red(bn_bwd_ws_floats(M1, (int)cmax));
    upd_red(head_bwd_ws_floats(M1, feat, out_ch) + bn_bwd_ws_floats(M1, feat));   // (+ the BN-backward records head_bwd leaves)
    upd_red(channel_sum_ws_floats(M1, (int)cmax));
    upd_red(loss_ws_doubles(M1) * 2);
    upd_red(sumsq_ws_doubles(n_flat) * 2);
    bufs[ws_red].ensure(ctx, red_need + 16);
    // wgrad slabs: the largest need over all layers
    auto conv_geom = [&](int ci, int& H, int& W) {         // spatial size of conv ci's output
        H = h >> (convs[ci].level - 1);
        W = w >> (convs[ci].level - 1);
    };
    for (size_t ci = 0; ci < convs.size(); ++ci) {
        int H, W;
        conv_geom((int)ci, H, W);
        WgradArgs a;
        a.N = n; a.H = H; a.W = W; a.Hx = H; a.Wx = W;
        a.Cx = convs[ci].cin_p; a.Cy = convs[ci].cout;
        a.R = 3; a.S = 1; a.pad = 1;
        if (convs[ci].stride == 2) { a.Cx *= 4; a.R = 2; }            // its 2x2 form on the space-to-depth input
        if (convs[ci].R == 1) { a.R = 1; a.pad = 0; }
        a.xop.pstride = a.Cx; a.yop.pstride = a.Cy;
        a.tap_stride = (int64_t)a.Cx * a.Cy;
        a.bf16x3 = true;              // (the split-at-staging plan is the largest)
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
    }
    for (int k = 0; k < D; ++k) {
        const int l = D - k;                       // decoder level; input at level l+1 resolution
        WgradArgs a;
        a.N = n; a.H = h >> l; a.W = w >> l; a.Hx = a.H * 2; a.Wx = a.W * 2;
        a.Cx = ups[k].cout; a.Cy = ups[k].cin;
        a.xop.pstride = 2 * ups[k].cout; a.yop.pstride = a.Cy;
        a.R = 2; a.S = 2; a.pad = 0;
        a.tap_stride = (int64_t)a.Cx * a.Cy;
        slab_need = std::max(slab_need, wgrad_slab_floats(a, IMPL_AUTO));
    }
    bufs[ws_slab].ensure(ctx, slab_need + 16);
    // per-layer regions for the partial sums of the conv-bias gradients + the table of the batched finisher
    // (the plane flows too: the plain U-Net's and the ResNet-encoder model's, whose encoder convs have no bias)
    if (arch == 0 || (arch == 2 && planesP)) {
        if (dbias_pool) { ctx->release(dbias_pool); dbias_pool = nullptr; }
        if (dbias_descs) { ctx->release(dbias_descs); dbias_descs = nullptr; }
        size_t need = 0;
        std::vector<FinishSumDesc> hd;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            int H, W;
            conv_geom((int)ci, H, W);
            const int64_t M = (int64_t)n * H * W;
            convs[ci].dbias_rec_off = need;
            need += align4(channel_sum_ws_floats(M, convs[ci].cout)) + 4;
        }
        for (int k = 0; k < depth; ++k) {               // the transposed convs' bias gradients (decoder level l = D - k)
            const int l = depth - k;
            ups[k].dbias_rec_off = need;
            need += align4(channel_sum_ws_floats((int64_t)n * (h >> (l - 1)) * (w >> (l - 1)), ups[k].cout)) + 4;
        }
        head_rec_off = need;
        need += align4(head_bwd_ws_floats((int64_t)n * h * w, feat, out_ch)) + 4;
        dbias_pool = static_cast<float*>(ctx->alloc(need * sizeof(float)));
        dbias_max_c = 0;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            int H, W;
            conv_geom((int)ci, H, W);
            const ConvBN& c = convs[ci];
            if (!c.has_bias) continue;
            hd.push_back(FinishSumDesc{reinterpret_cast<const double*>(dbias_pool + c.dbias_rec_off), bn_bwd_apply_records((int64_t)n * H * W, c.cout),
                                       (int64_t)c.cout, c.cout, grads + c.b_off});
            dbias_max_c = std::max(dbias_max_c, c.cout);
        }
        for (int k = 0; k < depth; ++k) {
            const int l = depth - k;
            const UpConv& u = ups[k];
            if (u.cout % 4) continue;                 // (launch_channel_sum keeps its own finish for such a layer)
            hd.push_back(FinishSumDesc{reinterpret_cast<const double*>(dbias_pool + u.dbias_rec_off),
                                       bn_bwd_apply_records((int64_t)n * (h >> (l - 1)) * (w >> (l - 1)), u.cout), (int64_t)u.cout, u.cout,
                                       grads + u.b_off});
            dbias_max_c = std::max(dbias_max_c, u.cout);
        }
        head_fin_deferred = feat % 4 == 0;
        if (head_fin_deferred) {                       // the head's dw (out_ch x feat) and db (out_ch), contiguous in every record
            const int recs = bn_bwd_apply_records((int64_t)n * h * w, feat);
            const int64_t stride = (int64_t)out_ch * feat + out_ch;
            const double* hp = reinterpret_cast<const double*>(dbias_pool + head_rec_off);
            hd.push_back(FinishSumDesc{hp, recs, stride, out_ch * feat, grads + head_w_off});
            hd.push_back(FinishSumDesc{hp + (size_t)out_ch * feat, recs, stride, out_ch, grads + head_b_off});
            dbias_max_c = std::max(dbias_max_c, out_ch * feat);
        }
        dbias_n = (int)hd.size();
        dbias_descs = ctx->alloc(hd.size() * sizeof(FinishSumDesc));
        RFI_CHECK_HIP(hipMemcpyAsync(dbias_descs, hd.data(), hd.size() * sizeof(FinishSumDesc), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // hd goes out of scope
    }
    if (arch == 2 && !planesP) prepare_resnet(n, h, w);
    pN = n; pH = h; pW = w;
}

// the main stream waits for the side stream's rebuild of the input-gradient-direction filter copies (refresh_dgrad_weights)
void rfi_model::wait_wd() {
    side_rebuild_wd();
    if (!wd_pending) return;
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, wd_ready, 0));
    wd_pending = false;
}

// the side-stream half of the split rebuild: dgrad layout + its B-operand images.  Started by forward() AFTER the first
// conv is enqueued (the stem is bound by its HBM writes: next to it the rebuild slowed it from 33 to 62 us; the convs that
// follow are matrix-bound) -- or by whoever needs the copies first (wait_wd)
void rfi_model::side_rebuild_wd() {
    if (!wd_side_todo) return;
    wd_side_todo = false;
    if (!wd_ready) RFI_CHECK_HIP(hipEventCreateWithFlags(&wd_ready, hipEventDisableTiming));
    side_begin();                                 // the side stream waits for everything on main so far (the optimiser step)
    struct Back { rfi_ctx* c; ~Back() { c->stream = c->main_stream; } } back{ctx};
    launch_weight_to_dgrad_batched(ctx, static_cast<const RelayoutDesc*>(relayout_descs), relayout_n, params,
                                   wd_pool, relayout_bytes, relayout_tiles);
    if (planesP) refresh_plane_weights(2);
    else refresh_ws_weights(ws_need(), 2);
    RFI_CHECK_HIP(hipEventRecord(wd_ready, ctx->side_stream));
    wd_pending = true;
}

void rfi_model::refresh_dgrad_weights() {
    if (!wd_dirty && (!use_w3() || x3_fresh) && ws_P == ws_need()) return;
    if (!relayout_descs) {          // one descriptor per conv-like layer, built once
        std::vector<RelayoutDesc> h;
        relayout_bytes = 0;
        for (auto& c : convs) {
            h.push_back({(int64_t)c.w_off, (int64_t)(c.wd - wd_pool), c.R * c.R, c.cout, c.cin_p, 1});
            relayout_bytes += 8.0 * c.R * c.R * c.cout * c.cin_p;
        }
        for (auto& u : ups) {
            h.push_back({(int64_t)u.w_off, (int64_t)(u.wd - wd_pool), 4, u.cout, u.cin, 0});
            relayout_bytes += 8.0 * 4 * u.cout * u.cin;
        }
        relayout_n = (int)h.size();
        relayout_tiles = relayout_assign_tiles(h.data(), relayout_n);
        relayout_descs = ctx->alloc(h.size() * sizeof(RelayoutDesc));
        RFI_CHECK_HIP(hipMemcpyAsync(relayout_descs, h.data(), h.size() * sizeof(RelayoutDesc),
                                     hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // h goes out of scope
    }
    // Split rebuild (plain U-Net, float32 tensors, overlap on): the forward pass needs the forward-direction copies only, so
    // the dgrad layout and its B-operand images are rebuilt on the SIDE stream (idle during the forward pass) under the first
    // convs; backward() waits for them (wait_wd).  RFI_NO_WD_SIDE=1: everything on the main stream, as before
    static const bool no_wd_side = getenv("RFI_NO_WD_SIDE") != nullptr;
    // (the plane flows: the forward pass reads the forward-direction images only, nothing but the input-gradient kernels reads
    // the dgrad layouts -- unless pre-split records of them are in use)
    const bool split_planes = planesP && wb_pool && wb_n_fwd > 0 && !use_w3() && training;
    const bool split = !no_wd_side && ctx->overlap && !ctx->profiling && ctx->stream == ctx->main_stream &&
                       (split_planes || (arch == 0 && !planesP && ws_need() != 0 && ws_pool && ws_P == ws_need() && ws_n_fwd > 0 &&
                                         (!use_w3() || (x3_descs && !x3_reads_wd && x3_for_ws_P == ws_P))));
    if (split) {
        wd_side_todo = true;                      // (forward() starts it behind the first conv: side_rebuild_wd)
    } else {
        wait_wd();
        launch_weight_to_dgrad_batched(ctx, static_cast<const RelayoutDesc*>(relayout_descs), relayout_n, params,
                                       wd_pool, relayout_bytes, relayout_tiles);
    }
    refresh_ws_weights(ws_need(), split ? 1 : 0);
    if (use_w3()) {       // pre-split records of both layouts of the conv-like layers (the round-2 kernels read them as is)
        if (!w3_pool) {
            size_t need = 0;
            for (auto& c : convs) need += weights_x3_floats(c.R * c.R, c.cout, c.cin_p) + weights_x3_floats(c.R * c.R, c.cin_p, c.cout);
            for (auto& u : ups) need += 2 * weights_x3_floats(4, u.cout, u.cin) + weights_x3_floats(4, u.cin, u.cout);
            w3_pool = static_cast<float*>(ctx->alloc((need + 64) * sizeof(float)));
            size_t o = 0;
            for (auto& c : convs) {
                c.w3 = w3_pool + o; o += weights_x3_floats(c.R * c.R, c.cout, c.cin_p);
                c.wd3 = w3_pool + o; o += weights_x3_floats(c.R * c.R, c.cin_p, c.cout);
            }
            for (auto& u : ups) {
                u.w3 = w3_pool + o; o += weights_x3_floats(4, u.cout, u.cin);
                u.wd3 = w3_pool + o; o += weights_x3_floats(4, u.cin, u.cout);
            }
        }
        // layers whose filters the wave-specialised kernels read (ws_by_w) need no records: launch_conv drops
        // ConvArgs::w3 for them, and a shape those kernels decline (maps under 8 x 8) gets a temporary split copy
        static const bool all_x3 = getenv("RFI_NO_WS") != nullptr || getenv("RFI_NO_GW") != nullptr;     // A/B runs
        if (x3_descs && (x3_for_ws_P != ws_P || x3_for_shape != pH * 65536 + pW)) {
            ctx->release(x3_descs);
            x3_descs = nullptr;
        }
        if (!x3_descs) {
            x3_for_ws_P = ws_P;
            x3_bytes = 0;
            std::vector<X3Desc> h;
            x3_skips_ws_layers = !all_x3 && ws_P == 3 && arch == 0;
            x3_reads_wd = false;
            x3_skipped.clear();
            // a layer is left out only if the wave-specialised kernels cannot decline it at the prepared shape: its maps are at
            // least 8 x 8 (conv_ws_eligible) -- a declined launch on a missing record would allocate a temporary copy and
            // synchronise the stream inside launch_conv.  (pH == 0: nothing prepared yet, keep everything)
            auto add = [&](const float* src, float* dst, int taps, int cout, int cin, int level = 0) {
                const bool small_map = pH == 0 || (level > 0 && ((pH >> (level - 1)) < 8 || (pW >> (level - 1)) < 8));
                if (x3_skips_ws_layers && ws_by_w.count(src) && !small_map) { x3_skipped.insert(src); return; }
                if (src < params || src >= params + n_flat) x3_reads_wd = true;
                h.push_back(X3Desc{src, dst, (int64_t)taps * cout, cin, (cin + 15) / 16});
                x3_bytes += (double)taps * cout * cin * 4 + (double)weights_x3_floats(taps, cout, cin) * 4;
            };
            for (auto& c : convs) {
                add(params + c.w_off, c.w3, c.R * c.R, c.cout, c.cin_p, c.level);
                add(c.wd, c.wd3, c.R * c.R, c.cin_p, c.cout, c.level);
            }
            for (auto& u : ups) {
                add(params + u.w_off, u.w3, 4, u.cout, u.cin);
                add(u.wd, u.wd3, 4, u.cin, u.cout);
            }
            x3_for_shape = pH * 65536 + pW;
            x3_n = (int)h.size();
            x3_descs = ctx->alloc((h.size() + 1) * sizeof(X3Desc));
            if (x3_n) RFI_CHECK_HIP(hipMemcpyAsync(x3_descs, h.data(), h.size() * sizeof(X3Desc), hipMemcpyHostToDevice, ctx->stream));
            RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // h goes out of scope
        }
        if (x3_n) launch_weights_to_x3_batched(ctx, static_cast<const X3Desc*>(x3_descs), x3_n, x3_bytes);
        x3_fresh = true;
    }
    if (arch == 2 && !planesP) refresh_resnet_weights();   // 2x2 forms of the stride-2 filters
    if (planesP) refresh_plane_weights(split ? 1 : 0);     // B-operand-order filters of the plane kernels
    wd_dirty = false;
}

// filters of every 3x3 stride-1 layer in MFMA B-operand order with three planes (conv_ws.hip), both directions, rebuilt
// with the other derived copies after each optimiser step by ONE batched launch.  Callers find them by the layer's
// float32 filter pointer (ws_set(ConvArgs) looks up ConvArgs::w), which every model's conv helper already passes around
void rfi_model::refresh_ws_weights(int P, int which) {
    if (P != ws_P) {                  // another arithmetic: its copies have another size
        if (ws_pool) { ctx->release(ws_pool); ws_pool = nullptr; }
        if (ws_descs) { ctx->release(ws_descs); ws_descs = nullptr; }
        ws_by_w.clear();
        ws_n = ws_n_fwd = 0;
        ws_P = P;
    }
    if (P == 0) return;
    if (!ws_pool) {
        // conv_ws: 3x3 stride-1 layers with channels % 16 == 0; gemm_ws (P = 3): the transposed convs
        auto conv_f = [&](const ConvBN& c) { return c.R == 3 && c.stride == 1 && c.cin_p % 16 == 0; };
        auto conv_d = [&](const ConvBN& c) { return c.R == 3 && c.stride == 1 && c.cout % 16 == 0; };
        auto up_ok = [&](const UpConv& u) { return P == 3 && u.cin % 16 == 0 && u.cout % 32 == 0; };
        // gemm_ws also runs the plain 1x1 stride-1 convs (the Bottleneck and pyramid convs of the detection backbone) as GEMMs
        auto one_ok = [&](const ConvBN& c) { return P == 3 && c.R == 1 && c.stride == 1 && c.cin_p % 32 == 0 && c.cout % 32 == 0; };
        size_t need = 0;
        for (const ConvBN& c : convs) {
            if (conv_f(c)) need += wb_elems(9, c.cout, c.cin_p, 0, P) + 32;
            if (conv_d(c)) need += wb_elems(9, c.cin_p, c.cout, 0, P) + 32;
            if (one_ok(c)) need += wb_elems(1, c.cout, c.cin_p, 0, P) + wb_elems(1, c.cin_p, c.cout, 0, P) + 64;
        }
        for (const UpConv& u : ups)
            if (up_ok(u)) need += wb_elems(1, 4 * u.cout, u.cin, 0, P) + wb_elems(4, u.cin, u.cout, 0, P) + 64;
        if (need == 0) return;
        ws_pool = static_cast<bf16_t*>(ctx->alloc(need * 2));
        RFI_CHECK_HIP(hipMemsetAsync(ws_pool, 0, need * 2, ctx->stream));
        // forward-direction images first (sources in `params`), then the input-gradient direction (sources in wd_pool): the
        // two halves can be rebuilt by separate launches (refresh_dgrad_weights puts the second on the side stream)
        std::vector<WBDesc> hd, hdg;
        size_t o = 0;
        ws_bytes = ws_bytes_fwd = 0;
        for (ConvBN& c : convs) {
            if (conv_f(c)) {
                c.ws3f = ws_pool + o;
                const size_t e = wb_elems(9, c.cout, c.cin_p, 0, P);
                o += e + 32;
                hd.push_back(WBDesc{params + c.w_off, c.ws3f, 9, c.cout, c.cin_p, {c.cin_p, 0}, P});
                ws_by_w[params + c.w_off] = c.ws3f;
                ws_bytes_fwd += 2.0 * e + 4.0 * 9 * c.cin_p * c.cout;
            }
            if (conv_d(c)) {
                c.ws3d = ws_pool + o;
                const size_t e = wb_elems(9, c.cin_p, c.cout, 0, P);
                o += e + 32;
                hdg.push_back(WBDesc{c.wd, c.ws3d, 9, c.cin_p, c.cout, {c.cout, 0}, P});
                ws_by_w[c.wd] = c.ws3d;
                ws_bytes += 2.0 * e + 4.0 * 9 * c.cin_p * c.cout;
            }
            if (one_ok(c)) {
                const size_t ef = wb_elems(1, c.cout, c.cin_p, 0, P), ed = wb_elems(1, c.cin_p, c.cout, 0, P);
                hd.push_back(WBDesc{params + c.w_off, ws_pool + o, 1, c.cout, c.cin_p, {c.cin_p, 0}, P});
                ws_by_w[params + c.w_off] = ws_pool + o;
                o += ef + 32;
                hdg.push_back(WBDesc{c.wd, ws_pool + o, 1, c.cin_p, c.cout, {c.cout, 0}, P});
                ws_by_w[c.wd] = ws_pool + o;
                o += ed + 32;
                ws_bytes_fwd += 2.0 * ef + 4.0 * c.cin_p * c.cout;
                ws_bytes += 2.0 * ed + 4.0 * c.cin_p * c.cout;
            }
        }
        // transposed convs (gemm_ws.hip): forward = ONE tap of 4 cout channels ([4][cout][cin] IS [4 cout][cin]); input
        // gradient = four taps of [cin][cout]
        for (UpConv& u : ups) {
            if (!up_ok(u)) continue;
            const size_t ef = wb_elems(1, 4 * u.cout, u.cin, 0, P), ed = wb_elems(4, u.cin, u.cout, 0, P);
            hd.push_back(WBDesc{params + u.w_off, ws_pool + o, 1, 4 * u.cout, u.cin, {u.cin, 0}, P});
            ws_by_w[params + u.w_off] = ws_pool + o;
            o += ef + 32;
            hdg.push_back(WBDesc{u.wd, ws_pool + o, 4, u.cin, u.cout, {u.cout, 0}, P});
            ws_by_w[u.wd] = ws_pool + o;
            o += ed + 32;
            ws_bytes_fwd += 2.0 * ef + 4.0 * 4 * u.cin * u.cout;
            ws_bytes += 2.0 * ed + 4.0 * 4 * u.cin * u.cout;
        }
        ws_n_fwd = (int)hd.size();
        ws_bytes += ws_bytes_fwd;
        hd.insert(hd.end(), hdg.begin(), hdg.end());
        ws_n = (int)hd.size();
        ws_descs = ctx->alloc(hd.size() * sizeof(WBDesc));
        RFI_CHECK_HIP(hipMemcpyAsync(ws_descs, hd.data(), hd.size() * sizeof(WBDesc), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // hd goes out of scope
    }
    const WBDesc* descs = static_cast<const WBDesc*>(ws_descs);
    if (which == 0 && ws_n) launch_weights_to_wb(ctx, descs, ws_n, ws_bytes);
    if (which == 1 && ws_n_fwd) launch_weights_to_wb(ctx, descs, ws_n_fwd, ws_bytes_fwd);
    if (which == 2 && ws_n > ws_n_fwd) launch_weights_to_wb(ctx, descs + ws_n_fwd, ws_n - ws_n_fwd, ws_bytes - ws_bytes_fwd);
}

// ------------------------------------------------------------------------------------ forward
namespace {

struct Shape { int N, H, W; };

void run_conv_bn(rfi_model* m, ConvBN& c, View in, InXform xf, Shape s, float* Y, bool train, hipEvent_t coeffs_done = nullptr) {
    ConvArgs a;
    a.x = in;
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
    a.Cin = c.cin_p; a.Cout = c.cout;
    a.w = m->params + c.w_off;
    a.w3 = m->use_w3() ? c.w3 : nullptr;
    m->ws_set(a);
    a.bias = m->params + c.b_off;
    a.y = MutView{Y, c.cout};
    a.Hout = s.H; a.Wout = s.W;
    a.R = 3; a.S = 1; a.pad = 1;
    a.xf = xf;
    a.algo_flops = 2.0 * s.N * s.H * s.W * 9.0 * c.cin * c.cout;
    float* ws = m->buf(m->ws_red);
    if (train) {            // batch statistics come out of the conv epilogue (no second pass over Y)
        a.stats = reinterpret_cast<double*>(ws);
        a.stats_max_records = (int)(bn_stats_ws_floats(c.cout) / ((size_t)c.cout * 4));
    }
    a.bf16 = m->compute_bf16;
    a.bf16x3 = m->compute_x3;
    launch_conv(m->ctx, a);
    const int64_t M = (int64_t)s.N * s.H * s.W;
    if (train) {
        if (a.stats_records == 0) launch_bn_stats(m->ctx, Y, M, c.cout, ws);   // direct-kernel fallback
        launch_bn_finalize(m->ctx, ws, M, c.cout, m->params + c.g_off, m->params + c.be_off,
                           c.running_mean(), c.running_var(), c.ema_repeats, c.mean(), c.invstd(),
                           c.scale(), c.shift(), nullptr, a.stats_records, coeffs_done);
        c.nbt += c.ema_repeats;
    } else {
        launch_bn_eval_coeffs(m->ctx, c.cout, m->params + c.g_off, m->params + c.be_off, c.running_mean(),
                              c.running_var(), c.scale(), c.shift());
        if (coeffs_done) RFI_CHECK_HIP(hipEventRecord(coeffs_done, m->ctx->stream));
    }
}


}  // namespace

// the first conv sees the input with its channels zero-padded to a multiple of 4 (16-byte pixels),
// so the 3-channel stem runs on the same MFMA kernels as every other layer
rfi::View rfi_model::network_input(const float* x_dev, int n, int h, int w) {
    const int cp = convs[0].cin_p;
    if (cp == in_ch) return View{x_dev, in_ch};
    launch_pad_channels(ctx, x_dev, (int64_t)n * h * w, in_ch, cp, buf(x_pad));
    return View{buf(x_pad), cp};
}

void rfi_model::forward(const float* x_dev, int n, int h, int w, bool train_mode) {
    prepare(n, h, w);
    refresh_dgrad_weights();          // derived filter copies (dgrad layout, 3 x bf16 records) follow the parameters
    if (arch == 1) return forward_cnn3(x_dev, n, h, w);
    if (arch == 3 || arch == 4) return forward_mask(x_dev, n, h, w);
    if (arch == 5) return forward_backbone(x_dev, n, h, w);
    if (arch == 6) return forward_mlp(x_dev, n);
    if (planesP) return forward_planes(x_dev, n, h, w, train_mode);
    const int D = depth, IB = i_bott;
    View cur = network_input(x_dev, n, h, w);
    if (arch == 2) cur = forward_resnet_encoder(cur, n, h, w, train_mode);
    else for (int l = 1; l <= D; ++l) {
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        run_conv_bn(this, c1, cur, InXform{}, s, buf(encY1[l]), train_mode);
        if (l == 1) side_rebuild_wd();
        // The pooled tensor feeds the next conv at once; the skip (the activated output in the decoder's concat buffer) is not
        // read before the decoder.  RFI_POOL_SPLIT=1: the main stream writes the pooled tensor only (a quarter of the bytes)
        // and the skip is written on the side stream under the next level's matrix-bound convs.  OFF by default: measured
        // 6.69 against 6.64 ms per step (round 4, two interleaved pairs) -- the second read of Y and the skip write next to
        // the convs cost them more than the 45 us the main stream saves
        static const bool no_split = getenv("RFI_POOL_SPLIT") == nullptr;
        const bool split_pool = !no_split && train_mode && ctx->overlap && ctx->stream == ctx->main_stream && !(s.H & 1) && !(s.W & 1);      // (training passes: the backward pass's side_join recycles the events)
        const hipEvent_t coeffs = split_pool ? next_fork_event() : nullptr;
        run_conv_bn(this, c2, View{buf(encY1[l]), c1.cout}, bn_xf(c1), s, buf(encY2[l]), train_mode, coeffs);
        const MutView skip{buf(concat[l]) + c2.cout, 2 * c2.cout};
        if (split_pool && coeffs) {
            launch_bn_relu_pool(ctx, buf(encY2[l]), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), MutView{}, buf(pool[l]), act_slope);
            if ((int)skip_done.size() <= l) skip_done.resize(l + 1, nullptr);
            if (!skip_done[l]) RFI_CHECK_HIP(hipEventCreateWithFlags(&skip_done[l], hipEventDisableTiming));
            side_begin_after(coeffs);             // (the side stream waits for c2's scale / shift only)
            struct Back { rfi_ctx* c; ~Back() { c->stream = c->main_stream; } } back{ctx};
            launch_bn_relu_pool(ctx, buf(encY2[l]), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), skip, nullptr, act_slope);
            RFI_CHECK_HIP(hipEventRecord(skip_done[l], ctx->side_stream));
            skip_pending |= 1u << l;
        } else {
            launch_bn_relu_pool(ctx, buf(encY2[l]), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), skip, buf(pool[l]), act_slope);
        }
        cur = View{buf(pool[l]), c2.cout};
    }
    {
        Shape s{n, h >> D, w >> D};
        ConvBN& c1 = convs[IB];
        ConvBN& c2 = convs[IB + 1];
        run_conv_bn(this, c1, cur, InXform{}, s, buf(bottY1), train_mode);
        run_conv_bn(this, c2, View{buf(bottY1), c1.cout}, bn_xf(c1), s, buf(bottY2), train_mode);
    }
    const float* prevY = buf(bottY2);
    ConvBN* prevBN = &convs[IB + 1];
    for (int l = D; l >= 1; --l) {
        const int k = D - l;
        UpConv& u = ups[k];
        Shape sin{n, h >> l, w >> l};
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvArgs a;
        a.x = View{prevY, u.cin};
        a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = sin.H; a.Win = sin.W;
        a.Cin = u.cin; a.Cout = u.cout;
        a.w = params + u.w_off;
        a.w3 = use_w3() ? u.w3 : nullptr;
        ws_set(a);
        a.bias = params + u.b_off;
        a.y = MutView{buf(concat[l]), 2 * u.cout};
        a.Hout = s.H; a.Wout = s.W;
        a.osy = 2; a.osx = 2;
        a.R = 1; a.S = 1; a.pad = 0;
        a.zgroups = 4;
        a.xf = bn_xf(*prevBN);
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
        ConvBN& c1 = convs[IB + 2 + 2 * k];
        ConvBN& c2 = convs[IB + 2 + 2 * k + 1];
        if (skip_pending & (1u << l)) {           // the skip half of concat[l] was written on the side stream
            RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, skip_done[l], 0));
            skip_pending &= ~(1u << l);
        }
        run_conv_bn(this, c1, View{buf(concat[l]), 2 * u.cout}, InXform{}, s, buf(decY1[l]), train_mode);
        run_conv_bn(this, c2, View{buf(decY1[l]), c1.cout}, bn_xf(c1), s, buf(decY2[l]), train_mode);
        prevY = buf(decY2[l]);
        prevBN = &c2;
    }
    const int64_t M1 = (int64_t)n * h * w;
    launch_head_fwd(ctx, prevY, M1, feat, prevBN->scale(), prevBN->shift(), params + head_w_off,
                    params + head_b_off, out_ch, buf(logits), act_slope);
    if (head_sigmoid) launch_sigmoid_fwd(ctx, buf(logits), M1 * out_ch, buf(probs));
}

void rfi_model::loss_forward(const uint8_t* labels_dev, int n, int h, int w) {
    RFI_REQUIRE(out_ch == 1, "loss: the reference's BCE+dice step is defined for out_channels == 1");
    const int64_t cnt = (int64_t)n * h * w * out_scale * out_scale;
    // UNetOverfit: BCE-with-logits + dice are applied to the model OUTPUT, i.e. to sigmoid(logits)
    if (loss_kind == 1) {
        launch_focal_reduce(ctx, buf(head_sigmoid ? probs : logits), labels_dev, cnt, focal_alpha, focal_gamma,
                            reinterpret_cast<double*>(buf(ws_red)), d_scalars);
        return;
    }
    launch_loss_reduce(ctx, buf(head_sigmoid ? probs : logits), labels_dev, cnt,
                       reinterpret_cast<double*>(buf(ws_red)), d_sums, d_scalars);
}

// ------------------------------------------------------------------------------------ backward
// Weight-gradient GEMMs have no consumer before the optimiser step, so they run on a SIDE stream
// while the main stream carries the dependent chain (BN backward -> dgrad -> next layer's BN
// backward ...).  The memory-bound BN / pooling / slab-reduce kernels then share the chip with an
// MFMA-bound kernel instead of running alone, and the one-wave-per-SIMD wgrad kernel gets co-resident
// waves.  Ordering: side waits on an event recorded after the producer of dY; the main stream may
// run at most 2 side launches ahead (gA/gB of a decoder level are rewritten by the encoder phase
// no sooner than 3 layers later); all side work is joined before clip+Adam / the all-reduce.
void rfi_model::side_begin() {
    if (!ctx->overlap) return;
    RFI_CHECK_HIP(hipEventRecord(ctx->fork_ev, ctx->main_stream));
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->fork_ev, 0));
    ctx->stream = ctx->side_stream;
}
hipEvent_t rfi_model::next_fork_event() {
    static const bool off = getenv("RFI_NO_STOP_EVENTS") != nullptr;        // A/B runs: event-record packets as before
    if (!ctx->overlap || off) return nullptr;
    if (ctx->fork_ring_used == ctx->fork_ring.size()) {
        hipEvent_t e;
        RFI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->fork_ring.push_back(e);
    }
    return ctx->fork_ring[ctx->fork_ring_used++];
}
void rfi_model::side_begin_after(hipEvent_t producer_done) {
    if (!producer_done) return side_begin();
    if (!ctx->overlap) return;
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->side_stream, producer_done, 0));
    ctx->stream = ctx->side_stream;
}
void rfi_model::side_end() {
    if (!ctx->overlap) return;
    const int ring = (int)ctx->side_done.size();
    RFI_CHECK_HIP(hipEventRecord(ctx->side_done[side_seq % ring], ctx->side_stream));
    ctx->stream = ctx->main_stream;
    if (side_bound > 0 && side_seq >= side_bound)
        RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->side_done[(side_seq - side_bound) % ring], 0));
    ++side_seq;
}
void rfi_model::side_join_lazy() {
    static const bool off = getenv("RFI_NO_LAZY_JOIN") != nullptr;          // A/B runs: join at the end of the pass
    if (off || (exchange_in_backward && ctx->exchange_active())) return side_join();      // (a bucket may leave right behind this pass)
    if (!ctx->overlap || side_seq == 0) return;
    if (!lazy_ev) RFI_CHECK_HIP(hipEventCreateWithFlags(&lazy_ev, hipEventDisableTiming));
    RFI_CHECK_HIP(hipEventRecord(lazy_ev, ctx->side_stream));
    lazy_pending = true;
    side_seq = 0;
}
void rfi_model::join_pending_side() {
    if (!lazy_pending) return;
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, lazy_ev, 0));
    lazy_pending = false;
}
void rfi_model::side_join() {
    if (!ctx->overlap || side_seq == 0) return;
    const int ring = (int)ctx->side_done.size();
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->side_done[(side_seq - 1) % ring], 0));
    side_seq = 0;
    ctx->fork_ring_used = 0;                      // (every wait on these events has been passed)
}

namespace rfi { void comm_bucket_allreduce(rfi_ctx* ctx, float* dptr, int64_t count); }

static hipEvent_t bucket_event(rfi_ctx* ctx) {
    if (ctx->bucket_ev_used == ctx->bucket_ev.size()) {
        hipEvent_t e;
        RFI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->bucket_ev.push_back(e);
    }
    return ctx->bucket_ev[ctx->bucket_ev_used++];
}
// Buckets arrive in the order the backward pass completes them (decreasing offsets).  Small ones are merged with
// their successors until RFI_BUCKET_MIN_FLOATS (default 2^20 = 4 MB) have accumulated -- a sub-megabyte all-reduce is
// bound by latency, not by the links -- and the last call (lo == 0) flushes what is left.  RFI_NO_BUCKETS=1: nothing
// leaves during the pass; exchange_join sends the whole buffer in one all-reduce (the fallback if the overlapped
// exchange misbehaves on a new communicator).
void rfi_model::bucket_ready(size_t lo, size_t hi) {
    if (!exchange_in_backward || !ctx->exchange_active() || hi <= lo) return;
    static const bool no_buckets = getenv("RFI_NO_BUCKETS") != nullptr;
    const char* mf = getenv("RFI_BUCKET_MIN_FLOATS");              // (read per call: tests switch it inside one process)
    const size_t min_floats = mf ? (size_t)atoll(mf) : (size_t)1 << 20;
    if (pend_hi > pend_lo) {
        RFI_REQUIRE(hi == pend_lo || lo == pend_hi, "bucket_ready: buckets must be adjacent");
        pend_lo = std::min(pend_lo, lo);
        pend_hi = std::max(pend_hi, hi);
    } else {
        pend_lo = lo;
        pend_hi = hi;
    }
    if (no_buckets) return;                       // (exchange_join flushes)
    if (pend_hi - pend_lo < min_floats && pend_lo != 0) return;
    flush_bucket();
}
void rfi_model::flush_bucket() {
    if (pend_hi <= pend_lo) return;
    const size_t lo = pend_lo, hi = pend_hi;
    pend_lo = pend_hi = 0;
    hipEvent_t em = bucket_event(ctx);
    RFI_CHECK_HIP(hipEventRecord(em, ctx->main_stream));
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->comm_stream, em, 0));
    if (ctx->overlap) {                           // weight gradients and their slab reductions run on the side stream
        hipEvent_t es = bucket_event(ctx);
        RFI_CHECK_HIP(hipEventRecord(es, ctx->side_stream));
        RFI_CHECK_HIP(hipStreamWaitEvent(ctx->comm_stream, es, 0));
    }
    comm_bucket_allreduce(ctx, grads + lo, (int64_t)(hi - lo));
}
void rfi_model::exchange_join() {
    if (!ctx->exchange_active()) return;
    flush_bucket();                               // (RFI_NO_BUCKETS, or a model whose last bucket does not start at 0)
    hipEvent_t e = bucket_event(ctx);
    RFI_CHECK_HIP(hipEventRecord(e, ctx->comm_stream));
    RFI_CHECK_HIP(hipStreamWaitEvent(ctx->main_stream, e, 0));
    ctx->bucket_ev_used = 0;                      // the pool is reused by the next step
}

namespace {

// launches inside a SideScope go to the side stream; if one throws, the context's stream is put back
struct SideScope {
    rfi_model* m;
    bool ended = false;
    explicit SideScope(rfi_model* model, hipEvent_t after = nullptr) : m(model) { m->side_begin_after(after); }
    void end() { m->side_end(); ended = true; }
    ~SideScope() { if (!ended) m->ctx->stream = m->ctx->main_stream; }
};

// given dA (grad w.r.t. the ACTIVATED output of conv c, overwritten with dY), produce dW/db/dgamma/
// dbeta into the grad buffer and, if dx != null, the gradient w.r.t. the conv's (activated) input.
// `have_records` > 0: the BatchNorm-backward sums of this layer already sit in the workspace (the kernel that
// produced dA folded them into its epilogue).  `next` / `next_Y`: the Conv+BN layer whose activated output dx is
// the gradient of (null: none); returns the number of records the dgrad left for it (0: none).
int backward_conv_bn(rfi_model* m, ConvBN& c, float* dA, const float* Y, View in, InXform in_xf,
                     Shape s, float* dx, int have_records, ConvBN* next, const float* next_Y,
                     const float* head_dl = nullptr, const float* head_w = nullptr) {
    rfi_ctx* ctx = m->ctx;
    const int64_t M = (int64_t)s.N * s.H * s.W;
    float* ws = m->buf(m->ws_red);
    if (have_records > 0)
        launch_bn_bwd_finalize_records(ctx, ws, have_records, M, c.cout, c.c1(), c.c2(), m->grads + c.g_off,
                                       m->grads + c.be_off);
    else
        launch_bn_bwd_reduce(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(), ws, c.c1(),
                             c.c2(), m->grads + c.g_off, m->grads + c.be_off, m->act_slope);
    // RFI_WGRAD_LATE=1 (default): the weight gradient is enqueued BEHIND the input-gradient conv of the same layer (the side
    // stream waits for it): two matrix-core kernels sharing the chip finish no sooner than one after the other, but a
    // weight gradient that runs next to the following layer's BatchNorm-backward passes (memory-bound) hides them
    static const bool late = !(getenv("RFI_WGRAD_LATE") && atoi(getenv("RFI_WGRAD_LATE")) == 0);
    // a kernel that carries a completion signal leaves a ~5 us bubble behind it on its stream: only the kernel the side stream
    // actually waits for gets one (RFI_ALL_STOP_EVENTS=1: every bn_bwd_apply too, as in round 3)
    static const bool all_stop = getenv("RFI_ALL_STOP_EVENTS") != nullptr;
    const hipEvent_t dy_done = (all_stop || !late || !dx) ? m->next_fork_event() : nullptr;      // completes with the kernel that writes dY (over dA)
    // (dbias_deferred: the partial sums of the conv-bias gradient stay in the layer's own region; backward() finishes every
    // layer's in one launch at the end of the pass)
    launch_bn_bwd_apply(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(),
                        m->params + c.g_off, c.c1(), c.c2(), m->dbias_deferred ? m->dbias_pool + c.dbias_rec_off : ws,
                        m->grads + c.b_off, m->act_slope, nullptr, 0, 0, dy_done, !m->dbias_deferred, head_dl, head_w);
    WgradArgs wa;
    wa.xop = in;
    wa.yop = View{dA, c.cout};
    wa.xf_x = in_xf;
    wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = s.H; wa.Wx = s.W;
    wa.Cx = c.cin_p; wa.Cy = c.cout;
    wa.R = 3; wa.S = 1; wa.pad = 1;
    wa.dw = m->grads + c.w_off;
    wa.tap_stride = (int64_t)c.cin_p * c.cout;
    wa.sy = c.cin_p; wa.sx = 1;
    wa.algo_flops = 2.0 * s.N * s.H * s.W * 9.0 * c.cin * c.cout;
    wa.slab = m->buf(m->ws_slab);
    wa.slab_floats = m->bufs[m->ws_slab].n;
    wa.bf16 = m->compute_bf16;
    wa.bf16x3 = m->compute_x3;
    const int ci = (int)(&c - m->convs.data());
    if (!late || !dx) m->wgrad_on_side(ci, wa, dy_done);
    int records = 0;
    if (dx) {
        ConvArgs a;
        a.x = View{dA, c.cout};
        a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
        a.Cin = c.cout; a.Cout = c.cin;     // dx exists only for layers whose cin == cin_p
        a.w = c.wd;
        a.w3 = m->use_w3() ? c.wd3 : nullptr;
        m->ws_set(a);
        a.bias = nullptr;
        a.y = MutView{dx, c.cin};
        a.Hout = s.H; a.Wout = s.W;
        a.R = 3; a.S = 1; a.pad = 1;
        a.bf16 = m->compute_bf16;
        a.bf16x3 = m->compute_x3;
        if (next && m->fuse_bn_bwd) {       // dx is next's dA: fold its BatchNorm-backward sums into this epilogue
            a.stats = reinterpret_cast<double*>(ws);
            a.stats_max_records = (int)(bn_stats_ws_floats(next->cout) / ((size_t)next->cout * 4));
            a.bwd_y = next_Y;
            a.bwd_scale = next->scale(); a.bwd_shift = next->shift();
            a.bwd_mean = next->mean(); a.bwd_invstd = next->invstd();
            a.bwd_slope = m->act_slope;
        }
        if (late) a.done = m->next_fork_event();      // the weight gradient starts when this kernel completes
        launch_conv(ctx, a);
        records = a.stats_records;
        if (late) m->wgrad_on_side(ci, wa, a.done_used ? a.done : nullptr, !a.done_used);
    }
    return records;
}

}  // namespace

// A layer's weight gradient has no consumer before the optimiser: it is a FILLER for the stretches in which the main stream
// runs memory-bound BatchNorm-backward kernels.  By default it is enqueued on the side stream when its layer is done; the
// defer map (RFI_WGRAD_DEFER="9>2,4>1": the weight gradient of convs[9] goes behind that of convs[2], ...; ">-1": the end of
// the pass) moves weight gradients of deep layers -- whose own BatchNorm chains are short, so that they only queue behind
// matrix-bound kernels -- to the shallow levels, whose BatchNorm chains outlast their own weight gradients.  Everything a
// weight-gradient kernel reads stays untouched until side_join, so the order is free.  Not with a gradient exchange: the
// buckets leave in layer order.
void rfi_model::wgrad_on_side(int ci, const rfi::WgradArgs& wa, hipEvent_t after, bool after_everything) {
    auto issue = [&](const rfi::WgradArgs& w, hipEvent_t ev, bool all) {
        SideScope side(this, all ? nullptr : ev);      // (no event: the side stream waits for everything enqueued on main so far)
        launch_wgrad(ctx, w);
        side.end();
    };
    const int to = (ci >= 0 && ci < (int)defer_to.size() && !ctx->exchange_active() && ctx->overlap) ? defer_to[ci] : -2;
    if (to != -2) deferred.push_back(DeferredWgrad{wa, after_everything ? nullptr : after, to});
    else issue(wa, after, after_everything);
    for (size_t i = 0; i < deferred.size();) {    // whatever was parked behind this layer's weight gradient
        if (deferred[i].to == ci && to == -2) {
            issue(deferred[i].a, deferred[i].after, false);
            deferred.erase(deferred.begin() + i);
        } else ++i;
    }
}
void rfi_model::flush_deferred_wgrads() {
    for (auto& d : deferred) {
        SideScope side(this, d.after);
        launch_wgrad(ctx, d.a);
        side.end();
    }
    deferred.clear();
}

void rfi_model::backward(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w) {
    join_pending_side();
    if (arch == 6) {
        backward_mlp(x_dev, n);
        bucket_ready(0, n_flat);
        return;
    }
    if (arch == 5) {
        backward_backbone(x_dev, n, h, w);
        bucket_ready(0, n_flat);
        return;
    }
    if (arch == 3 || arch == 4) {
        backward_mask(x_dev, labels_dev, n, h, w);
        bucket_ready(0, n_flat);
        return;
    }
    if (arch == 1) {
        backward_cnn3(x_dev, labels_dev, n, h, w);
        bucket_ready(0, n_flat);                  // three layers: one bucket
        return;
    }
    const int D = depth, IB = i_bott;
    const int64_t M1 = (int64_t)n * h * w;
    refresh_dgrad_weights();
    wait_wd();                        // (the input-gradient-direction filter copies were rebuilt on the side stream)
    if (planesP) return backward_planes(x_dev, labels_dev, n, h, w);
    static const int bound_env = getenv("RFI_SIDE_BOUND") ? atoi(getenv("RFI_SIDE_BOUND")) : 0;
    side_bound = arch == 0 ? bound_env : 2;       // (the ResNet-style encoder double-buffers by block parity: bound 2)
    deferred.clear();
    if (defer_to.empty()) {                       // parsed once per model
        defer_to.assign(convs.size(), -2);
        const char* e = getenv("RFI_WGRAD_DEFER");
        if (arch == 0 && e) {
            int a = 0, b = 0, nread = 0;
            while (*e && sscanf(e, "%d>%d%n", &a, &b, &nread) == 2) {
                if (a >= 0 && a < (int)convs.size() && b >= -1 && b < (int)convs.size()) defer_to[a] = b;
                e += nread;
                if (*e == ',') ++e;
            }
        }
    }
    static const bool no_defer = getenv("RFI_NO_DEFER_DBIAS") != nullptr;
    dbias_deferred = arch == 0 && dbias_pool && !no_defer && !ctx->exchange_active();
    // loss -> dlogits -> head
    if (loss_kind == 1)
        launch_focal_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, focal_alpha, focal_gamma, buf(dlogits));
    else
        launch_loss_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, d_sums, buf(dlogits));
    if (head_sigmoid) launch_sigmoid_bwd(ctx, buf(probs), M1 * out_ch, buf(dlogits));
    int head_records = 0;
    // a one-channel head sends d[pixel] * w[channel] down: where its BatchNorm-backward sums come out of launch_head_bwd's own
    // pass the gradient tensor is never written -- bn_bwd_apply recomputes it from the logit gradients (268 MB of HBM traffic
    // less at batch 64 x 128^2 x 32, in the one stretch of the step where no matrix-core kernel can run)
    static const bool no_head_fuse = getenv("RFI_NO_HEAD_FUSE") != nullptr;
    bool head_skip = out_ch == 1 && !no_head_fuse;
    {
        ConvBN& last = convs[IB + 2 + 2 * (D - 1) + 1];
        // (head partials behind the region where the next layer expects its BatchNorm-backward records)
        const bool hdefer = dbias_deferred && head_fin_deferred;
        head_records = launch_head_bwd(ctx, buf(decY2[1]), M1, feat, last.scale(), last.shift(), params + head_w_off,
                                       out_ch, buf(dlogits), buf(gA[1]),
                                       hdefer ? dbias_pool + head_rec_off : buf(ws_red) + bn_bwd_ws_floats(M1, feat),
                                       grads + head_w_off, grads + head_b_off, act_slope, last.mean(), last.invstd(),
                                       buf(ws_red), nullptr, &head_skip, !hdefer);
    }
    // decoders, shallow to deep
    int pending_records = head_records;   // BatchNorm-backward records a producing kernel left for the next layer
    for (int l = 1; l <= D; ++l) {
        const int k = D - l;
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        Shape sin{n, h >> l, w >> l};
        ConvBN& c1 = convs[IB + 2 + 2 * k];
        ConvBN& c2 = convs[IB + 2 + 2 * k + 1];
        UpConv& u = ups[k];
        // conv2: input = act(decY1) ; conv1: input = concat (materialised)
        const bool from_head = l == 1 && head_skip;
        int rec = backward_conv_bn(this, c2, buf(gA[l]), buf(decY2[l]), View{buf(decY1[l]), c1.cout}, bn_xf(c1), s,
                                   buf(gB[l]), pending_records, &c1, buf(decY1[l]), from_head ? buf(dlogits) : nullptr,
                                   from_head ? params + head_w_off : nullptr);
        backward_conv_bn(this, c1, buf(gB[l]), buf(decY1[l]), View{buf(concat[l]), 2 * u.cout}, InXform{}, s,
                         buf(dconcat[l]), rec, nullptr, nullptr);
        pending_records = 0;
        // up-conv: dUp = dconcat[..., 0:C]
        const float* prevY = (l == D) ? buf(bottY2) : buf(decY2[l + 1]);
        ConvBN& prevBN = (l == D) ? convs[IB + 1] : convs[IB + 2 + 2 * (k - 1) + 1];
        View dUp{buf(dconcat[l]), 2 * u.cout};
        {
            const bool defer = dbias_deferred && u.cout % 4 == 0;
            launch_channel_sum(ctx, dUp, (int64_t)s.N * s.H * s.W, u.cout, defer ? dbias_pool + u.dbias_rec_off : buf(ws_red),
                               grads + u.b_off, !defer);
        }
        WgradArgs wa;
        wa.xop = dUp;
        wa.yop = View{prevY, u.cin};
        wa.xf_y = bn_xf(prevBN);
        wa.N = sin.N; wa.H = sin.H; wa.W = sin.W; wa.Hx = s.H; wa.Wx = s.W;
        wa.Cx = u.cout; wa.Cy = u.cin;
        wa.R = 2; wa.S = 2; wa.pad = 0;
        wa.dw = grads + u.w_off;
        wa.tap_stride = (int64_t)u.cin * u.cout;
        wa.sy = 1; wa.sx = u.cin;          // -> [tap][cout][cin]
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        {
            SideScope side(this);
            launch_wgrad(ctx, wa);
            side.end();
        }
        ConvArgs a;
        a.x = dUp;
        a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = s.H; a.Win = s.W;
        a.Cin = u.cout; a.Cout = u.cin;
        a.w = u.wd;
        a.w3 = use_w3() ? u.wd3 : nullptr;
        ws_set(a);
        a.bias = nullptr;
        float* dprev = (l == D) ? buf(gBottA) : buf(gA[l + 1]);
        a.y = MutView{dprev, u.cin};
        a.Hout = sin.H; a.Wout = sin.W;
        a.R = 2; a.S = 2; a.pad = 0;
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        if (fuse_bn_bwd) {                  // dprev is prevBN's dA: its BatchNorm-backward sums come out of this epilogue
            a.stats = reinterpret_cast<double*>(buf(ws_red));
            a.stats_max_records = (int)(bn_stats_ws_floats(prevBN.cout) / ((size_t)prevBN.cout * 4));
            a.bwd_y = prevY;
            a.bwd_scale = prevBN.scale(); a.bwd_shift = prevBN.shift();
            a.bwd_mean = prevBN.mean(); a.bwd_invstd = prevBN.invstd();
            a.bwd_slope = act_slope;
        }
        launch_conv(ctx, a);
        pending_records = a.stats_records;
        // gradients of decoder level l (and, for l = 1, of the head) are complete: exchange them now
        bucket_ready(u.w_off, l == 1 ? n_flat : ups[k + 1].w_off);
    }
    // bottleneck
    {
        Shape s{n, h >> D, w >> D};
        ConvBN& c1 = convs[IB];
        ConvBN& c2 = convs[IB + 1];
        int rec = backward_conv_bn(this, c2, buf(gBottA), buf(bottY2), View{buf(bottY1), c1.cout}, bn_xf(c1), s,
                                   buf(gBottB), pending_records, &c1, buf(bottY1));
        backward_conv_bn(this, c1, buf(gBottB), buf(bottY1), View{buf(pool[D]), c1.cin}, InXform{}, s,
                         buf(dpool[D]), rec, nullptr, nullptr);
        bucket_ready(c1.w_off, ups[0].w_off);
    }
    if (arch == 2) {                  // ResNet-style encoder (model_resnet.cpp)
        backward_resnet_encoder(x_dev, n, h, w);
        flush_deferred_wgrads();
        side_join();
        return;
    }
    // encoders, deep to shallow.  They write their own gradient tensors (not the decoder's of the same level), so nothing
    // the weight-gradient kernels read is rewritten before side_join(): no run-ahead bound (RFI_SIDE_BOUND restores one)
    for (int l = D; l >= 1; --l) {
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        const std::vector<int>& gA = this->gAe;
        const std::vector<int>& gB = this->gBe;
        // max-pool routing + skip gradient, with the BatchNorm-backward sums of c2 from the same pass where the
        // shape allows (else the separate reduction inside backward_conv_bn)
        int have = launch_pool_bwd_merge_sums(ctx, buf(encY2[l]), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), c2.mean(),
                                              c2.invstd(), View{buf(dconcat[l]) + c2.cout, 2 * c2.cout}, buf(dpool[l]),
                                              buf(gA[l]), act_slope, buf(ws_red));
        if (!have)
            launch_pool_bwd_merge(ctx, buf(encY2[l]), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(),
                                  View{buf(dconcat[l]) + c2.cout, 2 * c2.cout}, buf(dpool[l]), buf(gA[l]), act_slope);
        int rec = backward_conv_bn(this, c2, buf(gA[l]), buf(encY2[l]), View{buf(encY1[l]), c1.cout}, bn_xf(c1), s,
                                   buf(gB[l]), have, &c1, buf(encY1[l]));
        View in = (l == 1) ? (c1.cin_p == in_ch ? View{x_dev, in_ch} : View{buf(x_pad), c1.cin_p})
                           : View{buf(pool[l - 1]), c1.cin};
        backward_conv_bn(this, c1, buf(gB[l]), buf(encY1[l]), in, InXform{}, s,
                         (l == 1) ? nullptr : buf(dpool[l - 1]), rec, nullptr, nullptr);
        bucket_ready(c1.w_off, convs[l == D ? IB : 2 * l].w_off);       // convs[2 l] = first conv of the next level / the bottleneck
    }
    flush_deferred_wgrads();
    if (dbias_deferred) launch_finish_channel_sums_batched(ctx, static_cast<const FinishSumDesc*>(dbias_descs), dbias_n, dbias_max_c);
    side_join();
    side_bound = 2;
}

// ------------------------------------------------------------------------------------ optimiser
void rfi_model::apply(const rfi_hyper& hp, float grad_scale) {
    join_pending_side();
    double* ws = reinterpret_cast<double*>(buf(ws_red));
    launch_sumsq(ctx, grads, (int64_t)n_flat, ws, d_sums + 4);
    adam_step += 1;
    AdamArgs a;
    a.p = params; a.g = grads; a.m = adam_m; a.v = adam_v; a.n = (int64_t)n_flat;
    a.beta2 = (float)hp.beta2; a.eps = (float)hp.eps; a.wd = (float)hp.weight_decay;
    a.max_norm = (float)hp.max_grad_norm; a.grad_scale = grad_scale;
    a.one_minus_beta1 = (float)(1.0 - hp.beta1);
    a.one_minus_beta2 = (float)(1.0 - hp.beta2);
    a.neg_step = (float)(-(hp.lr / (1.0 - std::pow(hp.beta1, (double)adam_step))));
    a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(hp.beta2, (double)adam_step));
    a.sumsq = d_sums + 4;
    a.norm_out = d_scalars + 1;
    launch_adam(ctx, a);
    wd_dirty = true;
    x3_fresh = false;
}
