// The U-Net step (rfi_toolbox/models/unet.py:41-77, scripts/train_model.py:139-151) on the PLANE kernels
// (planes.hpp): the data flow of the bfloat16 compute mode (P = 1: activations live in HBM as bf16) and of the
// float32-by-3xbf16 arithmetic on pre-split pieces (P = 3, `float32_planes`).
//
// What is stored per DoubleConv level (BatchNorm, loss, optimiser and every per-channel statistic stay float32):
//   Y1, Y2      raw conv outputs, float32 (BatchNorm statistics come out of the conv epilogue; the backward pass
//               needs the pre-BatchNorm values)
//   A1          act(BN(Y1)) as planes: written once by act_split, read by conv2 AND by conv2's weight gradient
//   skip, pool  act(BN(Y2)) and its 2x2 max-pool as planes, written by ONE kernel (bn_relu_pool_planes): read by
//               the decoder's first conv (second K-segment; torch.cat never happens), the next level's first conv
//               and both weight gradients
//   up          ConvTranspose output (float32, from the round-1 kernel) -> planes (first K-segment of the decoder)
//   dY          BatchNorm backward writes the gradient w.r.t. the raw conv output as planes: read by the
//               input-gradient conv and by the weight gradient
// The contraction kernels therefore never convert, split or transform an operand: they copy 16-byte pieces
// HBM -> LDS by LDS-DMA and feed v_mfma_f32_32x32x16_bf16.
#include <algorithm>

#include "model.hpp"

using namespace rfi;

namespace rfi {

void PlaneBuf::ensure(rfi_ctx* c, int64_t pixels, int C, int P) {
    const size_t need = plane_elems(pixels, C, P);
    nchunks = plane_chunks(C);
    pstride = (int64_t)nchunks * P * 16;
    if (need <= elems && p) return;
    if (p) c->release(p);
    ctx = c;
    p = static_cast<bf16_t*>(c->alloc(need * 2 + 64));
    // zero once: channel padding is never written by the BatchNorm-backward producer, and the 64-byte tail is
    // where the LDS-DMA of out-of-image halo pixels points
    RFI_CHECK_HIP(hipMemsetAsync(p, 0, need * 2 + 64, c->stream));
    elems = need;
}
void PlaneBuf::free() {
    if (p && ctx) ctx->release(p);
    p = nullptr;
    elems = 0;
}

}  // namespace rfi

void rfi_model::prepare_planes(int n, int h, int w) {
    const int P = planesP, D = depth, IB = i_bott;
    const bool rs = arch == 2;                     // ResNet-style encoder: its own tensors instead of the U-Net encoder's
    if (pl.empty()) {
        auto mk = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
        mk(pA1e); mk(pSkip); mk(pPool); mk(pUp); mk(pA1d); mk(pdYa); mk(pdYb); mk(pdYaE); mk(pdYbE);
        auto one = [&]() { pl.emplace_back(); return (int)pl.size() - 1; };
        pXin = one(); pA1b = one(); pdYbottA = one(); pdYbottB = one();
        upf.assign(D + 1, -1);
        for (int l = 1; l <= D; ++l) upf[l] = new_buf();
    }
    const int64_t M1 = (int64_t)n * h * w;
    pl[pXin].ensure(ctx, M1, in_ch, P);
    for (int l = 1; l <= D; ++l) {
        const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
        const int C = feat << (l - 1);
        for (int i : {pSkip[l], pUp[l], pA1d[l], pdYa[l], pdYb[l]}) pl[i].ensure(ctx, M, C, P);
        if (!rs) for (int i : {pA1e[l], pdYaE[l], pdYbE[l]}) pl[i].ensure(ctx, M, C, P);
        if (!rs || l == D) pl[pPool[l]].ensure(ctx, M / 4, C, P);
        bufs[upf[l]].ensure(ctx, (size_t)M * C);
    }
    const int64_t Mb = (int64_t)n * (h >> D) * (w >> D);
    for (int i : {pA1b, pdYbottA, pdYbottB}) pl[i].ensure(ctx, Mb, feat << D, P);
    static const bool no_y16 = getenv("RFI_NO_Y16") != nullptr;                 // A/B runs: float32 conv outputs
    y16_flow = P == 1 && feat % 4 == 0 && (!no_y16 || rs);
    if (y16_flow) {
        if (yB1 < 0) {
            auto mk1 = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
            mk1(yE1); mk1(yE2); mk1(yD1);
            pl.emplace_back(); yB1 = (int)pl.size() - 1;
            pl.emplace_back(); yD2top = (int)pl.size() - 1;
        }
        for (int l = 1; l <= D; ++l) {
            const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
            pl[yD1[l]].ensure(ctx, M, feat << (l - 1), 1);
            if (!rs) for (int i : {yE1[l], yE2[l]}) pl[i].ensure(ctx, M, feat << (l - 1), 1);
        }
        pl[yB1].ensure(ctx, Mb, feat << D, 1);
        pl[yD2top].ensure(ctx, M1, feat, 1);
    }
    static const bool no_g16 = getenv("RFI_NO_G16") != nullptr;                 // A/B runs: float32 gradient tensors
    g16_flow = y16_flow && feat % 16 == 0 && (!no_g16 || rs);
    if (g16_flow) {
        if (g16BottB < 0) {
            auto mk1 = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
            mk1(g16A); mk1(g16B); mk1(g16pool);
            pl.emplace_back(); g16BottB = (int)pl.size() - 1;
        }
        for (int l = 1; l <= D; ++l) {
            const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
            if (!rs || l == 1) pl[g16A[l]].ensure(ctx, M, feat << (l - 1), 1);
            pl[g16B[l]].ensure(ctx, M, feat << (l - 1), 1);
            if (!rs || l == D) pl[g16pool[l]].ensure(ctx, M / 4, feat << (l - 1), 1);
        }
        pl[g16BottB].ensure(ctx, Mb, feat << D, 1);
    }
    static const bool no_ctp = getenv("RFI_NO_CONVT_PLANES") != nullptr;        // A/B runs: the round-1 transposed-conv kernels
    convt_planes = g16_flow && feat % 32 == 0 && !no_ctp;
    if (convt_planes) {
        if (yB2 < 0) {
            auto mk1 = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
            mk1(yD2); mk1(pUpIn); mk1(g16cat);
            pl.emplace_back(); yB2 = (int)pl.size() - 1;
            pl.emplace_back(); g16BottA = (int)pl.size() - 1;
        }
        for (int l = 1; l <= D; ++l) {
            const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
            const int C = feat << (l - 1);
            if (l > 1) { pl[yD2[l]].ensure(ctx, M, C, 1); pl[g16A[l]].ensure(ctx, M, C, 1); }
            pl[pUpIn[l]].ensure(ctx, M / 4, 2 * C, 1);          // the input of decoder l's transposed conv: level l + 1, 2 C channels
            pl[g16cat[l]].ensure(ctx, M, 2 * C, 1);
        }
        pl[yB2].ensure(ctx, Mb, feat << D, 1);
        pl[g16BottA].ensure(ctx, Mb, feat << D, 1);
    }
    if (rs) {
        RFI_REQUIRE(P == 1 && y16_flow && g16_flow, "UNetResNet18 on the plane kernels: bfloat16 flow with init_features % 16 == 0 only");
        auto one = [&]() { pl.emplace_back(); return (int)pl.size() - 1; };
        if (rpStemY < 0) {
            rpStemY = one(); rpA0 = one(); rpdY0 = one(); rp_dA1 = one(); rp_dX = one(); rp_dz[0] = one(); rp_dz[1] = one();
            if (rpb.empty()) rpb.assign(blocks.size(), ResPlanes());
            for (size_t bi = 0; bi < blocks.size(); ++bi) {
                ResPlanes& r = rpb[bi];
                const ResBlock& b = blocks[bi];
                r.Y1 = one(); r.Y2 = one(); r.A1 = one(); r.dY1 = one(); r.dY2 = one();
                // a stage's output is the decoder's skip tensor; the last stage's is pooled (and copied there) by one kernel
                r.A = ((bi & 1) && b.level < D) ? pSkip[b.level] : one();
                if (b.stride == 2) { r.Yd = one(); r.dYd = one(); }
            }
        }
        for (int i : {rpStemY, rpA0, rpdY0, rp_dA1, rp_dX, rp_dz[0], rp_dz[1]}) pl[i].ensure(ctx, M1, feat, 1);   // (M C halves per level)
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const ResBlock& b = blocks[bi];
            const ResPlanes& r = rpb[bi];
            const int64_t M = (int64_t)n * (h >> (b.level - 1)) * (w >> (b.level - 1));
            for (int i : {r.Y1, r.Y2, r.A1, r.A, r.dY1, r.dY2}) pl[i].ensure(ctx, M, b.cout, 1);
            if (b.stride == 2) for (int i : {r.Yd, r.dYd}) pl[i].ensure(ctx, M, b.cout, 1);
        }
    }
    // weight-gradient slabs of the plane kernel
    size_t slab_need = 0;
    for (size_t ci = 0; ci < convs.size(); ++ci) {
        const ConvBN& c = convs[ci];
        const int lvl = c.level;
        PWgradArgs a;
        const bool two = (int)ci >= IB + 2 && (((int)ci - (IB + 2)) & 1) == 0;      // decoder conv1: [up | skip]
        a.nseg = two ? 2 : 1;
        a.seg_c[0] = two ? c.cin / 2 : c.cin;
        a.seg_c[1] = two ? c.cin / 2 : 0;
        a.xop[0].nchunks = plane_chunks(a.seg_c[0]);
        a.xop[1].nchunks = two ? plane_chunks(a.seg_c[1]) : 0;
        a.yop.nchunks = plane_chunks(c.cout);
        a.Cy = c.cout;
        a.N = n; a.H = h >> (lvl - 1); a.W = w >> (lvl - 1); a.Hx = a.H * c.stride; a.Wx = a.W * c.stride;
        a.R = c.R; a.S = c.stride; a.pad = c.R == 3 ? 1 : 0;
        if (c.stride == 2 || c.R != 3) a.P = 1;
        a.tap_stride = (int64_t)c.cin_p * c.cout;
        slab_need = std::max(slab_need, pwgrad_slab_floats(a));
    }
    if (convt_planes)
        for (int k = 0; k < D; ++k) {
            const int l = D - k;
            PWgradArgs a;
            a.nseg = 1; a.seg_c[0] = ups[k].cout;
            a.xop[0].nchunks = plane_chunks(ups[k].cout);
            a.yop.nchunks = plane_chunks(ups[k].cin);
            a.Cy = ups[k].cin; a.P = 1;
            a.N = n; a.H = h >> l; a.W = w >> l; a.Hx = a.H * 2; a.Wx = a.W * 2;
            a.R = 2; a.S = 2; a.pad = 0;
            a.tap_stride = (int64_t)ups[k].cin * ups[k].cout;
            slab_need = std::max(slab_need, pwgrad_slab_floats(a));
        }
    if (bufs[ws_slab].n < slab_need + 16) bufs[ws_slab].ensure(ctx, slab_need + 16);
}

// filters of every 3x3 layer in MFMA B-operand order, both directions, rebuilt with the dgrad layouts after each
// optimiser step by ONE batched launch
void rfi_model::refresh_plane_weights(int which) {
    const int P = planesP, IB = i_bott;
    auto two_seg = [&](size_t ci) { return (int)ci >= IB + 2 && (((int)ci - (IB + 2)) & 1) == 0; };      // decoder conv1: [up | skip]
    auto has_wBd = [&](const ConvBN& c) { return c.R == 3 && c.stride == 1; };      // (the other shapes' input gradients: class tables)
    if (arch == 2 && rpb.empty()) rpb.assign(blocks.size(), ResPlanes());
    if (!wb_pool) {
        // the table: every forward-direction image first (what the forward pass reads), then the input-gradient direction
        // (dgrad layouts, parity-class tables) -- the second half can be rebuilt on the side stream under the forward pass
        size_t need = 0, cls_need = 0;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            const ConvBN& c = convs[ci];
            const bool two = two_seg(ci);
            need += wb_elems(c.R * c.R, c.cout, two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0, P) + 32;
            if (has_wBd(c)) need += wb_elems(9, c.cin_p, c.cout, 0, P) + 32;
        }
        for (auto& b : blocks)
            if (b.stride == 2) {
                need += wb_elems(4, b.cin, b.cout, b.cout, 1) + 32 + 3 * (wb_elems(4, b.cin, b.cout, 0, 1) + 32);
                cls_need += s2_class_floats(b.cout, b.cin);
            }
        if (P == 1)
            for (auto& u : ups) need += wb_elems(1, 4 * u.cout, u.cin, 0, 1) + 32 + wb_elems(4, u.cin, u.cout, 0, 1) + 32;
        wb_pool = static_cast<bf16_t*>(ctx->alloc(need * 2));
        RFI_CHECK_HIP(hipMemsetAsync(wb_pool, 0, need * 2, ctx->stream));       // the zero tails stay zero
        if (cls_need && !rs_cls_pool) rs_cls_pool = static_cast<float*>(ctx->alloc(cls_need * sizeof(float)));
        std::vector<WBDesc> hd;
        size_t o = 0, co = 0;
        wb_bytes = wb_bytes_fwd = 0;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            ConvBN& c = convs[ci];
            const bool two = two_seg(ci);
            const int taps = c.R * c.R;
            c.wBf = wb_pool + o;
            const size_t ef = wb_elems(taps, c.cout, two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0, P);
            o += ef + 32;
            hd.push_back(WBDesc{params + c.w_off, c.wBf, taps, c.cout, c.cin_p, {two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0}, P});
            wb_bytes_fwd += 2.0 * ef + 4.0 * taps * c.cin_p * c.cout;
        }
        if (P == 1)
            for (auto& u : ups) {                 // forward layout [4][cout][cin] = one 1x1 contraction with 4 cout channels
                u.wBf = wb_pool + o;
                const size_t ef = wb_elems(1, 4 * u.cout, u.cin, 0, 1);
                o += ef + 32;
                hd.push_back(WBDesc{params + u.w_off, u.wBf, 1, 4 * u.cout, u.cin, {u.cin, 0}, 1});
                wb_bytes_fwd += 2.0 * ef + 16.0 * u.cin * u.cout;
            }
        wb_n_fwd = (int)hd.size();
        wb_bytes = wb_bytes_fwd;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            ConvBN& c = convs[ci];
            if (!has_wBd(c)) continue;
            c.wBd = wb_pool + o;
            const size_t ed = wb_elems(9, c.cin_p, c.cout, 0, P);
            o += ed + 32;
            hd.push_back(WBDesc{c.wd, c.wBd, 9, c.cin_p, c.cout, {c.cout, 0}, P});
            wb_bytes += 2.0 * ed + 4.0 * 9 * c.cin_p * c.cout;
        }
        if (P == 1)
            for (auto& u : ups) {                 // dgrad layout [4][cin][cout]
                u.wBd = wb_pool + o;
                const size_t ed = wb_elems(4, u.cin, u.cout, 0, 1);
                o += ed + 32;
                hd.push_back(WBDesc{u.wd, u.wBd, 4, u.cin, u.cout, {u.cout, 0}, 1});
                wb_bytes += 2.0 * ed + 16.0 * u.cin * u.cout;
            }
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const ResBlock& b = blocks[bi];
            if (b.stride != 2) continue;
            ResPlanes& r = rpb[bi];
            r.cls = rs_cls_pool + co;
            co += s2_class_floats(b.cout, b.cin);
            for (int c = 0; c < 4; ++c) {
                r.wBcls[c] = wb_pool + o;
                const size_t e = wb_elems(4, b.cin, b.cout, c == 0 ? b.cout : 0, 1);
                o += e + 32;
                hd.push_back(WBDesc{r.cls + s2_class_offset(c, b.cout, b.cin), r.wBcls[c], 4, b.cin, c == 0 ? 2 * b.cout : b.cout,
                                    {b.cout, c == 0 ? b.cout : 0}, 1});
                wb_bytes += 2.0 * e + 16.0 * b.cin * b.cout * (c == 0 ? 2 : 1);
            }
        }
        wb_n = (int)hd.size();
        wb_descs = ctx->alloc(hd.size() * sizeof(WBDesc));
        RFI_CHECK_HIP(hipMemcpyAsync(wb_descs, hd.data(), hd.size() * sizeof(WBDesc), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // hd goes out of scope
    }
    const WBDesc* descs = static_cast<const WBDesc*>(wb_descs);
    if (which != 2) launch_weights_to_wb(ctx, descs, wb_n_fwd, wb_bytes_fwd);
    if (which != 1) {
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const ResBlock& b = blocks[bi];
            if (b.stride == 2)
                launch_w_s2_classes(ctx, params + convs[b.c1].w_off, params + convs[b.cd].w_off, b.cout, b.cin, rpb[bi].cls);
        }
        if (wb_n > wb_n_fwd) launch_weights_to_wb(ctx, descs + wb_n_fwd, wb_n - wb_n_fwd, wb_bytes - wb_bytes_fwd);
    }
}

namespace {

struct Shape { int N, H, W; };

PlaneSeg seg_of(const PlaneBuf& b) { return PlaneSeg{b.p, b.pstride, b.nchunks}; }

// where a raw conv output lives: a float32 tensor (values rounded to bf16 when the bf16 flow is on) or a bfloat16 one
struct YT {
    float* f = nullptr;
    const PlaneBuf* h = nullptr;
    YRef ref() const { return h ? YRef(h->p, h->pstride) : YRef(f); }
};

// (s: the OUTPUT grid; a stride-2 layer reads an input of twice that size)
void run_pconv_bn(rfi_model* m, ConvBN& c, const PlaneSeg* in, int nseg, Shape s, YT Yt, bool train) {
    float* Y = Yt.f;
    PConvArgs a;
    a.x[0] = in[0];
    if (nseg > 1) a.x[1] = in[1];
    a.nseg = nseg; a.P = m->planesP;
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H * c.stride; a.Win = s.W * c.stride;
    a.R = c.R; a.S = c.stride; a.pad = c.R == 3 ? 1 : 0;
    a.Cout = c.cout;
    a.wB = c.wBf;
    a.bias = m->params + c.b_off;
    if (Yt.h) { a.y16 = Yt.h->p; a.y_pstride = (int)Yt.h->pstride; }
    else { a.y = Y; a.y_pstride = c.cout; a.round_y = m->y16_flow; }
    a.Hout = s.H; a.Wout = s.W;
    a.algo_flops = 2.0 * s.N * s.H * s.W * (double)(c.R * c.R) * c.cin * c.cout;
    float* ws = m->buf(m->ws_red);
    if (train) {
        a.stats = reinterpret_cast<double*>(ws);
        a.stats_max_records = (int)(bn_stats_ws_floats(c.cout) / ((size_t)c.cout * 4));
    }
    launch_pconv(m->ctx, a);
    const int64_t M = (int64_t)s.N * s.H * s.W;
    if (train) {
        RFI_REQUIRE(a.stats_records > 0 || !Yt.h, "plane conv: no statistics records for a bfloat16 output");
        if (a.stats_records == 0) launch_bn_stats(m->ctx, Y, M, c.cout, ws);
        launch_bn_finalize(m->ctx, ws, M, c.cout, m->params + c.g_off, m->params + c.be_off, c.running_mean(),
                           c.running_var(), c.ema_repeats, c.mean(), c.invstd(), c.scale(), c.shift(), nullptr,
                           a.stats_records);
        c.nbt += c.ema_repeats;
    } else {
        launch_bn_eval_coeffs(m->ctx, c.cout, m->params + c.g_off, m->params + c.be_off, c.running_mean(),
                              c.running_var(), c.scale(), c.shift());
    }
}

// Conv3x3+BN+act twice: Y1 = conv(in) ; A1 = planes(act(BN(Y1))) ; Y2 = conv(A1)
void double_conv(rfi_model* m, ConvBN& c1, ConvBN& c2, const PlaneSeg* in, int nseg, Shape s, YT Y1, PlaneBuf& A1,
                 YT Y2, bool train) {
    run_pconv_bn(m, c1, in, nseg, s, Y1, train);
    const int64_t M = (int64_t)s.N * s.H * s.W;
    launch_act_split(m->ctx, View{Y1.f, c1.cout}, M, c1.cout, m->bn_xf(c1), m->planesP, A1.p, A1.pstride,
                     Y1.h ? Y1.h->p : nullptr, Y1.h ? Y1.h->pstride : 0);
    const PlaneSeg a1 = seg_of(A1);
    run_pconv_bn(m, c2, &a1, 1, s, Y2, train);
}

}  // namespace

// ResNet-style encoder (model_resnet.cpp states the topology) on the bf16 flow.  Every tensor is bfloat16: raw conv outputs
// (statistics from the conv epilogues), activations as the operands of the next contraction.  Stride-2 layers read the
// full-resolution planes directly (the LDS halo tile has the stride): no space-to-depth copy.  A stage's output IS the
// decoder's skip tensor.  cur <- the pooled output of the last stage.
void rfi_model::forward_resnet_planes(PlaneSeg& cur, int n, int h, int w, bool train) {
    const int D = depth;
    {
        ConvBN& c = convs[0];
        YT y; y.h = &pl[rpStemY];
        run_pconv_bn(this, c, &cur, 1, Shape{n, h, w}, y, train);
        launch_act_split(ctx, View{nullptr, c.cout}, (int64_t)n * h * w, c.cout, bn_xf(c), 1, pl[rpA0].p, pl[rpA0].pstride, pl[rpStemY].p,
                         pl[rpStemY].pstride);
        side_rebuild_wd();                        // (the input-gradient-direction filter images: side stream, under the forward pass)
    }
    const PlaneBuf* a_in = &pl[rpA0];
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        ResBlock& b = blocks[bi];
        const ResPlanes& r = rpb[bi];
        ConvBN& c1 = convs[b.c1];
        ConvBN& c2 = convs[b.c2];
        Shape s{n, h >> (b.level - 1), w >> (b.level - 1)};
        const int64_t M = (int64_t)s.N * s.H * s.W;
        const PlaneSeg in = seg_of(*a_in);
        YT y1, y2; y1.h = &pl[r.Y1]; y2.h = &pl[r.Y2];
        run_pconv_bn(this, c1, &in, 1, s, y1, train);
        if (b.stride == 2) { YT yd; yd.h = &pl[r.Yd]; run_pconv_bn(this, convs[b.cd], &in, 1, s, yd, train); }
        launch_act_split(ctx, View{nullptr, c1.cout}, M, c1.cout, bn_xf(c1), 1, pl[r.A1].p, pl[r.A1].pstride, pl[r.Y1].p, pl[r.Y1].pstride);
        const PlaneSeg a1 = seg_of(pl[r.A1]);
        run_pconv_bn(this, c2, &a1, 1, s, y2, train);
        if (b.stride == 2) {
            ConvBN& cd = convs[b.cd];
            launch_bn_add_relu16(ctx, pl[r.Y2].p, pl[r.Y2].pstride, c2.scale(), c2.shift(), pl[r.Yd].p, pl[r.Yd].pstride, cd.scale(), cd.shift(),
                                 M, b.cout, pl[r.A].p, pl[r.A].pstride);
        } else {
            launch_bn_add_relu16(ctx, pl[r.Y2].p, pl[r.Y2].pstride, c2.scale(), c2.shift(), a_in->p, a_in->pstride, nullptr, nullptr, M, b.cout,
                                 pl[r.A].p, pl[r.A].pstride);
        }
        a_in = &pl[r.A];
    }
    // MaxPool2d(2) of the last stage (identity "BatchNorm": scale 1, shift 0; the values are >= 0), + its copy as the skip
    const ResBlock& lb = blocks.back();
    launch_bn_relu_pool_planes(ctx, nullptr, n, h >> (D - 1), w >> (D - 1), lb.cout, rs_ones, rs_zeros, 0.0f, 1, pl[pSkip[D]].p,
                               pl[pSkip[D]].pstride, pl[pPool[D]].p, pl[pPool[D]].pstride, a_in->p, a_in->pstride);
    cur = seg_of(pl[pPool[D]]);
}

void rfi_model::forward_planes(const float* x_dev, int n, int h, int w, bool train_mode) {
    const int D = depth, P = planesP;
    prepare_planes(n, h, w);
    launch_act_split(ctx, View{x_dev, in_ch}, (int64_t)n * h * w, in_ch, InXform{}, P, pl[pXin].p, pl[pXin].pstride);
    PlaneSeg cur = seg_of(pl[pXin]);
    const int IB = i_bott;
    // a raw conv output: the bfloat16 tensor pl[hi] when the bf16 flow is on (hi >= 0), else the float32 tensor bufs[fi]
    auto yt = [&](int fi, int hi) { YT y; if (y16_flow && hi >= 0) y.h = &pl[hi]; else y.f = buf(fi); return y; };
    if (arch == 2) forward_resnet_planes(cur, n, h, w, train_mode);
    else for (int l = 1; l <= D; ++l) {
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        const YT y2 = yt(encY2[l], y16_flow ? yE2[l] : -1);
        double_conv(this, c1, c2, &cur, 1, s, yt(encY1[l], y16_flow ? yE1[l] : -1), pl[pA1e[l]], y2, train_mode);
        if (l == 1) side_rebuild_wd();            // (the input-gradient-direction filter images: side stream, under the forward pass)
        launch_bn_relu_pool_planes(ctx, y2.f, s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), act_slope, P,
                                   pl[pSkip[l]].p, pl[pSkip[l]].pstride, pl[pPool[l]].p, pl[pPool[l]].pstride,
                                   y2.h ? y2.h->p : nullptr, y2.h ? y2.h->pstride : 0);
        cur = seg_of(pl[pPool[l]]);
    }
    {
        Shape s{n, h >> D, w >> D};
        double_conv(this, convs[IB], convs[IB + 1], &cur, 1, s, yt(bottY1, y16_flow ? yB1 : -1), pl[pA1b],
                    yt(bottY2, convt_planes ? yB2 : -1), train_mode);     // (read by the transposed conv: float32 tensor for the round-1 kernel)
    }
    const float* prevY = buf(bottY2);
    const PlaneBuf* prevY16 = convt_planes ? &pl[yB2] : nullptr;
    ConvBN* prevBN = &convs[IB + 1];
    for (int l = D; l >= 1; --l) {
        const int k = D - l;
        UpConv& u = ups[k];
        Shape sin{n, h >> l, w >> l};
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        if (convt_planes) {
            // ConvTranspose2d(k2, s2) on the plane kernels: the input activated once into planes, then ONE 1x1 contraction whose
            // 4 cout output channels are the four taps -- each lands on its own pixel of the 2 x 2 block (PConvArgs::zblocks)
            const int64_t Min = (int64_t)sin.N * sin.H * sin.W;
            launch_act_split(ctx, View{nullptr, u.cin}, Min, u.cin, bn_xf(*prevBN), 1, pl[pUpIn[l]].p, pl[pUpIn[l]].pstride, prevY16->p,
                             prevY16->pstride);
            PConvArgs a;
            a.x[0] = seg_of(pl[pUpIn[l]]);
            a.nseg = 1; a.P = 1;
            a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = sin.H; a.Win = sin.W;
            a.R = 1; a.S = 1; a.pad = 0;
            a.Cout = 4 * u.cout;
            a.zblocks = u.cout / 32;
            a.wB = u.wBf;
            a.bias = params + u.b_off;
            a.y16 = pl[pUp[l]].p; a.y_pstride = (int)pl[pUp[l]].pstride;
            a.Hout = s.H; a.Wout = s.W; a.osy = 2; a.osx = 2;
            a.algo_flops = 2.0 * Min * 4.0 * u.cin * u.cout;
            launch_pconv(ctx, a);
            ConvBN& c1 = convs[IB + 2 + 2 * k];
            ConvBN& c2 = convs[IB + 2 + 2 * k + 1];
            const PlaneSeg in2[2] = {seg_of(pl[pUp[l]]), seg_of(pl[pSkip[l]])};
            double_conv(this, c1, c2, in2, 2, s, yt(decY1[l], yD1[l]), pl[pA1d[l]], yt(decY2[l], l == 1 ? yD2top : yD2[l]), train_mode);
            prevY16 = l == 1 ? &pl[yD2top] : &pl[yD2[l]];
            prevBN = &c2;
            continue;
        }
        ConvArgs a;                               // ConvTranspose2d(k2,s2): the round-1 kernel, float32 tensors
        a.x = View{prevY, u.cin};
        a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = sin.H; a.Win = sin.W;
        a.Cin = u.cin; a.Cout = u.cout;
        a.w = params + u.w_off;
        a.w3 = use_w3() ? u.w3 : nullptr;
        a.bias = params + u.b_off;
        a.Hout = s.H; a.Wout = s.W;
        a.osy = 2; a.osx = 2;
        a.R = 1; a.S = 1; a.pad = 0;
        a.zgroups = 4;
        a.xf = bn_xf(*prevBN);
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        // bf16 data flow with whole 16-channel chunks: the kernel writes the up-conv output as the bf16 operand of the
        // decoder's first conv directly; otherwise float32 + one conversion pass (which also zero-fills chunk padding)
        const bool direct = P == 1 && u.cout % 16 == 0 && conv_mfma_eligible(a) && pl[pUp[l]].pstride % 4 == 0;
        if (direct) {
            a.y16 = pl[pUp[l]].p;
            a.y = MutView{nullptr, (int)pl[pUp[l]].pstride};
        } else {
            a.y = MutView{buf(upf[l]), u.cout};
        }
        launch_conv(ctx, a);
        const int64_t M = (int64_t)s.N * s.H * s.W;
        if (!direct)
            launch_act_split(ctx, View{buf(upf[l]), u.cout}, M, u.cout, InXform{}, P, pl[pUp[l]].p, pl[pUp[l]].pstride);
        ConvBN& c1 = convs[IB + 2 + 2 * k];
        ConvBN& c2 = convs[IB + 2 + 2 * k + 1];
        const PlaneSeg in2[2] = {seg_of(pl[pUp[l]]), seg_of(pl[pSkip[l]])};      // cat([up, skip], dim=1) as two K-segments
        // (decoder l's second output feeds the next transposed conv -- float32 tensor -- except the last: the head)
        double_conv(this, c1, c2, in2, 2, s, yt(decY1[l], y16_flow ? yD1[l] : -1), pl[pA1d[l]],
                    yt(decY2[l], y16_flow && l == 1 ? yD2top : -1), train_mode);
        prevY = buf(decY2[l]);
        prevBN = &c2;
    }
    const int64_t M1 = (int64_t)n * h * w;
    launch_head_fwd(ctx, yt(decY2[1], y16_flow ? yD2top : -1).ref(), M1, feat, prevBN->scale(), prevBN->shift(),
                    params + head_w_off, params + head_b_off, out_ch, buf(logits), act_slope);
    if (head_sigmoid) launch_sigmoid_fwd(ctx, buf(logits), M1 * out_ch, buf(probs));
}

namespace {

struct SideScopeP {
    rfi_model* m;
    bool ended = false;
    explicit SideScopeP(rfi_model* model, hipEvent_t after = nullptr) : m(model) { m->side_begin_after(after); }
    void end() { m->side_end(); ended = true; }
    ~SideScopeP() { if (!ended) m->ctx->stream = m->ctx->main_stream; }
};

// dA: gradient w.r.t. the ACTIVATED output of conv c (float32, left untouched).  Writes dW / db / dgamma / dbeta
// and, if dx != null, the gradient w.r.t. the conv's input (float32 raw, `cin` channels per pixel).
// a gradient tensor a conv kernel writes: float32 buffer, or (bf16 data flow) a dense bfloat16 [pixel][C] tensor
struct GT {
    float* f = nullptr;
    PlaneBuf* h = nullptr;
    GT(float* p) : f(p) {}
    GT(std::nullptr_t) {}
    GT(PlaneBuf& b) : h(&b) {}
    explicit operator bool() const { return f || h; }
    YRef ref() const { return h ? YRef(h->p, h->pstride) : YRef(f); }
};

// `next` (with its raw bfloat16 output next_Y): the layer whose activated output dx is the gradient of -- its BatchNorm-
// backward sums are then folded into the input-gradient conv's epilogue; returns the number of records left for it in
// the workspace (0: none, the caller's next call runs bn_bwd_reduce).
// slope_override: the activation slope of THIS layer's output instead of the model's (1: the output has no activation of its
// own -- a residual branch, whose masked gradient dA already is dz).  Layers with a stride or other taps (ResNet-style
// encoder): s is the OUTPUT grid, `in` the input of stride times that size, dx must be null (the caller runs the input
// gradient by parity classes).
int backward_pconv_bn(rfi_model* m, ConvBN& c, YRef dA, YRef Y, const PlaneSeg* in, int nseg, Shape s,
                      GT dx, PlaneBuf& dYp, int have_records = 0, ConvBN* next = nullptr, YRef next_Y = YRef((const float*)nullptr),
                      const float* slope_override = nullptr) {
    rfi_ctx* ctx = m->ctx;
    const int64_t M = (int64_t)s.N * s.H * s.W;
    float* ws = m->buf(m->ws_red);
    const float slope = slope_override ? *slope_override : m->act_slope;
    RFI_REQUIRE((c.R == 3 && c.stride == 1) || !dx, "backward_pconv_bn: strided / 1x1 layers have no single input-gradient launch");
    if (have_records > 0)           // the kernel that produced dA left the BatchNorm-backward sums in the workspace
        launch_bn_bwd_finalize_records(ctx, ws, have_records, M, c.cout, c.c1(), c.c2(), m->grads + c.g_off, m->grads + c.be_off);
    else
        launch_bn_bwd_reduce(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(), ws, c.c1(), c.c2(),
                             m->grads + c.g_off, m->grads + c.be_off, slope);
    const hipEvent_t dy_done = m->next_fork_event();         // completes with the kernel that writes dY
    // (dbias_deferred: the partial sums of the conv-bias gradient stay in the layer's own region; the pass finishes every
    // layer's with ONE batched launch at its end)
    const bool defer = m->dbias_deferred && c.has_bias;
    launch_bn_bwd_apply(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(),
                        m->params + c.g_off, c.c1(), c.c2(), defer ? m->dbias_pool + c.dbias_rec_off : ws,
                        c.has_bias ? m->grads + c.b_off : nullptr, slope, dYp.p, dYp.pstride, m->planesP, dy_done, !defer);
    PWgradArgs wa;
    wa.xop[0] = in[0];
    if (nseg > 1) wa.xop[1] = in[1];
    wa.nseg = nseg;
    wa.seg_c[0] = nseg > 1 ? c.cin / 2 : c.cin;
    wa.seg_c[1] = nseg > 1 ? c.cin / 2 : 0;
    wa.cx_layout = c.cin_p;
    wa.yop = PlaneSeg{dYp.p, dYp.pstride, dYp.nchunks};
    wa.Cy = c.cout;
    wa.P = m->planesP;
    wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = s.H * c.stride; wa.Wx = s.W * c.stride;
    wa.R = c.R; wa.S = c.stride; wa.pad = c.R == 3 ? 1 : 0;
    wa.dw = m->grads + c.w_off;
    wa.tap_stride = (int64_t)c.cin_p * c.cout;
    wa.sy = c.cin_p; wa.sx = 1;
    wa.algo_flops = 2.0 * s.N * s.H * s.W * (double)(c.R * c.R) * c.cin * c.cout;
    wa.slab = m->buf(m->ws_slab);
    wa.slab_floats = m->bufs[m->ws_slab].n;
    {
        SideScopeP side(m, dy_done);
        launch_pwgrad(ctx, wa);
        side.end();
    }
    if (dx) {
        PConvArgs a;
        a.x[0] = PlaneSeg{dYp.p, dYp.pstride, dYp.nchunks};
        a.nseg = 1; a.P = m->planesP;
        a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
        a.Cout = c.cin;                           // dx exists only for layers whose cin == cin_p
        a.wB = c.wBd;
        if (dx.h) { a.y16 = dx.h->p; a.y_pstride = (int)dx.h->pstride; }
        else { a.y = dx.f; a.y_pstride = c.cin; }
        a.Hout = s.H; a.Wout = s.W;
        a.algo_flops = 2.0 * s.N * s.H * s.W * 9.0 * c.cin * c.cout;
        static const bool no_fuse = getenv("RFI_NO_BN_FUSE") != nullptr;     // A/B runs: separate bn_bwd_reduce
        if (next && next_Y.bf16 && !no_fuse) {
            a.stats = reinterpret_cast<double*>(ws);          // (this layer's dy sums have been finished out of it)
            a.stats_max_records = (int)(bn_stats_ws_floats(next->cout) / ((size_t)next->cout * 4));
            a.bwd_y16 = static_cast<const bf16_t*>(next_Y.p);
            a.bwd_yps = next_Y.stride(next->cout);
            a.bwd_scale = next->scale(); a.bwd_shift = next->shift();
            a.bwd_mean = next->mean(); a.bwd_invstd = next->invstd();
            a.bwd_slope = m->act_slope;                       // (`next` is a conv + BatchNorm + activation layer)
        }
        launch_pconv(ctx, a);
        return a.stats_records;
    }
    return 0;
}

}  // namespace

// Backward pass of the ResNet-style encoder on the bf16 flow.  Per BasicBlock (g = gradient w.r.t. the block's output, a sum of
// up to three tensors that is never materialised):
//   dz  = g * (a_out > 0)                                           one pass (relu_mask_sum16), read by both BatchNorm branches
//   main branch     BN2 backward (identity activation) -> dY2 planes; conv2's weight gradient (side stream) and input gradient dA1
//                   with BN1's backward sums folded into its epilogue; BN1 backward -> dY1 planes; conv1's weight gradient
//   identity block  dX = conv1's input gradient; the block's input gradient is dX + dz (the next relu_mask_sum16 adds them)
//   stride-2 block  projection: BNd backward of the same dz -> dYd planes, its weight gradient; the input gradient of conv1 AND
//                   of the projection as four 2x2 contractions, one per parity class of the input pixel, written with output
//                   stride 2 (class 0 takes the projection as a second K segment): no zero-stuffed tensor, no scatter pass
void rfi_model::backward_resnet_planes(int n, int h, int w) {
    const int D = depth;
    const float one = 1.0f;
    // gradient w.r.t. the last stage's output: skip gradient + max-pool routing of dpool (-> rp_dX, as "dX from above")
    {
        const ResBlock& lb = blocks.back();
        const PlaneBuf& A = pl[rpb.back().A];
        launch_pool_bwd_merge(ctx, YRef(A.p, A.pstride), n, h >> (D - 1), w >> (D - 1), lb.cout, rs_ones, rs_zeros,
                              skip_grad(D, lb.cout), YRef(pl[g16pool[D]].p, pl[g16pool[D]].pstride), nullptr, 0.0f, pl[rp_dX].p);
    }
    // (the scratch tensors rp_dA1 / rp_dX / rp_dz serve every level: dense [M][C] views of them)
    auto view = [&](int idx, int C) { PlaneBuf v = pl[idx]; v.pstride = C; v.nchunks = C / 16; return v; };
    const bf16_t* g0 = pl[rp_dX].p;               // the terms of g (dense bfloat16 [M][C]) ...
    const bf16_t* g1 = nullptr;
    YRef g2((const float*)nullptr);               // ... and the decoder's skip gradient (a view), at stage boundaries
    for (int bi = (int)blocks.size() - 1; bi >= 0; --bi) {
        ResBlock& b = blocks[bi];
        const ResPlanes& r = rpb[bi];
        ConvBN& c1 = convs[b.c1];
        ConvBN& c2 = convs[b.c2];
        Shape s{n, h >> (b.level - 1), w >> (b.level - 1)};
        const int64_t M = (int64_t)s.N * s.H * s.W;
        const PlaneBuf& a_in = bi == 0 ? pl[rpA0] : pl[rpb[bi - 1].A];
        PlaneBuf& dz = pl[rp_dz[bi & 1]];
        launch_relu_mask_sum16(ctx, g0, b.cout, g1, b.cout, g2, pl[r.A].p, pl[r.A].pstride, M, b.cout, dz.p, b.cout);
        const YRef dzr(dz.p, (int64_t)b.cout);
        const PlaneSeg a1 = seg_of(pl[r.A1]);
        PlaneBuf dA1v = view(rp_dA1, b.cout), dXv = view(rp_dX, b.cin);
        const int rec1 = backward_pconv_bn(this, c2, dzr, YRef(pl[r.Y2].p, pl[r.Y2].pstride), &a1, 1, s, GT(dA1v), pl[r.dY2], 0, &c1,
                                           YRef(pl[r.Y1].p, pl[r.Y1].pstride), &one);
        const PlaneSeg in = seg_of(a_in);
        const YRef dA1(pl[rp_dA1].p, (int64_t)b.cout);
        if (b.stride == 1) {
            backward_pconv_bn(this, c1, dA1, YRef(pl[r.Y1].p, pl[r.Y1].pstride), &in, 1, s, GT(dXv), pl[r.dY1], rec1);
            g0 = pl[rp_dX].p; g1 = dz.p; g2 = YRef((const float*)nullptr);
        } else {
            ConvBN& cd = convs[b.cd];
            backward_pconv_bn(this, c1, dA1, YRef(pl[r.Y1].p, pl[r.Y1].pstride), &in, 1, s, GT(nullptr), pl[r.dY1], rec1);
            backward_pconv_bn(this, cd, dzr, YRef(pl[r.Yd].p, pl[r.Yd].pstride), &in, 1, s, GT(nullptr), pl[r.dYd], 0, nullptr,
                              YRef((const float*)nullptr), &one);
            for (int c = 0; c < 4; ++c) {
                PConvArgs a;
                a.x[0] = seg_of(pl[r.dY1]);
                if (c == 0) a.x[1] = seg_of(pl[r.dYd]);
                a.nseg = c == 0 ? 2 : 1; a.P = 1;
                a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
                a.R = 2; a.S = 1; a.pad = 0;
                a.Cout = b.cin;
                a.wB = r.wBcls[c];
                a.y16 = pl[rp_dX].p; a.y_pstride = b.cin;
                a.Hout = 2 * s.H; a.Wout = 2 * s.W;
                a.osy = 2; a.osx = 2; a.ooy = c >> 1; a.oox = c & 1;
                // (true work: 9 taps of conv1 + the projection over the M output pixels; the zero taps of the 2x2 forms are not counted)
                a.algo_flops = c == 0 ? 2.0 * M * (9.0 * b.cin * b.cout + (double)b.cin * b.cout) : 0.0;
                launch_pconv(ctx, a);
            }
            g0 = pl[rp_dX].p; g1 = nullptr;
            g2 = skip_grad(b.level - 1, b.cin);   // the previous stage's output is also a skip
        }
        bucket_ready(c1.w_off, (size_t)(bi + 1 < (int)blocks.size() ? convs[blocks[bi + 1].c1].w_off : convs[i_bott].w_off));
    }
    {                                             // stem: g = dX + dz of the first block, materialised once
        ConvBN& c = convs[0];
        Shape s{n, h, w};
        const int64_t M = (int64_t)n * h * w;
        PlaneBuf& g = pl[rp_dz[1]];               // (block 0 used rp_dz[0])
        launch_relu_mask_sum16(ctx, g0, c.cout, g1, c.cout, YRef((const float*)nullptr), nullptr, 0, M, c.cout, g.p, c.cout);
        const PlaneSeg in = seg_of(pl[pXin]);
        backward_pconv_bn(this, c, YRef(g.p, (int64_t)c.cout), YRef(pl[rpStemY].p, pl[rpStemY].pstride), &in, 1, s, GT(nullptr), pl[rpdY0]);
        bucket_ready(0, convs[blocks[0].c1].w_off);
    }
}

void rfi_model::backward_planes(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w) {
    (void)x_dev;
    const int D = depth, IB = i_bott;
    // every layer owns its dY plane tensor and the other side-stream inputs (forward planes, dconcat, raw conv outputs)
    // are not rewritten before side_join(): the main stream never has to wait for a weight gradient inside the pass
    static const int bound_env = getenv("RFI_SIDE_BOUND") ? atoi(getenv("RFI_SIDE_BOUND")) : 0;
    side_bound = bound_env;
    // a raw conv output of the forward pass: bfloat16 tensor pl[hi] in the bf16 flow, else float32 bufs[fi]
    auto yr = [&](int fi, int hi) { return (y16_flow && hi >= 0) ? YRef(pl[hi].p, pl[hi].pstride) : YRef(buf(fi)); };
    // a gradient tensor an input-gradient conv writes: bfloat16 tensor pl[hi] in the bf16 flow, else float32 bufs[fi]
    auto gt = [&](int fi, int hi) { return hi >= 0 ? GT(pl[hi]) : GT(buf(fi)); };
    const int64_t M1 = (int64_t)n * h * w;
    static const bool no_defer = getenv("RFI_NO_DEFER_DBIAS") != nullptr;
    dbias_deferred = dbias_pool && !no_defer && !ctx->exchange_active();    // (bias gradients must be final before their bucket leaves)
    if (loss_kind == 1)
        launch_focal_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, focal_alpha, focal_gamma, buf(dlogits));
    else
        launch_loss_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, d_sums, buf(dlogits));
    if (head_sigmoid) launch_sigmoid_bwd(ctx, buf(probs), M1 * out_ch, buf(dlogits));
    int head_records = 0;
    {
        ConvBN& last = convs[IB + 2 + 2 * (D - 1) + 1];
        const bool hdefer = dbias_deferred && head_fin_deferred;
        head_records = launch_head_bwd(ctx, yr(decY2[1], y16_flow ? yD2top : -1), M1, feat, last.scale(), last.shift(), params + head_w_off, out_ch,
                                       buf(dlogits), buf(gA[1]), hdefer ? dbias_pool + head_rec_off : buf(ws_red) + bn_bwd_ws_floats(M1, feat),
                                       grads + head_w_off, grads + head_b_off, act_slope, last.mean(), last.invstd(), buf(ws_red),
                                       g16_flow && out_ch == 1 ? pl[g16A[1]].p : nullptr, nullptr, !hdefer);
    }
    int up_records = 0;                           // BatchNorm-backward records a transposed conv's input-gradient kernel left for the layer below
    for (int l = 1; l <= D; ++l) {                // decoders, shallow to deep
        const int k = D - l;
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        Shape sin{n, h >> l, w >> l};
        ConvBN& c1 = convs[IB + 2 + 2 * k];
        ConvBN& c2 = convs[IB + 2 + 2 * k + 1];
        UpConv& u = ups[k];
        const PlaneSeg a1 = seg_of(pl[pA1d[l]]);
        // (with the transposed convs on the plane kernels the gradient arriving from decoder l - 1's is a bfloat16 tensor, the
        // BatchNorm-backward sums of c2 came out of that kernel's epilogue, and Y2 is a bfloat16 tensor)
        const bool up16 = convt_planes && l > 1;
        const int rec1 = backward_pconv_bn(this, c2, gt(gA[l], up16 || (g16_flow && out_ch == 1 && l == 1) ? g16A[l] : -1).ref(),
                                           yr(decY2[l], y16_flow && l == 1 ? yD2top : (up16 ? yD2[l] : -1)), &a1, 1, s,
                                           gt(gB[l], g16_flow ? g16B[l] : -1), pl[pdYa[l]], l == 1 ? head_records : up_records, &c1,
                                           yr(decY1[l], y16_flow ? yD1[l] : -1));
        const PlaneSeg in2[2] = {seg_of(pl[pUp[l]]), seg_of(pl[pSkip[l]])};
        backward_pconv_bn(this, c1, gt(gB[l], g16_flow ? g16B[l] : -1).ref(), yr(decY1[l], y16_flow ? yD1[l] : -1), in2, 2, s,
                          convt_planes ? GT(pl[g16cat[l]]) : GT(buf(dconcat[l])), pl[pdYb[l]], rec1);
        ConvBN& prevBN = (l == D) ? convs[IB + 1] : convs[IB + 2 + 2 * (k - 1) + 1];
        if (convt_planes) {
            // ConvTranspose on the plane kernels: dUp = the first C channels of the bfloat16 [up | skip] gradient
            const PlaneBuf& cat = pl[g16cat[l]];
            const PlaneSeg dUp{cat.p, cat.pstride, plane_chunks(u.cout)};
            const bool udefer = dbias_deferred && u.cout % 4 == 0;
            launch_channel_sum(ctx, YRef(cat.p, cat.pstride), (int64_t)s.N * s.H * s.W, u.cout, udefer ? dbias_pool + u.dbias_rec_off : buf(ws_red),
                               grads + u.b_off, !udefer);
            PWgradArgs wa;                        // dW[tap][cout][cin] = sum_pixels act(prev)[i, j, cin] * dUp[2 i + a, 2 j + b, cout]
            wa.xop[0] = dUp; wa.nseg = 1; wa.seg_c[0] = u.cout;
            wa.yop = seg_of(pl[pUpIn[l]]); wa.Cy = u.cin; wa.P = 1;
            wa.N = sin.N; wa.H = sin.H; wa.W = sin.W; wa.Hx = s.H; wa.Wx = s.W;
            wa.R = 2; wa.S = 2; wa.pad = 0;
            wa.dw = grads + u.w_off;
            wa.tap_stride = (int64_t)u.cin * u.cout;
            wa.sy = 1; wa.sx = u.cin;
            wa.algo_flops = 2.0 * sin.N * sin.H * sin.W * 4.0 * u.cin * u.cout;
            wa.slab = buf(ws_slab);
            wa.slab_floats = bufs[ws_slab].n;
            {
                SideScopeP side(this);
                launch_pwgrad(ctx, wa);
                side.end();
            }
            PConvArgs a;                          // the input gradient: a 2x2 stride-2 contraction of dUp, bfloat16 out, with the
            a.x[0] = dUp; a.nseg = 1; a.P = 1;    // BatchNorm-backward sums of the layer below in the epilogue
            a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = s.H; a.Win = s.W;
            a.R = 2; a.S = 2; a.pad = 0;
            a.Cout = u.cin;
            a.wB = u.wBd;
            PlaneBuf& dprev = (l == D) ? pl[g16BottA] : pl[g16A[l + 1]];
            const PlaneBuf& prevY16 = (l == D) ? pl[yB2] : pl[yD2[l + 1]];
            a.y16 = dprev.p; a.y_pstride = (int)dprev.pstride;
            a.Hout = sin.H; a.Wout = sin.W;
            a.algo_flops = 2.0 * sin.N * sin.H * sin.W * 4.0 * u.cin * u.cout;
            float* ws = buf(ws_red);
            a.stats = reinterpret_cast<double*>(ws);
            a.stats_max_records = (int)(bn_stats_ws_floats(u.cin) / ((size_t)u.cin * 4));
            a.bwd_y16 = prevY16.p;
            a.bwd_yps = prevY16.pstride;
            a.bwd_scale = prevBN.scale(); a.bwd_shift = prevBN.shift();
            a.bwd_mean = prevBN.mean(); a.bwd_invstd = prevBN.invstd();
            a.bwd_slope = act_slope;
            launch_pconv(ctx, a);
            up_records = a.stats_records;
            bucket_ready(u.w_off, l == 1 ? n_flat : ups[k + 1].w_off);
            continue;
        }
        // ConvTranspose: dUp = dconcat[..., 0:C]; the round-1 kernels on float32 tensors
        const float* prevY = (l == D) ? buf(bottY2) : buf(decY2[l + 1]);
        View dUp{buf(dconcat[l]), 2 * u.cout};
        {
            const bool udefer = dbias_deferred && u.cout % 4 == 0;
            launch_channel_sum(ctx, dUp, (int64_t)s.N * s.H * s.W, u.cout, udefer ? dbias_pool + u.dbias_rec_off : buf(ws_red), grads + u.b_off,
                               !udefer);
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
        wa.sy = 1; wa.sx = u.cin;
        wa.slab = buf(ws_slab);
        wa.slab_floats = bufs[ws_slab].n;
        wa.bf16 = compute_bf16;
        wa.bf16x3 = compute_x3;
        {
            SideScopeP side(this);
            launch_wgrad(ctx, wa);
            side.end();
        }
        ConvArgs a;
        a.x = dUp;
        a.N = sin.N; a.H = sin.H; a.W = sin.W; a.Hin = s.H; a.Win = s.W;
        a.Cin = u.cout; a.Cout = u.cin;
        a.w = u.wd;
        a.w3 = use_w3() ? u.wd3 : nullptr;
        float* dprev = (l == D) ? buf(gBottA) : buf(gA[l + 1]);
        a.y = MutView{dprev, u.cin};
        a.Hout = sin.H; a.Wout = sin.W;
        a.R = 2; a.S = 2; a.pad = 0;
        a.bf16 = compute_bf16;
        a.bf16x3 = compute_x3;
        launch_conv(ctx, a);
        bucket_ready(u.w_off, l == 1 ? n_flat : ups[k + 1].w_off);      // decoder level l (+ head) is complete
    }
    {                                             // bottleneck
        Shape s{n, h >> D, w >> D};
        const PlaneSeg a1 = seg_of(pl[pA1b]), p4 = seg_of(pl[pPool[D]]);
        const int rec1 = backward_pconv_bn(this, convs[IB + 1], gt(gBottA, convt_planes ? g16BottA : -1).ref(), yr(bottY2, convt_planes ? yB2 : -1), &a1, 1,
                                           s, gt(gBottB, g16_flow ? g16BottB : -1), pl[pdYbottA], convt_planes ? up_records : 0, &convs[IB],
                                           yr(bottY1, y16_flow ? yB1 : -1));
        backward_pconv_bn(this, convs[IB], gt(gBottB, g16_flow ? g16BottB : -1).ref(), yr(bottY1, y16_flow ? yB1 : -1), &p4, 1, s,
                          gt(dpool[D], g16_flow ? g16pool[D] : -1), pl[pdYbottB], rec1);
        bucket_ready(convs[IB].w_off, ups[0].w_off);
    }
    if (arch == 2) backward_resnet_planes(n, h, w);
    else for (int l = D; l >= 1; --l) {           // encoders, deep to shallow
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        const YRef dskip = skip_grad(l, c2.cout);
        const int have = launch_pool_bwd_merge_sums(ctx, yr(encY2[l], y16_flow ? yE2[l] : -1), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), c2.mean(),
                                                    c2.invstd(), dskip, gt(dpool[l], g16_flow ? g16pool[l] : -1).ref(),
                                                    buf(gA[l]), act_slope, buf(ws_red), g16_flow ? pl[g16A[l]].p : nullptr);
        if (!have)
            launch_pool_bwd_merge(ctx, yr(encY2[l], y16_flow ? yE2[l] : -1), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(),
                                  dskip, gt(dpool[l], g16_flow ? g16pool[l] : -1).ref(), buf(gA[l]), act_slope,
                                  g16_flow ? pl[g16A[l]].p : nullptr);
        const PlaneSeg a1 = seg_of(pl[pA1e[l]]);
        const int rec1 = backward_pconv_bn(this, c2, gt(gA[l], g16_flow ? g16A[l] : -1).ref(), yr(encY2[l], y16_flow ? yE2[l] : -1), &a1, 1, s, gt(gB[l], g16_flow ? g16B[l] : -1),
                                           pl[pdYaE[l]], have, &c1, yr(encY1[l], y16_flow ? yE1[l] : -1));
        const PlaneSeg in = seg_of(l == 1 ? pl[pXin] : pl[pPool[l - 1]]);
        backward_pconv_bn(this, c1, gt(gB[l], g16_flow ? g16B[l] : -1).ref(), yr(encY1[l], y16_flow ? yE1[l] : -1), &in, 1, s,
                          (l == 1) ? GT(nullptr) : gt(dpool[l - 1], g16_flow ? g16pool[l - 1] : -1),
                          pl[pdYbE[l]], rec1);
        bucket_ready(c1.w_off, convs[2 * l].w_off);
    }
    if (dbias_deferred)
        launch_finish_channel_sums_batched(ctx, static_cast<const FinishSumDesc*>(dbias_descs), dbias_n, dbias_max_c);
    side_join();
    side_bound = 2;
}
