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
    const int P = planesP, D = depth;
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
        for (int i : {pA1e[l], pSkip[l], pUp[l], pA1d[l], pdYa[l], pdYb[l], pdYaE[l], pdYbE[l]}) pl[i].ensure(ctx, M, C, P);
        pl[pPool[l]].ensure(ctx, M / 4, C, P);
        bufs[upf[l]].ensure(ctx, (size_t)M * C);
    }
    const int64_t Mb = (int64_t)n * (h >> D) * (w >> D);
    for (int i : {pA1b, pdYbottA, pdYbottB}) pl[i].ensure(ctx, Mb, feat << D, P);
    static const bool no_y16 = getenv("RFI_NO_Y16") != nullptr;                 // A/B runs: float32 conv outputs
    y16_flow = P == 1 && feat % 4 == 0 && !no_y16;
    if (y16_flow) {
        if (yB1 < 0) {
            auto mk1 = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
            mk1(yE1); mk1(yE2); mk1(yD1);
            pl.emplace_back(); yB1 = (int)pl.size() - 1;
            pl.emplace_back(); yD2top = (int)pl.size() - 1;
        }
        for (int l = 1; l <= D; ++l) {
            const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
            for (int i : {yE1[l], yE2[l], yD1[l]}) pl[i].ensure(ctx, M, feat << (l - 1), 1);
        }
        pl[yB1].ensure(ctx, Mb, feat << D, 1);
        pl[yD2top].ensure(ctx, M1, feat, 1);
    }
    static const bool no_g16 = getenv("RFI_NO_G16") != nullptr;                 // A/B runs: float32 gradient tensors
    g16_flow = y16_flow && feat % 16 == 0 && !no_g16;
    if (g16_flow) {
        if (g16BottB < 0) {
            auto mk1 = [&](std::vector<int>& v) { v.assign(D + 1, -1); for (int l = 1; l <= D; ++l) { pl.emplace_back(); v[l] = (int)pl.size() - 1; } };
            mk1(g16A); mk1(g16B); mk1(g16pool);
            pl.emplace_back(); g16BottB = (int)pl.size() - 1;
        }
        for (int l = 1; l <= D; ++l) {
            const int64_t M = (int64_t)n * (h >> (l - 1)) * (w >> (l - 1));
            pl[g16A[l]].ensure(ctx, M, feat << (l - 1), 1);
            pl[g16B[l]].ensure(ctx, M, feat << (l - 1), 1);
            pl[g16pool[l]].ensure(ctx, M / 4, feat << (l - 1), 1);
        }
        pl[g16BottB].ensure(ctx, Mb, feat << D, 1);
    }
    // weight-gradient slabs of the plane kernel
    size_t slab_need = 0;
    for (size_t ci = 0; ci < convs.size(); ++ci) {
        int lvl;
        if ((int)ci < 2 * D) lvl = (int)ci / 2 + 1;
        else if ((int)ci < 2 * D + 2) lvl = D + 1;
        else lvl = D - ((int)ci - (2 * D + 2)) / 2;
        PWgradArgs a;
        const bool two = (int)ci >= 2 * D + 2 && (((int)ci - (2 * D + 2)) & 1) == 0;      // decoder conv1: [up | skip]
        a.nseg = two ? 2 : 1;
        a.seg_c[0] = two ? convs[ci].cin / 2 : convs[ci].cin;
        a.seg_c[1] = two ? convs[ci].cin / 2 : 0;
        a.xop[0].nchunks = plane_chunks(a.seg_c[0]);
        a.xop[1].nchunks = two ? plane_chunks(a.seg_c[1]) : 0;
        a.yop.nchunks = plane_chunks(convs[ci].cout);
        a.Cy = convs[ci].cout;
        a.N = n; a.H = h >> (lvl - 1); a.W = w >> (lvl - 1); a.Hx = a.H; a.Wx = a.W;
        a.tap_stride = (int64_t)convs[ci].cin_p * convs[ci].cout;
        slab_need = std::max(slab_need, pwgrad_slab_floats(a));
    }
    if (bufs[ws_slab].n < slab_need + 16) bufs[ws_slab].ensure(ctx, slab_need + 16);
}

// filters of every 3x3 layer in MFMA B-operand order, both directions, rebuilt with the dgrad layouts after each
// optimiser step by ONE batched launch
void rfi_model::refresh_plane_weights() {
    const int P = planesP, D = depth;
    if (!wb_pool) {
        size_t need = 0;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            const ConvBN& c = convs[ci];
            const bool two = (int)ci >= 2 * D + 2 && (((int)ci - (2 * D + 2)) & 1) == 0;
            need += wb_elems(9, c.cout, two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0, P) + 32;
            need += wb_elems(9, c.cin_p, c.cout, 0, P) + 32;
        }
        wb_pool = static_cast<bf16_t*>(ctx->alloc(need * 2));
        RFI_CHECK_HIP(hipMemsetAsync(wb_pool, 0, need * 2, ctx->stream));       // the zero tails stay zero
        std::vector<WBDesc> hd;
        size_t o = 0;
        wb_bytes = 0;
        for (size_t ci = 0; ci < convs.size(); ++ci) {
            ConvBN& c = convs[ci];
            const bool two = (int)ci >= 2 * D + 2 && (((int)ci - (2 * D + 2)) & 1) == 0;
            c.wBf = wb_pool + o;
            const size_t ef = wb_elems(9, c.cout, two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0, P);
            o += ef + 32;
            hd.push_back(WBDesc{params + c.w_off, c.wBf, 9, c.cout, c.cin_p, {two ? c.cin / 2 : c.cin_p, two ? c.cin / 2 : 0}, P});
            c.wBd = wb_pool + o;
            const size_t ed = wb_elems(9, c.cin_p, c.cout, 0, P);
            o += ed + 32;
            hd.push_back(WBDesc{c.wd, c.wBd, 9, c.cin_p, c.cout, {c.cout, 0}, P});
            wb_bytes += 2.0 * (ef + ed) + 8.0 * 9 * c.cin_p * c.cout;
        }
        wb_n = (int)hd.size();
        wb_descs = ctx->alloc(hd.size() * sizeof(WBDesc));
        RFI_CHECK_HIP(hipMemcpyAsync(wb_descs, hd.data(), hd.size() * sizeof(WBDesc), hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // hd goes out of scope
    }
    launch_weights_to_wb(ctx, static_cast<const WBDesc*>(wb_descs), wb_n, wb_bytes);
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

void run_pconv_bn(rfi_model* m, ConvBN& c, const PlaneSeg* in, int nseg, Shape s, YT Yt, bool train) {
    float* Y = Yt.f;
    PConvArgs a;
    a.x[0] = in[0];
    if (nseg > 1) a.x[1] = in[1];
    a.nseg = nseg; a.P = m->planesP;
    a.N = s.N; a.H = s.H; a.W = s.W; a.Hin = s.H; a.Win = s.W;
    a.Cout = c.cout;
    a.wB = c.wBf;
    a.bias = m->params + c.b_off;
    if (Yt.h) { a.y16 = Yt.h->p; a.y_pstride = (int)Yt.h->pstride; }
    else { a.y = Y; a.y_pstride = c.cout; a.round_y = m->y16_flow; }
    a.Hout = s.H; a.Wout = s.W;
    a.algo_flops = 2.0 * s.N * s.H * s.W * 9.0 * c.cin * c.cout;
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

void rfi_model::forward_planes(const float* x_dev, int n, int h, int w, bool train_mode) {
    const int D = depth, P = planesP;
    prepare_planes(n, h, w);
    launch_act_split(ctx, View{x_dev, in_ch}, (int64_t)n * h * w, in_ch, InXform{}, P, pl[pXin].p, pl[pXin].pstride);
    PlaneSeg cur = seg_of(pl[pXin]);
    // a raw conv output: the bfloat16 tensor pl[hi] when the bf16 flow is on (hi >= 0), else the float32 tensor bufs[fi]
    auto yt = [&](int fi, int hi) { YT y; if (y16_flow && hi >= 0) y.h = &pl[hi]; else y.f = buf(fi); return y; };
    for (int l = 1; l <= D; ++l) {
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        const YT y2 = yt(encY2[l], y16_flow ? yE2[l] : -1);
        double_conv(this, c1, c2, &cur, 1, s, yt(encY1[l], y16_flow ? yE1[l] : -1), pl[pA1e[l]], y2, train_mode);
        launch_bn_relu_pool_planes(ctx, y2.f, s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), act_slope, P,
                                   pl[pSkip[l]].p, pl[pSkip[l]].pstride, pl[pPool[l]].p, pl[pPool[l]].pstride,
                                   y2.h ? y2.h->p : nullptr, y2.h ? y2.h->pstride : 0);
        cur = seg_of(pl[pPool[l]]);
    }
    {
        Shape s{n, h >> D, w >> D};
        double_conv(this, convs[2 * D], convs[2 * D + 1], &cur, 1, s, yt(bottY1, y16_flow ? yB1 : -1), pl[pA1b],
                    yt(bottY2, -1), train_mode);            // (read by the transposed conv: float32 tensor)
    }
    const float* prevY = buf(bottY2);
    ConvBN* prevBN = &convs[2 * D + 1];
    for (int l = D; l >= 1; --l) {
        const int k = D - l;
        UpConv& u = ups[k];
        Shape sin{n, h >> l, w >> l};
        Shape s{n, h >> (l - 1), w >> (l - 1)};
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
        ConvBN& c1 = convs[2 * D + 2 + 2 * k];
        ConvBN& c2 = convs[2 * D + 2 + 2 * k + 1];
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
int backward_pconv_bn(rfi_model* m, ConvBN& c, YRef dA, YRef Y, const PlaneSeg* in, int nseg, Shape s,
                      GT dx, PlaneBuf& dYp, int have_records = 0, ConvBN* next = nullptr, YRef next_Y = YRef((const float*)nullptr)) {
    rfi_ctx* ctx = m->ctx;
    const int64_t M = (int64_t)s.N * s.H * s.W;
    float* ws = m->buf(m->ws_red);
    if (have_records > 0)           // the kernel that produced dA left the BatchNorm-backward sums in the workspace
        launch_bn_bwd_finalize_records(ctx, ws, have_records, M, c.cout, c.c1(), c.c2(), m->grads + c.g_off, m->grads + c.be_off);
    else
        launch_bn_bwd_reduce(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(), ws, c.c1(), c.c2(),
                             m->grads + c.g_off, m->grads + c.be_off, m->act_slope);
    const hipEvent_t dy_done = m->next_fork_event();         // completes with the kernel that writes dY
    launch_bn_bwd_apply(ctx, dA, Y, M, c.cout, c.scale(), c.shift(), c.mean(), c.invstd(),
                        m->params + c.g_off, c.c1(), c.c2(), ws, m->grads + c.b_off, m->act_slope, dYp.p, dYp.pstride,
                        m->planesP, dy_done);
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
    wa.N = s.N; wa.H = s.H; wa.W = s.W; wa.Hx = s.H; wa.Wx = s.W;
    wa.dw = m->grads + c.w_off;
    wa.tap_stride = (int64_t)c.cin_p * c.cout;
    wa.sy = c.cin_p; wa.sx = 1;
    wa.algo_flops = 2.0 * s.N * s.H * s.W * 9.0 * c.cin * c.cout;
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
            a.bwd_slope = m->act_slope;
        }
        launch_pconv(ctx, a);
        return a.stats_records;
    }
    return 0;
}

}  // namespace

void rfi_model::backward_planes(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w) {
    (void)x_dev;
    const int D = depth;
    // every layer owns its dY plane tensor and the other side-stream inputs (forward planes, dconcat, raw conv outputs)
    // are not rewritten before side_join(): the main stream never has to wait for a weight gradient inside the pass
    static const int bound_env = getenv("RFI_SIDE_BOUND") ? atoi(getenv("RFI_SIDE_BOUND")) : 0;
    side_bound = bound_env;
    // a raw conv output of the forward pass: bfloat16 tensor pl[hi] in the bf16 flow, else float32 bufs[fi]
    auto yr = [&](int fi, int hi) { return (y16_flow && hi >= 0) ? YRef(pl[hi].p, pl[hi].pstride) : YRef(buf(fi)); };
    // a gradient tensor an input-gradient conv writes: bfloat16 tensor pl[hi] in the bf16 flow, else float32 bufs[fi]
    auto gt = [&](int fi, int hi) { return hi >= 0 ? GT(pl[hi]) : GT(buf(fi)); };
    const int64_t M1 = (int64_t)n * h * w;
    if (loss_kind == 1)
        launch_focal_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, focal_alpha, focal_gamma, buf(dlogits));
    else
        launch_loss_bwd(ctx, buf(head_sigmoid ? probs : logits), labels_dev, M1, d_sums, buf(dlogits));
    if (head_sigmoid) launch_sigmoid_bwd(ctx, buf(probs), M1 * out_ch, buf(dlogits));
    int head_records = 0;
    {
        ConvBN& last = convs[2 * D + 2 + 2 * (D - 1) + 1];
        head_records = launch_head_bwd(ctx, yr(decY2[1], y16_flow ? yD2top : -1), M1, feat, last.scale(), last.shift(), params + head_w_off, out_ch,
                                       buf(dlogits), buf(gA[1]), buf(ws_red) + bn_bwd_ws_floats(M1, feat), grads + head_w_off,
                                       grads + head_b_off, act_slope, last.mean(), last.invstd(), buf(ws_red),
                                       g16_flow && out_ch == 1 ? pl[g16A[1]].p : nullptr);
    }
    for (int l = 1; l <= D; ++l) {                // decoders, shallow to deep
        const int k = D - l;
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        Shape sin{n, h >> l, w >> l};
        ConvBN& c1 = convs[2 * D + 2 + 2 * k];
        ConvBN& c2 = convs[2 * D + 2 + 2 * k + 1];
        UpConv& u = ups[k];
        const PlaneSeg a1 = seg_of(pl[pA1d[l]]);
        const int rec1 = backward_pconv_bn(this, c2, gt(gA[l], g16_flow && out_ch == 1 && l == 1 ? g16A[1] : -1).ref(),
                                           yr(decY2[l], y16_flow && l == 1 ? yD2top : -1), &a1, 1, s,
                                           gt(gB[l], g16_flow ? g16B[l] : -1), pl[pdYa[l]], l == 1 ? head_records : 0, &c1,
                                           yr(decY1[l], y16_flow ? yD1[l] : -1));
        const PlaneSeg in2[2] = {seg_of(pl[pUp[l]]), seg_of(pl[pSkip[l]])};
        backward_pconv_bn(this, c1, gt(gB[l], g16_flow ? g16B[l] : -1).ref(), yr(decY1[l], y16_flow ? yD1[l] : -1), in2, 2, s, buf(dconcat[l]), pl[pdYb[l]],
                          rec1);
        // ConvTranspose: dUp = dconcat[..., 0:C]; the round-1 kernels on float32 tensors
        const float* prevY = (l == D) ? buf(bottY2) : buf(decY2[l + 1]);
        ConvBN& prevBN = (l == D) ? convs[2 * D + 1] : convs[2 * D + 2 + 2 * (k - 1) + 1];
        View dUp{buf(dconcat[l]), 2 * u.cout};
        launch_channel_sum(ctx, dUp, (int64_t)s.N * s.H * s.W, u.cout, buf(ws_red), grads + u.b_off);
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
        const int rec1 = backward_pconv_bn(this, convs[2 * D + 1], buf(gBottA), buf(bottY2), &a1, 1, s, gt(gBottB, g16_flow ? g16BottB : -1),
                                           pl[pdYbottA], 0, &convs[2 * D], yr(bottY1, y16_flow ? yB1 : -1));
        backward_pconv_bn(this, convs[2 * D], gt(gBottB, g16_flow ? g16BottB : -1).ref(), yr(bottY1, y16_flow ? yB1 : -1), &p4, 1, s,
                          gt(dpool[D], g16_flow ? g16pool[D] : -1), pl[pdYbottB], rec1);
        bucket_ready(convs[2 * D].w_off, ups[0].w_off);
    }
    for (int l = D; l >= 1; --l) {                // encoders, deep to shallow
        Shape s{n, h >> (l - 1), w >> (l - 1)};
        ConvBN& c1 = convs[2 * (l - 1)];
        ConvBN& c2 = convs[2 * (l - 1) + 1];
        const int have = launch_pool_bwd_merge_sums(ctx, yr(encY2[l], y16_flow ? yE2[l] : -1), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(), c2.mean(),
                                                    c2.invstd(), View{buf(dconcat[l]) + c2.cout, 2 * c2.cout}, gt(dpool[l], g16_flow ? g16pool[l] : -1).ref(),
                                                    buf(gA[l]), act_slope, buf(ws_red), g16_flow ? pl[g16A[l]].p : nullptr);
        if (!have)
            launch_pool_bwd_merge(ctx, yr(encY2[l], y16_flow ? yE2[l] : -1), s.N, s.H, s.W, c2.cout, c2.scale(), c2.shift(),
                                  View{buf(dconcat[l]) + c2.cout, 2 * c2.cout}, gt(dpool[l], g16_flow ? g16pool[l] : -1).ref(), buf(gA[l]), act_slope,
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
    side_join();
    side_bound = 2;
}
