// Region-proposal pieces of the Mask R-CNN path (BASELINE.json configs[3]).  NOT in the reference
// (no detector there, no torchvision here): defined by the published algorithms -- Ren et al. 2015 (Faster R-CNN: box
// parameterisation, RPN loss), Girshick 2015 (smooth L1), greedy IoU non-maximum suppression -- with the conventions of
// the de-facto implementation (box-coder weights 1, dw / dh clamped at log(1000 / 16), smooth-L1 beta 1/9, both loss
// terms divided by the number of sampled anchors).  Oracle: oracle/detection_ref.py; parity unpinned by the reference.
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kB = 256;

// boxes = decode(anchors, deltas), clipped to the image.  One thread per box; anchors repeat with period n_anchors.
__global__ __launch_bounds__(kB) void box_decode_kernel(const float* __restrict__ anchors, int64_t n_anchors,
                                                       const float* __restrict__ deltas, int64_t n, float clip_h, float clip_w,
                                                       float* __restrict__ out) {
    const float kClamp = 4.135166556742356f;         // log(1000 / 16)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = *reinterpret_cast<const float4*>(anchors + (i % n_anchors) * 4);
        const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
        const float w = a.z - a.x, h = a.w - a.y, cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
        const float dw = fminf(d.z, kClamp), dh = fminf(d.w, kClamp);
        const float pcx = d.x * w + cx, pcy = d.y * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        float4 b = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
        if (clip_w > 0.0f) {
            b.x = fminf(fmaxf(b.x, 0.0f), clip_w); b.z = fminf(fmaxf(b.z, 0.0f), clip_w);
            b.y = fminf(fmaxf(b.y, 0.0f), clip_h); b.w = fminf(fmaxf(b.w, 0.0f), clip_h);
        }
        *reinterpret_cast<float4*>(out + i * 4) = b;
    }
}

// Suppression matrix of n boxes given in descending score order: bit j of mask[i][j / 64] is set when j > i and
// IoU(i, j) > thr.  Block (bi, bj) handles rows 64 bi .. and columns 64 bj ..; the column boxes sit in LDS.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                     unsigned long long* __restrict__ mask, int words) {
    __shared__ float4 cols[64];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;                              // (j > i only)
    const int j0 = bj * 64, i = bi * 64 + threadIdx.x;
    if (j0 + (int)threadIdx.x < n) cols[threadIdx.x] = *reinterpret_cast<const float4*>(boxes + (int64_t)(j0 + threadIdx.x) * 4);
    __syncthreads();
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(boxes + (int64_t)i * 4);
    const float area_a = (a.z - a.x) * (a.w - a.y);
    unsigned long long bits = 0;
    const int cnt = min(64, n - j0);
    for (int k = 0; k < cnt; ++k) {
        if (j0 + k <= i) continue;
        const float4 b = cols[k];
        const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f), ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
        const float inter = iw * ih, uni = area_a + (b.z - b.x) * (b.w - b.y) - inter;
        if (inter > thr * uni) bits |= 1ull << k;     // IoU > thr without the division (uni >= 0)
    }
    mask[(int64_t)i * words + bj] = bits;
}

// RPN loss over the head output [P][5 A] (A objectness logits, then A x 4 box deltas per pixel): labels [P A] in
// {1 positive, 0 negative, -1 not sampled}, targets [P A][4].  Writes d(loss)/d(head output) and fp64 partials
// [block][2] = (sum BCE over sampled, sum smooth-L1 over positives), both terms scaled by inv_count.
__global__ __launch_bounds__(kB) void rpn_loss_kernel(const float* __restrict__ head, int64_t P, int A,
                                                     const signed char* __restrict__ labels, const float* __restrict__ targets,
                                                     float inv_count, const int* __restrict__ count_dev, float beta,
                                                     float* __restrict__ dhead, double* __restrict__ partial) {
    __shared__ double red[2][kB];
    if (count_dev) inv_count = 1.0f / (float)max(*count_dev, 1);         // the normaliser lives on the device (the sampler's count)
    double s_obj = 0.0, s_box = 0.0;
    const int64_t total = P * A;
    const int ps = 5 * A;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / A;
        const int a = (int)(i % A);
        const int lab = labels[i];
        const float x = head[p * ps + a];
        float dx = 0.0f;
        if (lab >= 0) {
            const float t = lab > 0 ? 1.0f : 0.0f;
            s_obj += (double)(fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x))));
            dx = (1.0f / (1.0f + expf(-x)) - t) * inv_count;
        }
        dhead[p * ps + a] = dx;
        const float4 d = *reinterpret_cast<const float4*>(head + p * ps + A + 4 * a);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lab > 0) {
            const float4 t4 = *reinterpret_cast<const float4*>(targets + i * 4);
            const float e[4] = {d.x - t4.x, d.y - t4.y, d.z - t4.z, d.w - t4.w};
            float ge[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ae = fabsf(e[k]);
                if (ae < beta) { s_box += (double)(0.5f * e[k] * e[k] / beta); ge[k] = e[k] / beta * inv_count; }
                else { s_box += (double)(ae - 0.5f * beta); ge[k] = (e[k] > 0.0f ? inv_count : -inv_count); }
            }
            g = make_float4(ge[0], ge[1], ge[2], ge[3]);
        }
        *reinterpret_cast<float4*>(dhead + p * ps + A + 4 * a) = g;
    }
    red[0][threadIdx.x] = s_obj;
    red[1][threadIdx.x] = s_box;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = red[0][0];
        partial[blockIdx.x * 2 + 1] = red[1][0];
    }
}
__global__ void rpn_loss_finish_kernel(const double* __restrict__ partial, int blocks, double inv_count, float* __restrict__ out2,
                                       const int* __restrict__ count_dev = nullptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (count_dev) inv_count = (double)(1.0f / (float)max(*count_dev, 1));
        double a = 0.0, b = 0.0;
        for (int i = 0; i < blocks; ++i) { a += partial[i * 2]; b += partial[i * 2 + 1]; }     // fixed order
        out2[0] = (float)(a * inv_count);
        out2[1] = (float)(b * inv_count);
    }
}

// Fast R-CNN loss over the box head's output [R][5 K1] (K1 = classes incl. background: K1 class logits, then K1 x 4 box
// deltas): mean cross-entropy over the RoIs + smooth L1 of the ground-truth class's deltas over the foreground RoIs,
// both / R.  One thread per RoI (K1 is small); gradient w.r.t. the head output; fp64 block partials.
__global__ __launch_bounds__(kB) void fastrcnn_loss_kernel(const float* __restrict__ head, int64_t R, int K1, const int* __restrict__ labels,
                                                          const float* __restrict__ targets, float inv_R, float beta,
                                                          float* __restrict__ dhead, double* __restrict__ partial) {
    __shared__ double red[2][kB];
    double s_cls = 0.0, s_box = 0.0;
    const int ps = 5 * K1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
        const float* x = head + i * ps;
        float* g = dhead + i * ps;
        const int lab = labels[i];
        float mx = x[0];
        for (int j = 1; j < K1; ++j) mx = fmaxf(mx, x[j]);
        float se = 0.0f;
        for (int j = 0; j < K1; ++j) se += expf(x[j] - mx);
        const float lse = mx + logf(se);
        s_cls += (double)(lse - x[lab]);
        for (int j = 0; j < K1; ++j) g[j] = (expf(x[j] - lse) - (j == lab ? 1.0f : 0.0f)) * inv_R;
        for (int j = 0; j < 4 * K1; ++j) g[K1 + j] = 0.0f;
        if (lab > 0) {
            for (int k = 0; k < 4; ++k) {
                const float e = x[K1 + 4 * lab + k] - targets[i * 4 + k], ae = fabsf(e);
                if (ae < beta) { s_box += (double)(0.5f * e * e / beta); g[K1 + 4 * lab + k] = e / beta * inv_R; }
                else { s_box += (double)(ae - 0.5f * beta); g[K1 + 4 * lab + k] = e > 0.0f ? inv_R : -inv_R; }
            }
        }
    }
    red[0][threadIdx.x] = s_cls;
    red[1][threadIdx.x] = s_box;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = red[0][0];
        partial[blockIdx.x * 2 + 1] = red[1][0];
    }
}

// ---- anchor <-> ground-truth matching (the Matcher of the usual implementation): IoU matrix never materialised
__device__ __forceinline__ float iou_of(const float4 a, const float4 b) {
    const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f), ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = iw * ih;
    return inter / ((a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter);
}
// best[g] = max over anchors of IoU(gt g, anchor): one block per ground-truth box
__global__ __launch_bounds__(kB) void gt_best_iou_kernel(const float* __restrict__ anchors, int64_t n, const float* __restrict__ gt,
                                                        float* __restrict__ best) {
    __shared__ float red[kB];
    const float4 g = *reinterpret_cast<const float4*>(gt + (int64_t)blockIdx.x * 4);
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += kB) m = fmaxf(m, iou_of(g, *reinterpret_cast<const float4*>(anchors + i * 4)));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) best[blockIdx.x] = red[0];
}
// per anchor: matched ground truth = first argmax of the IoU; label 1 (IoU >= hi, or -- low-quality rule -- the anchor
// attains some ground truth's best IoU), 0 (IoU < lo), -1 (between); matched index kept for labels 1 only (else -1)
__global__ __launch_bounds__(kB) void anchor_match_kernel(const float* __restrict__ anchors, int64_t n, const float* __restrict__ gt,
                                                         int G, const float* __restrict__ best, float hi, float lo, int low_quality,
                                                         signed char* __restrict__ labels, int* __restrict__ matched) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = *reinterpret_cast<const float4*>(anchors + i * 4);
        float mv = -1.0f;
        int mi = -1;
        bool lq = false;
        for (int g = 0; g < G; ++g) {
            const float v = iou_of(*reinterpret_cast<const float4*>(gt + (int64_t)g * 4), a);
            if (v > mv) { mv = v; mi = g; }
            lq = lq || (low_quality && v == best[g] && v > 0.0f);
        }
        int lab = G == 0 ? 0 : (mv >= hi ? 1 : (mv < lo ? 0 : -1));
        if (lq) lab = 1;
        labels[i] = (signed char)lab;
        matched[i] = lab == 1 ? mi : -1;
    }
}
// regression targets of the positive anchors: encode(gt[matched], anchor) with weights 1; zeros elsewhere
__global__ __launch_bounds__(kB) void box_encode_kernel(const float* __restrict__ anchors, int64_t n, const float* __restrict__ gt,
                                                       const int* __restrict__ matched, float* __restrict__ targets) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        const int m = matched[i];
        if (m >= 0) {
            const float4 a = *reinterpret_cast<const float4*>(anchors + i * 4), g = *reinterpret_cast<const float4*>(gt + (int64_t)m * 4);
            const float aw = a.z - a.x, ah = a.w - a.y, gw = g.z - g.x, gh = g.w - g.y;
            t = make_float4(((g.x + 0.5f * gw) - (a.x + 0.5f * aw)) / aw, ((g.y + 0.5f * gh) - (a.y + 0.5f * ah)) / ah, logf(gw / aw),
                            logf(gh / ah));
        }
        *reinterpret_cast<float4*>(targets + i * 4) = t;
    }
}

// ---- batched forms (one launch for every image of a batch): blockIdx.y = image.  Anchors are shared (stride 0) or per
// image ([B][n][4], `acount[b]` of them valid: the proposals of the RoI stage); ground truth [B][Gmax][4] with gcount[b]
__global__ __launch_bounds__(kB) void gt_best_iou_batched_kernel(const float* __restrict__ anchors, int64_t n, int64_t astride,
                                                                const int* __restrict__ acount, const float* __restrict__ gt, int Gmax,
                                                                const int* __restrict__ gcount, float* __restrict__ best) {
    __shared__ float red[kB];
    const int b = blockIdx.y, g = blockIdx.x;
    if (g >= gcount[b]) return;                       // (uniform for the block)
    const float* an = anchors + (int64_t)b * astride * 4;
    const int64_t na = acount ? acount[b] : n;
    const float4 gb = *reinterpret_cast<const float4*>(gt + ((int64_t)b * Gmax + g) * 4);
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < na; i += kB) m = fmaxf(m, iou_of(gb, *reinterpret_cast<const float4*>(an + i * 4)));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) best[(int64_t)b * Gmax + g] = red[0];
}
// labels as anchor_match_kernel; an entry beyond acount[b] gets label -2 (not a box), matched -1, zero targets
__global__ __launch_bounds__(kB) void anchor_match_batched_kernel(const float* __restrict__ anchors, int64_t n, int64_t astride,
                                                                 const int* __restrict__ acount, const float* __restrict__ gt, int Gmax,
                                                                 const int* __restrict__ gcount, const float* __restrict__ best, float hi,
                                                                 float lo, int low_quality, signed char* __restrict__ labels,
                                                                 int* __restrict__ matched, float* __restrict__ targets) {
    const int b = blockIdx.y;
    const float* an = anchors + (int64_t)b * astride * 4;
    const float* gb = gt + (int64_t)b * Gmax * 4;
    const int G = gcount[b];
    const int64_t na = acount ? acount[b] : n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lab = -2, mi = -1;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < na) {
            const float4 a = *reinterpret_cast<const float4*>(an + i * 4);
            float mv = -1.0f;
            bool lq = false;
            for (int g = 0; g < G; ++g) {
                const float v = iou_of(*reinterpret_cast<const float4*>(gb + (int64_t)g * 4), a);
                if (v > mv) { mv = v; mi = g; }
                lq = lq || (low_quality && v == best[(int64_t)b * Gmax + g] && v > 0.0f);
            }
            lab = G == 0 ? 0 : (mv >= hi ? 1 : (mv < lo ? 0 : -1));
            if (lq) lab = 1;
            if (lab != 1) mi = -1;
            if (mi >= 0 && targets) {
                const float4 g = *reinterpret_cast<const float4*>(gb + (int64_t)mi * 4);
                const float aw = a.z - a.x, ah = a.w - a.y, gw = g.z - g.x, gh = g.w - g.y;
                t = make_float4(((g.x + 0.5f * gw) - (a.x + 0.5f * aw)) / aw, ((g.y + 0.5f * gh) - (a.y + 0.5f * ah)) / ah, logf(gw / aw),
                                logf(gh / ah));
            }
        }
        labels[(int64_t)b * n + i] = (signed char)lab;
        matched[(int64_t)b * n + i] = mi;
        if (targets) *reinterpret_cast<float4*>(targets + ((int64_t)b * n + i) * 4) = t;
    }
}

// greedy NMS of B independent sets of at most K <= 256 boxes, each sorted by descending score: one workgroup per set builds
// the suppression matrix (bit j of row i: j > i and IoU > thr) in LDS, its first thread runs the greedy scan.  keep[b][i] = 1
// for the boxes kept, 0 for the suppressed ones and for i >= count[b]
constexpr int kNmsK = 256;
__global__ __launch_bounds__(kNmsK) void nms_batched_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int K, float thr,
                                                           unsigned char* __restrict__ keep) {
    __shared__ float4 bx[kNmsK];
    __shared__ unsigned long long sup[kNmsK][kNmsK / 64];
    const int b = blockIdx.x, i = threadIdx.x;
    const int n = min(count[b], K);
    if (i < n) bx[i] = *reinterpret_cast<const float4*>(boxes + ((int64_t)b * K + i) * 4);
    __syncthreads();
    if (i < n) {
        const float4 a = bx[i];
        const float area_a = (a.z - a.x) * (a.w - a.y);
        for (int w = 0; w < kNmsK / 64; ++w) {
            unsigned long long bits = 0;
            for (int k = 0; k < 64; ++k) {
                const int j = w * 64 + k;
                if (j <= i || j >= n) continue;
                const float4 c = bx[j];
                const float iw = fmaxf(fminf(a.z, c.z) - fmaxf(a.x, c.x), 0.0f), ih = fmaxf(fminf(a.w, c.w) - fmaxf(a.y, c.y), 0.0f);
                const float inter = iw * ih, uni = area_a + (c.z - c.x) * (c.w - c.y) - inter;
                if (inter > thr * uni) bits |= 1ull << k;
            }
            sup[i][w] = bits;
        }
    }
    __syncthreads();
    if (i == 0) {
        unsigned long long removed[kNmsK / 64] = {0ull, 0ull, 0ull, 0ull};
        for (int r = 0; r < K; ++r) {
            const bool kept = r < n && !((removed[r / 64] >> (r % 64)) & 1ull);
            keep[(int64_t)b * K + r] = kept ? 1 : 0;
            if (kept)
                for (int w = r / 64; w < kNmsK / 64; ++w) removed[w] |= sup[r][w];
        }
    }
}

}  // namespace

void launch_anchor_match_batched(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int* anchor_count,
                                 const float* gt, int B, int Gmax, const int* gt_count, float hi, float lo, bool low_quality,
                                 float* best_ws, signed char* labels, int* matched, float* targets) {
    RFI_REQUIRE(n > 0 && B > 0 && Gmax > 0, "anchor_match_batched: empty input");
    RFI_REQUIRE(!((reinterpret_cast<uintptr_t>(anchors) | reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(targets)) & 15),
                "anchor_match_batched: 16-byte aligned boxes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * n * (16.0 * 2 + 21));
    hipLaunchKernelGGL(gt_best_iou_batched_kernel, dim3(Gmax, B), dim3(kB), 0, ctx->stream, anchors, n, anchor_stride, anchor_count, gt,
                       Gmax, gt_count, best_ws);
    check_launch("gt_best_iou_batched");
    int64_t b = cdiv(n, kB);
    if (b > 256) b = 256;
    hipLaunchKernelGGL(anchor_match_batched_kernel, dim3((unsigned)b, B), dim3(kB), 0, ctx->stream, anchors, n, anchor_stride,
                       anchor_count, gt, Gmax, gt_count, best_ws, hi, lo, low_quality ? 1 : 0, labels, matched, targets);
    check_launch("anchor_match_batched");
}

void launch_nms_batched(rfi_ctx* ctx, const float* boxes, const int* count, int B, int K, float thr, unsigned char* keep) {
    RFI_REQUIRE(B > 0 && K > 0 && K <= kNmsK, "nms_batched: at most 256 boxes per set");
    RFI_REQUIRE(!(reinterpret_cast<uintptr_t>(boxes) & 15), "nms_batched: 16-byte aligned boxes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * K * 17);
    hipLaunchKernelGGL(nms_batched_kernel, dim3(B), dim3(kNmsK), 0, ctx->stream, boxes, count, K, thr, keep);
    check_launch("nms_batched");
}

void launch_anchor_match(rfi_ctx* ctx, const float* anchors, int64_t n, const float* gt, int G, float hi, float lo, bool low_quality,
                         float* best_ws, signed char* labels, int* matched, float* targets) {
    RFI_REQUIRE(n > 0 && G >= 0, "anchor_match: empty anchors");
    RFI_REQUIRE(!((reinterpret_cast<uintptr_t>(anchors) | reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(targets)) & 15),
                "anchor_match: 16-byte aligned boxes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * (16.0 * (G > 0 ? 2 : 1) + 21));
    if (G > 0) {
        hipLaunchKernelGGL(gt_best_iou_kernel, dim3(G), dim3(kB), 0, ctx->stream, anchors, n, gt, best_ws);
        check_launch("gt_best_iou");
    }
    int64_t b = cdiv(n, kB);
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(anchor_match_kernel, dim3((unsigned)b), dim3(kB), 0, ctx->stream, anchors, n, gt, G, best_ws, hi, lo,
                       low_quality ? 1 : 0, labels, matched);
    check_launch("anchor_match");
    if (targets) {
        hipLaunchKernelGGL(box_encode_kernel, dim3((unsigned)b), dim3(kB), 0, ctx->stream, anchors, n, gt, matched, targets);
        check_launch("box_encode");
    }
}

void launch_fastrcnn_loss(rfi_ctx* ctx, const float* head, int64_t R, int K1, const int* labels, const float* targets, float beta,
                          float* dhead, double* partial_ws, float* loss2_dev) {
    RFI_REQUIRE(R > 0 && K1 >= 2, "fastrcnn_loss: R > 0 and at least background + one class");
    const float inv = 1.0f / (float)R;
    int64_t blocks = cdiv(R, kB);
    if (blocks > 1024) blocks = 1024;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)R * (10.0 * K1 * 4 + 20));
    hipLaunchKernelGGL(fastrcnn_loss_kernel, dim3((unsigned)blocks), dim3(kB), 0, ctx->stream, head, R, K1, labels, targets, inv, beta,
                       dhead, partial_ws);
    check_launch("fastrcnn_loss");
    hipLaunchKernelGGL(rpn_loss_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, partial_ws, (int)blocks, (double)inv, loss2_dev,
                       (const int*)nullptr);
    check_launch("fastrcnn_loss_finish");
}

void launch_box_decode(rfi_ctx* ctx, const float* anchors, int64_t n_anchors, const float* deltas, int64_t n, float clip_h,
                       float clip_w, float* out) {
    RFI_REQUIRE(n > 0 && n_anchors > 0, "box_decode: empty input");
    RFI_REQUIRE(!((reinterpret_cast<uintptr_t>(anchors) | reinterpret_cast<uintptr_t>(deltas) | reinterpret_cast<uintptr_t>(out)) & 15),
                "box_decode: 16-byte aligned tensors");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 48);
    int64_t b = cdiv(n, kB);
    hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(kB), 0, ctx->stream, anchors, n_anchors, deltas, n,
                       clip_h, clip_w, out);
    check_launch("box_decode");
}
void launch_nms_mask(rfi_ctx* ctx, const float* boxes, int n, float thr, unsigned long long* mask) {
    RFI_REQUIRE(n > 0 && !(reinterpret_cast<uintptr_t>(boxes) & 15), "nms: empty input or unaligned boxes");
    const int words = (int)cdiv(n, 64);
    RFI_CHECK_HIP(hipMemsetAsync(mask, 0, (size_t)n * words * 8, ctx->stream));
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * words * 8);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, ctx->stream, boxes, n, thr, mask, words);
    check_launch("nms_mask");
}
size_t rpn_loss_ws_doubles() { return 2 * 1024; }
void launch_rpn_loss(rfi_ctx* ctx, const float* head, int64_t P, int A, const signed char* labels, const float* targets,
                     int64_t num_sampled, float beta, float* dhead, double* partial_ws, float* loss2_dev, const int* num_sampled_dev) {
    RFI_REQUIRE(P > 0 && A > 0 && A % 4 == 0, "rpn_loss: P > 0 and anchors per pixel a multiple of 4 (16-byte aligned delta groups)");
    RFI_REQUIRE(!((reinterpret_cast<uintptr_t>(head) | reinterpret_cast<uintptr_t>(targets) | reinterpret_cast<uintptr_t>(dhead)) & 15),
                "rpn_loss: 16-byte aligned tensors");
    const float inv = num_sampled > 0 ? 1.0f / (float)num_sampled : 0.0f;
    int64_t blocks = cdiv(P * A, kB);
    if (blocks > 1024) blocks = 1024;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)P * A * (5 * 8 + 17));
    hipLaunchKernelGGL(rpn_loss_kernel, dim3((unsigned)blocks), dim3(kB), 0, ctx->stream, head, P, A, labels, targets, inv, num_sampled_dev,
                       beta, dhead, partial_ws);
    check_launch("rpn_loss");
    hipLaunchKernelGGL(rpn_loss_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, partial_ws, (int)blocks, (double)inv, loss2_dev,
                       num_sampled_dev);
    check_launch("rpn_loss_finish");
}

}  // namespace rfi
