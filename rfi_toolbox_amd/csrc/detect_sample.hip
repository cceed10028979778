// The box bookkeeping of the Mask R-CNN training step on the device (BASELINE.json configs[3]): what
// round 3 did in NumPy between the GPU stages -- the two random samplers, the per-level top-k of the objectness scores,
// the post-NMS selection, the RoI lists -- with PCIe round trips inside the timed step.  NOT in the reference (it has no
// detector): the rules are the published ones (Ren et al. 2015: 256 anchors per image with at most half positive, 128 RoIs
// with at most a quarter foreground; Lin et al. 2017: top-k per pyramid level, NMS per level, best overall) with this
// package's conventions; oracle/mask_rcnn_ref.py restates them in NumPy -- parity unpinned by the reference.
//
// Everything here is small integer / box work on a few thousand elements per image: one workgroup per image (or per
// (image, level) set), 64-bit keys sorted in LDS by a bitonic network, results written in a fixed order -- no atomics on
// floats, no data-dependent launch geometry, nothing read back by the host.
//
// Random choices: element i of image b draws r = Philox4x32-10(counter (i, b, stream, step), key = seed).x and a class
// keeps its elements of SMALLEST (r, i) -- a function of (seed, step, image, element) only, which the oracle evaluates with
// the same generator (oracle/synth_ref.philox4x32_10).
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kB = 256;
typedef unsigned long long u64;

struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u64 p0 = (u64)0xD2511F53u * c.x;
        const u64 p1 = (u64)0xCD9E8D57u * c.z;
        const U4 n{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// ascending bitonic sort of s[0 .. n) in LDS, n a power of two; every thread of the block takes part
__device__ void bitonic_sort(u64* s, int n) {
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const u64 a = s[i], b = s[p];
                if ((a > b) == up) { s[i] = b; s[p] = a; }
            }
        }
    __syncthreads();
}

// order-preserving image of a float32 for DESCENDING order: smaller key = larger score; -0.0 and 0.0 tie
__device__ __forceinline__ unsigned desc_key(float s) {
    const unsigned bits = __float_as_uint((-s) + 0.0f);
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__device__ __forceinline__ int lower_bound(const u64* s, int n, u64 v) {      // first index with s[i] >= v
    int a = 0, b = n;
    while (a < b) { const int m = (a + b) >> 1; if (s[m] < v) a = m + 1; else b = m; }
    return a;
}

// ---------------------------------------------------------------- segmented sort: one workgroup per segment
__global__ __launch_bounds__(1024) void segsort_kernel(u64* __restrict__ keys, int stride) {
    extern __shared__ u64 s_keys[];
    u64* seg = keys + (size_t)blockIdx.x * stride;
    for (int i = threadIdx.x; i < stride; i += blockDim.x) s_keys[i] = seg[i];
    bitonic_sort(s_keys, stride);
    for (int i = threadIdx.x; i < stride; i += blockDim.x) seg[i] = s_keys[i];
}

// ---------------------------------------------------------------- sampler keys: class << 48 | random << 16 | index
// class 0: label 1 (positive), 1: label 0 (negative), 3: neither; entries beyond the image's count (or n) sort last
__global__ __launch_bounds__(kB) void sample_keys_kernel(const signed char* __restrict__ labels, int n, const int* __restrict__ count,
                                                        unsigned k0, unsigned k1, unsigned step, unsigned stream0, u64* __restrict__ keys,
                                                        int stride) {
    const int b = blockIdx.y;
    const int cnt = count ? min(count[b], n) : n;
    for (int i = blockIdx.x * kB + threadIdx.x; i < stride; i += gridDim.x * kB) {
        u64 key = ~0ull;
        if (i < cnt) {
            const int lab = labels[(size_t)b * n + i];
            const unsigned cls = lab == 1 ? 0u : (lab == 0 ? 1u : 3u);
            const unsigned r = philox4x32_10(U4{(unsigned)i, (unsigned)b, stream0 + (cls == 0 ? 0u : 1u), step}, k0, k1).x;
            key = ((u64)cls << 48) | ((u64)r << 16) | (u64)i;
        }
        keys[(size_t)b * stride + i] = key;
    }
}

struct RpnLevels {
    int L;
    int off[6];                        // anchors of level l of one image: [off[l], off[l + 1])
    signed char* lab[5];               // per level: labels [images][off[l + 1] - off[l]] (the layout rfi_op_rpn_loss reads)
    float* tgt[5];                     // ... and regression targets [..][4]
};
// per image: the sampled anchors keep their label, every other label >= 0 becomes -1; labels and targets are written level by
// level; *n_sampled += anchors sampled (integer atomic: order independent)
__global__ __launch_bounds__(kB) void rpn_sample_apply_kernel(const u64* __restrict__ keys, int n, int stride, int batch, int max_pos,
                                                             const signed char* __restrict__ labels, const float* __restrict__ targets,
                                                             RpnLevels lv, int* __restrict__ n_sampled) {
    extern __shared__ unsigned s_bits[];          // one bit per anchor: sampled
    __shared__ int s_cnt[4];
    const int b = blockIdx.x;
    const u64* k = keys + (size_t)b * stride;
    for (int i = threadIdx.x; i < (n + 31) / 32; i += kB) s_bits[i] = 0u;
    if (threadIdx.x == 0) {
        const int pos_avail = lower_bound(k, stride, 1ull << 48);
        const int neg_avail = lower_bound(k, stride, 2ull << 48) - pos_avail;
        const int npos = min(pos_avail, max_pos), nneg = min(neg_avail, batch - npos);
        s_cnt[0] = pos_avail; s_cnt[1] = npos; s_cnt[2] = nneg;
        atomicAdd(n_sampled, npos + nneg);
    }
    __syncthreads();
    const int pos_avail = s_cnt[0], npos = s_cnt[1], nneg = s_cnt[2];
    for (int p = threadIdx.x; p < npos + nneg; p += kB) {
        const int idx = (int)(k[p < npos ? p : pos_avail + (p - npos)] & 0xffffull);
        atomicOr(&s_bits[idx >> 5], 1u << (idx & 31));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kB) {
        int lab = labels[(size_t)b * n + i];
        if (lab >= 0 && !((s_bits[i >> 5] >> (i & 31)) & 1u)) lab = -1;
        int l = 0;
        while (l + 1 < lv.L && i >= lv.off[l + 1]) ++l;
        const int al = lv.off[l + 1] - lv.off[l];
        const size_t o = (size_t)b * al + (i - lv.off[l]);
        lv.lab[l][o] = (signed char)lab;
        *reinterpret_cast<float4*>(lv.tgt[l] + o * 4) = *reinterpret_cast<const float4*>(targets + ((size_t)b * n + i) * 4);
    }
}

// ---------------------------------------------------------------- per-level top-k of the objectness logits
// head [B][P][5 A]: score of anchor j = pixel * A + a is head[b][pixel][a]
__global__ __launch_bounds__(kB) void topk_keys_kernel(const float* __restrict__ head, int P, int A, u64* __restrict__ keys, int stride) {
    const int b = blockIdx.y, AL = P * A;
    for (int j = blockIdx.x * kB + threadIdx.x; j < stride; j += gridDim.x * kB) {
        u64 key = ~0ull;
        if (j < AL) key = ((u64)desc_key(head[((size_t)b * P + j / A) * 5 * A + j % A]) << 32) | (u64)j;
        keys[(size_t)b * stride + j] = key;
    }
}
// the K best of one (image, level): decode + clip, boxes under min_size to the back (order kept), count of the others
__global__ __launch_bounds__(kB) void topk_decode_kernel(const u64* __restrict__ keys, int stride, int P, int A, int K,
                                                        const float* __restrict__ head, const float* __restrict__ anchors, float clip_h,
                                                        float clip_w, float min_size, float* __restrict__ boxes, float* __restrict__ scores,
                                                        int* __restrict__ counts, int L, int lvl) {
    __shared__ int s_valid[kB];
    const float kClamp = 4.135166556742356f;         // log(1000 / 16)
    const int b = blockIdx.x, r = threadIdx.x, AL = P * A;
    const int kl = min(K, AL);
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = -INFINITY;
    bool ok = false;
    if (r < kl) {
        const int j = (int)(keys[(size_t)b * stride + r] & 0xffffffffull);
        const float* hp = head + ((size_t)b * P + j / A) * 5 * A;
        sc = hp[j % A];
        const float4 a = *reinterpret_cast<const float4*>(anchors + (size_t)j * 4);
        const float4 d = *reinterpret_cast<const float4*>(hp + A + 4 * (j % A));
        const float w = a.z - a.x, h = a.w - a.y, cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
        const float dw = fminf(d.z, kClamp), dh = fminf(d.w, kClamp);
        const float pcx = d.x * w + cx, pcy = d.y * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        box = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
        box.x = fminf(fmaxf(box.x, 0.0f), clip_w); box.z = fminf(fmaxf(box.z, 0.0f), clip_w);
        box.y = fminf(fmaxf(box.y, 0.0f), clip_h); box.w = fminf(fmaxf(box.w, 0.0f), clip_h);
        ok = (box.z - box.x) >= min_size && (box.w - box.y) >= min_size;
    }
    s_valid[r] = ok ? 1 : 0;
    __syncthreads();
    int before = 0, total = 0;                        // (K <= 256: a serial count per thread is a few hundred LDS reads)
    for (int q = 0; q < kl; ++q) {
        const int v = s_valid[q];
        total += v;
        before += (q < r) ? v : 0;
    }
    float* ob = boxes + (((size_t)b * L + lvl) * K) * 4;
    float* os = scores + ((size_t)b * L + lvl) * K;
    if (r < kl) {
        const int pos = ok ? before : total + (r - before);
        *reinterpret_cast<float4*>(ob + (size_t)pos * 4) = box;
        os[pos] = ok ? sc : -INFINITY;
    } else if (r < K) {
        *reinterpret_cast<float4*>(ob + (size_t)r * 4) = box;
        os[r] = -INFINITY;
    }
    if (r == 0) counts[(size_t)b * L + lvl] = total;
}

// ---------------------------------------------------------------- proposals of one image: the post_nms best kept boxes over
// its levels (descending score, ties by level-major position), then its ground-truth boxes
__global__ __launch_bounds__(kB) void proposals_select_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                             const unsigned char* __restrict__ keep, int LK, int npow2, int post_nms,
                                                             const float* __restrict__ gt, int Gmax, const int* __restrict__ gt_count,
                                                             int Pmax, float* __restrict__ props, int* __restrict__ pcount) {
    extern __shared__ u64 s_keys[];
    __shared__ int s_kept;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    int mine = 0;
    for (int e = threadIdx.x; e < npow2; e += kB) {
        u64 key = ~0ull;
        if (e < LK && keep[(size_t)b * LK + e]) {
            key = ((u64)desc_key(scores[(size_t)b * LK + e]) << 32) | (u64)e;
            ++mine;
        }
        s_keys[e] = key;
    }
    if (mine) atomicAdd(&s_kept, mine);
    bitonic_sort(s_keys, npow2);
    const int nsel = min(s_kept, post_nms), G = min(gt_count[b], Gmax);
    float* pb = props + (size_t)b * Pmax * 4;
    for (int r = threadIdx.x; r < Pmax; r += kB) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nsel) v = *reinterpret_cast<const float4*>(boxes + ((size_t)b * LK + (int)(s_keys[r] & 0xffffffffull)) * 4);
        else if (r < nsel + G) v = *reinterpret_cast<const float4*>(gt + ((size_t)b * Gmax + (r - nsel)) * 4);
        *reinterpret_cast<float4*>(pb + (size_t)r * 4) = v;
    }
    if (threadIdx.x == 0) pcount[b] = min(nsel + G, Pmax);
}

// ---------------------------------------------------------------- RoI sampler of one image: positives (matcher label 1) first,
// in ascending (random, index) order, then negatives; sel[b][0 .. nsel[b]) = proposal indices
__global__ __launch_bounds__(kB) void roi_sample_kernel(const signed char* __restrict__ labels, const int* __restrict__ pcount, int Pmax,
                                                       int npow2, int batch, int max_pos, unsigned k0, unsigned k1, unsigned step,
                                                       unsigned stream0, int* __restrict__ sel, int* __restrict__ nsel, int* __restrict__ npos_out) {
    extern __shared__ u64 s_keys[];
    __shared__ int s_cnt[3];
    const int b = blockIdx.x, cnt = min(pcount[b], Pmax);
    for (int i = threadIdx.x; i < npow2; i += kB) {
        u64 key = ~0ull;
        if (i < cnt) {
            const int lab = labels[(size_t)b * Pmax + i];
            const unsigned cls = lab == 1 ? 0u : (lab == 0 ? 1u : 3u);
            const unsigned r = philox4x32_10(U4{(unsigned)i, (unsigned)b, stream0 + (cls == 0 ? 0u : 1u), step}, k0, k1).x;
            key = ((u64)cls << 48) | ((u64)r << 16) | (u64)i;
        }
        s_keys[i] = key;
    }
    bitonic_sort(s_keys, npow2);
    if (threadIdx.x == 0) {
        const int pos_avail = lower_bound(s_keys, npow2, 1ull << 48);
        const int neg_avail = lower_bound(s_keys, npow2, 2ull << 48) - pos_avail;
        const int npos = min(pos_avail, max_pos), nneg = min(neg_avail, batch - npos);
        s_cnt[0] = pos_avail; s_cnt[1] = npos; s_cnt[2] = nneg;
        nsel[b] = npos + nneg;
        npos_out[b] = npos;
    }
    __syncthreads();
    const int pos_avail = s_cnt[0], npos = s_cnt[1], nneg = s_cnt[2];
    for (int p = threadIdx.x; p < batch; p += kB)
        sel[(size_t)b * batch + p] = p < npos + nneg ? (int)(s_keys[p < npos ? p : pos_avail + (p - npos)] & 0xffffull) : -1;
}

struct RoiOut {
    float* rois;                       // [R][5] (image, x1, y1, x2, y2), image-major, positives of an image first
    int* cls;                          // [R] class label (0 = background)
    float* tgt;                        // [R][4] regression targets (zeros for background)
    int* gt;                           // [R] matched ground-truth index within the image (-1: background)
    int* level;                        // [R] pyramid level 0 .. 3
    int* img_start;                    // [B + 1]
    float* rois_fg;                    // [Rf][5] the foreground rows of rois ...
    float* rois_gt;                    // [Rf][5] (global instance index gbase[image] + gt, x1, y1, x2, y2): rfi_op_mask_targets
    int* level_fg;                     // [Rf]
    int* fg_start;                     // [B + 1]
    int* counts;                       // [2] = (R, Rf)
};
// level k of a box: 0 + [area >= t1] + [area >= t2] + [area >= t3], area = max(w h, 1e-6) in float32
__global__ __launch_bounds__(kB) void roi_compact_kernel(const int* __restrict__ sel, const int* __restrict__ nsel, const int* __restrict__ npos,
                                                        int B, int batch, int Pmax, const float* __restrict__ props,
                                                        const int* __restrict__ matched, const float* __restrict__ targets,
                                                        const int* __restrict__ gt_labels, int Gmax, const int* __restrict__ gbase, float t1,
                                                        float t2, float t3, RoiOut o) {
    __shared__ int s_off[2];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int r0 = 0, f0 = 0;
        for (int i = 0; i < b; ++i) { r0 += nsel[i]; f0 += npos[i]; }
        s_off[0] = r0; s_off[1] = f0;
        o.img_start[b] = r0;
        o.fg_start[b] = f0;
        if (b == B - 1) {
            o.img_start[B] = r0 + nsel[b];
            o.fg_start[B] = f0 + npos[b];
            o.counts[0] = r0 + nsel[b];
            o.counts[1] = f0 + npos[b];
        }
    }
    __syncthreads();
    const int r0 = s_off[0], f0 = s_off[1], ns = nsel[b], np = npos[b];
    for (int p = threadIdx.x; p < ns; p += kB) {
        const int i = sel[(size_t)b * batch + p];
        const float4 bx = *reinterpret_cast<const float4*>(props + ((size_t)b * Pmax + i) * 4);
        const int m = p < np ? matched[(size_t)b * Pmax + i] : -1;
        const float area = fmaxf((bx.z - bx.x) * (bx.w - bx.y), 1e-6f);
        const int lvl = (area >= t1 ? 1 : 0) + (area >= t2 ? 1 : 0) + (area >= t3 ? 1 : 0);
        const int r = r0 + p;
        float* q = o.rois + (size_t)r * 5;
        q[0] = (float)b; q[1] = bx.x; q[2] = bx.y; q[3] = bx.z; q[4] = bx.w;
        o.cls[r] = m >= 0 ? gt_labels[(size_t)b * Gmax + m] : 0;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m >= 0) t = *reinterpret_cast<const float4*>(targets + ((size_t)b * Pmax + i) * 4);
        o.tgt[(size_t)r * 4 + 0] = t.x; o.tgt[(size_t)r * 4 + 1] = t.y; o.tgt[(size_t)r * 4 + 2] = t.z; o.tgt[(size_t)r * 4 + 3] = t.w;
        o.gt[r] = m;
        o.level[r] = lvl;
        if (p < np) {
            const int f = f0 + p;
            float* qf = o.rois_fg + (size_t)f * 5;
            float* qg = o.rois_gt + (size_t)f * 5;
            qf[0] = (float)b; qf[1] = bx.x; qf[2] = bx.y; qf[3] = bx.z; qf[4] = bx.w;
            qg[0] = (float)(gbase[b] + m); qg[1] = bx.x; qg[2] = bx.y; qg[3] = bx.z; qg[4] = bx.w;
            o.level_fg[f] = lvl;
        }
    }
}

int pow2_at_least(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace

// ======================================================================================== launch wrappers
void launch_segsort_u64(rfi_ctx* ctx, unsigned long long* keys, int n_segs, int stride) {
    RFI_REQUIRE(n_segs > 0 && stride >= 2 && stride <= 8192 && (stride & (stride - 1)) == 0, "segsort: segments of 2 .. 8192 keys, a power of two");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n_segs * stride * 16);
    const int threads = stride >= 2048 ? 1024 : (stride >= 512 ? 256 : 64);      // (a compare-exchange pair per thread and stage)
    hipLaunchKernelGGL(segsort_kernel, dim3(n_segs), dim3(threads), (size_t)stride * 8, ctx->stream, keys, stride);
    check_launch("segsort");
}

void launch_sample_keys(rfi_ctx* ctx, const signed char* labels, int B, int n, const int* count, unsigned long long seed, unsigned step,
                        unsigned stream0, unsigned long long* keys, int stride) {
    RFI_REQUIRE(B > 0 && n > 0 && n <= 65536 && stride >= n, "sample_keys: at most 65536 candidates per image, stride >= n");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * stride * 9);
    hipLaunchKernelGGL(sample_keys_kernel, dim3((unsigned)std::min<int64_t>(cdiv(stride, kB), 64), B), dim3(kB), 0, ctx->stream, labels, n, count,
                       (unsigned)seed, (unsigned)(seed >> 32), step, stream0, keys, stride);
    check_launch("sample_keys");
}

void launch_rpn_sample_apply(rfi_ctx* ctx, const unsigned long long* keys_sorted, int B, int n, int stride, int batch, int max_pos,
                             const signed char* labels, const float* targets, int L, const int* level_off, signed char* const* level_labels,
                             float* const* level_targets, int* n_sampled) {
    RFI_REQUIRE(B > 0 && n > 0 && n <= 65536 && L >= 1 && L <= 5 && level_off[0] == 0 && level_off[L] == n, "rpn_sample_apply: 1 .. 5 levels covering the anchors");
    RpnLevels lv{};
    lv.L = L;
    for (int l = 0; l <= L; ++l) lv.off[l] = level_off[l];
    for (int l = 0; l < L; ++l) { lv.lab[l] = level_labels[l]; lv.tgt[l] = level_targets[l]; }
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * n * 34);
    hipLaunchKernelGGL(rpn_sample_apply_kernel, dim3(B), dim3(kB), (size_t)((n + 31) / 32) * 4, ctx->stream, keys_sorted, n, stride, batch,
                       max_pos, labels, targets, lv, n_sampled);
    check_launch("rpn_sample_apply");
}

void launch_topk_keys(rfi_ctx* ctx, const float* head, int B, int P, int A, unsigned long long* keys, int stride) {
    RFI_REQUIRE(B > 0 && P > 0 && A > 0 && stride >= P * A, "topk_keys: stride >= anchors per image");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * stride * 12);
    hipLaunchKernelGGL(topk_keys_kernel, dim3((unsigned)std::min<int64_t>(cdiv(stride, kB), 64), B), dim3(kB), 0, ctx->stream, head, P, A, keys, stride);
    check_launch("topk_keys");
}

void launch_topk_decode(rfi_ctx* ctx, const unsigned long long* keys_sorted, int B, int stride, int P, int A, int K, const float* head,
                        const float* anchors, float clip_h, float clip_w, float min_size, float* boxes, float* scores, int* counts, int L,
                        int lvl) {
    RFI_REQUIRE(B > 0 && K > 0 && K <= kB && A % 4 == 0 && lvl >= 0 && lvl < L, "topk_decode: K <= 256, anchors per pixel a multiple of 4");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * K * 60);
    hipLaunchKernelGGL(topk_decode_kernel, dim3(B), dim3(kB), 0, ctx->stream, keys_sorted, stride, P, A, K, head, anchors, clip_h, clip_w,
                       min_size, boxes, scores, counts, L, lvl);
    check_launch("topk_decode");
}

void launch_proposals_select(rfi_ctx* ctx, const float* boxes, const float* scores, const unsigned char* keep, int B, int L, int K,
                             int post_nms, const float* gt, int Gmax, const int* gt_count, int Pmax, float* props, int* pcount) {
    const int LK = L * K, np2 = pow2_at_least(LK);
    RFI_REQUIRE(B > 0 && LK > 0 && np2 <= 8192 && Gmax > 0 && Pmax >= post_nms, "proposals_select: at most 8192 candidates per image");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * LK * 21);
    hipLaunchKernelGGL(proposals_select_kernel, dim3(B), dim3(kB), (size_t)np2 * 8, ctx->stream, boxes, scores, keep, LK, np2, post_nms, gt,
                       Gmax, gt_count, Pmax, props, pcount);
    check_launch("proposals_select");
}

void launch_roi_sample(rfi_ctx* ctx, const signed char* labels, const int* pcount, int B, int Pmax, int batch, int max_pos,
                       unsigned long long seed, unsigned step, unsigned stream0, int* sel, int* nsel, int* npos) {
    const int np2 = pow2_at_least(Pmax);
    RFI_REQUIRE(B > 0 && Pmax > 0 && np2 <= 8192 && batch > 0 && max_pos <= batch, "roi_sample: at most 8192 proposals per image");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * Pmax * 9);
    hipLaunchKernelGGL(roi_sample_kernel, dim3(B), dim3(kB), (size_t)np2 * 8, ctx->stream, labels, pcount, Pmax, np2, batch, max_pos,
                       (unsigned)seed, (unsigned)(seed >> 32), step, stream0, sel, nsel, npos);
    check_launch("roi_sample");
}

void launch_roi_compact(rfi_ctx* ctx, const int* sel, const int* nsel, const int* npos, int B, int batch, int Pmax, const float* props,
                        const int* matched, const float* targets, const int* gt_labels, int Gmax, const int* gbase, float t1, float t2,
                        float t3, float* rois, int* cls, float* tgt, int* gt, int* level, int* img_start, float* rois_fg, float* rois_gt,
                        int* level_fg, int* fg_start, int* counts) {
    RFI_REQUIRE(B > 0 && batch > 0, "roi_compact: empty batch");
    RoiOut o{rois, cls, tgt, gt, level, img_start, rois_fg, rois_gt, level_fg, fg_start, counts};
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)B * batch * 80);
    hipLaunchKernelGGL(roi_compact_kernel, dim3(B), dim3(kB), 0, ctx->stream, sel, nsel, npos, B, batch, Pmax, props, matched, targets,
                       gt_labels, Gmax, gbase, t1, t2, t3, o);
    check_launch("roi_compact");
}

}  // namespace rfi
