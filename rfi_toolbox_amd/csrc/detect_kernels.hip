// Building blocks of the Mask R-CNN path named by BASELINE.json configs[3].  NOT in the
// reference (it has no detector code; torchvision is absent from this image): defined by the published algorithms
// (He et al. 2017, "Mask R-CNN", RoIAlign; Lin et al. 2017, "Feature Pyramid Networks", top-down pathway) and
// checked against oracle/detection_ref.py -- parity unpinned by the reference.  NHWC float32 like every other
// tensor of the path.  Both are HBM-bound gather / elementwise kernels: 16-byte channel-contiguous accesses.
#include <algorithm>

#include "kernels.hpp"

namespace rfi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bilin { int y0, x0, y1, x1; float w00, w01, w10, w11; bool ok; };
// torchvision's roi_align sampling rule: a point outside [-1, H] x [-1, W] contributes nothing; otherwise it is
// clamped to the image and interpolated bilinearly
__device__ __forceinline__ Bilin bilin(float y, float x, int H, int W) {
    Bilin b;
    b.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y0 = (int)y, x0 = (int)x, y1, x1;
    if (y0 >= H - 1) { y1 = y0 = H - 1; y = (float)y0; } else y1 = y0 + 1;
    if (x0 >= W - 1) { x1 = x0 = W - 1; x = (float)x0; } else x1 = x0 + 1;
    const float ly = y - y0, lx = x - x0, hy = 1.0f - ly, hx = 1.0f - lx;
    b.y0 = y0; b.x0 = x0; b.y1 = y1; b.x1 = x1;
    b.w00 = hy * hx; b.w01 = hy * lx; b.w10 = ly * hx; b.w11 = ly * lx;
    return b;
}

struct RoiGeom { float y1, x1, bh, bw; int n; int gh, gw; };
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ rois, int r, float scale, int PH, int PW, int sr, bool aligned) {
    const float* q = rois + (int64_t)r * 5;                       // (batch index, x1, y1, x2, y2) in image coordinates
    const float off = aligned ? 0.5f : 0.0f;
    RoiGeom g;
    g.n = (int)q[0];
    g.x1 = q[1] * scale - off;
    g.y1 = q[2] * scale - off;
    float rw = q[3] * scale - off - g.x1, rh = q[4] * scale - off - g.y1;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    g.bh = rh / PH;
    g.bw = rw / PW;
    g.gh = sr > 0 ? sr : (int)ceilf(rh / PH);
    g.gw = sr > 0 ? sr : (int)ceilf(rw / PW);
    return g;
}

// thread = (roi, ph, pw, 4 channels)
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                           const float* __restrict__ rois, int R, float scale, int PH, int PW,
                                                           int sr, int aligned, float* __restrict__ out) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)R * PH * PW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int r = (int)(t / PH);
        const RoiGeom g = roi_geom(rois, r, scale, PH, PW, sr, aligned != 0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (g.n >= 0 && g.n < N) {
            const float* xb = x + (int64_t)g.n * H * W * C + c;
            for (int iy = 0; iy < g.gh; ++iy) {
                const float yy = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float xx = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
                    const Bilin b = bilin(yy, xx, H, W);
                    if (!b.ok) continue;
                    const f32x4 v00 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y0 * W + b.x0) * C);
                    const f32x4 v01 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y0 * W + b.x1) * C);
                    const f32x4 v10 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y1 * W + b.x0) * C);
                    const f32x4 v11 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y1 * W + b.x1) * C);
                    acc += v00 * b.w00 + v01 * b.w01 + v10 * b.w10 + v11 * b.w11;
                }
            }
            acc = acc * (1.0f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1));
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = acc;
    }
}

// mask-branch targets: RoIAlign (same sampling rules, scale 1) of ONE-channel uint8 instance masks, thresholded at 0.5.
// rois[r] = (instance index into masks[G][H][W], x1, y1, x2, y2); out[r][ph][pw] in {0, 1}.  thread = (roi, ph, pw)
__global__ __launch_bounds__(256) void mask_targets_kernel(const unsigned char* __restrict__ masks, int G, int H, int W,
                                                          const float* __restrict__ rois, int R, int PH, int PW, int sr,
                                                          unsigned char* __restrict__ out) {
    const int64_t total = (int64_t)R * PH * PW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int r = (int)(t / PH);
        const RoiGeom g = roi_geom(rois, r, 1.0f, PH, PW, sr, false);
        float acc = 0.0f;
        if (g.n >= 0 && g.n < G) {
            const unsigned char* mb = masks + (int64_t)g.n * H * W;
            for (int iy = 0; iy < g.gh; ++iy) {
                const float yy = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float xx = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
                    const Bilin b = bilin(yy, xx, H, W);
                    if (!b.ok) continue;
                    acc += (float)mb[(int64_t)b.y0 * W + b.x0] * b.w00 + (float)mb[(int64_t)b.y0 * W + b.x1] * b.w01 +
                           (float)mb[(int64_t)b.y1 * W + b.x0] * b.w10 + (float)mb[(int64_t)b.y1 * W + b.x1] * b.w11;
                }
            }
            acc = acc * (1.0f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1));
        }
        out[i] = acc >= 0.5f ? 1 : 0;
    }
}

// backward: scatter dout / count through the same bilinear weights (float atomics: the sum over overlapping
// RoIs is order dependent in the last bits, like the published implementations)
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int C,
                                                           const float* __restrict__ rois, int R, float scale, int PH, int PW,
                                                           int sr, int aligned, float* __restrict__ dx) {
    const int64_t total = (int64_t)R * PH * PW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int r = (int)(t / PH);
        const RoiGeom g = roi_geom(rois, r, scale, PH, PW, sr, aligned != 0);
        if (g.n < 0 || g.n >= N) continue;
        const float gv = dout[i] / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1);
        float* db = dx + (int64_t)g.n * H * W * C + c;
        if (g.gh <= 2 && g.gw <= 2) {
            // the usual 2 x 2 samples of a bin: where the bin is about a pixel wide or less they share corner pixels, so their
            // contributions are summed per pixel of a 3 x 3 window first (static indices only: everything stays in
            // registers) -- 4 to 9 atomics per bin instead of 16
            Bilin bs[4];
            int ymin = H, xmin = W, ymax = -1, xmax = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int iy = k >> 1, ix = k & 1;
                const float yy = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
                const float xx = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
                bs[k] = bilin(yy, xx, H, W);
                if (iy >= g.gh || ix >= g.gw) bs[k].ok = false;
                if (bs[k].ok) {
                    ymin = min(ymin, bs[k].y0); ymax = max(ymax, bs[k].y1);
                    xmin = min(xmin, bs[k].x0); xmax = max(xmax, bs[k].x1);
                } else {
                    bs[k].w00 = bs[k].w01 = bs[k].w10 = bs[k].w11 = 0.0f;
                }
            }
            if (ymax < 0) continue;                      // no sample inside the map
            if (ymax - ymin <= 2 && xmax - xmin <= 2) {
#pragma unroll
                for (int wy = 0; wy < 3; ++wy)
#pragma unroll
                    for (int wx = 0; wx < 3; ++wx) {
                        const int py = ymin + wy, px = xmin + wx;
                        float v = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v += (bs[k].y0 == py && bs[k].x0 == px) ? bs[k].w00 : 0.0f;
                            v += (bs[k].y0 == py && bs[k].x1 == px && bs[k].x1 != bs[k].x0) ? bs[k].w01 : 0.0f;
                            v += (bs[k].y1 == py && bs[k].x0 == px && bs[k].y1 != bs[k].y0) ? bs[k].w10 : 0.0f;
                            v += (bs[k].y1 == py && bs[k].x1 == px && bs[k].x1 != bs[k].x0 && bs[k].y1 != bs[k].y0) ? bs[k].w11 : 0.0f;
                        }
                        if (v != 0.0f && py <= ymax && px <= xmax) atomicAdd(db + ((int64_t)py * W + px) * C, gv * v);
                    }
                continue;
            }
        }
        for (int iy = 0; iy < g.gh; ++iy) {
            const float yy = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float xx = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
                const Bilin b = bilin(yy, xx, H, W);
                if (!b.ok) continue;
                atomicAdd(db + ((int64_t)b.y0 * W + b.x0) * C, gv * b.w00);
                atomicAdd(db + ((int64_t)b.y0 * W + b.x1) * C, gv * b.w01);
                atomicAdd(db + ((int64_t)b.y1 * W + b.x0) * C, gv * b.w10);
                atomicAdd(db + ((int64_t)b.y1 * W + b.x1) * C, gv * b.w11);
            }
        }
    }
}

// ---- the same gradient by GATHER: one thread per (image, feature pixel, 4 channels) sums what every RoI of its image sends to
// that pixel, in RoI order -- no atomics (the scatter form is bound by float atomics of overlapping RoIs contending in
// L2: 5.5 ms of the Mask R-CNN step), every pixel written exactly once (no memset), bit-reproducible.  Needs the RoIs
// sorted by image index (ascending).  The bilinear weight is separable, w(py, px) = wy * wx with the per-axis rule of
// bilin(); per bin the weights of its samples are summed first, so a (pixel, RoI) pair costs one 16-byte load of dout per
// bin within a pixel's reach.
struct Ax { int p0, p1; float w0, w1; bool ok; };
__device__ __forceinline__ Ax axis_of(float v, int L) {
    Ax a;
    a.ok = !(v < -1.0f || v > (float)L);
    if (v <= 0) v = 0;
    int p0 = (int)v, p1;
    if (p0 >= L - 1) { p1 = p0 = L - 1; v = (float)p0; } else p1 = p0 + 1;
    const float l = v - p0;
    a.p0 = p0; a.p1 = p1; a.w0 = 1.0f - l; a.w1 = l;
    return a;
}
// summed weight that the gh samples of bin `b` along one axis give to pixel `p` (start = roi start, bs = bin size)
__device__ __forceinline__ float bin_weight(float start, float bs, int b, int gcount, int p, int L) {
    float w = 0.0f;
    for (int s = 0; s < gcount; ++s) {
        const float v = start + b * bs + (s + 0.5f) * bs / gcount;          // (the forward kernel's expression, term for term)
        const Ax a = axis_of(v, L);
        if (!a.ok) continue;
        w += (a.p0 == p ? a.w0 : 0.0f) + ((a.p1 == p && a.p1 != a.p0) ? a.w1 : 0.0f);
    }
    return w;
}
constexpr int kGatherRois = 128;      // RoI geometries staged in LDS per pass
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(const float* __restrict__ dout, int N, int H, int W, int C,
                                                                  const float* __restrict__ rois, int R, float scale, int PH, int PW,
                                                                  int sr, int aligned, float* __restrict__ dx) {
    __shared__ RoiGeom sg[kGatherRois];
    __shared__ int s_range[2];
    const int n = blockIdx.y, C4 = C / 4;
    if (threadIdx.x == 0) {           // RoIs of image n: [lo, hi) by binary search on the (sorted) image indices
        int a = 0, b = R;
        while (a < b) { const int m = (a + b) >> 1; if ((int)rois[(int64_t)m * 5] < n) a = m + 1; else b = m; }
        s_range[0] = a;
        b = R;
        while (a < b) { const int m = (a + b) >> 1; if ((int)rois[(int64_t)m * 5] < n + 1) a = m + 1; else b = m; }
        s_range[1] = a;
    }
    __syncthreads();
    const int lo = s_range[0], hi = s_range[1];
    const int64_t items = (int64_t)H * W * C4;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < items; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < items;
        const int c = live ? (int)(i % C4) * 4 : 0;
        const int px = live ? (int)((i / C4) % W) : 0, py = live ? (int)(i / ((int64_t)C4 * W)) : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r0 = lo; r0 < hi; r0 += kGatherRois) {
            const int cnt = min(kGatherRois, hi - r0);
            __syncthreads();
            if ((int)threadIdx.x < cnt) sg[threadIdx.x] = roi_geom(rois, r0 + threadIdx.x, scale, PH, PW, sr, aligned != 0);
            __syncthreads();
            if (!live) continue;
            for (int k = 0; k < cnt; ++k) {
                const RoiGeom g = sg[k];
                // bins whose samples can touch this pixel (conservative; the exact test is in bin_weight).  A border pixel also
                // collects the samples clamped onto it from outside the map: it scans to the RoI's end on that side
                int by0 = (int)floorf(((float)py - 1.0f - g.y1) / g.bh) - 1, by1 = (int)ceilf(((float)py + 1.0f - g.y1) / g.bh) + 1;
                int bx0 = (int)floorf(((float)px - 1.0f - g.x1) / g.bw) - 1, bx1 = (int)ceilf(((float)px + 1.0f - g.x1) / g.bw) + 1;
                if (py == 0) by0 = 0;
                if (py == H - 1) by1 = PH - 1;
                if (px == 0) bx0 = 0;
                if (px == W - 1) bx1 = PW - 1;
                by0 = max(by0, 0); by1 = min(by1, PH - 1); bx0 = max(bx0, 0); bx1 = min(bx1, PW - 1);
                const float inv = 1.0f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1);
                const float* dr = dout + ((int64_t)(r0 + k) * PH * PW) * C + c;
                for (int by = by0; by <= by1; ++by) {
                    const float wy = bin_weight(g.y1, g.bh, by, g.gh, py, H);
                    if (wy == 0.0f) continue;
                    for (int bx = bx0; bx <= bx1; ++bx) {
                        const float wx = bin_weight(g.x1, g.bw, bx, g.gw, px, W);
                        if (wx == 0.0f) continue;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(dr + ((int64_t)by * PW + bx) * C);
                        acc += v * (wy * wx * inv);
                    }
                }
            }
        }
        if (live) *reinterpret_cast<f32x4*>(dx + (((int64_t)n * H + py) * W + px) * C + c) = acc;
    }
}

// ---- multi-level forms: every RoI reads (or sends its gradient to) the pyramid level `level[r]`; level k is an
// [N][H >> k][W >> k][C] map with spatial scale scale0 / 2^k.  One launch for all levels: no sort of the RoIs by level, no
// per-level slices, no host-side counts
struct MlMaps { const float* x[4]; float* dx[4]; int H[4], W[4]; float scale[4]; };
__global__ __launch_bounds__(256) void roi_align_ml_fwd_kernel(MlMaps m, int N, int C, const float* __restrict__ rois,
                                                              const int* __restrict__ level, const int* __restrict__ count, int PH,
                                                              int PW, int sr, float* __restrict__ out) {
    const int C4 = C / 4, R = *count;
    const int64_t total = (int64_t)R * PH * PW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH);
        const int r = (int)(t / PH);
        const int k = level[r];
        const int H = m.H[k], W = m.W[k];
        const RoiGeom g = roi_geom(rois, r, m.scale[k], PH, PW, sr, false);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (g.n >= 0 && g.n < N) {
            const float* xb = m.x[k] + (int64_t)g.n * H * W * C + c;
            for (int iy = 0; iy < g.gh; ++iy) {
                const float yy = g.y1 + ph * g.bh + (iy + 0.5f) * g.bh / g.gh;
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float xx = g.x1 + pw * g.bw + (ix + 0.5f) * g.bw / g.gw;
                    const Bilin b = bilin(yy, xx, H, W);
                    if (!b.ok) continue;
                    const f32x4 v00 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y0 * W + b.x0) * C);
                    const f32x4 v01 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y0 * W + b.x1) * C);
                    const f32x4 v10 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y1 * W + b.x0) * C);
                    const f32x4 v11 = *reinterpret_cast<const f32x4*>(xb + ((int64_t)b.y1 * W + b.x1) * C);
                    acc += v00 * b.w00 + v01 * b.w01 + v10 * b.w10 + v11 * b.w11;
                }
            }
            acc = acc * (1.0f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1));
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = acc;
    }
}
// the gradient by gather (roi_align_bwd_gather_kernel's scheme): blockIdx.z = level, blockIdx.y = image; the RoIs of image n
// are rows [img_start[n], img_start[n + 1]) and those of another level are skipped; the pixel's sum is ADDED to dx (the
// level's gradient map already holds the other terms): one read-modify-write per element by its own thread, no atomics.
// One WAVE per feature pixel, its lanes over the channel groups: which RoIs reach the pixel, which bins and with what
// weights is the same for every channel, so those tests run on wave-uniform values (a RoI elsewhere in the image costs a
// scalar compare, not 64 lanes of float arithmetic: 2.1 -> 0.4 ms per step of the detector)
constexpr int kMlGroups = 4;          // channel groups of 4 per lane: C <= 64 * 4 * kMlGroups = 1024
__global__ __launch_bounds__(256) void roi_align_ml_bwd_kernel(MlMaps m, int C, const float* __restrict__ dout, const float* __restrict__ rois,
                                                              const int* __restrict__ level, const int* __restrict__ img_start, int PH, int PW,
                                                              int sr) {
    __shared__ RoiGeom sg[kGatherRois];
    __shared__ int s_lvl[kGatherRois];
    __shared__ short4 s_reach[kGatherRois];          // pixel rows / columns a RoI's samples can touch: [y lo, y hi, x lo, x hi]
    const int k = blockIdx.z, n = blockIdx.y, C4 = C / 4;
    const int H = m.H[k], W = m.W[k];
    const int lo = img_start[n], hi = img_start[n + 1];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int npix = H * W;
    for (int pbase = blockIdx.x * 4; pbase < npix; pbase += gridDim.x * 4) {     // (uniform trip count for the block: barriers inside)
        const int pix = pbase + wave;
        const bool live = pix < npix;
        const int py = live ? pix / W : 0, px = live ? pix % W : 0;
        f32x4 acc[kMlGroups];
#pragma unroll
        for (int g4 = 0; g4 < kMlGroups; ++g4) acc[g4] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r0 = lo; r0 < hi; r0 += kGatherRois) {
            const int cnt = min(kGatherRois, hi - r0);
            __syncthreads();
            if ((int)threadIdx.x < cnt) {
                const RoiGeom g = roi_geom(rois, r0 + threadIdx.x, m.scale[k], PH, PW, sr, false);
                sg[threadIdx.x] = g;
                s_lvl[threadIdx.x] = level[r0 + threadIdx.x];
                // bilinear samples lie in [start, start + P * bin]; each touches floor and floor + 1 (clamped into the map)
                const float ylo = fmaxf(floorf(g.y1) - 1.0f, -2.0f), yhi = fminf(ceilf(g.y1 + PH * g.bh) + 1.0f, 32000.0f);
                const float xlo = fmaxf(floorf(g.x1) - 1.0f, -2.0f), xhi = fminf(ceilf(g.x1 + PW * g.bw) + 1.0f, 32000.0f);
                s_reach[threadIdx.x] = make_short4((short)ylo, (short)yhi, (short)xlo, (short)xhi);
            }
            __syncthreads();
            if (!live) continue;
            for (int q = 0; q < cnt; ++q) {
                if (s_lvl[q] != k) continue;
                const short4 rc = s_reach[q];
                if (py < rc.x || py > rc.y || px < rc.z || px > rc.w) continue;      // (most RoIs of the image are elsewhere)
                const RoiGeom g = sg[q];
                int by0 = (int)floorf(((float)py - 1.0f - g.y1) / g.bh) - 1, by1 = (int)ceilf(((float)py + 1.0f - g.y1) / g.bh) + 1;
                int bx0 = (int)floorf(((float)px - 1.0f - g.x1) / g.bw) - 1, bx1 = (int)ceilf(((float)px + 1.0f - g.x1) / g.bw) + 1;
                if (py == 0) by0 = 0;
                if (py == H - 1) by1 = PH - 1;
                if (px == 0) bx0 = 0;
                if (px == W - 1) bx1 = PW - 1;
                by0 = max(by0, 0); by1 = min(by1, PH - 1); bx0 = max(bx0, 0); bx1 = min(bx1, PW - 1);
                const float inv = 1.0f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1);
                const float* dr = dout + ((int64_t)(r0 + q) * PH * PW) * C;
                // the bins' summed weights once per (pixel, RoI), one bin per lane (rows in lanes 0 .. 31, columns in 32 .. 63),
                // read back lane by lane below -- every lane evaluating every bin's samples was most of this kernel's time
                const int nby = by1 - by0 + 1, nbx = bx1 - bx0 + 1;
                float wv = 0.0f;
                if (lane < 32) { if (lane < nby) wv = bin_weight(g.y1, g.bh, by0 + lane, g.gh, py, H); }
                else if (lane - 32 < nbx) wv = bin_weight(g.x1, g.bw, bx0 + lane - 32, g.gw, px, W);
                for (int by = by0; by <= by1; ++by) {
                    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), by - by0));
                    if (wy == 0.0f) continue;
                    for (int bx = bx0; bx <= bx1; ++bx) {
                        const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), 32 + bx - bx0));
                        if (wx == 0.0f) continue;
                        const float wgt = wy * wx * inv;
                        const float* dp = dr + ((int64_t)by * PW + bx) * C;
#pragma unroll
                        for (int g4 = 0; g4 < kMlGroups; ++g4) {
                            const int c4 = lane + 64 * g4;
                            if (c4 < C4) acc[g4] += *reinterpret_cast<const f32x4*>(dp + c4 * 4) * wgt;
                        }
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int g4 = 0; g4 < kMlGroups; ++g4) {
                const int c4 = lane + 64 * g4;
                if (c4 < C4) {
                    f32x4* d = reinterpret_cast<f32x4*>(m.dx[k] + (((int64_t)n * H + py) * W + px) * C + c4 * 4);
                    *d = *d + acc[g4];
                }
            }
        }
    }
}

// FPN top-down pathway: out = lateral + nearest-neighbour 2x upsampling of the coarser level
__global__ __launch_bounds__(256) void fpn_merge_fwd_kernel(const float* __restrict__ lateral, const float* __restrict__ top,
                                                           int N, int H, int W, int C, float* __restrict__ out) {
    const int C4 = C / 4, Ht = (H + 1) / 2, Wt = (W + 1) / 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const f32x4 a = *reinterpret_cast<const f32x4*>(lateral + i * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(top + (((int64_t)n * Ht + (y >> 1)) * Wt + (x >> 1)) * C + c);
        *reinterpret_cast<f32x4*>(out + i * 4) = a + b;
    }
}
// backward w.r.t. the coarser level: every coarse pixel gathers its (up to) 2x2 children in fixed order (no atomics);
// the gradient w.r.t. the lateral input is dout itself
__global__ __launch_bounds__(256) void fpn_merge_bwd_top_kernel(const float* __restrict__ dout, int N, int H, int W, int C,
                                                               float* __restrict__ dtop) {
    const int C4 = C / 4, Ht = (H + 1) / 2, Wt = (W + 1) / 2;
    const int64_t total = (int64_t)N * Ht * Wt * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int xt = (int)(t % Wt); t /= Wt;
        const int yt = (int)(t % Ht);
        const int n = (int)(t / Ht);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = 2 * yt + (k >> 1), x = 2 * xt + (k & 1);
            if (y < H && x < W) acc += *reinterpret_cast<const f32x4*>(dout + (((int64_t)n * H + y) * W + x) * C + c);
        }
        *reinterpret_cast<f32x4*>(dtop + i * 4) = acc;
    }
}

int grid_of(int64_t total) {
    int64_t b = cdiv(total, 256);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

void launch_roi_align_fwd(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, const float* rois, int R, float scale,
                          int PH, int PW, int sampling_ratio, bool aligned, float* out) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align: C % 4 == 0 and positive sizes");
    if (R == 0) return;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)R * PH * PW * C * 4 * 5);
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(grid_of((int64_t)R * PH * PW * C / 4)), dim3(256), 0, ctx->stream, x, N, H, W, C,
                       rois, R, scale, PH, PW, sampling_ratio, aligned ? 1 : 0, out);
    check_launch("roi_align_fwd");
}
void launch_roi_align_bwd_sorted(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, const float* rois, int R, float scale,
                                 int PH, int PW, int sr, bool aligned, float* dx) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_align_backward_sorted: C % 4 == 0 and positive sizes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)R * PH * PW * C * 4 * 4 + (double)N * H * W * C * 4);
    const int64_t items = (int64_t)H * W * (C / 4);
    int bx = (int)std::min<int64_t>(cdiv(items, 256), 4096);
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3(bx, N), dim3(256), 0, ctx->stream, dout, N, H, W, C, rois, R, scale, PH, PW,
                       sr, aligned ? 1 : 0, dx);
    check_launch("roi_align_bwd_gather");
}
void launch_mask_targets(rfi_ctx* ctx, const unsigned char* masks, int G, int H, int W, const float* rois, int R, int PH, int PW,
                         int sr, unsigned char* out) {
    RFI_REQUIRE(G > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "mask_targets: positive sizes");
    if (R == 0) return;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)R * PH * PW * (1.0 + 4.0 * (sr > 0 ? sr * sr : 4)));
    hipLaunchKernelGGL(mask_targets_kernel, dim3(grid_of((int64_t)R * PH * PW)), dim3(256), 0, ctx->stream, masks, G, H, W, rois, R,
                       PH, PW, sr, out);
    check_launch("mask_targets");
}
void launch_roi_align_bwd(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, const float* rois, int R, float scale,
                          int PH, int PW, int sampling_ratio, bool aligned, float* dx) {
    RFI_REQUIRE(N > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align: positive sizes");
    RFI_CHECK_HIP(hipMemsetAsync(dx, 0, (size_t)N * H * W * C * sizeof(float), ctx->stream));
    if (R == 0) return;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)R * PH * PW * C * 4 * 5);
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(grid_of((int64_t)R * PH * PW * C)), dim3(256), 0, ctx->stream, dout, N, H, W, C,
                       rois, R, scale, PH, PW, sampling_ratio, aligned ? 1 : 0, dx);
    check_launch("roi_align_bwd");
}
static MlMaps ml_maps(const float* const* x, float* const* dx, int H0, int W0, float scale0) {
    MlMaps m{};
    for (int k = 0; k < 4; ++k) {
        m.x[k] = x ? x[k] : nullptr;
        m.dx[k] = dx ? dx[k] : nullptr;
        m.H[k] = H0 >> k;
        m.W[k] = W0 >> k;
        m.scale[k] = scale0 / (float)(1 << k);
    }
    return m;
}
// rows beyond *count_dev (the RoI count lives on the device) are not written; max_rois sizes the grid
void launch_roi_align_ml_fwd(rfi_ctx* ctx, const float* const* maps, int N, int H0, int W0, int C, float scale0, const float* rois,
                             const int* level, const int* count_dev, int max_rois, int PH, int PW, int sr, float* out) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && (H0 >> 3) > 0 && (W0 >> 3) > 0 && PH > 0 && PW > 0 && max_rois > 0, "roi_align_ml: C % 4 == 0, four levels, positive sizes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)max_rois * PH * PW * C * 4 * 5);
    hipLaunchKernelGGL(roi_align_ml_fwd_kernel, dim3(grid_of((int64_t)max_rois * PH * PW * C / 4)), dim3(256), 0, ctx->stream,
                       ml_maps(maps, nullptr, H0, W0, scale0), N, C, rois, level, count_dev, PH, PW, sr, out);
    check_launch("roi_align_ml_fwd");
}
void launch_roi_align_ml_bwd(rfi_ctx* ctx, float* const* dmaps, int N, int H0, int W0, int C, float scale0, const float* dout,
                             const float* rois, const int* level, const int* img_start, int max_rois, int PH, int PW, int sr) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && (H0 >> 3) > 0 && (W0 >> 3) > 0 && PH > 0 && PW > 0, "roi_align_ml_backward: C % 4 == 0, four levels, positive sizes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)max_rois * PH * PW * C * 4 * 4 + (double)N * H0 * W0 * C * 8 * 1.33);
    RFI_REQUIRE(C <= 64 * 4 * kMlGroups && PH <= 32 && PW <= 32, "roi_align_ml_backward: at most 1024 channels and 32 x 32 bins");
    int bx = (int)std::min<int64_t>(cdiv((int64_t)H0 * W0, 4), 1024);          // four pixels (waves) per workgroup
    hipLaunchKernelGGL(roi_align_ml_bwd_kernel, dim3(bx, N, 4), dim3(256), 0, ctx->stream, ml_maps(nullptr, dmaps, H0, W0, scale0), C, dout,
                       rois, level, img_start, PH, PW, sr);
    check_launch("roi_align_ml_bwd");
}
void launch_fpn_merge_fwd(rfi_ctx* ctx, const float* lateral, const float* top, int N, int H, int W, int C, float* out) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0, "fpn_merge: C % 4 == 0 and positive sizes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * 9);
    hipLaunchKernelGGL(fpn_merge_fwd_kernel, dim3(grid_of((int64_t)N * H * W * C / 4)), dim3(256), 0, ctx->stream, lateral, top, N,
                       H, W, C, out);
    check_launch("fpn_merge_fwd");
}
void launch_fpn_merge_bwd_top(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, float* dtop) {
    RFI_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0, "fpn_merge: C % 4 == 0 and positive sizes");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * 5);
    hipLaunchKernelGGL(fpn_merge_bwd_top_kernel, dim3(grid_of((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4)), dim3(256), 0,
                       ctx->stream, dout, N, H, W, C, dtop);
    check_launch("fpn_merge_bwd_top");
}

}  // namespace rfi
