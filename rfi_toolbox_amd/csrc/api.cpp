// extern "C" surface of librfi_hip.so (declared in include/rfi_hip.h).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <random>

#include "model.hpp"

using namespace rfi;

// ------------------------------------------------------------------------------------ errors
namespace rfi {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace rfi

// ------------------------------------------------------------------------------------ ctx
void rfi_ctx::activate() const { RFI_CHECK_HIP(hipSetDevice(device)); }

void* rfi_ctx::alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    RFI_CHECK_HIP(hipMalloc(&p, bytes));
    allocs[p] = bytes;
    return p;
}
void rfi_ctx::release(void* p) {
    if (!p) return;
    auto it = allocs.find(p);
    RFI_REQUIRE(it != allocs.end(), "rfi_free: pointer was not allocated by this context");
    allocs.erase(it);
    RFI_CHECK_HIP(hipFree(p));
}
void* rfi_ctx::get_scratch(size_t bytes) {
    if (bytes > scratch_bytes) {
        if (scratch) {
            RFI_CHECK_HIP(hipStreamSynchronize(stream));
            release(scratch);
        }
        scratch = alloc(bytes);
        scratch_bytes = bytes;
    }
    return scratch;
}
hipEvent_t rfi_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    RFI_CHECK_HIP(hipEventCreate(&e));
    return e;
}
void rfi_ctx::drain_profile() {
    for (auto& pe : pending) {
        RFI_CHECK_HIP(hipEventSynchronize(pe.b));
        float ms = 0;
        RFI_CHECK_HIP(hipEventElapsedTime(&ms, pe.a, pe.b));
        fam[pe.family].ms += ms;
        launches.push_back({pe.family, (double)ms, pe.flops, pe.bytes, pe.label});
        event_pool.push_back(pe.a);
        event_pool.push_back(pe.b);
    }
    pending.clear();
}

static const char* kFamilyNames[FAM_COUNT] = {"conv_igemm_mfma", "wgrad_igemm_mfma", "conv_direct_valu",
                                              "batchnorm", "elementwise", "slab_reduce", "optimizer",
                                              "preprocess", "metrics", "comm"};

extern "C" {

int rfi_abi_version(void) { return RFI_HIP_ABI_VERSION; }
const char* rfi_last_error(void) { return g_last_error.c_str(); }

int rfi_device_count(int* count) {
    return guarded([&] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) n = 0;
        *count = n;
    });
}

int rfi_ctx_create(int device_id, rfi_ctx** out) {
    return guarded([&] {
        RFI_REQUIRE(out, "rfi_ctx_create: null out pointer");
        int n = 0;
        RFI_CHECK_HIP(hipGetDeviceCount(&n));
        RFI_REQUIRE(device_id >= 0 && device_id < n, "rfi_ctx_create: no such GPU (device " +
                                                         std::to_string(device_id) + " of " + std::to_string(n) + ")");
        auto* c = new rfi_ctx();
        c->device = device_id;
        c->activate();
        RFI_CHECK_HIP(hipGetDeviceProperties(&c->prop, device_id));
        // the main stream carries the dependent chain: highest priority; the side stream fills in
        int pr_least = 0, pr_greatest = 0;
        RFI_CHECK_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
        RFI_CHECK_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr_greatest));
        c->main_stream = c->stream;
        RFI_CHECK_HIP(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, pr_least));
        RFI_CHECK_HIP(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, pr_least));
        RFI_CHECK_HIP(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
        c->side_done.resize(4);
        for (auto& e : c->side_done) RFI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (const char* e = getenv("RFI_NO_OVERLAP")) c->overlap = !(e[0] == '1');
        RFI_CHECK_HIP(hipEventCreate(&c->t0));
        RFI_CHECK_HIP(hipEventCreate(&c->t1));
        RFI_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->pinned), 4096, hipHostMallocDefault));
        c->zero_page = c->alloc(4096);
        RFI_CHECK_HIP(hipMemsetAsync(c->zero_page, 0, 4096, c->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(c->stream));
        *out = c;
    });
}

int rfi_ctx_destroy(rfi_ctx* ctx) {
    return guarded([&] {
        if (!ctx) return;
        ctx->activate();
        hipStreamSynchronize(ctx->stream);
        for (auto& pe : ctx->pending) { hipEventDestroy(pe.a); hipEventDestroy(pe.b); }
        for (auto e : ctx->event_pool) hipEventDestroy(e);
        for (auto& kv : ctx->allocs) hipFree(kv.first);
        if (ctx->pinned) hipHostFree(ctx->pinned);
        if (ctx->readback_ev) (void)hipEventDestroy(ctx->readback_ev);
        hipEventDestroy(ctx->t0);
        hipEventDestroy(ctx->t1);
        hipStreamSynchronize(ctx->side_stream);
        hipEventDestroy(ctx->fork_ev);
        for (auto e : ctx->fork_ring) hipEventDestroy(e);
        for (auto e : ctx->side_done) hipEventDestroy(e);
        hipStreamSynchronize(ctx->comm_stream);
        for (auto e : ctx->bucket_ev) hipEventDestroy(e);
        hipStreamDestroy(ctx->comm_stream);
        hipStreamDestroy(ctx->side_stream);
        hipStreamDestroy(ctx->main_stream);
        delete ctx;
    });
}

int rfi_ctx_set_overlap(rfi_ctx* ctx, int enabled) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->main_stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->side_stream));
        ctx->overlap = enabled != 0;
    });
}
int rfi_ctx_synchronize(rfi_ctx* ctx) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->side_stream) RFI_CHECK_HIP(hipStreamSynchronize(ctx->side_stream));
        if (ctx->comm_stream) RFI_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream));
    });
}
int rfi_ctx_stream(rfi_ctx* ctx, void** s) {
    return guarded([&] { *s = reinterpret_cast<void*>(ctx->stream); });
}
int rfi_ctx_device_name(rfi_ctx* ctx, char* buf, size_t buflen) {
    return guarded([&] {
        std::string s = std::string(ctx->prop.name) + " " + ctx->prop.gcnArchName + " CUs=" +
                        std::to_string(ctx->prop.multiProcessorCount);
        std::snprintf(buf, buflen, "%s", s.c_str());
    });
}

int rfi_malloc(rfi_ctx* ctx, size_t bytes, void** dptr) {
    return guarded([&] {
        ctx->activate();
        *dptr = ctx->alloc(bytes);
    });
}
int rfi_free(rfi_ctx* ctx, void* dptr) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->release(dptr);
    });
}
int rfi_memcpy(rfi_ctx* ctx, void* dst, int dst_mem, const void* src, int src_mem, size_t bytes) {
    return guarded([&] {
        ctx->activate();
        hipMemcpyKind k = (dst_mem == RFI_DEVICE)
                              ? (src_mem == RFI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice)
                              : (src_mem == RFI_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
        RFI_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int rfi_memset(rfi_ctx* ctx, void* dptr, int value, size_t bytes) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    });
}

int rfi_timer_start(rfi_ctx* ctx) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipEventRecord(ctx->t0, ctx->stream));
    });
}
int rfi_timer_stop(rfi_ctx* ctx, float* elapsed_ms) {
    return guarded([&] {
        ctx->activate();
        RFI_CHECK_HIP(hipEventRecord(ctx->t1, ctx->stream));
        RFI_CHECK_HIP(hipEventSynchronize(ctx->t1));
        RFI_CHECK_HIP(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    });
}

int rfi_profile_enable(rfi_ctx* ctx, int on) {
    return guarded([&] {
        ctx->activate();
        if (!on) ctx->drain_profile();
        ctx->profiling = on != 0;
    });
}
int rfi_profile_reset(rfi_ctx* ctx) {
    return guarded([&] {
        ctx->activate();
        ctx->drain_profile();
        for (auto& f : ctx->fam) f = FamilyStat();
        ctx->launches.clear();
    });
}
int rfi_profile_dump(rfi_ctx* ctx, const char* csv_path) {
    return guarded([&] {
        ctx->activate();
        ctx->drain_profile();
        FILE* f = std::fopen(csv_path, "w");
        RFI_REQUIRE(f, std::string("cannot open ") + csv_path);
        std::fprintf(f, "index,family,label,ms,gflop,tflops,mbytes,gbs\n");
        int i = 0;
        for (auto& l : ctx->launches)
            std::fprintf(f, "%d,%s,%s,%.6f,%.4f,%.3f,%.3f,%.1f\n", i++, kFamilyNames[l.family], l.label.c_str(), l.ms,
                         l.flops * 1e-9, l.ms > 0 ? l.flops / (l.ms * 1e-3) * 1e-12 : 0.0, l.bytes * 1e-6,
                         l.ms > 0 ? l.bytes / (l.ms * 1e-3) * 1e-9 : 0.0);
        std::fclose(f);
    });
}
int rfi_profile_family_count(void) { return FAM_COUNT; }
const char* rfi_profile_family_name(int family) {
    return (family >= 0 && family < FAM_COUNT) ? kFamilyNames[family] : "";
}
int rfi_profile_get(rfi_ctx* ctx, int family, int64_t* launches, double* total_ms, double* flops,
                    double* bytes) {
    return guarded([&] {
        RFI_REQUIRE(family >= 0 && family < FAM_COUNT, "rfi_profile_get: bad family");
        ctx->activate();
        ctx->drain_profile();
        const FamilyStat& f = ctx->fam[family];
        if (launches) *launches = f.launches;
        if (total_ms) *total_ms = f.ms;
        if (flops) *flops = f.flops;
        if (bytes) *bytes = f.bytes;
    });
}

// ------------------------------------------------------------------------------------ model
int rfi_unet_create(rfi_ctx* ctx, int in_channels, int out_channels, int init_features, int depth,
                    rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_unet_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->in_ch = in_channels;
        m->out_ch = out_channels;
        m->feat = init_features;
        m->depth = depth;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;   // nothing to free through the dtor path that is not tracked by ctx
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_cnn3_create(rfi_ctx* ctx, int in_channels, int out_channels, int width, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_cnn3_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 1;
        m->in_ch = in_channels;
        m->out_ch = out_channels;
        m->feat = width;
        m->depth = 0;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_unet_resnet_create(rfi_ctx* ctx, int in_channels, int out_channels, int init_features, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_unet_resnet_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 2;
        m->in_ch = in_channels;
        m->out_ch = out_channels;
        m->feat = init_features;
        m->depth = 4;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_mask_head_create(rfi_ctx* ctx, int in_channels, int conv_layers, int out_channels, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_mask_head_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 3;
        m->in_ch = in_channels;
        m->out_ch = out_channels;
        m->feat = in_channels;
        m->depth = conv_layers;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_box_head_create(rfi_ctx* ctx, int in_features, int hidden, int fc_layers, int num_outputs, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_box_head_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 6;
        m->in_ch = in_features;
        m->feat = hidden;
        m->depth = fc_layers;
        m->out_ch = num_outputs;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_resnet50_fpn_create(rfi_ctx* ctx, int in_channels, int base_width, int fpn_channels, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_resnet50_fpn_create: null argument");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 5;
        m->in_ch = in_channels;
        m->feat = base_width;
        m->out_ch = fpn_channels;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_rpn_head_create(rfi_ctx* ctx, int in_channels, int conv_layers, int anchors_per_pixel, rfi_model** out) {
    return guarded([&] {
        RFI_REQUIRE(ctx && out, "rfi_rpn_head_create: null argument");
        RFI_REQUIRE(anchors_per_pixel > 0 && anchors_per_pixel % 4 == 0, "RPNHead: anchors per pixel must be a positive multiple of 4");
        auto* m = new rfi_model();
        m->ctx = ctx;
        m->arch = 4;
        m->in_ch = in_channels;
        m->out_ch = 5 * anchors_per_pixel;
        m->feat = in_channels;
        m->depth = conv_layers;
        try {
            m->build();
        } catch (...) {
            m->ctx = nullptr;
            delete m;
            throw;
        }
        *out = m;
    });
}
int rfi_model_input_grad(rfi_model* m, float* dx, int dx_mem) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 3 || m->arch == 4 || m->arch == 6, "input_grad: only the mask head computes the gradient w.r.t. its input");
        RFI_REQUIRE(m->pN > 0 && dx, "input_grad: no backward pass has run");
        m->ctx->activate();
        const size_t cnt = (size_t)m->pN * m->pH * m->pW * m->in_ch;
        if (dx_mem == RFI_DEVICE) launch_copy_d2d(m->ctx, dx, m->buf(m->mkGx), cnt * sizeof(float));
        else RFI_CHECK_HIP(hipMemcpyAsync(dx, m->buf(m->mkGx), cnt * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
        if (dx_mem != RFI_DEVICE || getenv("RFI_SYNC_ALWAYS")) RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));   // (sync_if_host below)
    });
}
int rfi_model_destroy(rfi_model* m) {
    return guarded([&] {
        if (!m) return;
        m->ctx->activate();
        RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
        delete m;
    });
}

namespace {

// reference layout <-> library layout for one entry; host staging vectors
// cin_p >= cin: library rows are zero-padded to cin_p input channels
void to_lib_conv(const float* oihw, int cout, int cin, int R, std::vector<float>& out, int cin_p = -1) {
    if (cin_p < 0) cin_p = cin;
    out.assign((size_t)R * R * cout * cin_p, 0.0f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < R * R; ++t)
                out[((size_t)t * cout + co) * cin_p + ci] = oihw[((size_t)co * cin + ci) * R * R + t];
}
void from_lib_conv(const float* lib, int cout, int cin, int R, float* oihw, int cin_p = -1) {
    if (cin_p < 0) cin_p = cin;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < R * R; ++t)
                oihw[((size_t)co * cin + ci) * R * R + t] = lib[((size_t)t * cout + co) * cin_p + ci];
}
// ConvTranspose2d weight is [cin][cout][2][2]
void to_lib_convt(const float* iohw, int cin, int cout, std::vector<float>& out) {
    out.resize((size_t)4 * cout * cin);
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
            for (int t = 0; t < 4; ++t)
                out[((size_t)t * cout + co) * cin + ci] = iohw[((size_t)ci * cout + co) * 4 + t];
}
void from_lib_convt(const float* lib, int cin, int cout, float* iohw) {
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
            for (int t = 0; t < 4; ++t)
                iohw[((size_t)ci * cout + co) * 4 + t] = lib[((size_t)t * cout + co) * cin + ci];
}

// where a float entry lives inside a flat buffer (params / grads / adam m / adam v)
size_t flat_offset(const rfi_model* m, const Entry& e) {
    switch (e.kind) {
        case 0: case 7: return m->convs[e.layer].w_off;
        case 1: return m->ups[e.layer].w_off;
        case 6: return m->head_w_off;
        case 2:
            switch (e.which) {
                case 0: return m->convs[e.layer].b_off;
                case 1: return m->convs[e.layer].g_off;
                case 2: return m->convs[e.layer].be_off;
                case 3: return m->ups[e.layer].b_off;
                default: return m->head_b_off;
            }
        default: throw Error("entry " + e.name + " is not a parameter");
    }
}

void upload(rfi_model* m, float* dst, const float* src, size_t n) {
    RFI_CHECK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
    RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
}
void download(rfi_model* m, float* dst, const float* src, size_t n) {
    RFI_CHECK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
    RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
}

const Entry& find_entry(rfi_model* m, const char* name) {
    RFI_REQUIRE(name, "null entry name");
    auto it = m->entry_index.find(name);
    RFI_REQUIRE(it != m->entry_index.end(), std::string("unexpected key in state_dict: ") + name);
    return m->entries[it->second];
}

void store_from_flat(rfi_model* m, const float* flat, const Entry& e, void* host, size_t bytes) {
    RFI_REQUIRE(bytes == (size_t)e.numel() * sizeof(float),
                "size mismatch for " + e.name + ": expected " + std::to_string(e.numel() * 4) + " bytes, got " +
                    std::to_string(bytes));
    const size_t off = flat_offset(m, e);
    const int cin_p = e.kind == 0 ? m->convs[e.layer].cin_p : 0;
    std::vector<float> tmp(e.kind == 0 ? (size_t)e.dims[2] * e.dims[3] * e.dims[0] * cin_p : (size_t)e.numel());
    download(m, tmp.data(), flat + off, tmp.size());
    float* out = static_cast<float*>(host);
    if (e.kind == 0) from_lib_conv(tmp.data(), (int)e.dims[0], (int)e.dims[1], (int)e.dims[2], out, cin_p);
    else if (e.kind == 1) from_lib_convt(tmp.data(), (int)e.dims[0], (int)e.dims[1], out);
    else std::memcpy(out, tmp.data(), bytes);
}

}  // namespace

int rfi_model_init(rfi_model* m, uint64_t seed) {
    return guarded([&] {
        m->ctx->activate();
        std::mt19937_64 rng(seed);
        std::vector<float> flat(m->n_flat, 0.0f);
        auto uni = [&](float bound) {
            return (float)((std::generate_canonical<double, 53>(rng) * 2.0 - 1.0) * bound);
        };
        float last_bound = 0;
        for (const Entry& e : m->entries) {
            if (e.kind == 0 || e.kind == 1 || e.kind == 6 || e.kind == 7) {
                // kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)), fan_in = dims[1]*kh*kw
                const double fan_in = (double)e.dims[1] * e.dims[2] * e.dims[3];
                last_bound = (float)(1.0 / std::sqrt(fan_in));
                float* w = flat.data() + flat_offset(m, e);
                if (e.kind == 0) {                      // [tap][cout][cin_p], padded channels stay 0
                    const int cin_p = m->convs[e.layer].cin_p, cin = (int)e.dims[1];
                    for (int64_t r = 0; r < e.dims[2] * e.dims[3] * e.dims[0]; ++r)
                        for (int ci = 0; ci < cin; ++ci) w[r * cin_p + ci] = uni(last_bound);
                } else {
                    for (int64_t i = 0; i < e.numel(); ++i) w[i] = uni(last_bound);   // layout-agnostic iid
                }
            } else if (e.kind == 2) {
                float* v = flat.data() + flat_offset(m, e);
                for (int64_t i = 0; i < e.numel(); ++i)
                    v[i] = (e.which == 1) ? 1.0f : (e.which == 2 ? 0.0f : uni(last_bound));
            }
        }
        upload(m, m->params, flat.data(), m->n_flat);
        m->reset_channel_state();
        RFI_CHECK_HIP(hipMemsetAsync(m->adam_m, 0, m->n_flat * sizeof(float), m->ctx->stream));
        RFI_CHECK_HIP(hipMemsetAsync(m->adam_v, 0, m->n_flat * sizeof(float), m->ctx->stream));
        m->adam_step = 0;
        m->wd_dirty = true;
        m->x3_fresh = false;
    });
}

int rfi_model_entry_count(rfi_model* m, int* n) {
    return guarded([&] { *n = (int)m->entries.size(); });
}
int rfi_model_entry_info(rfi_model* m, int index, const char** name, int* ndim, int64_t dims[4],
                         int* is_int64, int* is_parameter) {
    return guarded([&] {
        RFI_REQUIRE(index >= 0 && index < (int)m->entries.size(), "entry index out of range");
        const Entry& e = m->entries[index];
        if (name) *name = e.name.c_str();
        if (ndim) *ndim = e.ndim;
        if (dims)
            for (int i = 0; i < 4; ++i) dims[i] = i < e.ndim ? e.dims[i] : 1;
        if (is_int64) *is_int64 = e.kind == 5;
        if (is_parameter) *is_parameter = (e.kind == 0 || e.kind == 1 || e.kind == 2 || e.kind == 6 || e.kind == 7);
    });
}
int rfi_model_param_count(rfi_model* m, int64_t* n) {
    return guarded([&] { *n = m->n_params; });
}

int rfi_model_load_entry(rfi_model* m, const char* name, const void* host, size_t bytes) {
    return guarded([&] {
        m->ctx->activate();
        const Entry& e = find_entry(m, name);
        if (e.kind == 5) {
            RFI_REQUIRE(bytes == sizeof(int64_t), "size mismatch for " + e.name);
            m->convs[e.layer].nbt = *static_cast<const int64_t*>(host);
            return;
        }
        RFI_REQUIRE(bytes == (size_t)e.numel() * sizeof(float),
                    "size mismatch for " + e.name + ": expected " + std::to_string(e.numel() * 4) +
                        " bytes, got " + std::to_string(bytes));
        const float* src = static_cast<const float*>(host);
        if (e.kind == 3 || e.kind == 4 || e.kind == 8 || e.kind == 9) {     // (8 / 9: frozen BatchNorm weight / bias)
            ConvBN& c = m->convs[e.layer];
            float* dst = e.kind == 3 ? c.running_mean() : (e.kind == 4 ? c.running_var() : (e.kind == 8 ? c.mean() : c.invstd()));
            upload(m, dst, src, (size_t)c.cout);
            m->frozen_dirty = true;
            return;
        }
        std::vector<float> tmp;
        if (e.kind == 0) to_lib_conv(src, (int)e.dims[0], (int)e.dims[1], (int)e.dims[2], tmp, m->convs[e.layer].cin_p);
        else if (e.kind == 1) to_lib_convt(src, (int)e.dims[0], (int)e.dims[1], tmp);
        else tmp.assign(src, src + e.numel());
        upload(m, m->params + flat_offset(m, e), tmp.data(), tmp.size());
        m->wd_dirty = true;
        m->x3_fresh = false;
    });
}

int rfi_model_store_entry(rfi_model* m, const char* name, void* host, size_t bytes) {
    return guarded([&] {
        m->ctx->activate();
        const Entry& e = find_entry(m, name);
        if (e.kind == 5) {
            RFI_REQUIRE(bytes == sizeof(int64_t), "size mismatch for " + e.name);
            *static_cast<int64_t*>(host) = m->convs[e.layer].nbt;
            return;
        }
        if (e.kind == 3 || e.kind == 4 || e.kind == 8 || e.kind == 9) {
            RFI_REQUIRE(bytes == (size_t)e.numel() * sizeof(float), "size mismatch for " + e.name);
            ConvBN& c = m->convs[e.layer];
            const float* srcd = e.kind == 3 ? c.running_mean() : (e.kind == 4 ? c.running_var() : (e.kind == 8 ? c.mean() : c.invstd()));
            download(m, static_cast<float*>(host), srcd, (size_t)c.cout);
            return;
        }
        store_from_flat(m, m->params, e, host, bytes);
    });
}

int rfi_model_store_grad(rfi_model* m, const char* name, void* host, size_t bytes) {
    return guarded([&] {
        m->ctx->activate();
        m->join_pending_side();
        store_from_flat(m, m->grads, find_entry(m, name), host, bytes);
    });
}
int rfi_model_store_adam(rfi_model* m, const char* name, void* host_m, void* host_v, size_t bytes,
                         int64_t* step) {
    return guarded([&] {
        m->ctx->activate();
        const Entry& e = find_entry(m, name);
        if (host_m) store_from_flat(m, m->adam_m, e, host_m, bytes);
        if (host_v) store_from_flat(m, m->adam_v, e, host_v, bytes);
        if (step) *step = m->adam_step;
    });
}

int rfi_model_load_adam(rfi_model* m, const char* name, const void* host_m, const void* host_v,
                        size_t bytes) {
    return guarded([&] {
        m->ctx->activate();
        const Entry& e = find_entry(m, name);
        RFI_REQUIRE(bytes == (size_t)e.numel() * sizeof(float), "size mismatch for " + e.name);
        const size_t off = flat_offset(m, e);
        for (int which = 0; which < 2; ++which) {
            const float* src = static_cast<const float*>(which ? host_v : host_m);
            if (!src) continue;
            std::vector<float> tmp;
            if (e.kind == 0) to_lib_conv(src, (int)e.dims[0], (int)e.dims[1], (int)e.dims[2], tmp, m->convs[e.layer].cin_p);
            else if (e.kind == 1) to_lib_convt(src, (int)e.dims[0], (int)e.dims[1], tmp);
            else tmp.assign(src, src + e.numel());
            upload(m, (which ? m->adam_v : m->adam_m) + off, tmp.data(), tmp.size());
        }
    });
}
int rfi_model_set_adam_step(rfi_model* m, int64_t step) {
    return guarded([&] {
        RFI_REQUIRE(step >= 0, "Adam step must be >= 0");
        m->adam_step = step;
    });
}

int rfi_model_set_training(rfi_model* m, int training) {
    return guarded([&] { m->training = training != 0; });
}
int rfi_model_set_activation(rfi_model* m, float negative_slope) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 0, "set_activation: U-Net models only");
        RFI_REQUIRE(negative_slope >= 0.0f && negative_slope < 1.0f, "set_activation: negative_slope must be in [0, 1)");
        m->act_slope = negative_slope;
    });
}
int rfi_model_set_compute_dtype(rfi_model* m, int dtype) {
    return guarded([&] {
        RFI_REQUIRE(dtype >= 0 && dtype <= 4,
                    "set_compute_dtype: 0 native float32 MFMA, 1 bfloat16 (bf16 activations in HBM, plane kernels), "
                    "2 float32 by 3 x bfloat16 splitting in registers (default), 3 the same arithmetic on pre-split "
                    "plane tensors, 4 bfloat16 operands rounded in registers (float32 storage)");
        m->compute_bf16 = dtype == 1 || dtype == 4;
        m->compute_x3 = dtype == 2 || dtype == 3;
        if (m->arch == 0) m->set_planes(dtype == 1 ? 1 : (dtype == 3 ? 3 : 0));
        // the ResNet-encoder U-Net has the bfloat16 flow only, for widths in whole 16-channel chunks (else: dtype 4's kernels)
        if (m->arch == 2) m->set_planes(dtype == 1 && m->feat % 16 == 0 ? 1 : 0);
    });
}
int rfi_model_set_loss(rfi_model* m, int kind, float alpha, float gamma) {
    return guarded([&] {
        RFI_REQUIRE(kind == 0 || kind == 1, "set_loss: 0 (BCE-with-logits + dice) or 1 (sigmoid focal loss)");
        RFI_REQUIRE(kind == 0 || (gamma >= 0.0f && alpha <= 1.0f), "set_loss: focal needs gamma >= 0 and alpha <= 1");
        m->loss_kind = kind;
        m->focal_alpha = alpha;
        m->focal_gamma = gamma;
    });
}
int rfi_model_set_head_sigmoid(rfi_model* m, int enabled) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 0, "set_head_sigmoid: U-Net models only");
        m->head_sigmoid = enabled != 0;
    });
}

namespace {

// a copy on the model's stream: between two device buffers as a kernel (launch_copy_d2d), else hipMemcpyAsync
void copy_on_stream(rfi_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (kind == hipMemcpyDeviceToDevice) launch_copy_d2d(ctx, dst, src, bytes);
    else RFI_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
}
const float* stage_input(rfi_model* m, const float* x, int x_mem, int n, int h, int w, bool nchw) {
    const size_t cnt = (size_t)n * h * w * m->in_ch;
    const float* dev = x;
    if (x_mem == RFI_HOST) {
        float* st = m->buf(nchw ? m->x_stage2 : m->x_stage);
        RFI_CHECK_HIP(hipMemcpyAsync(st, x, cnt * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
        dev = st;
    }
    if (nchw) {
        launch_nchw_to_nhwc(m->ctx, dev, n, m->in_ch, h, w, m->buf(m->x_stage));
        dev = m->buf(m->x_stage);
    }
    return dev;
}
const uint8_t* stage_labels(rfi_model* m, const uint8_t* y, int y_mem, int n, int h, int w) {
    if (y_mem == RFI_DEVICE) return y;
    uint8_t* st = reinterpret_cast<uint8_t*>(m->buf(m->lab_stage));
    RFI_CHECK_HIP(hipMemcpyAsync(st, y, (size_t)n * h * w * m->out_scale * m->out_scale, hipMemcpyHostToDevice, m->ctx->stream));
    return st;
}
// Entry points whose every tensor argument is a DEVICE pointer return when their work is enqueued (the caller's next
// call that hands data to the host -- rfi_memcpy, a loss scalar -- synchronises the stream); with a host pointer they
// return when the data is there.  RFI_SYNC_ALWAYS=1: synchronise always (round 2's behaviour).
void sync_if_host(rfi_model* m, int mem_a, int mem_b = RFI_DEVICE) {
    static const bool always = getenv("RFI_SYNC_ALWAYS") != nullptr;
    if (always || mem_a != RFI_DEVICE || mem_b != RFI_DEVICE) RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
}
void emit_logits(rfi_model* m, float* out, int out_mem, int n, int h, int w, bool nchw) {
    h *= m->out_scale; w *= m->out_scale;         // (the mask head's output map is twice its input map)
    const size_t cnt = (size_t)n * h * w * m->out_ch;
    const float* src = m->buf(m->head_sigmoid ? m->probs : m->logits);   // the model's OUTPUT
    if (nchw && m->out_ch > 1) {
        launch_nhwc_to_nchw(m->ctx, src, n, m->out_ch, h, w, m->buf(m->out_stage));
        src = m->buf(m->out_stage);
    }
    copy_on_stream(m->ctx, out, src, cnt * sizeof(float), out_mem == RFI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
    sync_if_host(m, out_mem);
}
float read_scalar(rfi_model* m, const float* dev) {
    RFI_CHECK_HIP(hipMemcpyAsync(m->ctx->pinned, dev, sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
    RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    return m->ctx->pinned[0];
}
void check_hyper(const rfi_hyper* hp) {
    RFI_REQUIRE(hp, "null hyper-parameters");
    RFI_REQUIRE(hp->lr >= 0 && hp->eps >= 0 && hp->weight_decay >= 0, "Adam: lr, eps, weight_decay must be >= 0");
    RFI_REQUIRE(hp->beta1 >= 0 && hp->beta1 < 1 && hp->beta2 >= 0 && hp->beta2 < 1, "Adam: betas must be in [0,1)");
}


// The optimiser half of a full training step, shared by rfi_train_step and rfi_train_step_async: when the
// context holds a communicator of more than one rank the flat gradient buffer is summed over the ranks
// (RCCL, on this context's stream) and clip + Adam see the MEAN gradient (grad_scale = 1 / world).
void backward_with_exchange(rfi_model* m, const float* x, const uint8_t* y, int n, int h, int w) {
    struct Flag { rfi_model* m; ~Flag() { m->exchange_in_backward = false; } } flag{m};
    m->exchange_in_backward = true;               // the buckets leave as the backward pass finishes them
    m->backward(x, y, n, h, w);
}
void exchange_and_apply(rfi_model* m, const rfi_hyper& hp) {
    if (m->ctx->exchange_active()) {
        // the buckets were all-reduced on the communication stream while the backward pass ran (rfi_model::
        // bucket_ready); exchange_join makes the main stream wait for the last of them
        m->exchange_join();
        m->apply(hp, 1.0f / (float)m->ctx->exchange_world());
    } else {
        m->apply(hp, 1.0f);
    }
}

}  // namespace

int rfi_model_forward_nhwc(rfi_model* m, const float* x, int x_mem, int n, int h, int w, float* logits,
                           int logits_mem) {
    return guarded([&] {
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        m->forward(xd, n, h, w, m->training);
        emit_logits(m, logits, logits_mem, n, h, w, false);
    });
}
int rfi_model_forward_nchw(rfi_model* m, const float* x, int x_mem, int n, int h, int w, float* logits,
                           int logits_mem) {
    return guarded([&] {
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, true);
        m->forward(xd, n, h, w, m->training);
        emit_logits(m, logits, logits_mem, n, h, w, true);
    });
}

int rfi_train_forward_backward(rfi_model* m, const float* x, int x_mem, const uint8_t* labels,
                               int labels_mem, int n, int h, int w, float* loss_out) {
    return guarded([&] {
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        const uint8_t* yd = stage_labels(m, labels, labels_mem, n, h, w);
        m->forward(xd, n, h, w, true);
        m->loss_forward(yd, n, h, w);
        m->backward(xd, yd, n, h, w);
        if (loss_out) *loss_out = m->last_loss = read_scalar(m, m->d_scalars);
    });
}
int rfi_train_apply(rfi_model* m, const rfi_hyper* hp, float grad_scale, float* grad_norm_out) {
    return guarded([&] {
        check_hyper(hp);
        m->ctx->activate();
        RFI_REQUIRE(m->pN > 0, "rfi_train_apply: no gradients (call rfi_train_forward_backward first)");
        m->apply(*hp, grad_scale);
        if (grad_norm_out) *grad_norm_out = m->last_norm = read_scalar(m, m->d_scalars + 1);
    });
}
int rfi_train_step(rfi_model* m, const float* x, int x_mem, const uint8_t* labels, int labels_mem,
                   int n, int h, int w, const rfi_hyper* hp, float* loss_out) {
    return guarded([&] {
        check_hyper(hp);
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        const uint8_t* yd = stage_labels(m, labels, labels_mem, n, h, w);
        m->forward(xd, n, h, w, true);
        m->loss_forward(yd, n, h, w);
        backward_with_exchange(m, xd, yd, n, h, w);
        exchange_and_apply(m, *hp);
        RFI_CHECK_HIP(hipMemcpyAsync(m->ctx->pinned, m->d_scalars, 2 * sizeof(float), hipMemcpyDeviceToHost,
                                     m->ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
        m->last_loss = m->ctx->pinned[0];
        m->last_norm = m->ctx->pinned[1];
        if (loss_out) *loss_out = m->last_loss;
    });
}
int rfi_train_step_async(rfi_model* m, const float* x_dev, const uint8_t* labels_dev, int n, int h,
                         int w, const rfi_hyper* hp) {
    return guarded([&] {
        check_hyper(hp);
        m->ctx->activate();
        m->prepare(n, h, w);
        m->forward(x_dev, n, h, w, true);
        m->loss_forward(labels_dev, n, h, w);
        backward_with_exchange(m, x_dev, labels_dev, n, h, w);
        exchange_and_apply(m, *hp);
    });
}
int rfi_model_backward_dlogits(rfi_model* m, const float* x, int x_mem, const float* dlogits, int dlogits_mem, int n, int h,
                               int w) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 3 || m->arch == 4 || m->arch == 6, "backward_dlogits: mask / RPN / box heads only");
        RFI_REQUIRE(m->pN == n && m->pH == h && m->pW == w, "backward_dlogits: run the forward pass on this input first");
        m->ctx->activate();
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        const size_t cnt = (size_t)n * h * w * m->out_scale * m->out_scale * m->out_ch;
        copy_on_stream(m->ctx, m->buf(m->dlogits), dlogits, cnt * sizeof(float),
                       dlogits_mem == RFI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
        struct Flag { rfi_model* m; ~Flag() { m->ext_dlogits = false; } } flag{m};
        m->ext_dlogits = true;
        m->backward(xd, nullptr, n, h, w);
        sync_if_host(m, x_mem, dlogits_mem);
    });
}
int rfi_backbone_forward(rfi_model* m, const float* x, int x_mem, int n, int h, int w, float* const feats[5], int feats_mem) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 5 && feats, "backbone_forward: ResNet-50-FPN models only");
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        m->forward(xd, n, h, w, false);
        for (int i = 0; i < 5; ++i) {
            if (!feats[i]) continue;
            const int lvl = i + 2;
            const size_t cnt = (size_t)n * (h >> lvl) * (w >> lvl) * m->out_ch;
            copy_on_stream(m->ctx, feats[i], m->buf(i < 4 ? m->fP[i] : m->fP6), cnt * sizeof(float),
                           feats_mem == RFI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
        }
        sync_if_host(m, x_mem, feats_mem);
    });
}
int rfi_backbone_backward(rfi_model* m, const float* x, int x_mem, int n, int h, int w, const float* const dfeats[5], int dfeats_mem) {
    return guarded([&] {
        RFI_REQUIRE(m->arch == 5 && dfeats, "backbone_backward: ResNet-50-FPN models only");
        RFI_REQUIRE(m->pN == n && m->pH == h && m->pW == w, "backbone_backward: run the forward pass on this input first");
        m->ctx->activate();
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        for (int i = 0; i < 5; ++i) {
            const int lvl = i + 2;
            const size_t cnt = (size_t)n * (h >> lvl) * (w >> lvl) * m->out_ch;
            float* dst = m->buf(i < 4 ? m->fdP[i] : m->fdP6);
            if (dfeats[i])
                copy_on_stream(m->ctx, dst, dfeats[i], cnt * sizeof(float),
                               dfeats_mem == RFI_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
            else
                RFI_CHECK_HIP(hipMemsetAsync(dst, 0, cnt * sizeof(float), m->ctx->stream));
        }
        m->backward(xd, nullptr, n, h, w);
        sync_if_host(m, x_mem, dfeats_mem);
    });
}
int rfi_model_last_loss(rfi_model* m, float* loss_out, float* grad_norm_out) {
    return guarded([&] {
        m->ctx->activate();
        RFI_CHECK_HIP(hipMemcpyAsync(m->ctx->pinned, m->d_scalars, 2 * sizeof(float), hipMemcpyDeviceToHost,
                                     m->ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
        if (loss_out) *loss_out = m->ctx->pinned[0];
        if (grad_norm_out) *grad_norm_out = m->ctx->pinned[1];
    });
}
int rfi_model_loss(rfi_model* m, const float* x, int x_mem, const uint8_t* labels, int labels_mem, int n,
                   int h, int w, float* loss_out) {
    return guarded([&] {
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        const uint8_t* yd = stage_labels(m, labels, labels_mem, n, h, w);
        m->forward(xd, n, h, w, m->training);
        m->loss_forward(yd, n, h, w);
        if (loss_out) *loss_out = read_scalar(m, m->d_scalars);
    });
}

int rfi_model_eval_batch(rfi_model* m, const float* x, int x_mem, const uint8_t* labels, int labels_mem, int n,
                         int h, int w, float threshold, int64_t* tp, int64_t* fp, int64_t* fn) {
    return guarded([&] {
        RFI_REQUIRE(m->out_ch == 1, "eval_batch: defined for out_channels == 1");
        m->ctx->activate();
        m->prepare(n, h, w);
        const float* xd = stage_input(m, x, x_mem, n, h, w, false);
        const uint8_t* yd = stage_labels(m, labels, labels_mem, n, h, w);
        m->forward(xd, n, h, w, m->training);
        const int64_t cnt = (int64_t)n * h * w * m->out_scale * m->out_scale;
        uint8_t* mask = reinterpret_cast<uint8_t*>(m->buf(m->out_stage));      // cnt bytes fit (cnt floats)
        // evaluate_model.py:44-47 thresholds sigmoid(model output), whatever the model returns
        launch_threshold(m->ctx, m->buf(m->head_sigmoid ? m->probs : m->logits), cnt, threshold, mask);
        auto* d3 = reinterpret_cast<unsigned long long*>(m->d_sums + 5);      // 3 spare 64-bit words
        launch_confusion(m->ctx, mask, RFI_U8, yd, RFI_U8, cnt, d3);
        unsigned long long h3[3];
        RFI_CHECK_HIP(hipMemcpyAsync(h3, d3, sizeof(h3), hipMemcpyDeviceToHost, m->ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
        *tp = (int64_t)h3[0]; *fp = (int64_t)h3[1]; *fn = (int64_t)h3[2];
    });
}

int rfi_model_grad_accumulate(rfi_model* m, int phase) {
    return guarded([&] {
        RFI_REQUIRE(phase >= 0 && phase <= 2, "grad_accumulate: phase 0 (begin), 1 (add the current gradients), 2 (end: sum -> gradients)");
        m->ctx->activate();
        m->join_pending_side();
        const size_t bytes = m->n_flat * sizeof(float);
        if (!m->grad_acc) m->grad_acc = static_cast<float*>(m->ctx->alloc(bytes));
        if (phase == 0) RFI_CHECK_HIP(hipMemsetAsync(m->grad_acc, 0, bytes, m->ctx->stream));
        else if (phase == 1) launch_add_inplace(m->ctx, m->grad_acc, m->grads, (int64_t)m->n_flat);
        else launch_copy_d2d(m->ctx, m->grads, m->grad_acc, bytes);
    });
}
int rfi_model_grad_buffer(rfi_model* m, float** dptr, int64_t* n_floats) {
    return guarded([&] {
        m->ctx->activate();
        m->join_pending_side();                   // (the caller is about to read or reduce the gradients)
        *dptr = m->grads;
        *n_floats = (int64_t)m->n_flat;
    });
}
int rfi_model_param_buffer(rfi_model* m, float** dptr, int64_t* n_floats) {
    return guarded([&] {
        *dptr = m->params;
        *n_floats = (int64_t)m->n_flat;
    });
}

int rfi_model_debug_tensor(rfi_model* m, const char* name, float* host, size_t host_floats,
                           int64_t* n_floats) {
    return guarded([&] {
        m->ctx->activate();
        RFI_REQUIRE(name && m->pN > 0, "debug_tensor: no prepared shape");
        std::string s(name), base = s;
        int idx = -1;
        auto dot = s.find('.');
        if (dot != std::string::npos) {
            base = s.substr(0, dot);
            idx = std::stoi(s.substr(dot + 1));
        }
        const float* src = nullptr;
        size_t n = 0;
        const int D = m->depth;
        RFI_REQUIRE(m->arch == 0 || base == "logits" || base == "dlogits" || base == "chan" ||
                        (m->arch == 2 && base != "encY1" && base != "encY2" && base != "pool" && base != "dpool"),
                    "debug_tensor: this model exposes only logits / dlogits / chan (and the decoder tensors of the ResNet-encoder U-Net)");
        RFI_REQUIRE(!(m->y16_flow && m->planesP == 1 && (base == "encY1" || base == "encY2" || base == "decY1" || base == "bottY1" ||
                                                        (base == "decY2" && idx == 1))),
                    "debug_tensor: this conv output is stored as bfloat16 in the bfloat16 compute mode");
        RFI_REQUIRE(!(m->g16_flow && m->planesP == 1 && (base == "gB" || base == "dpool" || base == "gBottB" || base == "gA")),
                    "debug_tensor: this gradient tensor is stored as bfloat16 in the bfloat16 compute mode");
        RFI_REQUIRE(!(m->convt_planes && m->planesP == 1 && (base == "decY2" || base == "bottY2" || base == "gBottA")),
                    "debug_tensor: with the transposed convs on the plane kernels this tensor is stored as bfloat16");
        auto level = [&](const std::vector<int>& v, size_t chmul) {
            RFI_REQUIRE(idx >= 1 && idx <= D, "debug_tensor: level out of range");
            const size_t M = (size_t)m->pN * (m->pH >> (idx - 1)) * (m->pW >> (idx - 1));
            src = m->buf(v[idx]);
            n = M * ((size_t)m->feat << (idx - 1)) * chmul;
        };
        const size_t Mb = (size_t)m->pN * (m->pH >> D) * (m->pW >> D), Cb = (size_t)m->feat << D;
        const size_t M1 = (size_t)m->pN * m->pH * m->pW * m->out_scale * m->out_scale;     // pixels of the OUTPUT map
        if (base == "encY1") level(m->encY1, 1);
        else if (base == "encY2") level(m->encY2, 1);
        else if (base == "decY1") level(m->decY1, 1);
        else if (base == "decY2") level(m->decY2, 1);
        else if (base == "gA") level(!m->planesP && m->arch == 0 ? m->gAe : m->gA, 1);     // (as left by the encoder phase)
        else if (base == "gB") level(!m->planesP && m->arch == 0 ? m->gBe : m->gB, 1);
        else if (base == "concat") level(m->concat, 2);
        else if (base == "dconcat") {
            level(m->dconcat, 2);
            if (m->convt_planes && m->planesP == 1 && host) {      // stored as bfloat16: its values as float32 (tools/race_probe.py watches it)
                const PlaneBuf& g = m->pl[m->g16cat[idx]];
                launch_planes_to_f32(m->ctx, g.p, g.pstride, (int64_t)(n / ((size_t)2 * (m->feat << (idx - 1)))), 2 * (m->feat << (idx - 1)), 1,
                                     m->buf(m->dconcat[idx]), 2 * (m->feat << (idx - 1)));
            }
        }
        else if (base == "pool") { level(m->pool, 1); n /= 4; }
        else if (base == "dpool") { level(m->dpool, 1); n /= 4; }
        else if (base == "bottY1") { src = m->buf(m->bottY1); n = Mb * Cb; }
        else if (base == "bottY2") { src = m->buf(m->bottY2); n = Mb * Cb; }
        else if (base == "gBottA") { src = m->buf(m->gBottA); n = Mb * Cb; }
        else if (base == "gBottB") { src = m->buf(m->gBottB); n = Mb * Cb; }
        else if (base == "logits") { src = m->buf(m->logits); n = M1 * m->out_ch; }
        else if (base == "dlogits") { src = m->buf(m->dlogits); n = M1 * m->out_ch; }
        else if (base == "chan") {
            RFI_REQUIRE(idx >= 0 && idx < (int)m->convs.size(), "debug_tensor: conv index out of range");
            src = m->convs[idx].chan;
            n = (size_t)8 * m->convs[idx].cout;
        } else {
            throw Error("debug_tensor: unknown tensor " + s);
        }
        if (n_floats) *n_floats = (int64_t)n;
        if (host) {
            RFI_REQUIRE(host_floats >= n, "debug_tensor: host buffer too small");
            download(m, host, src, n);
        }
    });
}

int rfi_model_algorithmic_flops(rfi_model* m, int n, int h, int w, double* fwd, double* step) {
    return guarded([&] {
        // 2*M*K*N over every conv / convT / head, each layer evaluated once
        double f = 0, stem = 0;
        const int D = m->depth;
        if (m->arch == 6) {
            for (auto& c : m->convs) f += 2.0 * n * c.cin * c.cout;
            f += 2.0 * n * (double)m->feat * m->out_ch;
            if (fwd) *fwd = f;
            if (step) *step = 3.0 * f;
            return;
        }
        if (m->arch == 5) {             // every conv once at its output resolution (the stem's 7x7 has stride 2)
            for (auto& c : m->convs) f += 2.0 * n * (double)(h >> c.level) * (w >> c.level) * c.R * c.R * c.cin * c.cout;
            if (fwd) *fwd = f;
            if (step) *step = 3.0 * f - 2.0 * n * (double)(h >> 1) * (w >> 1) * 49.0 * m->convs[0].cin * m->convs[0].cout;
            return;
        }
        if (m->arch == 3 || m->arch == 4) {   // mask / RPN head: L 3x3 convs, (the transposed conv,) the 1x1 head
            const double M = (double)n * h * w, C = m->in_ch, s2 = (double)m->out_scale * m->out_scale;
            f = m->depth * 2.0 * M * 9.0 * C * C + (m->arch == 3 ? 2.0 * M * 4.0 * C * C : 0.0) + 2.0 * s2 * M * C * m->out_ch;
            if (fwd) *fwd = f;
            if (step) *step = 3.0 * f;  // the input gradient is computed too (it feeds the RoIAlign adjoint)
            return;
        }
        if (m->arch == 1) {             // 3-layer CNN: two 3x3 convs at full resolution + the 1x1 head
            const double M = (double)n * h * w;
            stem = 2.0 * M * 9.0 * m->convs[0].cin * m->convs[0].cout;
            f = stem + 2.0 * M * 9.0 * m->convs[1].cin * m->convs[1].cout + 2.0 * M * (double)m->feat * m->out_ch;
            if (fwd) *fwd = f;
            if (step) *step = 3.0 * f - stem;
            return;
        }
        for (size_t ci = 0; ci < m->convs.size(); ++ci) {
            const int lvl = m->convs[ci].level, R = m->convs[ci].R;       // M = OUTPUT pixels (stride-2 convs included)
            const double M = (double)n * (h >> (lvl - 1)) * (w >> (lvl - 1));
            const double fl = 2.0 * M * R * R * m->convs[ci].cin * m->convs[ci].cout;
            f += fl;
            if (ci == 0) stem = fl;
        }
        for (int k = 0; k < D; ++k) {
            const int l = D - k;
            const double M = (double)n * (h >> l) * (w >> l);
            f += 2.0 * M * 4.0 * m->ups[k].cin * m->ups[k].cout;
        }
        f += 2.0 * n * h * w * (double)m->feat * m->out_ch;
        if (fwd) *fwd = f;
        if (step) *step = 3.0 * f - stem;     // fwd + dgrad + wgrad, no dgrad for the first layer
    });
}

// ------------------------------------------------------------------------------------ RCCL
struct Id128 { char bytes[128]; };   // ncclUniqueId is passed BY VALUE to ncclCommInitRank
namespace {

struct NcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;

void load_nccl() {
    if (g_nccl.lib) return;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
        g_nccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_nccl.lib) break;
    }
    RFI_REQUIRE(g_nccl.lib, std::string("cannot dlopen librccl.so: ") + dlerror());
    auto sym = [&](const char* s) {
        void* p = dlsym(g_nccl.lib, s);
        RFI_REQUIRE(p, std::string("librccl.so lacks symbol ") + s);
        return p;
    };
    g_nccl.GetUniqueId = reinterpret_cast<decltype(g_nccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_nccl.CommInitRank = reinterpret_cast<decltype(g_nccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_nccl.AllReduce = reinterpret_cast<decltype(g_nccl.AllReduce)>(sym("ncclAllReduce"));
    g_nccl.CommDestroy = reinterpret_cast<decltype(g_nccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_nccl.GetErrorString = reinterpret_cast<decltype(g_nccl.GetErrorString)>(sym("ncclGetErrorString"));
}
void nccl_check(int rc, const char* what) {
    if (rc != 0) throw Error(std::string(what) + " failed: " + g_nccl.GetErrorString(rc));
}
}  // namespace

int rfi_comm_unique_id(void* id_buf128) {
    return guarded([&] {
        load_nccl();
        nccl_check(g_nccl.GetUniqueId(id_buf128), "ncclGetUniqueId");
    });
}
int rfi_comm_init(rfi_ctx* ctx, const void* id_buf128, int rank, int world_size) {
    return guarded([&] {
        RFI_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "rfi_comm_init: bad rank/world");
        load_nccl();
        ctx->activate();
        Id128 id;
        std::memcpy(id.bytes, id_buf128, 128);
        void* comm = nullptr;
        nccl_check(g_nccl.CommInitRank(&comm, world_size, id, rank), "ncclCommInitRank");
        ctx->nccl_comm = comm;
        ctx->rank = rank;
        ctx->world = world_size;
    });
}
int rfi_comm_destroy(rfi_ctx* ctx) {
    return guarded([&] {
        if (ctx->nccl_comm) {
            ctx->activate();
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->main_stream) (void)hipStreamSynchronize(ctx->main_stream);
            if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);      // (weight gradients feed the buckets)
            if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
            nccl_check(g_nccl.CommDestroy(ctx->nccl_comm), "ncclCommDestroy");
            ctx->nccl_comm = nullptr;
            ctx->world = 1;
        }
    });
}
namespace {
void comm_allreduce_sum(rfi_ctx* ctx, float* dptr, int64_t count) {
    RFI_REQUIRE(ctx->nccl_comm, "rfi_comm_allreduce: communicator not initialised");
    ctx->activate();
    ProfScope ps(ctx, FAM_COMM, 0, (double)count * 4);
    // ncclFloat32 = 7, ncclSum = 0
    nccl_check(g_nccl.AllReduce(dptr, dptr, (size_t)count, 7, 0, ctx->nccl_comm, ctx->stream), "ncclAllReduce");
}
}  // namespace
int rfi_comm_allreduce_sum_f32(rfi_ctx* ctx, float* dptr, int64_t count) {
    return guarded([&] { comm_allreduce_sum(ctx, dptr, count); });
}
int rfi_comm_emulate(rfi_ctx* ctx, int world) {
    return guarded([&] {
        RFI_REQUIRE(world == 0 || world >= 2, "rfi_comm_emulate: world must be 0 (off) or >= 2");
        RFI_REQUIRE(!(ctx->nccl_comm && ctx->world > 1), "rfi_comm_emulate: a real communicator is active");
        ctx->activate();
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->main_stream));
        ctx->comm_emulate = world;
    });
}
}  // extern "C"
namespace rfi {
// one bucket of the gradient exchange, on the context's communication stream (the caller orders it by events)
void comm_bucket_allreduce(rfi_ctx* ctx, float* dptr, int64_t count) {
    hipStream_t keep = ctx->stream;
    ctx->stream = ctx->comm_stream;
    struct Restore { rfi_ctx* c; hipStream_t s; ~Restore() { c->stream = s; } } restore{ctx, keep};
    if (ctx->comm_emulate > 1) {
        launch_scale_inplace(ctx, dptr, count, (float)ctx->comm_emulate);
        return;
    }
    load_nccl();
    ProfScope ps(ctx, FAM_COMM, 0, (double)count * 4);
    nccl_check(g_nccl.AllReduce(dptr, dptr, (size_t)count, 7, 0, ctx->nccl_comm, ctx->comm_stream), "ncclAllReduce");
}
}  // namespace rfi
extern "C" {
int rfi_model_allreduce_grads(rfi_model* m) {
    {
        const int rc = guarded([&] { m->ctx->activate(); m->join_pending_side(); });
        if (rc) return rc;
    }
    if (m->ctx->comm_emulate > 1)          // rfi_comm_emulate: the explicit exchange of the split API is emulated like the buckets
        return guarded([&] {
            m->ctx->activate();
            launch_scale_inplace(m->ctx, m->grads, (int64_t)m->n_flat, (float)m->ctx->comm_emulate);
        });
    return rfi_comm_allreduce_sum_f32(m->ctx, m->grads, (int64_t)m->n_flat);
}

// ------------------------------------------------------------------------------------ preprocessing / metrics
int rfi_preprocess_patches(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype, int n,
                           int ps_h, int ps_w, float* out_nhwc, int out_mem) {
    return guarded([&] {
        RFI_REQUIRE(n >= 0 && ps_h > 0 && ps_w > 0, "preprocess: bad shape");
        if (n == 0) return;
        ctx->activate();
        const size_t px = (size_t)n * ps_h * ps_w;
        const size_t esz = dtype == RFI_C128 ? 16 : (dtype == RFI_F32 ? 4 : 8);
        void* din = const_cast<void*>(patches);
        float* dout = out_nhwc;
        void *tmp_in = nullptr, *tmp_out = nullptr;
        if (patches_mem == RFI_HOST) {
            tmp_in = ctx->alloc(px * esz);
            RFI_CHECK_HIP(hipMemcpyAsync(tmp_in, patches, px * esz, hipMemcpyHostToDevice, ctx->stream));
            din = tmp_in;
        }
        if (out_mem == RFI_HOST) {
            tmp_out = ctx->alloc(px * 3 * sizeof(float));
            dout = static_cast<float*>(tmp_out);
        }
        void* mm = ctx->get_scratch((size_t)n * 4 * sizeof(unsigned long long));
        launch_preprocess(ctx, din, dtype, n, ps_h, ps_w, static_cast<float*>(mm), dout);
        if (out_mem == RFI_HOST)
            RFI_CHECK_HIP(hipMemcpyAsync(out_nhwc, dout, px * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (tmp_in || tmp_out) {            // device-to-device calls stay asynchronous on the ctx stream
            RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            if (tmp_in) ctx->release(tmp_in);
            if (tmp_out) ctx->release(tmp_out);
        }
    });
}

namespace {
// device copy of `bytes` host bytes (or the pointer itself when already on the device)
struct Staged {
    rfi_ctx* ctx; void* dev; bool owned;
    Staged(rfi_ctx* c, const void* p, int mem, size_t bytes) : ctx(c), dev(const_cast<void*>(p)), owned(false) {
        if (mem == RFI_HOST && p && bytes) {
            dev = ctx->alloc(bytes);
            owned = true;
            RFI_CHECK_HIP(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ~Staged() {
        if (owned) {
            (void)hipStreamSynchronize(ctx->stream);
            try { ctx->release(dev); } catch (...) {}
        }
    }
};
void check_table(const rfi_patch_src* t, int n, int n_planes, int c, int tt, int ps) {
    RFI_REQUIRE(n_planes > 0 && c > 0 && tt > 0 && ps > 0, "preprocess_gather: bad shape");
    RFI_REQUIRE((int64_t)n_planes * c * tt < ((int64_t)1 << 40), "preprocess_gather: waterfall too large");
    for (int i = 0; i < n; ++i) {
        const rfi_patch_src& e = t[i];
        RFI_REQUIRE(e.plane >= 0 && e.plane < n_planes && e.view >= 0 && e.view <= 3 && e.row0 >= 0 && e.col0 >= 0,
                    "preprocess_gather: bad table entry " + std::to_string(i));
    }
}
}  // namespace

int rfi_patch_any_flag(rfi_ctx* ctx, const uint8_t* flags, int flags_mem, int n_planes, int c, int t,
                       const rfi_patch_src* table_host, int n, int ps, uint8_t* any_out_host) {
    return guarded([&] {
        RFI_REQUIRE(n >= 0 && flags && (n == 0 || (table_host && any_out_host)), "patch_any_flag: null argument");
        if (n == 0) return;
        check_table(table_host, n, n_planes, c, t, ps);
        ctx->activate();
        Staged fl(ctx, flags, flags_mem, (size_t)n_planes * c * t);
        Staged tb(ctx, table_host, RFI_HOST, (size_t)n * sizeof(rfi_patch_src));
        unsigned* d_any = static_cast<unsigned*>(ctx->alloc((size_t)n * sizeof(unsigned)));
        launch_patch_any_flag(ctx, static_cast<const uint8_t*>(fl.dev), static_cast<const rfi_patch_src*>(tb.dev), c, t,
                              n, ps, d_any);
        std::vector<unsigned> h((size_t)n);
        RFI_CHECK_HIP(hipMemcpyAsync(h.data(), d_any, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->release(d_any);
        for (int i = 0; i < n; ++i) any_out_host[i] = h[(size_t)i] ? 1 : 0;
    });
}

int rfi_preprocess_gather(rfi_ctx* ctx, const void* planes, int planes_mem, int dtype, int n_planes, int c,
                          int t, const uint8_t* flags, int flags_mem, const rfi_patch_src* table_host, int n,
                          int ps, float* out_nhwc, int out_mem, uint8_t* out_labels, int labels_mem) {
    return guarded([&] {
        RFI_REQUIRE(n >= 0 && planes && (n == 0 || (table_host && out_nhwc)), "preprocess_gather: null argument");
        RFI_REQUIRE(dtype >= RFI_C128 && dtype <= RFI_F32, "preprocess_gather: unknown dtype");
        RFI_REQUIRE(!out_labels || flags, "preprocess_gather: labels requested without flags");
        if (n == 0) return;
        check_table(table_host, n, n_planes, c, t, ps);
        ctx->activate();
        const size_t esz = dtype == RFI_C128 ? 16 : (dtype == RFI_F32 ? 4 : 8);
        const size_t px = (size_t)n * ps * ps;
        Staged pl(ctx, planes, planes_mem, (size_t)n_planes * c * t * esz);
        Staged fl(ctx, flags, flags_mem, (size_t)n_planes * c * t);
        Staged tb(ctx, table_host, RFI_HOST, (size_t)n * sizeof(rfi_patch_src));
        float* dout = out_nhwc;
        uint8_t* dlab = out_labels;
        void *tmp_out = nullptr, *tmp_lab = nullptr;
        if (out_mem == RFI_HOST) dout = static_cast<float*>(tmp_out = ctx->alloc(px * 3 * sizeof(float)));
        if (out_labels && labels_mem == RFI_HOST) dlab = static_cast<uint8_t*>(tmp_lab = ctx->alloc(px));
        void* mm = ctx->get_scratch((size_t)n * 4 * sizeof(unsigned long long));
        const auto* table_dev = static_cast<const rfi_patch_src*>(tb.dev);
        launch_preprocess(ctx, pl.dev, dtype, n, ps, ps, static_cast<float*>(mm), dout, table_dev, c, t);
        if (out_labels) launch_gather_labels(ctx, static_cast<const uint8_t*>(fl.dev), table_dev, c, t, n, ps, dlab);
        if (tmp_out)
            RFI_CHECK_HIP(hipMemcpyAsync(out_nhwc, dout, px * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (tmp_lab) RFI_CHECK_HIP(hipMemcpyAsync(out_labels, dlab, px, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));     // the staged table dies with this call
        if (tmp_out) ctx->release(tmp_out);
        if (tmp_lab) ctx->release(tmp_lab);
    });
}

namespace {
// med / mad of every patch of w (n x per doubles) over its non-NaN values -> d_med, d_mad
void median_and_mad(rfi_ctx* ctx, const double* w, int n, int per, bool finite_only, double* d_med, double* d_mad,
                    int* d_cnt, bool f32 = false) {
    launch_patch_median(ctx, w, n, per, false, nullptr, finite_only, d_med, d_cnt, f32);
    launch_patch_median(ctx, w, n, per, true, d_med, finite_only, d_mad, nullptr, f32);
}
}  // namespace

int rfi_preprocess_real(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype, int n, int ps_h, int ps_w,
                        int stretch, int normalize_before, int normalize_after, double flag_sigma, float* out_nhwc,
                        int out_mem, uint8_t* flags_out, int flags_mem) {
    return guarded([&] {
        RFI_REQUIRE(dtype == RFI_F64 || dtype == RFI_F32, "preprocess_real: input must be float64 or float32");
        // float32 input: NumPy keeps it in float32 arithmetic (preprocessor.py:608-706 on a float32 array); here the values
        // ride in doubles and every result is rounded to float32 (order_stats.hip)
        const bool f32 = dtype == RFI_F32;
        RFI_REQUIRE(stretch >= 0 && stretch <= 2, "preprocess_real: stretch must be 0 (none), 1 (SQRT) or 2 (LOG10)");
        RFI_REQUIRE(n >= 0 && ps_h > 0 && ps_w > 0 && (n == 0 || (patches && out_nhwc)), "preprocess_real: bad argument");
        if (n == 0) return;
        ctx->activate();
        const int per = ps_h * ps_w;
        const size_t px = (size_t)n * per, esz = dtype == RFI_F64 ? 8 : 4;
        Staged in(ctx, patches, patches_mem, px * esz);
        double* w = static_cast<double*>(ctx->alloc(px * sizeof(double)));
        double* stat = static_cast<double*>(ctx->alloc((size_t)n * 2 * sizeof(double) + (size_t)n * sizeof(int)));
        double *d_med = stat, *d_mad = stat + n;
        int* d_cnt = reinterpret_cast<int*>(stat + 2 * (size_t)n);
        launch_to_abs_f64(ctx, in.dev, dtype, (int64_t)px, w);            // real dtypes: widening copy
        auto normalise = [&] {
            launch_patch_median(ctx, w, n, per, false, nullptr, false, d_med, nullptr, f32);
            launch_scale_by_median(ctx, w, n, per, d_med, f32);
        };
        if (normalize_before) normalise();
        if (stretch) {
            launch_stretch(ctx, w, (int64_t)px, stretch, f32);
            median_and_mad(ctx, w, n, per, true, d_med, d_mad, d_cnt, f32);
            launch_replace_inf(ctx, w, n, per, d_mad, d_cnt);
        }
        if (normalize_after) normalise();
        float* dout = out_nhwc;
        uint8_t* dfl = flags_out;
        void *tmp_out = nullptr, *tmp_fl = nullptr;
        if (out_mem == RFI_HOST) dout = static_cast<float*>(tmp_out = ctx->alloc(px * 3 * sizeof(float)));
        if (flags_out && flags_mem == RFI_HOST) dfl = static_cast<uint8_t*>(tmp_fl = ctx->alloc(px));
        if (flags_out) {
            median_and_mad(ctx, w, n, per, false, d_med, d_mad, nullptr, f32);
            launch_mad_flags(ctx, w, n, per, d_med, d_mad, flag_sigma, dfl, f32);
        }
        void* mm = ctx->get_scratch((size_t)n * 4 * sizeof(unsigned long long));
        float* w32 = nullptr;
        if (f32) {                                  // the channel kernels' float32 form on the float32 values
            w32 = static_cast<float*>(ctx->alloc(px * sizeof(float)));
            launch_narrow_f32(ctx, w, (int64_t)px, w32);
            launch_preprocess(ctx, w32, RFI_F32, n, ps_h, ps_w, static_cast<float*>(mm), dout);
        } else {
            launch_preprocess(ctx, w, RFI_F64, n, ps_h, ps_w, static_cast<float*>(mm), dout);
        }
        if (tmp_out) RFI_CHECK_HIP(hipMemcpyAsync(out_nhwc, dout, px * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (tmp_fl) RFI_CHECK_HIP(hipMemcpyAsync(flags_out, dfl, px, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->release(w);
        ctx->release(stat);
        if (w32) ctx->release(w32);
        if (tmp_out) ctx->release(tmp_out);
        if (tmp_fl) ctx->release(tmp_fl);
    });
}

int rfi_mad_flags(rfi_ctx* ctx, const void* patches, int patches_mem, int dtype, int n, int ps_h, int ps_w,
                  double flag_sigma, uint8_t* flags_out, int flags_mem) {
    return guarded([&] {
        RFI_REQUIRE(dtype >= RFI_C128 && dtype <= RFI_F32, "mad_flags: unknown dtype");
        RFI_REQUIRE(n >= 0 && ps_h > 0 && ps_w > 0 && (n == 0 || (patches && flags_out)), "mad_flags: bad argument");
        if (n == 0) return;
        ctx->activate();
        const int per = ps_h * ps_w;
        const size_t px = (size_t)n * per, esz = dtype == RFI_C128 ? 16 : (dtype == RFI_F32 ? 4 : 8);
        Staged in(ctx, patches, patches_mem, px * esz);
        double* w = static_cast<double*>(ctx->alloc(px * sizeof(double)));
        double* stat = static_cast<double*>(ctx->alloc((size_t)n * 2 * sizeof(double)));
        uint8_t* dfl = flags_out;
        void* tmp_fl = nullptr;
        if (flags_mem == RFI_HOST) dfl = static_cast<uint8_t*>(tmp_fl = ctx->alloc(px));
        launch_to_abs_f64(ctx, in.dev, dtype, (int64_t)px, w);
        median_and_mad(ctx, w, n, per, false, stat, stat + n, nullptr);
        launch_mad_flags(ctx, w, n, per, stat, stat + n, flag_sigma, dfl);
        if (tmp_fl) RFI_CHECK_HIP(hipMemcpyAsync(flags_out, dfl, px, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->release(w);
        ctx->release(stat);
        if (tmp_fl) ctx->release(tmp_fl);
    });
}

int rfi_generate_waterfalls(rfi_ctx* ctx, uint64_t seed, int n_samples, int n_pol, int c, int t, double noise_mjy,
                            int bandpass, int bandpass_order, double pol_corr, const rfi_event* events_host,
                            const int32_t* event_offsets_host, int out_dtype, void* planes_out, int planes_mem,
                            uint8_t* flags_out, int flags_mem) {
    return guarded([&] {
        RFI_REQUIRE(n_samples >= 0 && n_pol > 0 && c > 0 && t > 0, "generate_waterfalls: bad shape");
        RFI_REQUIRE(out_dtype == RFI_C128 || out_dtype == RFI_C64, "generate_waterfalls: output must be complex128/64");
        RFI_REQUIRE(planes_out && flags_out && event_offsets_host, "generate_waterfalls: null argument");
        if (n_samples == 0) return;
        const int n_events = event_offsets_host[n_samples];
        RFI_REQUIRE(event_offsets_host[0] == 0 && n_events >= 0 && (n_events == 0 || events_host),
                    "generate_waterfalls: bad event table");
        for (int s = 0; s < n_samples; ++s)
            RFI_REQUIRE(event_offsets_host[s] <= event_offsets_host[s + 1], "generate_waterfalls: offsets must ascend");
        ctx->activate();
        const size_t px = (size_t)n_samples * n_pol * c * t, esz = out_dtype == RFI_C128 ? 16 : 8;
        Staged ev(ctx, events_host, RFI_HOST, (size_t)std::max(n_events, 1) * sizeof(rfi_event));
        Staged of(ctx, event_offsets_host, RFI_HOST, (size_t)(n_samples + 1) * sizeof(int32_t));
        void* dpl = planes_out;
        uint8_t* dfl = flags_out;
        void *tmp_p = nullptr, *tmp_f = nullptr;
        if (planes_mem == RFI_HOST) dpl = tmp_p = ctx->alloc(px * esz);
        if (flags_mem == RFI_HOST) dfl = static_cast<uint8_t*>(tmp_f = ctx->alloc(px));
        launch_synth(ctx, seed, n_samples, n_pol, c, t, noise_mjy, bandpass, bandpass_order, pol_corr,
                     static_cast<const rfi_event*>(n_events ? ev.dev : nullptr), static_cast<const int*>(of.dev),
                     out_dtype, dpl, dfl);
        if (tmp_p) RFI_CHECK_HIP(hipMemcpyAsync(planes_out, dpl, px * esz, hipMemcpyDeviceToHost, ctx->stream));
        if (tmp_f) RFI_CHECK_HIP(hipMemcpyAsync(flags_out, dfl, px, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (tmp_p) ctx->release(tmp_p);
        if (tmp_f) ctx->release(tmp_f);
    });
}

int rfi_confusion_counts(rfi_ctx* ctx, const void* pred, int pred_dtype, int pred_mem, const void* truth,
                         int truth_dtype, int truth_mem, int64_t count, int64_t* tp, int64_t* fp,
                         int64_t* fn) {
    return guarded([&] {
        RFI_REQUIRE(count >= 0, "confusion: negative count");
        RFI_REQUIRE((pred_dtype == RFI_U8 || pred_dtype == RFI_FLOAT32) &&
                        (truth_dtype == RFI_U8 || truth_dtype == RFI_FLOAT32), "confusion: dtype must be u8 or f32");
        ctx->activate();
        const void *dp = pred, *dt = truth;
        void *tp_ = nullptr, *tt_ = nullptr;
        if (pred_mem == RFI_HOST && count) {
            const size_t b = (size_t)count * (pred_dtype ? 4 : 1);
            tp_ = ctx->alloc(b);
            RFI_CHECK_HIP(hipMemcpyAsync(tp_, pred, b, hipMemcpyHostToDevice, ctx->stream));
            dp = tp_;
        }
        if (truth_mem == RFI_HOST && count) {
            const size_t b = (size_t)count * (truth_dtype ? 4 : 1);
            tt_ = ctx->alloc(b);
            RFI_CHECK_HIP(hipMemcpyAsync(tt_, truth, b, hipMemcpyHostToDevice, ctx->stream));
            dt = tt_;
        }
        auto* d3 = static_cast<unsigned long long*>(ctx->alloc(3 * sizeof(unsigned long long)));
        launch_confusion(ctx, dp, pred_dtype, dt, truth_dtype, count, d3);
        unsigned long long h3[3];
        RFI_CHECK_HIP(hipMemcpyAsync(h3, d3, sizeof(h3), hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        *tp = (int64_t)h3[0]; *fp = (int64_t)h3[1]; *fn = (int64_t)h3[2];
        ctx->release(d3);
        if (tp_) ctx->release(tp_);
        if (tt_) ctx->release(tt_);
    });
}
int rfi_threshold_logits(rfi_ctx* ctx, const float* logits_dev, int64_t count, float threshold,
                         uint8_t* mask_dev) {
    return guarded([&] {
        ctx->activate();
        launch_threshold(ctx, logits_dev, count, threshold, mask_dev);
    });
}

// ------------------------------------------------------------------------------------ kernel-level ops
namespace {
struct Scratch {
    rfi_ctx* c;
    std::vector<void*> v;
    explicit Scratch(rfi_ctx* ctx) : c(ctx) {}
    float* get(size_t floats) {
        void* p = c->alloc(floats * sizeof(float));
        v.push_back(p);
        return static_cast<float*>(p);
    }
    ~Scratch() {
        hipStreamSynchronize(c->stream);
        for (void* p : v) c->release(p);
    }
};
float* upload_lib_weight(rfi_ctx* ctx, Scratch& s, const float* dev_ref, size_t numel, bool convt,
                         int d0, int d1, int R) {
    // dev_ref holds the reference layout ON DEVICE; bounce through the host to convert
    std::vector<float> h(numel), lib;
    RFI_CHECK_HIP(hipMemcpyAsync(h.data(), dev_ref, numel * 4, hipMemcpyDeviceToHost, ctx->stream));
    RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (convt) to_lib_convt(h.data(), d0, d1, lib);
    else to_lib_conv(h.data(), d0, d1, R, lib);
    float* d = s.get(numel);
    RFI_CHECK_HIP(hipMemcpyAsync(d, lib.data(), numel * 4, hipMemcpyHostToDevice, ctx->stream));
    RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return d;
}

// ---- the stride-2 layers on the plane kernels (bfloat16 flow of the ResNet-encoder model; impl 6): temporary plane copies
struct PlaneTmp {                 // a zero-tailed bf16 tensor [pixels][chunks * 16]
    bf16_t* p = nullptr;
    int64_t ps = 0;
    int nchunks = 0;
    PlaneSeg seg() const { return PlaneSeg{p, ps, nchunks}; }
};
PlaneTmp plane_tmp(rfi_ctx* ctx, Scratch& s, int64_t pixels, int C) {
    PlaneTmp t;
    t.nchunks = plane_chunks(C);
    t.ps = (int64_t)t.nchunks * 16;
    const size_t bytes = (size_t)pixels * t.ps * 2 + 64;
    t.p = reinterpret_cast<bf16_t*>(s.get((bytes + 3) / 4));
    RFI_CHECK_HIP(hipMemsetAsync(t.p, 0, bytes, ctx->stream));
    return t;
}
PlaneTmp planes_of(rfi_ctx* ctx, Scratch& s, const float* x, int64_t pixels, int C) {
    PlaneTmp t = plane_tmp(ctx, s, pixels, C);
    launch_act_split(ctx, View{x, C}, pixels, C, InXform{}, 1, t.p, t.ps);
    return t;
}
bf16_t* wb_of(rfi_ctx* ctx, Scratch& s, const float* src, int taps, int Cout, int Cin, int seg0, int seg1) {
    const size_t e = wb_elems(taps, Cout, seg0, seg1, 1);
    bf16_t* wb = reinterpret_cast<bf16_t*>(s.get((e * 2 + 64 + 3) / 4));
    RFI_CHECK_HIP(hipMemsetAsync(wb, 0, e * 2 + 64, ctx->stream));
    launch_weights_to_wb_one(ctx, WBDesc{src, wb, taps, Cout, Cin, {seg0, seg1}, 1});
    return wb;
}
}  // namespace

int rfi_op_conv3x3(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin,
                   const float* w_oihw, const float* bias, int cout, const float* in_scale,
                   const float* in_shift, int in_relu, float* y) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        ConvArgs a;
        a.x = View{x, cin};
        a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w; a.Cin = cin; a.Cout = cout;
        a.w = upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3);
        a.bias = bias;
        a.y = MutView{y, cout};
        a.Hout = h; a.Wout = w;
        a.xf = InXform{in_scale, in_shift, in_relu};
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_conv1x1(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin, const float* w_oihw, const float* bias,
                   int cout, const float* in_scale, const float* in_shift, int in_relu, float* y) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        ConvArgs a;
        a.x = View{x, cin};
        a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w; a.Cin = cin; a.Cout = cout;
        a.w = upload_lib_weight(ctx, s, w_oihw, (size_t)cin * cout, false, cout, cin, 1);
        a.bias = bias;
        a.y = MutView{y, cout};
        a.Hout = h; a.Wout = w;
        a.R = 1; a.S = 1; a.pad = 0;
        a.xf = InXform{in_scale, in_shift, in_relu};
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_conv3x3_dgrad(rfi_ctx* ctx, int impl, const float* dy, int n, int h, int w, int cout,
                         const float* w_oihw, int cin, float* dx) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        float* wf = upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3);
        float* wd = s.get((size_t)9 * cin * cout);
        launch_weight_to_dgrad(ctx, wf, 9, cout, cin, 1, wd);
        ConvArgs a;
        a.x = View{dy, cout};
        a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w; a.Cin = cout; a.Cout = cin;
        a.w = wd;
        a.y = MutView{dx, cin};
        a.Hout = h; a.Wout = w;
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_conv3x3_wgrad(rfi_ctx* ctx, int impl, const float* x, const float* dy, int n, int h, int w,
                         int cin, int cout, const float* in_scale, const float* in_shift, int in_relu,
                         float* dw_oihw) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        WgradArgs a;
        a.xop = View{x, cin};
        a.yop = View{dy, cout};
        a.xf_x = InXform{in_scale, in_shift, in_relu};
        a.N = n; a.H = h; a.W = w; a.Hx = h; a.Wx = w; a.Cx = cin; a.Cy = cout;
        a.tap_stride = (int64_t)cin * cout;
        a.sy = cin; a.sx = 1;
        const size_t numel = (size_t)9 * cin * cout;
        a.dw = s.get(numel);
        a.slab_floats = wgrad_slab_floats(a, impl);
        a.slab = s.get(a.slab_floats);
        launch_wgrad(ctx, a, impl);
        std::vector<float> lib(numel), ref(numel);
        RFI_CHECK_HIP(hipMemcpyAsync(lib.data(), a.dw, numel * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        from_lib_conv(lib.data(), cout, cin, 3, ref.data());
        RFI_CHECK_HIP(hipMemcpyAsync(dw_oihw, ref.data(), numel * 4, hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    });
}
// ---- stride-2 convolutions of the ResNet-style encoder (model_resnet.cpp): 3x3 / pad 1 as a 2x2 convolution on the
// space-to-depth input, 1x1 / pad 0 on a channel slice of it.  h, w: INPUT size (even); outputs are h/2 x w/2.
int rfi_op_conv_s2(rfi_ctx* ctx, int impl, int ksize, const float* x, int n, int h, int w, int cin, const float* w_oihw,
                   int cout, float* y) {
    return guarded([&] {
        RFI_REQUIRE(ksize == 3 || ksize == 1, "conv_s2: kernel size 3 or 1");
        ctx->activate();
        Scratch s(ctx);
        if (impl == IMPL_PLANES_BF16) {           // the strided contraction on the full-resolution planes, bfloat16 output
            RFI_REQUIRE(cout % 4 == 0, "conv_s2 on planes: cout % 4 == 0");
            const int64_t Mo = (int64_t)n * (h / 2) * (w / 2);
            const PlaneTmp xp = planes_of(ctx, s, x, (int64_t)n * h * w, cin);
            const float* wl = ksize == 3 ? upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3) : w_oihw;
            PlaneTmp yp = plane_tmp(ctx, s, Mo, cout);
            PConvArgs a;
            a.x[0] = xp.seg(); a.nseg = 1; a.P = 1;
            a.N = n; a.H = h / 2; a.W = w / 2; a.Hin = h; a.Win = w; a.Hout = h / 2; a.Wout = w / 2;
            a.R = ksize; a.S = 2; a.pad = ksize == 3 ? 1 : 0;
            a.Cout = cout;
            a.wB = wb_of(ctx, s, wl, ksize * ksize, cout, cin, cin, 0);
            a.y16 = yp.p; a.y_pstride = (int)yp.ps;
            launch_pconv(ctx, a);
            launch_planes_to_f32(ctx, yp.p, yp.ps, Mo, cout, 1, y, cout);
            return;
        }
        float* xs = s.get((size_t)n * h * w * cin);
        launch_s2d(ctx, x, n, h, w, cin, xs);
        ConvArgs a;
        a.N = n; a.H = h / 2; a.W = w / 2; a.Hin = h / 2; a.Win = w / 2; a.Cout = cout;
        a.x = View{xs, 4 * cin};
        a.y = MutView{y, cout};
        a.Hout = h / 2; a.Wout = w / 2;
        a.S = 1;
        if (ksize == 3) {
            float* w3 = upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3);
            float* w2 = s.get((size_t)16 * cin * cout);
            launch_w_s2d(ctx, w3, cout, cin, w2, true);
            a.w = w2; a.Cin = 4 * cin; a.R = 2; a.pad = 1;
        } else {
            a.w = w_oihw; a.Cin = cin; a.R = 1; a.pad = 0;          // [cout][cin][1][1] == [1 tap][cout][cin]
        }
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_conv_s2_dgrad(rfi_ctx* ctx, int impl, int ksize, const float* dy, int n, int h, int w, int cout,
                         const float* w_oihw, int cin, float* dx) {
    return guarded([&] {
        RFI_REQUIRE(ksize == 3 || ksize == 1, "conv_s2_dgrad: kernel size 3 or 1");
        ctx->activate();
        Scratch s(ctx);
        if (impl == IMPL_PLANES_BF16) {           // four 2x2 contractions of dY, one per parity class of the input pixel
            RFI_REQUIRE(cin % 4 == 0, "conv_s2_dgrad on planes: cin % 4 == 0");
            const int64_t Mo = (int64_t)n * (h / 2) * (w / 2), Mi = (int64_t)n * h * w;
            const PlaneTmp dyp = planes_of(ctx, s, dy, Mo, cout);
            const float* w3 = ksize == 3 ? upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3) : nullptr;
            float* cls = s.get(s2_class_floats(cout, cin));
            launch_w_s2_classes(ctx, w3, ksize == 1 ? w_oihw : nullptr, cout, cin, cls);     // (a 1x1 layer: the second K segment of class 0)
            PlaneTmp dxp = plane_tmp(ctx, s, Mi, cin);
            bf16_t* dx16 = dxp.p;                 // dense [Mi][cin] bfloat16 (cin % 16 != 0: rows of cin elements inside the allocation)
            for (int c = 0; c < 4; ++c) {
                PConvArgs a;
                a.x[0] = dyp.seg();
                if (c == 0) a.x[1] = dyp.seg();
                a.nseg = c == 0 ? 2 : 1; a.P = 1;
                a.N = n; a.H = h / 2; a.W = w / 2; a.Hin = h / 2; a.Win = w / 2;
                a.R = 2; a.S = 1; a.pad = 0;
                a.Cout = cin;
                a.wB = wb_of(ctx, s, cls + s2_class_offset(c, cout, cin), 4, cin, c == 0 ? 2 * cout : cout, cout, c == 0 ? cout : 0);
                a.y16 = dx16; a.y_pstride = cin;
                a.Hout = h; a.Wout = w; a.osy = 2; a.osx = 2; a.ooy = c >> 1; a.oox = c & 1;
                launch_pconv(ctx, a);
            }
            launch_planes_to_f32(ctx, dx16, cin, Mi, cin, 1, dx, cin);
            return;
        }
        ConvArgs a;
        a.N = n; a.H = h / 2; a.W = w / 2; a.Hin = h / 2; a.Win = w / 2; a.Cin = cout;
        a.x = View{dy, cout};
        a.Hout = h / 2; a.Wout = w / 2;
        a.S = 1;
        float* dxp = s.get((size_t)n * h * w * cin);
        float* ds = nullptr;
        if (ksize == 3) {
            float* w3 = upload_lib_weight(ctx, s, w_oihw, (size_t)9 * cin * cout, false, cout, cin, 3);
            float* w2 = s.get((size_t)16 * cin * cout);
            float* wd = s.get((size_t)16 * cin * cout);
            launch_w_s2d(ctx, w3, cout, cin, w2, true);
            launch_weight_to_dgrad(ctx, w2, 4, cout, 4 * cin, 1, wd);
            a.w = wd; a.Cout = 4 * cin; a.R = 2; a.pad = 0;
            a.y = MutView{dxp, 4 * cin};
        } else {
            float* wd = s.get((size_t)cin * cout);
            launch_weight_to_dgrad(ctx, w_oihw, 1, cout, cin, 0, wd);
            RFI_CHECK_HIP(hipMemsetAsync(dxp, 0, (size_t)n * h * w * cin * sizeof(float), ctx->stream));
            ds = s.get((size_t)n * (h / 2) * (w / 2) * cin);
            a.w = wd; a.Cout = cin; a.R = 1; a.pad = 0;
            a.y = MutView{ds, cin};
        }
        launch_conv(ctx, a, impl);
        launch_d2s_add(ctx, dxp, ds, View{}, n, h, w, cin, dx);
    });
}
int rfi_op_conv_s2_wgrad(rfi_ctx* ctx, int impl, int ksize, const float* x, const float* dy, int n, int h, int w, int cin,
                         int cout, float* dw_oihw) {
    return guarded([&] {
        RFI_REQUIRE(ksize == 3 || ksize == 1, "conv_s2_wgrad: kernel size 3 or 1");
        ctx->activate();
        Scratch s(ctx);
        if (impl == IMPL_PLANES_BF16) {           // the strided weight gradient on the full-resolution planes
            const PlaneTmp xp = planes_of(ctx, s, x, (int64_t)n * h * w, cin);
            const PlaneTmp dyp = planes_of(ctx, s, dy, (int64_t)n * (h / 2) * (w / 2), cout);
            PWgradArgs a;
            a.xop[0] = xp.seg(); a.nseg = 1; a.seg_c[0] = cin;
            a.yop = dyp.seg(); a.Cy = cout; a.P = 1;
            a.N = n; a.H = h / 2; a.W = w / 2; a.Hx = h; a.Wx = w;
            a.R = ksize; a.S = 2; a.pad = ksize == 3 ? 1 : 0;
            a.tap_stride = (int64_t)cin * cout; a.sy = cin; a.sx = 1;
            const size_t numel = (size_t)ksize * ksize * cin * cout;
            a.dw = s.get(numel);
            a.slab_floats = pwgrad_slab_floats(a);
            a.slab = s.get(a.slab_floats);
            launch_pwgrad(ctx, a);
            if (ksize == 1) {
                RFI_CHECK_HIP(hipMemcpyAsync(dw_oihw, a.dw, numel * 4, hipMemcpyDeviceToDevice, ctx->stream));
                return;
            }
            std::vector<float> lib(numel), ref(numel);
            RFI_CHECK_HIP(hipMemcpyAsync(lib.data(), a.dw, numel * 4, hipMemcpyDeviceToHost, ctx->stream));
            RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            from_lib_conv(lib.data(), cout, cin, 3, ref.data());
            RFI_CHECK_HIP(hipMemcpyAsync(dw_oihw, ref.data(), numel * 4, hipMemcpyHostToDevice, ctx->stream));
            RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            return;
        }
        float* xs = s.get((size_t)n * h * w * cin);
        launch_s2d(ctx, x, n, h, w, cin, xs);
        WgradArgs a;
        a.xop = View{xs, 4 * cin};
        a.yop = View{dy, cout};
        a.N = n; a.H = h / 2; a.W = w / 2; a.Hx = h / 2; a.Wx = w / 2; a.Cy = cout;
        a.S = 1; a.sx = 1;
        if (ksize == 3) { a.Cx = 4 * cin; a.R = 2; a.pad = 1; }
        else { a.Cx = cin; a.R = 1; a.pad = 0; }
        a.tap_stride = (int64_t)a.Cx * cout;
        a.sy = a.Cx;
        const size_t numel = (size_t)a.R * a.R * a.Cx * cout;
        a.dw = s.get(numel);
        a.slab_floats = wgrad_slab_floats(a, impl);
        a.slab = s.get(a.slab_floats);
        launch_wgrad(ctx, a, impl);
        if (ksize == 1) {
            RFI_CHECK_HIP(hipMemcpyAsync(dw_oihw, a.dw, numel * 4, hipMemcpyDeviceToDevice, ctx->stream));
            return;
        }
        const size_t n3 = (size_t)9 * cin * cout;
        float* w3 = s.get(n3);
        launch_w_s2d(ctx, w3, cout, cin, a.dw, false);
        std::vector<float> lib(n3), ref(n3);
        RFI_CHECK_HIP(hipMemcpyAsync(lib.data(), w3, n3 * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        from_lib_conv(lib.data(), cout, cin, 3, ref.data());
        RFI_CHECK_HIP(hipMemcpyAsync(dw_oihw, ref.data(), n3 * 4, hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int rfi_op_convt2x2(rfi_ctx* ctx, int impl, const float* x, int n, int h, int w, int cin,
                    const float* w_iohw, const float* bias, int cout, float* y) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        if (impl == IMPL_PLANES_BF16) {           // ONE 1x1 contraction on planes: the four taps are 4 cout output channels
            RFI_REQUIRE(cout % 32 == 0, "convt2x2 on planes: cout % 32 == 0");
            const int64_t Mi = (int64_t)n * h * w;
            const PlaneTmp xp = planes_of(ctx, s, x, Mi, cin);
            const float* wl = upload_lib_weight(ctx, s, w_iohw, (size_t)4 * cin * cout, true, cin, cout, 2);      // [4][cout][cin]
            PlaneTmp yp = plane_tmp(ctx, s, 4 * Mi, cout);
            PConvArgs a;
            a.x[0] = xp.seg(); a.nseg = 1; a.P = 1;
            a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w;
            a.R = 1; a.S = 1; a.pad = 0;
            a.Cout = 4 * cout; a.zblocks = cout / 32;
            a.wB = wb_of(ctx, s, wl, 1, 4 * cout, cin, cin, 0);
            a.bias = bias;
            a.y16 = yp.p; a.y_pstride = (int)yp.ps;
            a.Hout = 2 * h; a.Wout = 2 * w; a.osy = 2; a.osx = 2;
            launch_pconv(ctx, a);
            launch_planes_to_f32(ctx, yp.p, yp.ps, 4 * Mi, cout, 1, y, cout);
            return;
        }
        ConvArgs a;
        a.x = View{x, cin};
        a.N = n; a.H = h; a.W = w; a.Hin = h; a.Win = w; a.Cin = cin; a.Cout = cout;
        a.w = upload_lib_weight(ctx, s, w_iohw, (size_t)4 * cin * cout, true, cin, cout, 2);
        a.bias = bias;
        a.y = MutView{y, cout};
        a.Hout = 2 * h; a.Wout = 2 * w;
        a.osy = 2; a.osx = 2;
        a.R = 1; a.S = 1; a.pad = 0; a.zgroups = 4;
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_convt2x2_dgrad(rfi_ctx* ctx, int impl, const float* dy, int n, int h, int w, int cout,
                          const float* w_iohw, int cin, float* dx) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        float* wf = upload_lib_weight(ctx, s, w_iohw, (size_t)4 * cin * cout, true, cin, cout, 2);
        float* wd = s.get((size_t)4 * cin * cout);
        launch_weight_to_dgrad(ctx, wf, 4, cout, cin, 0, wd);
        if (impl == IMPL_PLANES_BF16) {           // a 2x2 stride-2 contraction of dy on planes, bfloat16 out
            RFI_REQUIRE(cin % 4 == 0, "convt2x2_dgrad on planes: cin % 4 == 0");
            const int64_t Mi = (int64_t)n * h * w;
            const PlaneTmp dyp = planes_of(ctx, s, dy, 4 * Mi, cout);
            PlaneTmp dxp = plane_tmp(ctx, s, Mi, cin);
            PConvArgs a;
            a.x[0] = dyp.seg(); a.nseg = 1; a.P = 1;
            a.N = n; a.H = h; a.W = w; a.Hin = 2 * h; a.Win = 2 * w; a.Hout = h; a.Wout = w;
            a.R = 2; a.S = 2; a.pad = 0;
            a.Cout = cin;
            a.wB = wb_of(ctx, s, wd, 4, cin, cout, cout, 0);
            a.y16 = dxp.p; a.y_pstride = (int)dxp.ps;
            launch_pconv(ctx, a);
            launch_planes_to_f32(ctx, dxp.p, dxp.ps, Mi, cin, 1, dx, cin);
            return;
        }
        ConvArgs a;               // (n,h,w) is the INPUT grid of the convT, dy is (n,2h,2w,cout)
        a.x = View{dy, cout};
        a.N = n; a.H = h; a.W = w; a.Hin = 2 * h; a.Win = 2 * w; a.Cin = cout; a.Cout = cin;
        a.w = wd;
        a.y = MutView{dx, cin};
        a.Hout = h; a.Wout = w;
        a.R = 2; a.S = 2; a.pad = 0;
        launch_conv(ctx, a, impl);
    });
}
int rfi_op_convt2x2_wgrad(rfi_ctx* ctx, int impl, const float* x, const float* dy, int n, int h, int w,
                          int cin, int cout, float* dw_iohw) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        WgradArgs a;
        a.xop = View{dy, cout};
        a.yop = View{x, cin};
        a.N = n; a.H = h; a.W = w; a.Hx = 2 * h; a.Wx = 2 * w; a.Cx = cout; a.Cy = cin;
        a.R = 2; a.S = 2; a.pad = 0;
        a.tap_stride = (int64_t)cin * cout;
        a.sy = 1; a.sx = cin;
        const size_t numel = (size_t)4 * cin * cout;
        a.dw = s.get(numel);
        if (impl == IMPL_PLANES_BF16) {           // the 2x2 stride-2 weight gradient on planes (Xop = the output gradient)
            const PlaneTmp dyp = planes_of(ctx, s, dy, (int64_t)4 * n * h * w, cout);
            const PlaneTmp xp = planes_of(ctx, s, x, (int64_t)n * h * w, cin);
            PWgradArgs pa;
            pa.xop[0] = dyp.seg(); pa.nseg = 1; pa.seg_c[0] = cout;
            pa.yop = xp.seg(); pa.Cy = cin; pa.P = 1;
            pa.N = n; pa.H = h; pa.W = w; pa.Hx = 2 * h; pa.Wx = 2 * w;
            pa.R = 2; pa.S = 2; pa.pad = 0;
            pa.dw = a.dw; pa.tap_stride = a.tap_stride; pa.sy = 1; pa.sx = cin;
            pa.slab_floats = pwgrad_slab_floats(pa);
            pa.slab = s.get(pa.slab_floats);
            launch_pwgrad(ctx, pa);
        } else {
        a.slab_floats = wgrad_slab_floats(a, impl);
        a.slab = s.get(a.slab_floats);
        launch_wgrad(ctx, a, impl);
        }
        std::vector<float> lib(numel), ref(numel);
        RFI_CHECK_HIP(hipMemcpyAsync(lib.data(), a.dw, numel * 4, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        from_lib_convt(lib.data(), cin, cout, ref.data());
        RFI_CHECK_HIP(hipMemcpyAsync(dw_iohw, ref.data(), numel * 4, hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int rfi_op_roi_align(rfi_ctx* ctx, const float* x, int n, int h, int w, int c, const float* rois, int r,
                     float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, float* out) {
    return guarded([&] {
        ctx->activate();
        launch_roi_align_fwd(ctx, x, n, h, w, c, rois, r, spatial_scale, ph, pw, sampling_ratio, aligned != 0, out);
    });
}
int rfi_op_roi_align_backward(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, const float* rois, int r,
                              float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, float* dx) {
    return guarded([&] {
        ctx->activate();
        launch_roi_align_bwd(ctx, dout, n, h, w, c, rois, r, spatial_scale, ph, pw, sampling_ratio, aligned != 0, dx);
    });
}
int rfi_op_roi_align_backward_sorted(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, const float* rois_sorted, int r,
                                     float spatial_scale, int ph, int pw, int sampling_ratio, int aligned, float* dx) {
    return guarded([&] {
        ctx->activate();
        launch_roi_align_bwd_sorted(ctx, dout, n, h, w, c, rois_sorted, r, spatial_scale, ph, pw, sampling_ratio, aligned != 0, dx);
    });
}
int rfi_op_mask_targets(rfi_ctx* ctx, const uint8_t* masks, int g, int h, int w, const float* rois, int r, int ph, int pw,
                        int sampling_ratio, uint8_t* out) {
    return guarded([&] {
        ctx->activate();
        launch_mask_targets(ctx, masks, g, h, w, rois, r, ph, pw, sampling_ratio, out);
    });
}
int rfi_op_box_decode(rfi_ctx* ctx, const float* anchors, int64_t n_anchors, const float* deltas, int64_t n, float clip_h,
                      float clip_w, float* boxes) {
    return guarded([&] {
        ctx->activate();
        launch_box_decode(ctx, anchors, n_anchors, deltas, n, clip_h, clip_w, boxes);
    });
}
int rfi_op_add_inplace(rfi_ctx* ctx, float* x, const float* y, int64_t n) {
    return guarded([&] {
        ctx->activate();
        launch_add_inplace(ctx, x, y, n);
    });
}
int rfi_op_fastrcnn_loss(rfi_ctx* ctx, const float* head, int64_t rois, int num_classes, const int32_t* labels, const float* targets,
                         float beta, float* dhead, float* loss_classifier, float* loss_box_reg) {
    return guarded([&] {
        ctx->activate();
        double* ws = static_cast<double*>(ctx->alloc(rpn_loss_ws_doubles() * 8 + 16));
        struct Free { rfi_ctx* c; void* p; ~Free() { try { c->release(p); } catch (...) {} } } fr{ctx, ws};
        float* out2 = reinterpret_cast<float*>(ws + rpn_loss_ws_doubles());
        launch_fastrcnn_loss(ctx, head, rois, num_classes, labels, targets, beta, dhead, ws, out2);
        float h2[2];
        RFI_CHECK_HIP(hipMemcpyAsync(h2, out2, sizeof(h2), hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (loss_classifier) *loss_classifier = h2[0];
        if (loss_box_reg) *loss_box_reg = h2[1];
    });
}
int rfi_op_anchor_match(rfi_ctx* ctx, const float* anchors, int64_t n, const float* gt_boxes, int n_gt, float fg_iou, float bg_iou,
                        int allow_low_quality, int8_t* labels, int32_t* matched, float* targets) {
    return guarded([&] {
        ctx->activate();
        float* ws = static_cast<float*>(ctx->alloc((size_t)(n_gt > 0 ? n_gt : 1) * sizeof(float)));
        struct Free { rfi_ctx* c; void* p; ~Free() { (void)hipStreamSynchronize(c->stream); try { c->release(p); } catch (...) {} } } fr{ctx, ws};
        launch_anchor_match(ctx, anchors, n, gt_boxes, n_gt, fg_iou, bg_iou, allow_low_quality != 0, ws,
                            reinterpret_cast<signed char*>(labels), matched, targets);
    });
}
int rfi_op_nms(rfi_ctx* ctx, const float* boxes_sorted, int n, float iou_threshold, int32_t* keep_host, int* n_keep) {
    return guarded([&] {
        RFI_REQUIRE(keep_host && n_keep, "nms: null output");
        ctx->activate();
        if (n <= 0) { *n_keep = 0; return; }
        const int words = (n + 63) / 64;
        auto* mask = static_cast<unsigned long long*>(ctx->alloc((size_t)n * words * 8));
        struct Free { rfi_ctx* c; void* p; ~Free() { try { c->release(p); } catch (...) {} } } fr{ctx, mask};
        launch_nms_mask(ctx, boxes_sorted, n, iou_threshold, mask);
        std::vector<unsigned long long> h((size_t)n * words), removed(words, 0ull);
        RFI_CHECK_HIP(hipMemcpyAsync(h.data(), mask, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        int k = 0;                                   // greedy scan in score order: keep i unless a kept box suppressed it
        for (int i = 0; i < n; ++i) {
            if (removed[i / 64] >> (i % 64) & 1ull) continue;
            keep_host[k++] = i;
            const unsigned long long* row = h.data() + (size_t)i * words;
            for (int w = i / 64; w < words; ++w) removed[w] |= row[w];
        }
        *n_keep = k;
    });
}
int rfi_op_anchor_match_batched(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int32_t* anchor_count,
                                const float* gt_boxes, int images, int gt_max, const int32_t* gt_count, float fg_iou, float bg_iou,
                                int allow_low_quality, int8_t* labels, int32_t* matched, float* targets) {
    return guarded([&] {
        ctx->activate();
        float* best = static_cast<float*>(ctx->alloc((size_t)images * gt_max * 4 + 16));
        struct Free { rfi_ctx* c; void* p; ~Free() { try { c->release(p); } catch (...) {} } } fr{ctx, best};
        launch_anchor_match_batched(ctx, anchors, n, anchor_stride, anchor_count, gt_boxes, images, gt_max, gt_count, fg_iou, bg_iou,
                                    allow_low_quality != 0, best, reinterpret_cast<signed char*>(labels), matched, targets);
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // (the workspace is released on return)
    });
}
int rfi_op_nms_batched(rfi_ctx* ctx, const float* boxes_sorted, const int32_t* count, int sets, int k, float iou_threshold, uint8_t* keep) {
    return guarded([&] {
        ctx->activate();
        launch_nms_batched(ctx, boxes_sorted, count, sets, k, iou_threshold, keep);
    });
}
int rfi_op_rpn_loss(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                    const float* targets, int64_t num_sampled, float beta, float* dhead, float* loss_objectness,
                    float* loss_box) {
    return guarded([&] {
        ctx->activate();
        double* ws = static_cast<double*>(ctx->alloc(rpn_loss_ws_doubles() * 8 + 16));
        struct Free { rfi_ctx* c; void* p; ~Free() { try { c->release(p); } catch (...) {} } } fr{ctx, ws};
        float* out2 = reinterpret_cast<float*>(ws + rpn_loss_ws_doubles());
        launch_rpn_loss(ctx, head, pixels, anchors_per_pixel, reinterpret_cast<const signed char*>(labels), targets, num_sampled, beta,
                        dhead, ws, out2);
        float h2[2];
        RFI_CHECK_HIP(hipMemcpyAsync(h2, out2, sizeof(h2), hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (loss_objectness) *loss_objectness = h2[0];
        if (loss_box) *loss_box = h2[1];
    });
}
int rfi_op_rpn_loss_dev(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                        const float* targets, int64_t num_sampled, float beta, float* dhead, void* workspace, float* loss2_dev) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(workspace && loss2_dev, "rpn_loss_dev: workspace (rfi_op_rpn_loss_ws_bytes) and a 2-float device output");
        launch_rpn_loss(ctx, head, pixels, anchors_per_pixel, reinterpret_cast<const signed char*>(labels), targets, num_sampled, beta,
                        dhead, static_cast<double*>(workspace), loss2_dev);
    });
}
size_t rfi_op_rpn_loss_ws_bytes(void) { return rpn_loss_ws_doubles() * sizeof(double); }
// ---- the detector's box bookkeeping on the device (detect_sample.hip): nothing here synchronises or allocates
int rfi_op_rpn_loss_devcount(rfi_ctx* ctx, const float* head, int64_t pixels, int anchors_per_pixel, const int8_t* labels,
                             const float* targets, const int32_t* num_sampled_dev, float beta, float* dhead, void* workspace,
                             float* loss2_dev) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(workspace && loss2_dev && num_sampled_dev, "rpn_loss_devcount: workspace, a 2-float device output and the device count");
        launch_rpn_loss(ctx, head, pixels, anchors_per_pixel, reinterpret_cast<const signed char*>(labels), targets, 1, beta, dhead,
                        static_cast<double*>(workspace), loss2_dev, num_sampled_dev);
    });
}
int rfi_op_fastrcnn_loss_dev(rfi_ctx* ctx, const float* head, int64_t rois, int num_classes, const int32_t* labels, const float* targets,
                             float beta, float* dhead, void* workspace, float* loss2_dev) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(workspace && loss2_dev, "fastrcnn_loss_dev: workspace (rfi_op_rpn_loss_ws_bytes) and a 2-float device output");
        launch_fastrcnn_loss(ctx, head, rois, num_classes, labels, targets, beta, dhead, static_cast<double*>(workspace), loss2_dev);
    });
}
int rfi_op_anchor_match_batched_ws(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int32_t* anchor_count,
                                   const float* gt_boxes, int images, int gt_max, const int32_t* gt_count, float fg_iou, float bg_iou,
                                   int allow_low_quality, float* best_ws, int8_t* labels, int32_t* matched, float* targets) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(best_ws, "anchor_match_batched_ws: workspace of images x gt_max floats");
        launch_anchor_match_batched(ctx, anchors, n, anchor_stride, anchor_count, gt_boxes, images, gt_max, gt_count, fg_iou, bg_iou,
                                    allow_low_quality != 0, best_ws, reinterpret_cast<signed char*>(labels), matched, targets);
    });
}
int rfi_op_segsort_u64(rfi_ctx* ctx, uint64_t* keys, int n_segs, int stride) {
    return guarded([&] {
        ctx->activate();
        launch_segsort_u64(ctx, reinterpret_cast<unsigned long long*>(keys), n_segs, stride);
    });
}
int rfi_op_sample_keys(rfi_ctx* ctx, const int8_t* labels, int images, int n, const int32_t* count, uint64_t seed, uint32_t step,
                       uint32_t stream0, uint64_t* keys, int stride) {
    return guarded([&] {
        ctx->activate();
        launch_sample_keys(ctx, reinterpret_cast<const signed char*>(labels), images, n, count, seed, step, stream0,
                           reinterpret_cast<unsigned long long*>(keys), stride);
    });
}
int rfi_op_rpn_sample_apply(rfi_ctx* ctx, const uint64_t* keys_sorted, int images, int n, int stride, int batch, int max_pos,
                            const int8_t* labels, const float* targets, int levels, const int32_t* level_off_host,
                            int8_t* const* level_labels_host, float* const* level_targets_host, int32_t* n_sampled) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(level_off_host && level_labels_host && level_targets_host && n_sampled, "rpn_sample_apply: null table");
        launch_rpn_sample_apply(ctx, reinterpret_cast<const unsigned long long*>(keys_sorted), images, n, stride, batch, max_pos,
                                reinterpret_cast<const signed char*>(labels), targets, levels, level_off_host,
                                reinterpret_cast<signed char* const*>(level_labels_host), level_targets_host, n_sampled);
    });
}
int rfi_op_topk_keys(rfi_ctx* ctx, const float* head, int images, int pixels, int anchors_per_pixel, uint64_t* keys, int stride) {
    return guarded([&] {
        ctx->activate();
        launch_topk_keys(ctx, head, images, pixels, anchors_per_pixel, reinterpret_cast<unsigned long long*>(keys), stride);
    });
}
int rfi_op_topk_decode(rfi_ctx* ctx, const uint64_t* keys_sorted, int images, int stride, int pixels, int anchors_per_pixel, int k,
                       const float* head, const float* anchors, float clip_h, float clip_w, float min_size, float* boxes, float* scores,
                       int32_t* counts, int levels, int level) {
    return guarded([&] {
        ctx->activate();
        launch_topk_decode(ctx, reinterpret_cast<const unsigned long long*>(keys_sorted), images, stride, pixels, anchors_per_pixel, k, head,
                           anchors, clip_h, clip_w, min_size, boxes, scores, counts, levels, level);
    });
}
int rfi_op_proposals_select(rfi_ctx* ctx, const float* boxes, const float* scores, const uint8_t* keep, int images, int levels, int k,
                            int post_nms, const float* gt_boxes, int gt_max, const int32_t* gt_count, int pmax, float* props,
                            int32_t* pcount) {
    return guarded([&] {
        ctx->activate();
        launch_proposals_select(ctx, boxes, scores, keep, images, levels, k, post_nms, gt_boxes, gt_max, gt_count, pmax, props, pcount);
    });
}
int rfi_op_roi_sample(rfi_ctx* ctx, const int8_t* labels, const int32_t* pcount, int images, int pmax, int batch, int max_pos,
                      uint64_t seed, uint32_t step, uint32_t stream0, int32_t* sel, int32_t* nsel, int32_t* npos) {
    return guarded([&] {
        ctx->activate();
        launch_roi_sample(ctx, reinterpret_cast<const signed char*>(labels), pcount, images, pmax, batch, max_pos, seed, step, stream0, sel,
                          nsel, npos);
    });
}
int rfi_op_roi_compact(rfi_ctx* ctx, const int32_t* sel, const int32_t* nsel, const int32_t* npos, int images, int batch, int pmax,
                       const float* props, const int32_t* matched, const float* targets, const int32_t* gt_labels, int gt_max,
                       const int32_t* gt_base, float t1, float t2, float t3, float* rois, int32_t* cls, float* tgt, int32_t* gt,
                       int32_t* level, int32_t* img_start, float* rois_fg, float* rois_gt, int32_t* level_fg, int32_t* fg_start,
                       int32_t* counts) {
    return guarded([&] {
        ctx->activate();
        launch_roi_compact(ctx, sel, nsel, npos, images, batch, pmax, props, matched, targets, gt_labels, gt_max, gt_base, t1, t2, t3, rois,
                           cls, tgt, gt, level, img_start, rois_fg, rois_gt, level_fg, fg_start, counts);
    });
}
int rfi_op_bn_add_relu16(rfi_ctx* ctx, const uint16_t* y, const float* scale, const float* shift, const uint16_t* s, const float* s_scale,
                         const float* s_shift, int64_t m, int c, uint16_t* out) {
    return guarded([&] {
        ctx->activate();
        launch_bn_add_relu16(ctx, y, c, scale, shift, s, c, s_scale, s_shift, m, c, out, c);
    });
}
int rfi_op_relu_mask_sum16(rfi_ctx* ctx, const uint16_t* g0, const uint16_t* g1, const float* g2_f32, const uint16_t* g2_bf16,
                           int64_t g2_stride, const uint16_t* a, int64_t m, int c, uint16_t* dz) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(!(g2_f32 && g2_bf16), "relu_mask_sum16: one third term at most");
        const YRef g2 = g2_bf16 ? YRef(g2_bf16, g2_stride) : YRef(g2_f32, g2_f32 ? g2_stride : (int64_t)0);
        launch_relu_mask_sum16(ctx, g0, c, g1, c, g2, a, c, m, c, dz, c);
    });
}
int rfi_op_bn_backward16(rfi_ctx* ctx, const uint16_t* da, const uint16_t* y, int64_t m, int c, const float* gamma, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float slope, uint16_t* dy, float* dgamma,
                         float* dbeta, float* dbias) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(m > 0 && c > 0 && c % 16 == 0, "bn_backward16: channels in whole 16-channel chunks");
        Scratch s(ctx);
        float* ws = s.get(std::max(bn_bwd_ws_floats(m, c), channel_sum_ws_floats(m, c)) + 16);
        float* c12 = s.get((size_t)2 * c);
        launch_bn_bwd_reduce(ctx, YRef(da, (int64_t)c), YRef(y, (int64_t)c), m, c, scale, shift, mean, invstd, ws, c12, c12 + c, dgamma, dbeta, slope);
        launch_bn_bwd_apply(ctx, YRef(da, (int64_t)c), YRef(y, (int64_t)c), m, c, scale, shift, mean, invstd, gamma, c12, c12 + c, ws, dbias, slope, dy,
                            (int64_t)c, 1);
    });
}
int rfi_op_roi_align_ml(rfi_ctx* ctx, const float* const* maps_host, int n, int h0, int w0, int c, float scale0, const float* rois,
                        const int32_t* level, const int32_t* count_dev, int max_rois, int ph, int pw, int sampling_ratio, float* out) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(maps_host && count_dev, "roi_align_ml: null table");
        launch_roi_align_ml_fwd(ctx, maps_host, n, h0, w0, c, scale0, rois, level, count_dev, max_rois, ph, pw, sampling_ratio, out);
    });
}
int rfi_op_roi_align_ml_backward(rfi_ctx* ctx, float* const* dmaps_host, int n, int h0, int w0, int c, float scale0, const float* dout,
                                 const float* rois, const int32_t* level, const int32_t* img_start, int max_rois, int ph, int pw,
                                 int sampling_ratio) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(dmaps_host && img_start, "roi_align_ml_backward: null table");
        launch_roi_align_ml_bwd(ctx, dmaps_host, n, h0, w0, c, scale0, dout, rois, level, img_start, max_rois, ph, pw, sampling_ratio);
    });
}
// a small device -> host copy that does NOT stall the host at once: begin enqueues the copy into pinned memory and an event
// behind it; whatever the caller enqueues next runs on; end waits for the event only (not for the later work)
int rfi_readback_begin(rfi_ctx* ctx, const void* src_dev, size_t bytes) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(src_dev && bytes > 0 && bytes <= 2048, "readback_begin: 1 .. 2048 bytes");
        if (!ctx->readback_ev) RFI_CHECK_HIP(hipEventCreateWithFlags(&ctx->readback_ev, hipEventDisableTiming));
        RFI_CHECK_HIP(hipMemcpyAsync(reinterpret_cast<char*>(ctx->pinned) + 2048, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipEventRecord(ctx->readback_ev, ctx->stream));
        ctx->readback_bytes = bytes;
    });
}
int rfi_readback_end(rfi_ctx* ctx, void* dst_host, size_t bytes) {
    return guarded([&] {
        ctx->activate();
        RFI_REQUIRE(ctx->readback_ev && dst_host && bytes == ctx->readback_bytes, "readback_end: no matching readback_begin");
        RFI_CHECK_HIP(hipEventSynchronize(ctx->readback_ev));
        std::memcpy(dst_host, reinterpret_cast<char*>(ctx->pinned) + 2048, bytes);
        ctx->readback_bytes = 0;
    });
}
int rfi_op_fpn_merge(rfi_ctx* ctx, const float* lateral, const float* top, int n, int h, int w, int c, float* out) {
    return guarded([&] {
        ctx->activate();
        launch_fpn_merge_fwd(ctx, lateral, top, n, h, w, c, out);
    });
}
int rfi_op_fpn_merge_backward(rfi_ctx* ctx, const float* dout, int n, int h, int w, int c, float* dtop) {
    return guarded([&] {
        ctx->activate();
        launch_fpn_merge_bwd_top(ctx, dout, n, h, w, c, dtop);
    });
}
int rfi_op_bn_stats(rfi_ctx* ctx, const float* y, int64_t m, int c, float* mean, float* var_biased) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        float* ws = s.get(bn_stats_ws_floats(c));
        float* tmp = s.get((size_t)6 * c);
        std::vector<float> ones((size_t)c, 1.0f), zeros((size_t)c, 0.0f);
        RFI_CHECK_HIP(hipMemcpyAsync(tmp, ones.data(), c * 4, hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipMemcpyAsync(tmp + c, zeros.data(), c * 4, hipMemcpyHostToDevice, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        launch_bn_stats(ctx, y, m, c, ws);
        launch_bn_finalize(ctx, ws, m, c, tmp, tmp + c, nullptr, nullptr, 0, mean, tmp + 2 * c, tmp + 3 * c,
                           tmp + 4 * c, var_biased);
    });
}

int rfi_op_bn_relu_pool(rfi_ctx* ctx, const float* y, int n, int h, int w, int c, const float* scale,
                        const float* shift, float* skip, float* pooled) {
    return guarded([&] {
        ctx->activate();
        launch_bn_relu_pool(ctx, y, n, h, w, c, scale, shift, MutView{skip, c}, pooled);
    });
}
int rfi_op_pool_bwd_merge(rfi_ctx* ctx, const float* y, int n, int h, int w, int c, const float* scale,
                          const float* shift, const float* dskip, const float* dpool, float* da) {
    return guarded([&] {
        ctx->activate();
        launch_pool_bwd_merge(ctx, y, n, h, w, c, scale, shift, View{dskip, c}, dpool, da);
    });
}
int rfi_op_bn_relu_backward(rfi_ctx* ctx, const float* y, int64_t m, int c, const float* gamma,
                            const float* beta, float* da_inout, float* dgamma, float* dbeta, float* dbias) {
    return guarded([&] {
        ctx->activate();
        Scratch s(ctx);
        size_t wsf = bn_stats_ws_floats(c);
        if (bn_bwd_ws_floats(m, c) > wsf) wsf = bn_bwd_ws_floats(m, c);
        float* ws = s.get(wsf);
        float* t = s.get((size_t)6 * c);     // mean | invstd | scale | shift | c1 | c2
        launch_bn_stats(ctx, y, m, c, ws);
        launch_bn_finalize(ctx, ws, m, c, gamma, beta, nullptr, nullptr, 0, t, t + c, t + 2 * c, t + 3 * c,
                           nullptr);
        launch_bn_bwd_reduce(ctx, da_inout, y, m, c, t + 2 * c, t + 3 * c, t, t + c, ws, t + 4 * c, t + 5 * c,
                             dgamma, dbeta);
        launch_bn_bwd_apply(ctx, da_inout, y, m, c, t + 2 * c, t + 3 * c, t, t + c, gamma, t + 4 * c, t + 5 * c,
                            ws, dbias);
    });
}

}  // extern "C"
