// Shared host-side declarations for librfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/rfi_hip.h"

namespace rfi {

// ---------------------------------------------------------------- errors
void set_last_error(const std::string& msg);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define RFI_CHECK_HIP(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            throw ::rfi::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +   \
                               " (" __FILE__ ":" + std::to_string(__LINE__) + ")");         \
    } while (0)

#define RFI_REQUIRE(cond, msg)                                                              \
    do {                                                                                    \
        if (!(cond)) throw ::rfi::Error(std::string(msg));                                  \
    } while (0)

// run `body`, translate exceptions into the C-ABI int + rfi_last_error()
template <typename F>
static inline int guarded(F&& body) {
    try {
        body();
        return 0;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return 1;
    } catch (...) {
        set_last_error("unknown C++ exception");
        return 1;
    }
}

// ---------------------------------------------------------------- profiling families
enum Family : int {
    FAM_CONV_MFMA = 0,    // implicit-GEMM conv fwd / dgrad / convT (MFMA)
    FAM_WGRAD_MFMA,       // implicit-GEMM weight gradient (MFMA)
    FAM_CONV_DIRECT,      // VALU direct conv (stem, odd shapes)
    FAM_BN,               // batch-norm statistics / finalize / backward
    FAM_ELEMWISE,         // pool, head, loss, relayout, copies
    FAM_REDUCE,           // partial-slab reductions
    FAM_OPTIM,            // grad-norm, clip, Adam
    FAM_PREPROCESS,
    FAM_METRICS,
    FAM_COMM,
    FAM_COUNT
};

struct FamilyStat {
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};

struct PendingEvent {
    hipEvent_t a, b;
    int family;
    double flops, bytes;
    std::string label;
};

struct LaunchRecord {        // one profiled launch, in stream order
    int family;
    double ms, flops, bytes;
    std::string label;
};

}  // namespace rfi

// ---------------------------------------------------------------- context
struct rfi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    std::unordered_map<void*, size_t> allocs;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // profiling
    bool profiling = false;
    rfi::FamilyStat fam[rfi::FAM_COUNT];
    std::vector<rfi::PendingEvent> pending;
    std::vector<rfi::LaunchRecord> launches;
    std::vector<hipEvent_t> event_pool;
    // pinned scratch for small D2H readbacks
    float* pinned = nullptr;
    hipEvent_t readback_ev = nullptr;        // rfi_readback_begin / _end: the upper half of `pinned` + this event
    size_t readback_bytes = 0;
    // grow-only device scratch (per-patch min/max words of the preprocessing kernels)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    void* get_scratch(size_t bytes);
    // 4 KiB of zeros in HBM: where the LDS-DMA staging of the plane kernels points lanes whose halo pixel
    // lies outside the image (a DMA lane cannot be zero-filled conditionally)
    void* zero_page = nullptr;
    // side stream: weight-gradient GEMMs of the backward pass run here, next to the main stream's
    // dgrad / batch-norm chain (model.cpp); `stream` is swapped to it for those launches
    hipStream_t main_stream = nullptr, side_stream = nullptr;
    hipEvent_t fork_ev = nullptr;
    std::vector<hipEvent_t> fork_ring;   // stop events of producer kernels (launch_*(..., done)): one per side launch of a pass
    size_t fork_ring_used = 0;
    std::vector<hipEvent_t> side_done;   // ring, one per in-flight side launch
    bool overlap = true;
    // RCCL.  The gradient exchange is BUCKETED: contiguous ranges of the flat gradient buffer are all-reduced on
    // comm_stream as soon as the backward pass has finished them (reverse layer order), next to the remaining
    // backward kernels.  comm_emulate > 1 (rfi_comm_emulate, tests on one GPU): no communicator; every "all-reduce"
    // multiplies its range by comm_emulate, so with grad_scale = 1 / comm_emulate a step must reproduce the plain
    // step bit for bit iff every element is exchanged exactly once and in order with its producers and consumers.
    void* nccl_comm = nullptr;
    int rank = 0, world = 1;
    int comm_emulate = 0;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> bucket_ev;   // pool: (main, side, done) events of the buckets of one step
    size_t bucket_ev_used = 0;
    bool exchange_active() const { return (nccl_comm && world > 1) || comm_emulate > 1; }
    int exchange_world() const { return comm_emulate > 1 ? comm_emulate : world; }

    void* alloc(size_t bytes);
    void release(void* p);
    void activate() const;   // hipSetDevice
    hipEvent_t get_event();
    void drain_profile();
};

namespace rfi {

// RAII: brackets one kernel launch with events when profiling is on
struct ProfScope {
    rfi_ctx* c;
    int fam;
    hipEvent_t a = nullptr, b = nullptr;
    double fl, by;
    std::string label;
    ProfScope(rfi_ctx* ctx, int family, double flops = 0, double bytes = 0, std::string lbl = std::string())
        : c(ctx), fam(family), fl(flops), by(bytes) {
        if (c->profiling) {
            label = std::move(lbl);
            a = c->get_event();
            b = c->get_event();
            (void)hipEventRecord(a, c->stream);
            c->fam[fam].flops += flops;
            c->fam[fam].bytes += bytes;
        }
    }
    ~ProfScope() {
        if (c->profiling) {
            (void)hipEventRecord(b, c->stream);
            c->fam[fam].launches += 1;
            c->pending.push_back({a, b, fam, fl, by, std::move(label)});
        }
    }
};

static inline void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(std::string("launch of ") + what + " failed: " + hipGetErrorString(e));
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// "do once per device": function attributes such as hipFuncAttributeMaxDynamicSharedMemorySize are
// per device, and a process may hold contexts on several GPUs (and host threads)
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t done[4] = {0, 0, 0, 0};     // up to 256 device ordinals
    template <typename F>
    void run(int device, F&& f) {
        std::lock_guard<std::mutex> g(mu);
        const unsigned d = (unsigned)device & 255u;
        if (done[d >> 6] >> (d & 63) & 1u) return;
        f();
        done[d >> 6] |= uint64_t(1) << (d & 63);
    }
};

}  // namespace rfi
