// C ABI (include/dau_conv.h) and pass orchestration.
//
// Sequencing follows what the reference does between its op boundary and its kernels
// (plugins/tensorflow/src/dau_conv_op.cpp:150-324, dau_conv_grad_op.cpp:115-318,
//  src/dau_conv/base_dau_conv_layer.cu:15-127 Forward_gpu, :130-363 Backward_gpu), minus
// the per-call handle creation, side streams, host syncs and workspace re-allocation.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "dau_common.hpp"
#include "dau_tiled.hpp"

using namespace dau;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define DAU_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(DAU_INTERNAL, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off = align_up(off + count * sizeof(T));
        return p;
    }
};

// image n0 of an activation tensor whose elements are esize bytes (float32, or bfloat16 behind the float* of the ABI)
inline const float* slab_ptr(const float* base, size_t elements, size_t esize) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + elements * esize);
}
inline float* slab_ptr(float* base, size_t elements, size_t esize) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + elements * esize);
}

// unit_testing edge rule of the numpy oracle (dau_conv_test.py:110-136)
int edge_disabled(int size) {
    if (size >= 64) return size % 64 == 0;
    if (size >= 32) return size % 32 == 0;
    if (size >= 16) return size % 16 == 0;
    if (size >= 8) return size % 8 == 0;
    return 0;
}

}  // namespace

// The tiled kernels of one offset bucket.  A plan holds one set per bucket up to the static one (the bucket
// max_kernel_size allows); which set runs is decided per call from the actual max|mu| (see run_sets below).
struct BucketSet {
    int bucket = 0;
    bool fwd_ok = false, dot_ok = false;
    TiledConfig tiled_fwd;   // gather-sum y  : S -> F
    TiledConfig tiled_dx;    // gather-sum dx : F -> S
    TiledDotConfig tiled_dot;
    // DAU_FLAG_DENSE_BF16, bucket 4 only: the gather-sum passes run as a densified bf16 implicit GEMM (k_dense_bf16.hip)
    bool dense_ok = false;
    // ... and, from three units on (DAU_FLAG_DENSE_WGRAD_NEVER / _ALWAYS: never / from one unit on), the parameter gradients as
    // dense correlations on the same matrix cores (k_dense_wgrad.hip)
    bool wgrad_ok = false;
    WgradConfig wgrad;
    DenseConfig dense_fwd, dense_dx;
    // ... and the same two forms for calls whose offsets lie within +-3 (7 x 7 taps / displacements instead of 9 x 9; the call's
    // device guard decides between the two: (-1, 3] and (3, 4])
    bool dense3_ok = false, wgrad3_ok = false;
    DenseConfig dense3_fwd, dense3_dx;
    WgradConfig wgrad3;
    // bucket 4 only: calls whose offsets lie within +-2 / +-3 / +-4 run the gather-sum passes as the two-limb f16 GEMM of that
    // radius (k_dense_split.hip: fp32 accuracy; index r - 2); the call's device guard decides, everything else takes the exact
    // kernels.  Which radii a plan holds: those that pay for its unit count (split_pays below), all / none by flag.
    bool split_ok[3] = {false, false, false};
    DenseConfig split_fwd[3], split_dx[3];
    bool any_split() const { return split_ok[0] || split_ok[1] || split_ok[2]; }
    // Batch slabs.  Every pass stages its whole input before it gathers; where that staged copy would exceed the workspace
    // budget (DAU_WORKSPACE_BUDGET_GB at plan creation, default 12: only the 512 x 512 configurations get there) the pass
    // runs slab by slab over the batch -- the configs above are made for `slab_*` images, the passes loop -- so that the
    // workspace holds one slab's staged copy.  Forward and dx are per-image; the parameter sums add up over the slabs.
    int slab_gather = 0, slab_dot = 0;     // images per slab (the whole batch unless the budget says otherwise)
};
constexpr int kBuckets[] = {4, 8, 16, 18, 20, 24, 32};
constexpr int kNumBuckets = 7;

// the two-limb f16 dense gather-sum, one set of entry points per offset radius (index r - 2)
struct SplitFns {
    bool (*configure)(int, int, int, int, int, int, int, int, bool, DenseConfig*);
    size_t (*workspace_bytes)(const DenseConfig&);
    void (*init)(const DenseConfig&);
    void (*prepare)(hipStream_t, const DenseConfig&, const float*, const float*, bool, const UnitRef*, void*, const Guard&);
    void (*run)(hipStream_t, const DenseConfig&, float*, void*, const Guard&);
};
const SplitFns kSplit[3] = {
    {s2::split_gather_configure, s2::split_gather_workspace_bytes, s2::split_gather_init, s2::split_gather_prepare, s2::split_gather_run},
    {s3::split_gather_configure, s3::split_gather_workspace_bytes, s3::split_gather_init, s3::split_gather_prepare, s3::split_gather_run},
    {s4::split_gather_configure, s4::split_gather_workspace_bytes, s4::split_gather_init, s4::split_gather_prepare, s4::split_gather_run},
};
// Does the dense form of radius r pay against the exact gather?  MFMA work per pass in fp32-rate MAC units: the dense GEMM runs
// (2r+1)^2 taps x 3 limb products at 16x the fp32 rate over the PADDED tile (8-row / 8-column blocks, 128 output channels, 16 input
// channels) plus its staging, at ~55 % of the f16 roof; the exact gather 4 MACs per live unit at ~60 % of the fp32 roof on maps
// of 32 pixels and more, ~45 % on smaller ones (measured, same box, gather + staging per pass against the exact gather: NS radius 3
// 5.4 + 0.5 against 8.9 ms at four units, radius 4 8.5 + 0.5 against 9.4; radius 2 2.9 + 0.5 against 5.4 at two units; C1 27 x 27
// radius 3 0.36 / 0.42 + 0.07 against 0.60 / 0.51).  On whole tiles radius 2 pays from two units per channel pair on, radius 3 from
// three, radius 4 from four.
bool split_pays(int r, int Cin, int Cout, int G_live, int H, int W) {
    const double taps = (2.0 * r + 1) * (2.0 * r + 1);
    const double hp = (H + 7) / 8 * 8, wp = (W + 7) / 8 * 8, fp = (Cout + 127) / 128 * 128, sp = (Cin + 15) / 16 * 16;
    const double dense = hp * wp * (fp * sp * taps * 3.0 / 16.0 / 0.55 + sp * 430.0);
    const double exact = (double)H * W * Cout * Cin * G_live * 4.0 / ((H < 32 || W < 32) ? 0.45 : 0.60);
    return dense < 1.1 * exact;
}

struct dau_conv_plan {
    dau_conv_desc d;
    Shape sh;
    int bucket;        // static offset bucket R: the largest displacement max_kernel_size allows
    int blur_k;        // prefilter support
    int drop_col, drop_row;
    int algo_fwd, algo_bwd;
    int nsets = 0;             // bucket sets, ascending; sets[nsets - 1] is the static bucket
    BucketSet sets[kNumBuckets];
    bool dynamic = false;      // pick the set from the actual offsets (tiled kernels only)
    // pinned host mirror (HostStatus, dau_common.hpp): the status of the most recent completed call -- the offset-bucket hint
    // of the next call; a stale or torn value only costs speed, never correctness (the device-side guards decide which set
    // really runs) -- and the STICKY record of the worst status any completed call has left since the last report.
    HostStatus* host_status = nullptr;
    // dynamic-LDS limits raised on these devices (bit d: done on device d; the first call on a device does it behind the lock)
    mutable std::atomic<unsigned long long> attrs_devices{0};
    mutable std::mutex attrs_mutex;
    const BucketSet& top() const { return sets[nsets - 1]; }
    long units() const { return (long)sh.S * sh.G * sh.F; }
    // optional benchmark timing (dau_conv_profile_begin/_end); mutable because the passes take a const plan
    mutable bool profiling = false;
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[DAU_PROFILE_SLOTS];
    mutable size_t prof_used[DAU_PROFILE_SLOTS] = {0, 0, 0};
    mutable int prof_passes[DAU_PROFILE_SLOTS] = {0, 0, 0};   // passes (a pass may take several window launches)
};

namespace {

// event bracket around one dominant kernel launch when profiling is on
struct ProfScope {
    const dau_conv_plan* p;
    int slot;
    hipStream_t st;
    hipEvent_t stop = nullptr;
    ProfScope(const dau_conv_plan* plan, int slot_, hipStream_t st_) : p(plan), slot(slot_), st(st_) {
        if (!p->profiling) return;
        auto& pool = p->prof_events[slot];
        size_t& used = p->prof_used[slot];
        if (used == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.emplace_back(a, b);
        }
        (void)hipEventRecord(pool[used].first, st);
        stop = pool[used].second;
        ++used;
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, st);
    }
};

// The bucket sets one call enqueues.  Without a hint (first call, dynamic selection off) it is the static set, unguarded.
// With a hint -- max|mu| of the most recent completed call, read from pinned host memory without a sync -- it is the
// smallest set that covers the hint, guarded by (-1, R_hint], followed by the static set guarded by (R_hint, inf): the
// device decides between them from the max|mu| of THIS call, so results never depend on the hint.  This replaces the
// reference's blocking amax + D2H copy per call (dau_conv_op.cpp:223-253) and makes a layer with a large
// max_kernel_size but small offsets run the small-offset kernels (the reference's tests rely on that,
// dau_conv_test.py:433,436).
struct Candidate {
    const BucketSet* set;
    Guard guard;
    bool r3 = false;          // the set's radius-3 dense bf16 member (sets[0] only)
    int split_r = 0;          // 2, 3, 4: the set's two-limb f16 dense member of that radius (sets[0], gather-sum passes only)
};

// dau_conv_last_status has just reported the whole mirror: forget it, sticky record included.
void clear_host_status(const dau_conv_plan* p) {
    if (!p->host_status) return;
    volatile unsigned* h = reinterpret_cast<volatile unsigned*>(p->host_status);
    h[2] = 0u; h[1] = 0u; h[0] = 0u; h[4] = 0u; h[5] = 0u;
}
// dau_conv_check_status has just reported the status of ONE workspace's call: forget the "most recent call" part, and of the sticky
// record only what is this very report (the same out-of-range maximum, the NaN flag if this call had one) -- it may also hold the
// not-yet-reported error of another layer or stream that shares the plan, which dau_conv_last_status must still see.
void clear_reported_status(const dau_conv_plan* p, unsigned max_bits, bool nan_seen) {
    if (!p->host_status) return;
    volatile unsigned* h = reinterpret_cast<volatile unsigned*>(p->host_status);
    h[2] = 0u; h[1] = 0u; h[0] = 0u;
    if (h[4] == max_bits) h[4] = 0u;
    if (nan_seen) h[5] = 0u;
}

// pass_kind: 0 = gather-sum (needs fwd_ok), 1 = gather-dot (needs dot_ok)
// A plan with DAU_FLAG_DENSE_BF16 has one member whose ARITHMETIC differs (bucket 4: bf16 products): that member is enqueued,
// guarded by (-1, 4], on every call, hint or no hint, so that which arithmetic a call gets depends on its own offsets only.
// The dense forms exist for |mu| <= 3 as well (49 taps instead of 81): that member goes first, guarded by (-1, 3].
constexpr int kMaxCandidates = 6;
int pick_candidates(const dau_conv_plan* p, const Status* dev_status, int pass_kind, Candidate out[kMaxCandidates]) {
    const BucketSet* top = &p->top();
    out[0] = Candidate{top, Guard{nullptr, 0.0f, 0.0f}};
    if (!p->dynamic) return 1;
    const BucketSet& s0 = p->sets[0];
    const bool d3 = pass_kind == 0 ? s0.dense3_ok : s0.wgrad3_ok;
    const bool split = pass_kind == 0 && s0.any_split();
    if (p->nsets < 2 && !d3 && !split) return 1;
    const BucketSet* dense = (pass_kind == 0 ? s0.dense_ok : s0.wgrad_ok) ? &s0 : nullptr;
    const BucketSet* hinted = nullptr;
    if (p->host_status && p->nsets >= 2) {
        const volatile unsigned* h = reinterpret_cast<const volatile unsigned*>(p->host_status);
        float mx = -1.0f;
        if (h[2] == 1u && h[1] == 0u) {
            const unsigned bits = h[0];
            std::memcpy(&mx, &bits, sizeof(float));
        }
        if (mx >= 0.0f)
            for (int i = 0; i + 1 < p->nsets && !hinted; ++i) {
                const BucketSet& b = p->sets[i];
                if (mx <= (float)b.bucket && (pass_kind == 0 ? b.fwd_ok : b.dot_ok)) hinted = &b;
            }
    }
    int n = 0;
    float lo = -1.0f;
    // the members with an arithmetic of their own are candidates of EVERY call (hint or no hint), smallest radius first: which
    // arithmetic a call gets depends on its own offsets only
    if (split)
        for (int r = 2; r <= 4; ++r)
            if (s0.split_ok[r - 2]) { out[n] = Candidate{&s0, Guard{dev_status, lo, (float)r}}; out[n++].split_r = r; lo = (float)r; }
    if (d3) { out[n] = Candidate{&s0, Guard{dev_status, lo, 3.0f}}; out[n++].r3 = true; lo = 3.0f; }
    if (dense && hinted != dense && dense != top) { out[n++] = Candidate{dense, Guard{dev_status, lo, (float)dense->bucket}}; lo = (float)dense->bucket; }
    if (hinted && lo < (float)hinted->bucket) { out[n++] = Candidate{hinted, Guard{dev_status, lo, (float)hinted->bucket}}; lo = (float)hinted->bucket; }
    if (n == 0) return 1;                                    // no hint, nothing dense: the static set, unguarded
    out[n++] = Candidate{top, Guard{dev_status, lo, INFINITY}};
    return n;
}

// first call of a plan on a device: raise the dynamic-LDS limit of every kernel its sets can launch (function attributes are
// per device, hence not at plan creation, which must also work without a device).  Plans are shared between threads (the
// TF plan cache hands one plan to every Compute of a shape): the first call on a device does this behind the plan's lock.
int ensure_attrs(const dau_conv_plan* p) {
    int dev = 0;
    DAU_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (p->attrs_devices.load(std::memory_order_acquire) & bit) return DAU_OK;
    std::lock_guard<std::mutex> lock(p->attrs_mutex);
    if (p->attrs_devices.load(std::memory_order_relaxed) & bit) return DAU_OK;
    (void)hipGetLastError();
    for (int i = 0; i < p->nsets; ++i) {
        if (p->sets[i].fwd_ok) { tiled_gather_init(p->sets[i].tiled_fwd); tiled_gather_init(p->sets[i].tiled_dx); }
        if (p->sets[i].dot_ok) tiled_dot_init(p->sets[i].tiled_dot);
        if (p->sets[i].dense_ok) { r4::dense_gather_init(p->sets[i].dense_fwd); r4::dense_gather_init(p->sets[i].dense_dx); }
        if (p->sets[i].wgrad_ok) r4::dense_wgrad_init(p->sets[i].wgrad);
        if (p->sets[i].dense3_ok) { r3::dense_gather_init(p->sets[i].dense3_fwd); r3::dense_gather_init(p->sets[i].dense3_dx); }
        if (p->sets[i].wgrad3_ok) r3::dense_wgrad_init(p->sets[i].wgrad3);
        for (int r = 0; r < 3; ++r)
            if (p->sets[i].split_ok[r]) { kSplit[r].init(p->sets[i].split_fwd[r]); kSplit[r].init(p->sets[i].split_dx[r]); }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return fail(DAU_INTERNAL, "raising the dynamic-LDS limit of the bucket-%d kernels failed: %s", p->sets[i].bucket,
                        hipGetErrorString(e));
    }
    p->attrs_devices.fetch_or(bit, std::memory_order_release);
    return DAU_OK;
}

struct FwdWs {
    Status* status;
    float* filters;
    UnitRef* table;
    float* xb;          // direct: blurred input, NCHW
    void* tiled;        // tiled: staged planes + packed units
    size_t bytes;
};

FwdWs carve_forward(const dau_conv_plan* p, void* ws) {
    Carver c(ws);
    FwdWs w{};
    w.status = c.take<Status>(1);
    w.filters = c.take<float>(kFilterFloats);
    w.table = c.take<UnitRef>(p->units());
    if (p->algo_fwd == DAU_ALGO_TILED) {
        size_t need = 0;
        for (int i = 0; i < p->nsets; ++i) {
            if (p->sets[i].fwd_ok) need = std::max(need, tiled_gather_workspace_bytes(p->sets[i].tiled_fwd));
            if (p->sets[i].dense_ok) need = std::max(need, r4::dense_gather_workspace_bytes(p->sets[i].dense_fwd));
            if (p->sets[i].dense3_ok) need = std::max(need, r3::dense_gather_workspace_bytes(p->sets[i].dense3_fwd));
            for (int r = 0; r < 3; ++r)
                if (p->sets[i].split_ok[r]) need = std::max(need, kSplit[r].workspace_bytes(p->sets[i].split_fwd[r]));
        }
        w.tiled = c.take<char>(need);
    } else {
        w.xb = c.take<float>((size_t)p->sh.N * p->sh.S * p->sh.H * p->sh.W);
    }
    w.bytes = c.off;
    return w;
}

struct BwdWs {
    Status* status;
    float* filters;
    UnitRef* table_bare;   // [S][G][F], w = 1  (parameter gradients)
    UnitRef* table_t;      // [F][G][S], negated offsets, times w  (input gradient)
    float* r4;             // [4][S][G][F]
    float* xk4;            // direct: [N*S][4][H][W]
    float* eb;             // direct: blurred error, NCHW
    void* tiled_dx;
    void* tiled_dot;
    size_t bytes;
};

BwdWs carve_backward(const dau_conv_plan* p, void* ws) {
    Carver c(ws);
    BwdWs w{};
    const Shape& s = p->sh;
    w.status = c.take<Status>(1);
    w.filters = c.take<float>(kFilterFloats);
    w.table_bare = c.take<UnitRef>(p->units());
    w.table_t = c.take<UnitRef>(p->units());
    w.r4 = c.take<float>(kNumK * p->units());
    if (p->algo_bwd == DAU_ALGO_TILED) {
        size_t need = 0;
        for (int i = 0; i < p->nsets; ++i)
            if (p->sets[i].dot_ok) need = std::max(need, tiled_dot_workspace_bytes(p->sets[i].tiled_dot));
        for (int i = 0; i < p->nsets; ++i) {
            if (p->sets[i].wgrad_ok) need = std::max(need, r4::dense_wgrad_workspace_bytes(p->sets[i].wgrad));
            if (p->sets[i].wgrad3_ok) need = std::max(need, r3::dense_wgrad_workspace_bytes(p->sets[i].wgrad3));
        }
        w.tiled_dot = c.take<char>(need);
    } else {
        w.xk4 = c.take<float>((size_t)kNumK * s.N * s.S * s.H * s.W);
    }
    if (p->algo_fwd == DAU_ALGO_TILED) {
        size_t need = 0;
        for (int i = 0; i < p->nsets; ++i) {
            if (p->sets[i].fwd_ok) need = std::max(need, tiled_gather_workspace_bytes(p->sets[i].tiled_dx));
            if (p->sets[i].dense_ok) need = std::max(need, r4::dense_gather_workspace_bytes(p->sets[i].dense_dx));
            if (p->sets[i].dense3_ok) need = std::max(need, r3::dense_gather_workspace_bytes(p->sets[i].dense3_dx));
            for (int r = 0; r < 3; ++r)
                if (p->sets[i].split_ok[r]) need = std::max(need, kSplit[r].workspace_bytes(p->sets[i].split_dx[r]));
        }
        w.tiled_dx = c.take<char>(need);
    } else {
        w.eb = c.take<float>((size_t)s.N * s.F * s.H * s.W);
    }
    w.bytes = c.off;
    return w;
}

}  // namespace

extern "C" {

int dau_conv_abi_version(void) { return DAU_CONV_ABI_VERSION; }

#ifndef DAU_BUILD_ID
#define DAU_BUILD_ID "unknown"
#endif
const char* dau_conv_build_id(void) { return DAU_BUILD_ID; }

const char* dau_conv_last_error(void) { return g_last_error.c_str(); }

int dau_conv_filter_support(float sigma) { return 2 * (int)std::ceil(5.0f * sigma) + 1; }   // base_dau_conv_layer.cpp:146

int dau_conv_plan_create(const dau_conv_desc* desc, dau_conv_plan** plan_out) {
    if (!desc || !plan_out) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (desc->struct_size != (int32_t)sizeof(dau_conv_desc))
        return fail(DAU_INVALID_ARGUMENT, "dau_conv_desc.struct_size %d != %zu", desc->struct_size, sizeof(dau_conv_desc));
    if (desc->batch < 1 || desc->in_channels < 1 || desc->out_channels < 1 || desc->units_per_channel < 1 ||
        desc->height < 1 || desc->width < 1)
        return fail(DAU_INVALID_ARGUMENT, "all of N,S,F,G,H,W must be >= 1");
    if (desc->number_units_ignore < 0 || desc->number_units_ignore >= desc->units_per_channel)
        return fail(DAU_INVALID_ARGUMENT, "number_units_ignore must be in [0, G)");
    if (desc->max_kernel_size < 3 || desc->max_kernel_size % 2 == 0)
        return fail(DAU_INVALID_ARGUMENT, "kernel_size must be odd and >= 3");
    // offset bucket from the largest displacement the attrs allow (dau_conv_op.cpp:236-253)
    const int half = desc->max_kernel_size / 2;
    int bucket;
    if (half <= 4) bucket = 4;
    else if (half <= 8) bucket = 8;
    else if (half <= 16) bucket = 16;
    else if (half <= 18) bucket = 18;
    else if (half <= 20) bucket = 20;
    else if (half <= 24) bucket = 24;
    else if (half <= 32) bucket = 32;
    else
        return fail(DAU_INVALID_ARGUMENT,
                    "DAUConv: offsets larger than the 32 px the kernels stage (set max_kernel_size <= 65)");
    if (!(desc->sigma_hint > 0.0f))  // DAU_CHECK(sigma > 0) base_dau_conv_layer.cpp:143
        return fail(DAU_FAILED_PRECONDITION, "Must use sigma > 0 - initialize it with appropriate value");
    const int blur_k = dau_conv_filter_support(desc->sigma_hint);
    if (blur_k > kMaxBlurSupport)
        return fail(DAU_INVALID_ARGUMENT, "sigma %.3f needs a %dx%d prefilter; at most %dx%d is supported", desc->sigma_hint,
                    blur_k, blur_k, kMaxBlurSupport, kMaxBlurSupport);
    if (desc->algo < DAU_ALGO_AUTO || desc->algo > DAU_ALGO_TILED) return fail(DAU_INVALID_ARGUMENT, "unknown algo");

    dau_conv_plan* p = new (std::nothrow) dau_conv_plan();
    if (!p) return fail(DAU_INTERNAL, "out of host memory");
    p->d = *desc;
    p->sh = Shape{desc->batch, desc->in_channels, desc->out_channels, desc->units_per_channel, desc->height, desc->width};
    p->bucket = bucket;
    p->blur_k = blur_k;
    const bool ut = desc->flags & DAU_FLAG_UNIT_TESTING;
    p->drop_col = ut ? edge_disabled(desc->width) : 0;
    p->drop_row = ut ? edge_disabled(desc->height) : 0;

    const Shape& s = p->sh;
    const bool bf16 = (desc->flags & DAU_FLAG_IO_BF16) != 0;
    const char* budget_env = getenv("DAU_WORKSPACE_BUDGET_GB");
    const double budget_bytes = (budget_env ? atof(budget_env) : 12.0) * 1e9;
    for (int b : kBuckets) {
        if (b > bucket) break;
        BucketSet& bs = p->sets[p->nsets++];
        bs.bucket = b;
        // slab candidates: the whole batch, then its even divisors (image pairs stay together), largest first
        const bool want_dense = (desc->flags & DAU_FLAG_DENSE_BF16) && desc->algo != DAU_ALGO_DIRECT;
        const bool split_forced = (desc->flags & DAU_FLAG_DENSE_SPLIT_F16) != 0;
        const bool split_allowed = !(desc->flags & (DAU_FLAG_NO_DENSE_SPLIT | DAU_FLAG_DENSE_BF16)) && desc->algo != DAU_ALGO_DIRECT &&
                                   DAU_TUNE_INT("DAU_DENSE_SPLIT", 1) != 0;
        auto configure_gather = [&](int n) {
            bs.fwd_ok = tiled_gather_configure(n, s.S, s.F, s.G, s.H, s.W, b, blur_k, bf16, &bs.tiled_fwd) &&
                        tiled_gather_configure(n, s.F, s.S, s.G, s.H, s.W, b, blur_k, bf16, &bs.tiled_dx);
            bs.dense_ok = want_dense && bs.fwd_ok &&
                          r4::dense_gather_configure(n, s.S, s.F, s.G, s.H, s.W, b, blur_k, bf16, &bs.dense_fwd) &&
                          r4::dense_gather_configure(n, s.F, s.S, s.G, s.H, s.W, b, blur_k, bf16, &bs.dense_dx);
            bs.dense3_ok = bs.dense_ok && b == 4 && DAU_TUNE_INT("DAU_DENSE_R3", 1) != 0 &&
                           r3::dense_gather_configure(n, s.S, s.F, s.G, s.H, s.W, 3, blur_k, bf16, &bs.dense3_fwd) &&
                           r3::dense_gather_configure(n, s.F, s.S, s.G, s.H, s.W, 3, blur_k, bf16, &bs.dense3_dx);
            size_t need = 0;
            if (bs.fwd_ok) need = std::max(tiled_gather_workspace_bytes(bs.tiled_fwd), tiled_gather_workspace_bytes(bs.tiled_dx));
            for (int r = 2; r <= 4; ++r) {
                bool& ok = bs.split_ok[r - 2];
                const int g_live = s.G - desc->number_units_ignore;
                ok = split_allowed && bs.fwd_ok && b == 4 &&
                     (split_forced || (split_pays(r, s.S, s.F, g_live, s.H, s.W) && split_pays(r, s.F, s.S, g_live, s.H, s.W))) &&
                     kSplit[r - 2].configure(n, s.S, s.F, s.G, s.H, s.W, r, blur_k, bf16, &bs.split_fwd[r - 2]) &&
                     kSplit[r - 2].configure(n, s.F, s.S, s.G, s.H, s.W, r, blur_k, bf16, &bs.split_dx[r - 2]);
                if (ok) need = std::max(need, std::max(kSplit[r - 2].workspace_bytes(bs.split_fwd[r - 2]), kSplit[r - 2].workspace_bytes(bs.split_dx[r - 2])));
            }
            if (bs.dense_ok) need = std::max(need, std::max(r4::dense_gather_workspace_bytes(bs.dense_fwd), r4::dense_gather_workspace_bytes(bs.dense_dx)));
            if (bs.dense3_ok) need = std::max(need, std::max(r3::dense_gather_workspace_bytes(bs.dense3_fwd), r3::dense_gather_workspace_bytes(bs.dense3_dx)));
            return need;
        };
        auto configure_dot = [&](int n) {
            Shape sn = s; sn.N = n;
            bs.dot_ok = tiled_dot_configure(sn, b, blur_k, bf16, desc->number_units_ignore, &bs.tiled_dot);
            return bs.dot_ok ? tiled_dot_workspace_bytes(bs.tiled_dot) : (size_t)0;
        };
        auto pick_slab = [&](auto&& configure) {
            int chosen = s.N;
            for (int n = s.N; n >= 2; --n) {
                if (s.N % n || (n != s.N && (n & 1))) continue;
                chosen = n;
                if ((double)configure(n) <= budget_bytes) break;
            }
            configure(chosen);
            return chosen;
        };
        bs.slab_gather = pick_slab(configure_gather);
        bs.slab_dot = pick_slab(configure_dot);
        {
            // dense parameter gradients: the bf16 layer's bucket-4 set, whole batch in one slab, three or more units (its cost
            // does not depend on the unit count: 15.3 ms at the north-star size against 16.0 ms for the exact gather-dot of a
            // four-unit block, 9.7 ms of two units); DAU_FLAG_DENSE_WGRAD_NEVER / _ALWAYS: never / from one unit on
            const int min_units = (desc->flags & DAU_FLAG_DENSE_WGRAD_NEVER) ? 1 << 30 : (desc->flags & DAU_FLAG_DENSE_WGRAD_ALWAYS) ? 1 : 3;
            bs.wgrad_ok = want_dense && bf16 && b == 4 && bs.dense_ok && bs.dot_ok && bs.slab_dot == s.N && s.G >= min_units &&
                          r4::dense_wgrad_configure(s, blur_k, bf16, &bs.wgrad) &&
                          (double)r4::dense_wgrad_workspace_bytes(bs.wgrad) <= budget_bytes;
            bs.wgrad3_ok = bs.wgrad_ok && bs.dense3_ok && r3::dense_wgrad_configure(s, blur_k, bf16, &bs.wgrad3) &&
                           (double)r3::dense_wgrad_workspace_bytes(bs.wgrad3) <= budget_bytes;
        }
    }
    if ((desc->flags & DAU_FLAG_DENSE_BF16) && !bf16) {
        delete p;
        return fail(DAU_INVALID_ARGUMENT, "DAU_FLAG_DENSE_BF16 needs DAU_FLAG_IO_BF16 (it is the bf16 layer's gather-sum)");
    }
    if ((desc->flags & DAU_FLAG_DENSE_SPLIT_F16) && (desc->flags & (DAU_FLAG_DENSE_BF16 | DAU_FLAG_NO_DENSE_SPLIT))) {
        delete p;
        return fail(DAU_INVALID_ARGUMENT, "DAU_FLAG_DENSE_SPLIT_F16 excludes DAU_FLAG_DENSE_BF16 and DAU_FLAG_NO_DENSE_SPLIT");
    }
    if ((desc->flags & (DAU_FLAG_DENSE_WGRAD_NEVER | DAU_FLAG_DENSE_WGRAD_ALWAYS)) &&
        (!(desc->flags & DAU_FLAG_DENSE_BF16) || (desc->flags & DAU_FLAG_DENSE_WGRAD_NEVER && desc->flags & DAU_FLAG_DENSE_WGRAD_ALWAYS))) {
        delete p;
        return fail(DAU_INVALID_ARGUMENT, "DAU_FLAG_DENSE_WGRAD_NEVER / _ALWAYS qualify DAU_FLAG_DENSE_BF16 and exclude each other");
    }
    const bool fwd_ok = p->top().fwd_ok, dot_ok = p->top().dot_ok;
    if (bf16 && (desc->algo == DAU_ALGO_DIRECT || !(fwd_ok && dot_ok))) {
        delete p;
        return fail(DAU_INVALID_ARGUMENT, "DAU_FLAG_IO_BF16 needs the tiled kernels, which do not support this shape / algo");
    }
    if (desc->algo == DAU_ALGO_TILED && !(fwd_ok && dot_ok)) {
        delete p;
        return fail(DAU_INVALID_ARGUMENT, "DAU_ALGO_TILED does not support this shape");
    }
    p->algo_fwd = (desc->algo != DAU_ALGO_DIRECT && fwd_ok) ? DAU_ALGO_TILED : DAU_ALGO_DIRECT;
    p->algo_bwd = (desc->algo != DAU_ALGO_DIRECT && dot_ok) ? DAU_ALGO_TILED : DAU_ALGO_DIRECT;
    // dynamic bucket selection: tiled kernels, more than one bucket (or the two radii of the dense forms), not switched off (DAU_FLAG_STATIC_BUCKET; tuning build:
    // DAU_DYNAMIC_BUCKET=0 in the environment at plan creation).  The pinned status mirror needs a device; without one
    // (header-only checks on a CPU box) the plan simply has no hint.
    p->dynamic = (p->nsets > 1 || p->sets[0].dense3_ok || p->sets[0].any_split()) && !(desc->flags & DAU_FLAG_STATIC_BUCKET) && DAU_TUNE_INT("DAU_DYNAMIC_BUCKET", 1) != 0 &&
                 (p->algo_fwd == DAU_ALGO_TILED || p->algo_bwd == DAU_ALGO_TILED);
    void* hs = nullptr;
    // portable + mapped: a plan may be used on any device, and every device's prepare_units_kernel writes the mirror
    if (hipHostMalloc(&hs, sizeof(HostStatus), hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && hs) {
        std::memset(hs, 0, sizeof(HostStatus));
        p->host_status = static_cast<HostStatus*>(hs);
    } else {
        (void)hipGetLastError();   // no device: not an error of this call
    }
    *plan_out = p;
    return DAU_OK;
}

int dau_conv_plan_destroy(dau_conv_plan* plan) {
    if (plan) {
        for (auto& pool : plan->prof_events)
            for (auto& ev : pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        if (plan->host_status) (void)hipHostFree(plan->host_status);
    }
    delete plan;
    return DAU_OK;
}

int dau_conv_profile_begin(dau_conv_plan* plan) {
    if (!plan) return fail(DAU_INVALID_ARGUMENT, "null argument");
    for (size_t& u : plan->prof_used) u = 0;
    for (int& n : plan->prof_passes) n = 0;
    plan->profiling = true;
    return DAU_OK;
}

int dau_conv_profile_end(dau_conv_plan* plan, double* ms_out, int32_t* passes_out) {
    if (!plan || !ms_out || !passes_out) return fail(DAU_INVALID_ARGUMENT, "null argument");
    plan->profiling = false;
    for (int slot = 0; slot < DAU_PROFILE_SLOTS; ++slot) {
        double total = 0.0;
        for (size_t i = 0; i < plan->prof_used[slot]; ++i) {
            auto& ev = plan->prof_events[slot][i];
            DAU_HIP(hipEventSynchronize(ev.second));
            float ms = 0.0f;
            DAU_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
            total += ms;
        }
        ms_out[slot] = total;
        passes_out[slot] = plan->prof_passes[slot];
        plan->prof_used[slot] = 0;
        plan->prof_passes[slot] = 0;
    }
    return DAU_OK;
}

int dau_conv_plan_get_info(const dau_conv_plan* plan, dau_conv_plan_info* info) {
    if (!plan || !info) return fail(DAU_INVALID_ARGUMENT, "null argument");
    info->offset_bucket = plan->bucket;
    info->blur_support = plan->blur_k;
    info->algo_forward = plan->algo_fwd;
    info->algo_backward = plan->algo_bwd;
    info->drop_last_col = plan->drop_col;
    info->drop_last_row = plan->drop_row;
    info->gather_patch = plan->algo_fwd == DAU_ALGO_TILED ? plan->top().tiled_fwd.tiles_x * plan->top().tiled_fwd.tile_w : 0;
    info->gather_stack = plan->algo_fwd == DAU_ALGO_TILED ? plan->top().tiled_fwd.stack : 0;
    info->dot_windows = plan->algo_bwd == DAU_ALGO_TILED ? plan->top().tiled_dot.windows : 0;
    info->gather_windows = plan->algo_fwd == DAU_ALGO_TILED ? plan->top().tiled_fwd.windows : 0;
    info->bucket_sets = plan->dynamic ? plan->nsets : 1;
    // the dense member is bucket 4: reachable as the static set itself, or through the per-call selection
    const bool dense_reachable = plan->sets[0].dense_ok && (plan->nsets == 1 || plan->dynamic);
    info->gather_dense_bf16 = dense_reachable ? (plan->sets[0].wgrad_ok ? 2 : 1) : 0;
    info->batch_slab_gather = plan->top().slab_gather;
    info->batch_slab_dot = plan->top().slab_dot;
    info->dot_region = plan->top().dot_ok ? plan->top().tiled_dot.region_cols * 100 + plan->top().tiled_dot.region_rows : 0;
    info->gather_fblock = plan->algo_fwd == DAU_ALGO_TILED ? plan->top().tiled_fwd.fblock : 0;
    info->gather_variant = plan->algo_fwd == DAU_ALGO_TILED ? plan->top().tiled_fwd.variant : -1;
    info->dense_bf16_radius3 = dense_reachable && plan->dynamic && plan->sets[0].dense3_ok ? (plan->sets[0].wgrad3_ok ? 2 : 1) : 0;
    info->gather_dense_split = 0;
    if (plan->dynamic && plan->algo_fwd == DAU_ALGO_TILED)
        for (int r = 2; r <= 4; ++r)
            if (plan->sets[0].split_ok[r - 2]) info->gather_dense_split |= 1 << r;
    return DAU_OK;
}

int dau_conv_workspace_bytes(const dau_conv_plan* plan, int pass, size_t* bytes_out) {
    if (!plan || !bytes_out) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (pass == DAU_PASS_FORWARD) *bytes_out = carve_forward(plan, nullptr).bytes;
    else if (pass == DAU_PASS_BACKWARD) *bytes_out = carve_backward(plan, nullptr).bytes;
    else return fail(DAU_INVALID_ARGUMENT, "pass must be DAU_PASS_FORWARD or DAU_PASS_BACKWARD");
    return DAU_OK;
}

int dau_conv_forward(const dau_conv_plan* p, void* stream, const float* x, const float* w, const float* mu1,
                     const float* mu2, const float* sigma, float* y, void* workspace, size_t workspace_bytes) {
    if (!p || !x || !w || !mu1 || !mu2 || !sigma || !y || !workspace) return fail(DAU_INVALID_ARGUMENT, "null argument");
    FwdWs ws = carve_forward(p, workspace);
    if (workspace_bytes < ws.bytes)
        return fail(DAU_INVALID_ARGUMENT, "workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Shape& s = p->sh;
    if (int rc = ensure_attrs(p)) return rc;
    DAU_HIP(hipMemsetAsync(ws.status, 0, sizeof(Status), st));
    launch_synth_filters(st, sigma, p->blur_k, p->d.flags, ws.filters);
    launch_prepare_units(st, w, mu1, mu2, s, p->d.number_units_ignore, p->d.flags, p->bucket, false, ws.table, ws.status,
                         p->host_status);
    if (p->algo_fwd == DAU_ALGO_TILED) {
        Candidate cand[kMaxCandidates];
        const int ncand = pick_candidates(p, ws.status, 0, cand);
        if (p->profiling) ++p->prof_passes[0];
        const size_t esize = (p->d.flags & DAU_FLAG_IO_BF16) ? 2 : 4;
        for (int ci = 0; ci < ncand; ++ci) {
            const BucketSet& bs = *cand[ci].set;
            for (int n0 = 0; n0 < s.N; n0 += bs.slab_gather) {             // one slab unless the staged copy exceeds the budget
                const float* xs = slab_ptr(x, (size_t)n0 * s.S * s.H * s.W, esize);
                float* ys = slab_ptr(y, (size_t)n0 * s.F * s.H * s.W, esize);
                if (const int r = cand[ci].split_r) {                      // offsets within +-r: two-limb f16 GEMM, fp32 accuracy
                    kSplit[r - 2].prepare(st, bs.split_fwd[r - 2], xs, ws.filters, false, ws.table, ws.tiled, cand[ci].guard);
                    ProfScope prof(p, 0, st);
                    kSplit[r - 2].run(st, bs.split_fwd[r - 2], ys, ws.tiled, cand[ci].guard);
                    continue;
                }
                if (cand[ci].r3) {                                         // bf16 layer, offsets within +-3: 7 x 7 dense kernel
                    r3::dense_gather_prepare(st, bs.dense3_fwd, xs, ws.filters, false, ws.table, ws.tiled, cand[ci].guard);
                    ProfScope prof(p, 0, st);
                    r3::dense_gather_run(st, bs.dense3_fwd, ys, ws.tiled, cand[ci].guard);
                    continue;
                }
                if (bs.dense_ok) {                                         // bf16 layer, offsets within +-4: dense implicit GEMM
                    r4::dense_gather_prepare(st, bs.dense_fwd, xs, ws.filters, false, ws.table, ws.tiled, cand[ci].guard);
                    ProfScope prof(p, 0, st);
                    r4::dense_gather_run(st, bs.dense_fwd, ys, ws.tiled, cand[ci].guard);
                    continue;
                }
                const TiledConfig& cfg = bs.tiled_fwd;
                for (int window = 0; window < tiled_gather_windows(cfg); ++window) {   // one pass unless the bucket is 32
                    tiled_gather_prepare(st, cfg, xs, ws.filters, false, ws.table, ws.tiled, window, cand[ci].guard);
                    ProfScope prof(p, 0, st);
                    tiled_gather_run(st, cfg, ys, ws.tiled, window > 0, cand[ci].guard);
                }
            }
        }
    } else {
        launch_blur_direct(st, x, (long)s.N * s.S, s.H, s.W, ws.filters + 0 * kFilterPlane, 1, p->blur_k, ws.xb);
        if (p->profiling) ++p->prof_passes[0];
        ProfScope prof(p, 0, st);
        launch_gather_sum_direct(st, ws.xb, ws.table, s.N, s.S, s.F, s.G, s.H, s.W, y);
    }
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

namespace {

// raw parameter-gradient sums r4[k][s][g][f] = offset_and_dot(x * D_k, dy') with bare bilinear factors
// kinds: 4, or 3 when the caller does not want dsigma (only the dense form computes fewer: its GEMMs are per kind)
int run_param_sums(const dau_conv_plan* p, hipStream_t st, const float* x, const float* dy, const float* mu1,
                   const float* mu2, const BwdWs& ws, float* r4, int kinds) {
    const Shape& s = p->sh;
    const int flags = p->d.flags;
    launch_prepare_units(st, nullptr, mu1, mu2, s, p->d.number_units_ignore, flags, p->bucket, false, ws.table_bare,
                         ws.status, p->host_status);
    if (p->profiling) ++p->prof_passes[2];
    if (p->algo_bwd == DAU_ALGO_TILED) {
        Candidate cand[kMaxCandidates];
        const int ncand = pick_candidates(p, ws.status, 1, cand);
        const size_t esize = (flags & DAU_FLAG_IO_BF16) ? 2 : 4;
        for (int ci = 0; ci < ncand; ++ci) {
            const BucketSet& bs = *cand[ci].set;
            const TiledDotConfig& cfg = bs.tiled_dot;
            if (cand[ci].r3) {                                                 // offsets within +-3: 49 displacements
                ProfScope prof(p, 2, st);
                r3::dense_wgrad_run(st, bs.wgrad3, x, dy, ws.filters, ws.table_bare, p->drop_col, p->drop_row, r4, ws.tiled_dot, cand[ci].guard, kinds);
                continue;
            }
            if (bs.wgrad_ok) {                                                 // bf16 layer, offsets within +-4, many units
                ProfScope prof(p, 2, st);
                r4::dense_wgrad_run(st, bs.wgrad, x, dy, ws.filters, ws.table_bare, p->drop_col, p->drop_row, r4, ws.tiled_dot, cand[ci].guard, kinds);
                continue;
            }
            for (int n0 = 0; n0 < s.N; n0 += bs.slab_dot) {                // the sums of the slabs add up in r4
                tiled_dot_prepare(st, cfg, slab_ptr(x, (size_t)n0 * s.S * s.H * s.W, esize),
                                  slab_ptr(dy, (size_t)n0 * s.F * s.H * s.W, esize), ws.filters, ws.table_bare, p->drop_col,
                                  p->drop_row, ws.tiled_dot, cand[ci].guard);
                ProfScope prof(p, 2, st);
                tiled_dot_run(st, cfg, r4, ws.tiled_dot, cand[ci].guard, n0 > 0);
            }
        }
    } else {
        launch_blur_direct(st, x, (long)s.N * s.S, s.H, s.W, ws.filters + 1 * kFilterPlane, kNumK, p->blur_k, ws.xk4);
        ProfScope prof(p, 2, st);
        launch_gather_dot_direct(st, ws.xk4, dy, ws.table_bare, s, p->drop_col, p->drop_row, r4);
    }
    return DAU_OK;
}

int check_backward_ws(const dau_conv_plan* p, void* workspace, size_t workspace_bytes, BwdWs* ws) {
    *ws = carve_backward(p, workspace);
    if (workspace_bytes < ws->bytes)
        return fail(DAU_INVALID_ARGUMENT, "workspace too small: %zu < %zu", workspace_bytes, ws->bytes);
    return DAU_OK;
}

}  // namespace

int dau_conv_backward(const dau_conv_plan* p, void* stream, const float* x, const float* dy, const float* w,
                      const float* mu1, const float* mu2, const float* sigma, float* dx, float* dw, float* dmu1,
                      float* dmu2, float* dsigma, void* workspace, size_t workspace_bytes, int need_mask) {
    if (!p || !x || !dy || !w || !mu1 || !mu2 || !sigma || !workspace) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (((need_mask & DAU_NEED_DX) && !dx) || ((need_mask & DAU_NEED_DW) && !dw) ||
        ((need_mask & DAU_NEED_DMU1) && !dmu1) || ((need_mask & DAU_NEED_DMU2) && !dmu2) ||
        ((need_mask & DAU_NEED_DSIGMA) && !dsigma))
        return fail(DAU_INVALID_ARGUMENT, "need_mask asks for a gradient whose output pointer is null");
    BwdWs ws;
    if (int rc = check_backward_ws(p, workspace, workspace_bytes, &ws)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Shape& s = p->sh;
    const int flags = p->d.flags;
    if (int rc = ensure_attrs(p)) return rc;
    DAU_HIP(hipMemsetAsync(ws.status, 0, sizeof(Status), st));
    launch_synth_filters(st, sigma, p->blur_k, flags, ws.filters);

    const int param_mask = DAU_NEED_DW | DAU_NEED_DMU1 | DAU_NEED_DMU2 | DAU_NEED_DSIGMA;
    if (need_mask & param_mask) {
        run_param_sums(p, st, x, dy, mu1, mu2, ws, ws.r4, (need_mask & DAU_NEED_DSIGMA) ? 4 : 3);
        launch_finalize_grads(st, ws.r4, w, s, p->d.number_units_ignore, p->d.mu_learning_rate_factor, need_mask,
                              flags & DAU_FLAG_SINGLE_DIM_KERNEL, dw, dmu1, dmu2, dsigma);
    }
    if (need_mask & DAU_NEED_DX) {
        // input gradient: the forward gather on the mirrored-Gaussian-blurred error with the
        // parameters read as [F,G,S] and negated offsets (base_dau_conv_layer.cu:299-325)
        const bool fresh = !(need_mask & param_mask);     // this call has not looked at the offsets yet
        launch_prepare_units(st, w, mu1, mu2, s, 0, flags, p->bucket, true, ws.table_t, fresh ? ws.status : nullptr,
                             p->host_status);
        if (p->profiling) ++p->prof_passes[1];
        if (p->algo_fwd == DAU_ALGO_TILED) {
            Candidate cand[kMaxCandidates];
            const int ncand = pick_candidates(p, ws.status, 0, cand);
            const size_t esize = (flags & DAU_FLAG_IO_BF16) ? 2 : 4;
            for (int ci = 0; ci < ncand; ++ci) {
                const BucketSet& bs = *cand[ci].set;
                for (int n0 = 0; n0 < s.N; n0 += bs.slab_gather) {
                    const float* dys = slab_ptr(dy, (size_t)n0 * s.F * s.H * s.W, esize);
                    float* dxs = slab_ptr(dx, (size_t)n0 * s.S * s.H * s.W, esize);
                    if (const int r = cand[ci].split_r) {
                        kSplit[r - 2].prepare(st, bs.split_dx[r - 2], dys, ws.filters, true, ws.table_t, ws.tiled_dx, cand[ci].guard);
                        ProfScope prof(p, 1, st);
                        kSplit[r - 2].run(st, bs.split_dx[r - 2], dxs, ws.tiled_dx, cand[ci].guard);
                        continue;
                    }
                    if (cand[ci].r3) {
                        r3::dense_gather_prepare(st, bs.dense3_dx, dys, ws.filters, true, ws.table_t, ws.tiled_dx, cand[ci].guard);
                        ProfScope prof(p, 1, st);
                        r3::dense_gather_run(st, bs.dense3_dx, dxs, ws.tiled_dx, cand[ci].guard);
                        continue;
                    }
                    if (bs.dense_ok) {
                        r4::dense_gather_prepare(st, bs.dense_dx, dys, ws.filters, true, ws.table_t, ws.tiled_dx, cand[ci].guard);
                        ProfScope prof(p, 1, st);
                        r4::dense_gather_run(st, bs.dense_dx, dxs, ws.tiled_dx, cand[ci].guard);
                        continue;
                    }
                    const TiledConfig& cfg = bs.tiled_dx;
                    for (int window = 0; window < tiled_gather_windows(cfg); ++window) {
                        tiled_gather_prepare(st, cfg, dys, ws.filters, true, ws.table_t, ws.tiled_dx, window, cand[ci].guard);
                        ProfScope prof(p, 1, st);
                        tiled_gather_run(st, cfg, dxs, ws.tiled_dx, window > 0, cand[ci].guard);
                    }
                }
            }
        } else {
            launch_blur_direct(st, dy, (long)s.N * s.F, s.H, s.W, ws.filters + 5 * kFilterPlane, 1, p->blur_k, ws.eb);
            ProfScope prof(p, 1, st);
            launch_gather_sum_direct(st, ws.eb, ws.table_t, s.N, s.F, s.S, s.G, s.H, s.W, dx);
        }
    }
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

int dau_conv_backward_param_sums(const dau_conv_plan* p, void* stream, const float* x, const float* dy, const float* mu1,
                                 const float* mu2, const float* sigma, float* sums_out, void* workspace,
                                 size_t workspace_bytes) {
    if (!p || !x || !dy || !mu1 || !mu2 || !sigma || !sums_out || !workspace) return fail(DAU_INVALID_ARGUMENT, "null argument");
    BwdWs ws;
    if (int rc = check_backward_ws(p, workspace, workspace_bytes, &ws)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = ensure_attrs(p)) return rc;
    DAU_HIP(hipMemsetAsync(ws.status, 0, sizeof(Status), st));
    launch_synth_filters(st, sigma, p->blur_k, p->d.flags, ws.filters);
    run_param_sums(p, st, x, dy, mu1, mu2, ws, sums_out, 4);
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

int dau_conv_finalize_param_grads(const dau_conv_plan* p, void* stream, const float* sums, const float* w, float* dw,
                                  float* dmu1, float* dmu2, float* dsigma, int need_mask) {
    if (!p || !sums || !w) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (((need_mask & DAU_NEED_DW) && !dw) || ((need_mask & DAU_NEED_DMU1) && !dmu1) ||
        ((need_mask & DAU_NEED_DMU2) && !dmu2) || ((need_mask & DAU_NEED_DSIGMA) && !dsigma))
        return fail(DAU_INVALID_ARGUMENT, "need_mask asks for a gradient whose output pointer is null");
    launch_finalize_grads(static_cast<hipStream_t>(stream), sums, w, p->sh, p->d.number_units_ignore,
                          p->d.mu_learning_rate_factor, need_mask, p->d.flags & DAU_FLAG_SINGLE_DIM_KERNEL, dw, dmu1, dmu2,
                          dsigma);
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

int dau_conv_check_status(const dau_conv_plan* p, void* stream, const void* workspace, float* max_abs_mu_out) {
    if (!p || !workspace) return fail(DAU_INVALID_ARGUMENT, "null argument");
    Status h;
    DAU_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    DAU_HIP(hipMemcpy(&h, workspace, sizeof(Status), hipMemcpyDeviceToHost));
    float mx;
    std::memcpy(&mx, &h.max_abs_mu_bits, sizeof(float));
    if (max_abs_mu_out) *max_abs_mu_out = mx;
    if (h.nan_seen || mx > (float)p->bucket) clear_reported_status(p, h.max_abs_mu_bits, h.nan_seen != 0);   // reported here
    if (h.nan_seen) return fail(DAU_FAILED_PRECONDITION, "DAUConvOp ERROR: got NaN value in offset (mu1,mu2) variable");
    if (mx > (float)p->bucket)
        return fail(DAU_INVALID_ARGUMENT,
                    "DAUConvOp ERROR: actual offsets (%.3f) larger than what max_kernel_size=%d allows (setup max_kernel_size "
                    "and dau_unit_border_bound correctly to avoid this)",
                    mx, p->d.max_kernel_size);
    return DAU_OK;
}

int dau_conv_last_status(const dau_conv_plan* p, float* max_abs_mu_out, int32_t* valid_out) {
    if (!p) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (max_abs_mu_out) *max_abs_mu_out = 0.0f;
    if (valid_out) *valid_out = 0;
    if (!p->host_status) return DAU_OK;
    const volatile unsigned* h = reinterpret_cast<const volatile unsigned*>(p->host_status);
    if (h[2] != 1u && h[4] == 0u && h[5] == 0u) return DAU_OK;   // no call has completed yet
    // the sticky record first: the worst status of ANY completed call since the last report (a later good call of another
    // layer sharing this plan must not hide it); then the most recent call
    const unsigned bad_bits = h[4], bad_nan = h[5];
    unsigned bits = h[0], nan_seen = h[1] | bad_nan;
    if (bad_bits > bits) bits = bad_bits;                // non-negative float bits order like unsigned integers
    float mx;
    std::memcpy(&mx, &bits, sizeof(float));
    if (max_abs_mu_out) *max_abs_mu_out = mx;
    if (valid_out) *valid_out = 1;
    // a bad status is reported once: the mirror goes back to "nothing reported" until the next call completes
    if (nan_seen || mx > (float)p->bucket) clear_host_status(p);
    if (nan_seen) return fail(DAU_FAILED_PRECONDITION, "DAUConvOp ERROR: got NaN value in offset (mu1,mu2) variable");
    if (mx > (float)p->bucket)
        return fail(DAU_INVALID_ARGUMENT,
                    "DAUConvOp ERROR: actual offsets (%.3f) larger than what max_kernel_size=%d allows (setup max_kernel_size "
                    "and dau_unit_border_bound correctly to avoid this)",
                    mx, p->d.max_kernel_size);
    return DAU_OK;
}

int dau_conv_filters(const dau_conv_plan* p, void* stream, const float* sigma, float* filters_out) {
    if (!p || !sigma || !filters_out) return fail(DAU_INVALID_ARGUMENT, "null argument");
    // the six k x k planes are written straight into the caller's buffer: no scratch, no sync
    launch_synth_filters_compact(static_cast<hipStream_t>(stream), sigma, p->blur_k, p->d.flags, filters_out);
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

int dau_conv_unit_table(const dau_conv_plan* p, void* stream, const float* w, const float* mu1, const float* mu2, int form,
                        void* table_out) {
    if (!p || !mu1 || !mu2 || !table_out) return fail(DAU_INVALID_ARGUMENT, "null argument");
    if (form != 0 && form != 1) return fail(DAU_INVALID_ARGUMENT, "form must be 0 ([S][G][F]) or 1 ([F][G][S], negated offsets)");
    // the kernel every forward / backward call runs first, writing into the caller's buffer instead of the workspace
    // (the input-gradient pass ignores no unit: the reference transposes the ignored units' zero weights along)
    launch_prepare_units(static_cast<hipStream_t>(stream), w, mu1, mu2, p->sh, form == 1 ? 0 : p->d.number_units_ignore, p->d.flags,
                         p->bucket, form == 1, static_cast<UnitRef*>(table_out), nullptr, nullptr);
    DAU_HIP(hipPeekAtLastError());
    return DAU_OK;
}

}  // extern "C"
