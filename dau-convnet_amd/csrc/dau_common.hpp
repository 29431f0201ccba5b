// Shared declarations of the HIP implementation behind include/dau_conv.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "dau_conv.h"

// Tuning knobs.  The shipped library's behaviour is fully determined by dau_conv_desc (plus DAU_WORKSPACE_BUDGET_GB): the
// environment variables that pin a kernel variant, a chunking or a staging tile for same-box A/B runs and for the variant
// tests exist only in the TUNING build (-DDAU_TUNING: libdau_conv_hip_tuning.so, `make tuning`), where these macros read the
// environment at plan creation.  In the release build they are their defaults and the names are not in the binary
// (tests/test_capi_symbols.py checks that).
#ifdef DAU_TUNING
#include <cstdlib>
#define DAU_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define DAU_TUNE_SET(name) (getenv(name) != nullptr)
#else
#define DAU_TUNE_INT(name, dflt) (dflt)
#define DAU_TUNE_SET(name) (false)
#endif

namespace dau {

// bfloat16 I/O (DAU_FLAG_IO_BF16): activations are stored as the upper 16 bits of an fp32
__device__ __forceinline__ float load_act(const float* base, long idx, bool bf16) {
    if (!bf16) return base[idx];
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[idx] << 16);
}
// the stored bits of an activation, untouched (no arithmetic on a value still in flight), and their conversion
template <bool BF> struct RawAct { typedef float type; };
template <> struct RawAct<true> { typedef unsigned type; };   // (zero-extended by the load; a 16-bit pair would be packed = arithmetic)
template <bool BF>
__device__ __forceinline__ typename RawAct<BF>::type load_raw(const float* base, long idx) {
    if constexpr (!BF) return base[idx];
    else return reinterpret_cast<const unsigned short*>(base)[idx];
}
__device__ __forceinline__ float act_of(float v) { return v; }
__device__ __forceinline__ float act_of(unsigned v) { return __uint_as_float(v << 16); }
template <bool BF>
__device__ __forceinline__ float load_act_t(const float* base, long idx) {
    if constexpr (!BF) return base[idx];
    else return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[idx] << 16);
}
// v where keep, +0 elsewhere, as a bit mask: with a select hipcc moves the load that produced v under a branch on `keep`
__device__ __forceinline__ float mask_act(float v, bool keep) { return __uint_as_float(__float_as_uint(v) & (keep ? 0xffffffffu : 0u)); }
// Load phase of the staging kernels: rows_ x cols_ items of one plane split over the nw waves of its wave group, the loads of
// kLoadBatch items issued back to back before the first of them is stored to LDS.  These kernels are HBM bound, and a wave with
// one or two loads in flight keeps far fewer bytes in the air than the memory latency needs (Little's law: ~50 KB per CU).
// ld(r, x) -> value must be BRANCH FREE (clamped address + select): a branch around a load makes hipcc wait for it at the
// join, one load at a time.  It returns the loaded bits untouched (RawAct / load_raw); st converts and masks them.  Items past the end load item (rows_-1, cols_-1) again and store nothing.  st(r, x, value).
// A wave per row when the rows are wide, a flat index when they are narrow.
#ifndef DAU_LOAD_BATCH
#define DAU_LOAD_BATCH 6
#endif
constexpr int kLoadBatch = DAU_LOAD_BATCH;
template <class V, class Load, class Store>
__device__ __forceinline__ void load_phase(int rows_, int cols_, int wave, int nw, int lane, Load&& ld, Store&& st) {
    if (cols_ >= 56) {
        for (int x0 = 0; x0 < cols_; x0 += 64) {
            const int x = x0 + lane, xc = x < cols_ ? x : cols_ - 1;
            for (int r0 = wave; r0 < rows_; r0 += nw * kLoadBatch) {
                V v[kLoadBatch];
#pragma unroll
                for (int u = 0; u < kLoadBatch; ++u) { const int r = r0 + u * nw; v[u] = ld(r < rows_ ? r : rows_ - 1, xc); }
#pragma unroll
                for (int u = 0; u < kLoadBatch; ++u) { const int r = r0 + u * nw; if (r < rows_ && x < cols_) st(r, x, v[u]); }
            }
        }
    } else {
        const int total = rows_ * cols_;
        for (int t0 = wave * 64 + lane; t0 < total; t0 += nw * 64 * kLoadBatch) {
            V v[kLoadBatch];
            int rr[kLoadBatch];
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) {
                const int t = t0 + u * nw * 64, tc = t < total ? t : total - 1;
                rr[u] = tc / cols_;
                v[u] = ld(rr[u], tc - rr[u] * cols_);
            }
#pragma unroll
            for (int u = 0; u < kLoadBatch; ++u) { const int t = t0 + u * nw * 64; if (t < total) st(rr[u], t - rr[u] * cols_, v[u]); }
        }
    }
}
__device__ __forceinline__ void store_act(float* base, long idx, float v, bool bf16, bool accumulate) {
    if (!bf16) { base[idx] = accumulate ? base[idx] + v : v; return; }
    unsigned short* p = reinterpret_cast<unsigned short*>(base) + idx;
    if (accumulate) v += __uint_as_float((unsigned)*p << 16);
    unsigned u = __float_as_uint(v);
    // round to nearest even on the bits; a NaN must stay a NaN (the rounding add can carry a NaN payload into the sign /
    // exponent: 0xFFFFFFFF -> +0, 0x7F800001 -> +inf), so it is stored as the quiet NaN pattern instead
    u = (v != v) ? 0x7fc00000u : u + 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}


constexpr int kMaxBlurSupport = 17;                       // convolve.cu:40 caps the prefilter at 17x17
constexpr int kFilterPlane = kMaxBlurSupport * kMaxBlurSupport;
constexpr int kNumK = 4;                                  // gradient kinds {w, mu1, mu2, sigma}
// The prefilters are separable: Gn = gx (x) gy, Dmu1 = ax (x) gy, Dmu2 = gx (x) ay, Dsigma = cx (x) gy + gx (x) by
// (SURVEY.md Appendix A item 1 rewritten with 1-D factors).  synth_filters_kernel also emits these taps, after
// the six 2-D planes: eight arrays of kTapPitch floats.
constexpr int kTapPitch = 32;
enum Tap1d { kTapGX = 0, kTapGY, kTapAX, kTapAY, kTapCX, kTapBY, kTapGXR, kTapGYR, kNumTap1d };
constexpr int kTaps1dOffset = 6 * kFilterPlane;
constexpr int kFilterFloats = kTaps1dOffset + kNumTap1d * kTapPitch;

// Device-side status block at the head of every workspace (see dau_conv_check_status).
struct Status {
    unsigned int max_abs_mu_bits;  // float bits of max(|mu1|,|mu2|); valid because |x| bits order like uints
    unsigned int nan_seen;
    unsigned int pad[2];
};

// Pinned host mirror of a plan (written by the last workgroup of prepare_units_kernel, read by dau_conv_last_status and as
// the next call's offset-bucket hint): the most recent completed call's status, and the sticky record of bad ones.
struct HostStatus {
    unsigned int max_abs_mu_bits, nan_seen, valid, pad0;      // most recent completed call
    unsigned int bad_max_abs_mu_bits, bad_nan_seen, pad1[2];  // worst status since the host last reported one (sticky)
};

// Offset-bucket guard of a launch.  The bucket a call needs depends on max|mu|, which only the device knows when the
// kernels are enqueued (prepare_units_kernel leaves it in the status block).  The host enqueues the kernel sets of up to
// two candidate buckets; every kernel of a set starts with guard_pass() and returns at once unless the actual max|mu|
// falls into (lo, hi] -- exactly one set does the work, without a device->host sync (the reference blocks on a D2H copy
// of the amax for this, dau_conv_op.cpp:229-253).  status == nullptr: unguarded.
struct Guard {
    const Status* status;
    float lo, hi;
};
__device__ __forceinline__ bool guard_pass(const Guard& g) {
    if (!g.status) return true;
    const float mx = __uint_as_float(g.status->max_abs_mu_bits);
    return mx > g.lo && mx <= g.hi;
}

// One prepared unit for the gather kernels: integer displacement and the four
// bilinear weights already multiplied by w (dau_conv_forward_core.hpp:2155-2213).
struct UnitRef {
    int ox, oy;
    float w00, w01, w10, w11;
};

struct Shape {
    int N, S, F, G, H, W;
};

// ---- launchers implemented in the kernel TUs --------------------------------------
// k_filters.hip
void launch_synth_filters(hipStream_t st, const float* sigma_dev, int k, int flags, float* filters6);
void launch_synth_filters_compact(hipStream_t st, const float* sigma_dev, int k, int flags, float* planes6);
// k_units.hip
// host_status (may be null): pinned host copy of the status block, written by the last workgroup to finish
void launch_prepare_units(hipStream_t st, const float* w, const float* mu1, const float* mu2, Shape sh,
                          int ignore, int flags, int bucket, bool transposed_negated, UnitRef* table,
                          Status* status, HostStatus* host_status);
void launch_finalize_grads(hipStream_t st, const float* r4, const float* w, Shape sh, int ignore, float lr,
                           int need_mask, bool single_dim, float* dw, float* dmu1, float* dmu2, float* dsigma);
// k_direct.hip  (DAU_ALGO_DIRECT: plain kernels, any shape)
void launch_blur_direct(hipStream_t st, const float* x, long planes, int H, int W, const float* filters,
                        int nfilt, int k, float* out);
void launch_gather_sum_direct(hipStream_t st, const float* xb, const UnitRef* table, int N, int Sin, int Fout,
                              int G, int H, int W, float* y);
void launch_gather_dot_direct(hipStream_t st, const float* xk4, const float* err, const UnitRef* table, Shape sh,
                              int drop_col, int drop_row, float* r4);

}  // namespace dau
