// Per-unit bookkeeping: mu -> integer displacement + bilinear factors (bit-exact float32,
// reference: include/dau_conv/dau_conv_impl/dau_conv_forward_core.hpp:2025-2028, 2135-2213;
// backward variant with w = 1: dau_conv_backward_core.hpp:2078-2081), the max|mu| / NaN
// status that replaces the ops' blocking amax (dau_conv_op.cpp:223-262), and the
// elementwise tail of Backward_gpu (base_dau_conv_layer.cu:335-355).
#include "dau_common.hpp"

namespace dau {

__device__ __forceinline__ void unit_math(float m1, float m2, bool interp, int& ox, int& oy, float& b00,
                                          float& b01, float& b10, float& b11) {
    const float fox = floorf(m1), foy = floorf(m2);
    float fx = m1 - fox, fy = m2 - foy;
    if (!interp) { fx = 0.0f; fy = 0.0f; }
    ox = (int)fox; oy = (int)foy;
    b00 = (1.0f - fx) * (1.0f - fy);
    b01 = fx * (1.0f - fy);
    b10 = (1.0f - fx) * fy;
    b11 = fx * fy;
}

// table index: SGF order ((s*G+g)*F+f) or, for the input-gradient pass, FGS order
// ((f*G+g)*S+s) with negated offsets (base_dau_conv_layer.cu:299-325).
// weight_mode: 0 = multiply by w, 1 = bare factors (parameter-gradient kernels).
__global__ void prepare_units_kernel(const float* __restrict__ w, const float* __restrict__ mu1,
                                     const float* __restrict__ mu2, int S, int G, int F, int ignore,
                                     int flags, int bucket, int transposed_negated, int weight_mode,
                                     UnitRef* __restrict__ table, Status* __restrict__ status,
                                     HostStatus* __restrict__ host_status) {
    const long units = (long)S * G * F;
    unsigned int local_max = 0, local_nan = 0;
    // the loop index is the DESTINATION slot, so that the 24-byte table entries are written in order; in the transposed
    // case the three parameter reads are the strided side instead (12 B per unit, served from L2)
    for (long dst = blockIdx.x * (long)blockDim.x + threadIdx.x; dst < units; dst += (long)gridDim.x * blockDim.x) {
        int f, g, s;
        if (transposed_negated) { s = (int)(dst % S); g = (int)((dst / S) % G); f = (int)(dst / ((long)S * G)); }
        else { f = (int)(dst % F); g = (int)((dst / F) % G); s = (int)(dst / ((long)F * G)); }
        const long u = ((long)s * G + g) * F + f;
        float m1 = mu1[u], m2 = mu2[u];
        const bool is_nan = (m1 != m1) || (m2 != m2);
        if (is_nan) { local_nan = 1; m1 = 0.0f; m2 = 0.0f; }
        const float amax = fmaxf(fabsf(m1), fabsf(m2));
        local_max = max(local_max, __float_as_uint(amax));
        // safety clamp: a displacement beyond the bucket would read outside the staged tile
        m1 = fminf(fmaxf(m1, -(float)bucket), (float)bucket);
        m2 = fminf(fmaxf(m2, -(float)bucket), (float)bucket);
        if (transposed_negated) { m1 = -m1; m2 = -m2; }
        int ox, oy; float b00, b01, b10, b11;
        unit_math(m1, m2, flags & DAU_FLAG_USE_INTERPOLATION, ox, oy, b00, b01, b10, b11);
        float wv = weight_mode == 0 ? w[u] : 1.0f;
        if (g >= G - ignore) wv = 0.0f;
        UnitRef r;
        r.ox = ox; r.oy = oy;
        // premultiplied tap weights (dau_conv_forward_core.hpp:2155-2213)
        r.w00 = wv * b00; r.w01 = wv * b01; r.w10 = wv * b10; r.w11 = wv * b11;
        table[dst] = r;
    }
    if (status) {
        // one atomic per workgroup: thousands of same-address atomics serialise in L2 and were most of this kernel's time
        __shared__ unsigned int smax[16], snan[16];
        for (int m = 32; m >= 1; m >>= 1) {
            local_max = max(local_max, (unsigned int)__shfl_xor((int)local_max, m));
            local_nan |= (unsigned int)__shfl_xor((int)local_nan, m);
        }
        const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
        if ((threadIdx.x & 63) == 0) { smax[wave] = local_max; snan[wave] = local_nan; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < nw; ++i) { local_max = max(local_max, smax[i]); local_nan |= snan[i]; }
            atomicMax(&status->max_abs_mu_bits, local_max);
            if (local_nan) atomicOr(&status->nan_seen, 1u);
            if (host_status) {
                // the workgroup that finishes last mirrors the result into pinned host memory (read without a sync by
                // dau_conv_last_status and as the next call's offset-bucket hint); pad[0] counts finished workgroups
                __threadfence();
                if (atomicAdd(&status->pad[0], 1u) == gridDim.x - 1) {
                    const unsigned mx = atomicMax(&status->max_abs_mu_bits, 0u), nn = atomicOr(&status->nan_seen, 0u);
                    volatile unsigned* h = reinterpret_cast<volatile unsigned*>(host_status);
                    h[0] = mx; h[1] = nn; h[2] = 1u;   // [2]: a completed call has reported
                    // a bad status is also recorded STICKY ([4] worst max|mu| beyond the bucket, [5] NaN seen): every later
                    // call of this plan overwrites [0..2], only the host's report clears [4..5]
                    // (plain stores: calls of several streams / devices may race here and with the host's clear.  The NaN flag is
                    // a store of 1; of two racing out-of-range maxima either may stay -- both lie beyond the bucket, so the error is
                    // reported either way, only the number in its message may be the smaller one)
                    if (nn) h[5] = 1u;
                    if (mx > __float_as_uint((float)bucket) && mx > h[4]) h[4] = mx;
                }
            }
        }
    }
}

void launch_prepare_units(hipStream_t st, const float* w, const float* mu1, const float* mu2, Shape sh,
                          int ignore, int flags, int bucket, bool transposed_negated, UnitRef* table,
                          Status* status, HostStatus* host_status) {
    const long units = (long)sh.S * sh.G * sh.F;
    const int block = 256;
    const int grid = (int)((units + block - 1) / block < 512 ? (units + block - 1) / block : 512);
    hipLaunchKernelGGL(prepare_units_kernel, dim3(grid), dim3(block), 0, st, w, mu1, mu2, sh.S, sh.G, sh.F,
                       ignore, flags, bucket, transposed_negated ? 1 : 0, w == nullptr ? 1 : 0, table, status,
                       status ? host_status : nullptr);
}

// dw = r0 ; dmu1 = w*r1*lr ; dmu2 = w*r2*lr ; dsigma = w*r3 ; ignored units -> 0 ; NaN in dmu -> 0.
// dmu2 is left at zero for single_dim_kernel (dau_conv_grad_op.cpp:293-294).
__global__ void finalize_grads_kernel(const float* __restrict__ r4, const float* __restrict__ w, int S, int G,
                                      int F, int ignore, float lr, int need_mask, int single_dim,
                                      float* __restrict__ dw, float* __restrict__ dmu1,
                                      float* __restrict__ dmu2, float* __restrict__ dsigma) {
    const long units = (long)S * G * F;
    for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const int g = (int)((u / F) % G);
        const bool live = g < G - ignore;
        const float wv = w[u];
        if (dw && (need_mask & DAU_NEED_DW)) dw[u] = live ? r4[u] : 0.0f;
        if (dmu1 && (need_mask & DAU_NEED_DMU1)) {
            float v = live ? r4[units + u] * wv * lr : 0.0f;
            dmu1[u] = (v != v) ? 0.0f : v;
        }
        if (dmu2 && (need_mask & DAU_NEED_DMU2)) {
            float v = (live && !single_dim) ? r4[2 * units + u] * wv * lr : 0.0f;
            dmu2[u] = (v != v) ? 0.0f : v;
        }
        if (dsigma && (need_mask & DAU_NEED_DSIGMA)) dsigma[u] = live ? r4[3 * units + u] * wv : 0.0f;
    }
}

void launch_finalize_grads(hipStream_t st, const float* r4, const float* w, Shape sh, int ignore, float lr,
                           int need_mask, bool single_dim, float* dw, float* dmu1, float* dmu2, float* dsigma) {
    const long units = (long)sh.S * sh.G * sh.F;
    const int block = 256;
    const int grid = (int)((units + block - 1) / block < 2048 ? (units + block - 1) / block : 2048);
    hipLaunchKernelGGL(finalize_grads_kernel, dim3(grid), dim3(block), 0, st, r4, w, sh.S, sh.G, sh.F, ignore, lr,
                       need_mask, single_dim ? 1 : 0, dw, dmu1, dmu2, dsigma);
}

}  // namespace dau
