// Gaussian prefilter synthesis on the device (no host sync on sigma).
// Same math as BaseDAUKernelCompute::get_kernels with w = 1, mu = 0 and
// unit_normalization = true (reference: src/dau_conv/base_dau_conv_layer.cu:402-448, 583-704),
// evaluated in double and rounded to float once.
#include "dau_common.hpp"

namespace dau {

__global__ void synth_filters_kernel(const float* __restrict__ sigma_dev, int k, int flags,
                                     float* __restrict__ out) {
    // one wave; lane t strides over the k*k taps, sums via wave reduction
    const int lane = threadIdx.x;
    const int n = k * k, c = (k - 1) / 2;
    const double sigma = (double)sigma_dev[0];
    const bool single_dim = flags & DAU_FLAG_SINGLE_DIM_KERNEL;
    const bool forbid_pos = flags & DAU_FLAG_FORBID_POSITIVE_DIM1;
    const double inv_s2 = 1.0 / (sigma * sigma), inv_s3 = inv_s2 / sigma;
    double Z = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int t = lane; t < n; t += 64) {
        const double u = (t % k) - c, v = (t / k) - c;
        double g = exp(-(u * u + v * v) * 0.5 * inv_s2);
        if (single_dim && v != 0) g = 0;
        if (forbid_pos && u > 0) g = 0;
        Z += g; s1 += u * inv_s2 * g; s2 += v * inv_s2 * g; s3 += (u * u + v * v) * inv_s3 * g;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        Z += __shfl_xor(Z, m); s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); s3 += __shfl_xor(s3, m);
    }
    s1 /= Z; s2 /= Z; s3 /= Z;
    for (int t = lane; t < n; t += 64) {
        const int i = t % k, j = t / k;
        const double u = i - c, v = j - c;
        double g = exp(-(u * u + v * v) * 0.5 * inv_s2);
        if (single_dim && v != 0) g = 0;
        if (forbid_pos && u > 0) g = 0;
        const double gn = g / Z;
        out[0 * kFilterPlane + t] = (float)gn;                                        // Gn
        out[1 * kFilterPlane + t] = (float)gn;                                        // Dw
        out[2 * kFilterPlane + t] = (float)(u * inv_s2 * g / Z - gn * s1);            // Dmu1
        out[3 * kFilterPlane + t] = (float)(v * inv_s2 * g / Z - gn * s2);            // Dmu2
        out[4 * kFilterPlane + t] = (float)((u * u + v * v) * inv_s3 * g / Z - gn * s3);  // Dsigma
        out[5 * kFilterPlane + (k - 1 - j) * k + (k - 1 - i)] = (float)gn;            // Gerr = flip(Gn)
    }
}

void launch_synth_filters(hipStream_t st, const float* sigma_dev, int k, int flags, float* filters6) {
    hipLaunchKernelGGL(synth_filters_kernel, dim3(1), dim3(64), 0, st, sigma_dev, k, flags, filters6);
}

}  // namespace dau
