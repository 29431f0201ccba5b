// Gaussian prefilter synthesis on the device (no host sync on sigma).
// Same math as BaseDAUKernelCompute::get_kernels with w = 1, mu = 0 and
// unit_normalization = true (reference: src/dau_conv/base_dau_conv_layer.cu:402-448, 583-704),
// evaluated in double and rounded to float once.
#include "dau_common.hpp"

namespace dau {

// plane_pitch: floats between the six 2-D planes of `out`; taps: the eight 1-D factor arrays (may be null)
__global__ void synth_filters_kernel(const float* __restrict__ sigma_dev, int k, int flags, int plane_pitch,
                                     float* __restrict__ out, float* __restrict__ taps) {
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
        out[0 * plane_pitch + t] = (float)gn;                                        // Gn
        out[1 * plane_pitch + t] = (float)gn;                                        // Dw
        out[2 * plane_pitch + t] = (float)(u * inv_s2 * g / Z - gn * s1);            // Dmu1
        out[3 * plane_pitch + t] = (float)(v * inv_s2 * g / Z - gn * s2);            // Dmu2
        out[4 * plane_pitch + t] = (float)((u * u + v * v) * inv_s3 * g / Z - gn * s3);  // Dsigma
        out[5 * plane_pitch + (k - 1 - j) * k + (k - 1 - i)] = (float)gn;            // Gerr = flip(Gn)
    }
    // 1-D factors (exact: the masks of single_dim_kernel / forbid_positive_dim1 are separable too)
    if (taps && lane < k) {
        const double t = lane - c;
        double gxs = 0, gys = 0, sx1 = 0, sy1 = 0, sx2 = 0, sy2 = 0;
        for (int q = 0; q < k; ++q) {
            const double d = q - c;
            const double e = exp(-d * d * 0.5 * inv_s2);
            const double ex = (forbid_pos && d > 0) ? 0.0 : e;
            const double ey = (single_dim && d != 0) ? 0.0 : e;
            gxs += ex; gys += ey;
            sx1 += d * inv_s2 * ex; sy1 += d * inv_s2 * ey;
            sx2 += d * d * inv_s3 * ex; sy2 += d * d * inv_s3 * ey;
        }
        const double e = exp(-t * t * 0.5 * inv_s2);
        const double gx = ((forbid_pos && t > 0) ? 0.0 : e) / gxs;
        const double gy = ((single_dim && t != 0) ? 0.0 : e) / gys;
        const double m1 = sx1 / gxs, m2 = sy1 / gys, m3 = sx2 / gxs + sy2 / gys;   // = s1, s2, s3 of the 2-D form
        taps[kTapGX * kTapPitch + lane] = (float)gx;
        taps[kTapGY * kTapPitch + lane] = (float)gy;
        taps[kTapAX * kTapPitch + lane] = (float)((t * inv_s2 - m1) * gx);
        taps[kTapAY * kTapPitch + lane] = (float)((t * inv_s2 - m2) * gy);
        taps[kTapCX * kTapPitch + lane] = (float)((t * t * inv_s3 - m3) * gx);
        taps[kTapBY * kTapPitch + lane] = (float)(t * t * inv_s3 * gy);
        taps[kTapGXR * kTapPitch + (k - 1 - lane)] = (float)gx;
        taps[kTapGYR * kTapPitch + (k - 1 - lane)] = (float)gy;
    }
}

void launch_synth_filters(hipStream_t st, const float* sigma_dev, int k, int flags, float* filters6) {
    hipLaunchKernelGGL(synth_filters_kernel, dim3(1), dim3(64), 0, st, sigma_dev, k, flags, kFilterPlane, filters6,
                       filters6 + kTaps1dOffset);
}

// the six planes only, k*k floats each, back to back (dau_conv_filters: the caller's buffer, no scratch)
void launch_synth_filters_compact(hipStream_t st, const float* sigma_dev, int k, int flags, float* planes6) {
    hipLaunchKernelGGL(synth_filters_kernel, dim3(1), dim3(64), 0, st, sigma_dev, k, flags, k * k, planes6,
                       static_cast<float*>(nullptr));
}

}  // namespace dau
