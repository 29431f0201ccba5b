// DAU_ALGO_DIRECT: straightforward HIP kernels that accept every shape.  They are the
// always-available device path (small / odd shapes) and the on-device cross-check for the
// tiled kernels.  Math: SURVEY.md Appendix A items 2, 4, 5 (reference:
// src/dau_conv/util/convolve.cu:48-131, dau_conv_forward_core.hpp:804-1605,
// dau_conv_backward_core.hpp:1017-1820).
#include "dau_common.hpp"

namespace dau {

// ---- zero padded correlation with nfilt filters: out[(plane*nfilt + kf)][y][x] ------------
template <int NF>
__global__ void blur_direct_kernel(const float* __restrict__ x, long planes, int H, int W,
                                   const float* __restrict__ filters, int k, float* __restrict__ out) {
    __shared__ float filt[NF * kFilterPlane];
    for (int t = threadIdx.x; t < NF * kFilterPlane; t += blockDim.x) {
        const int kf = t / kFilterPlane, r = t % kFilterPlane;
        filt[t] = r < k * k ? filters[kf * kFilterPlane + r] : 0.0f;
    }
    __syncthreads();
    const long HW = (long)H * W;
    const int c = (k - 1) / 2;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < planes * HW;
         idx += (long)gridDim.x * blockDim.x) {
        const long p = idx / HW;
        const int px = (int)(idx % HW), yy = px / W, xx = px % W;
        const float* xp = x + p * HW;
        float acc[NF];
#pragma unroll
        for (int kf = 0; kf < NF; ++kf) acc[kf] = 0.0f;
        for (int j = 0; j < k; ++j) {
            const int sy = yy + j - c;
            if (sy < 0 || sy >= H) continue;
            for (int i = 0; i < k; ++i) {
                const int sx = xx + i - c;
                if (sx < 0 || sx >= W) continue;
                const float v = xp[(long)sy * W + sx];
#pragma unroll
                for (int kf = 0; kf < NF; ++kf) acc[kf] = fmaf(v, filt[kf * kFilterPlane + j * k + i], acc[kf]);
            }
        }
#pragma unroll
        for (int kf = 0; kf < NF; ++kf) out[(p * NF + kf) * HW + px] = acc[kf];
    }
}

void launch_blur_direct(hipStream_t st, const float* x, long planes, int H, int W, const float* filters,
                        int nfilt, int k, float* out) {
    const long total = planes * (long)H * W;
    const int block = 256;
    const int grid = (int)((total + block - 1) / block < 65536 ? (total + block - 1) / block : 65536);
    if (nfilt == 1)
        hipLaunchKernelGGL(blur_direct_kernel<1>, dim3(grid), dim3(block), 0, st, x, planes, H, W, filters, k, out);
    else
        hipLaunchKernelGGL(blur_direct_kernel<4>, dim3(grid), dim3(block), 0, st, x, planes, H, W, filters, k, out);
}

// ---- offset-and-sum: y[n,f,p] = sum_{s,g} sum_taps w' * xb[n,s,p+o+tap] -------------------
// block = 256 pixels of one (n,f) plane; the unit table entry is wave-uniform (scalar loads).
// The grid is one-dimensional over (n*Fout + f, pixel block): grid.y would cap N*Fout at 65535.
__global__ void gather_sum_direct_kernel(const float* __restrict__ xb, const UnitRef* __restrict__ table,
                                         int Sin, int Fout, int G, int H, int W, int pxblocks, float* __restrict__ y) {
    const int nf = blockIdx.x / pxblocks;
    const int n = nf / Fout, f = nf % Fout;
    const int px = (blockIdx.x % pxblocks) * blockDim.x + threadIdx.x;
    const long HW = (long)H * W;
    if (px >= HW) return;
    const int yy = px / W, xx = px % W;
    float acc = 0.0f;
    for (int s = 0; s < Sin; ++s) {
        const float* plane = xb + ((long)n * Sin + s) * HW;
        for (int g = 0; g < G; ++g) {
            const UnitRef u = table[((long)s * G + g) * Fout + f];
            const int sy = yy + u.oy, sx = xx + u.ox;
            const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
            const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
            const float* q = plane + (long)sy * W + sx;
            const float v00 = (y0 && x0) ? q[0] : 0.0f;
            const float v01 = (y0 && x1) ? q[1] : 0.0f;
            const float v10 = (y1 && x0) ? q[W] : 0.0f;
            const float v11 = (y1 && x1) ? q[W + 1] : 0.0f;
            acc = fmaf(u.w00, v00, acc);
            acc = fmaf(u.w01, v01, acc);
            acc = fmaf(u.w10, v10, acc);
            acc = fmaf(u.w11, v11, acc);
        }
    }
    y[((long)n * Fout + f) * HW + px] = acc;
}

void launch_gather_sum_direct(hipStream_t st, const float* xb, const UnitRef* table, int N, int Sin, int Fout,
                              int G, int H, int W, float* y) {
    // table is indexed [Sin][G][Fout] (f fastest) for both passes
    const int block = 256;
    const long HW = (long)H * W;
    const int pxblocks = (int)((HW + block - 1) / block);
    dim3 grid((unsigned)((long)pxblocks * N * Fout));
    hipLaunchKernelGGL(gather_sum_direct_kernel, grid, dim3(block), 0, st, xb, table, Sin, Fout, G, H, W, pxblocks, y);
}

// ---- offset-and-dot: r_k[s,g,f] = sum_{n,p} err'[n,f,p] * bilinear(xk[n,s,k], p + o) ---------
// one block per (s,f) pair; xk4 layout [N*S][4][H][W]; table holds bare bilinear factors.
template <int G_MAX>
__global__ void gather_dot_direct_kernel(const float* __restrict__ xk4, const float* __restrict__ err,
                                         const UnitRef* __restrict__ table, int N, int S, int F, int G, int H,
                                         int W, int drop_col, int drop_row, float* __restrict__ r4) {
    const int s = blockIdx.x / F, f = blockIdx.x % F;
    const long HW = (long)H * W;
    const long units = (long)S * G * F;
    __shared__ double red[256];
    for (int g0 = 0; g0 < G; g0 += G_MAX) {
        double acc[G_MAX][kNumK];
#pragma unroll
        for (int g = 0; g < G_MAX; ++g)
#pragma unroll
            for (int kk = 0; kk < kNumK; ++kk) acc[g][kk] = 0.0;
        for (long idx = threadIdx.x; idx < (long)N * HW; idx += blockDim.x) {
            const int n = (int)(idx / HW), px = (int)(idx % HW), yy = px / W, xx = px % W;
            float e = err[((long)n * F + f) * HW + px];
            if ((drop_col && xx == W - 1) || (drop_row && yy == H - 1)) e = 0.0f;
            const float* xp = xk4 + ((long)n * S + s) * kNumK * HW;
#pragma unroll
            for (int g = 0; g < G_MAX; ++g) {
                if (g0 + g >= G) break;
                const UnitRef u = table[((long)s * G + g0 + g) * F + f];
                const int sy = yy + u.oy, sx = xx + u.ox;
                const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
                const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
#pragma unroll
                for (int kk = 0; kk < kNumK; ++kk) {
                    const float* q = xp + kk * HW + (long)sy * W + sx;
                    float v = 0.0f;
                    if (y0 && x0) v = fmaf(u.w00, q[0], v);
                    if (y0 && x1) v = fmaf(u.w01, q[1], v);
                    if (y1 && x0) v = fmaf(u.w10, q[W], v);
                    if (y1 && x1) v = fmaf(u.w11, q[W + 1], v);
                    acc[g][kk] += (double)(e * v);
                }
            }
        }
        for (int g = 0; g < G_MAX; ++g) {
            if (g0 + g >= G) break;
            for (int kk = 0; kk < kNumK; ++kk) {
                red[threadIdx.x] = acc[g][kk];
                __syncthreads();
                for (int m = blockDim.x / 2; m >= 1; m >>= 1) {
                    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
                    __syncthreads();
                }
                if (threadIdx.x == 0) r4[kk * units + ((long)s * G + g0 + g) * F + f] = (float)red[0];
                __syncthreads();
            }
        }
    }
}

void launch_gather_dot_direct(hipStream_t st, const float* xk4, const float* err, const UnitRef* table, Shape sh,
                              int drop_col, int drop_row, float* r4) {
    hipLaunchKernelGGL(gather_dot_direct_kernel<2>, dim3(sh.S * sh.F), dim3(256), 0, st, xk4, err, table, sh.N,
                       sh.S, sh.F, sh.G, sh.H, sh.W, drop_col, drop_row, r4);
}

}  // namespace dau
