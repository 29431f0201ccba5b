/*
 * dau_oracle.c -- CPU restatement of the DAU convolution forward/backward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call it, and only as the checker / timed CPU baseline.  The product path
 * (dau-convnet_amd/) never routes through this file and has no CPU fallback.
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ hold inputs and outputs produced by
 * the reference's own numpy oracle (class DAUConvPython,
 * plugins/tensorflow/tests/dau_conv_test.py:13-295) executed in the build
 * container by tests/golden/make_golden.py; tests/test_oracle_golden.py checks
 * every function below against them.
 *
 * What is restated (reference file:line it follows):
 *   dau_oracle_filters        dau_conv_test.py:177-220 (_get_filters),
 *                             base_dau_conv_layer.cu:402-448,583-704 (get_kernels),
 *                             support rule base_dau_conv_layer.cpp:146-147
 *   dau_oracle_blur           dau_conv_test.py:86-88 (scipy correlate, mode='constant'),
 *                             convolve.cu:48-131 (zero padded correlation)
 *   dau_oracle_unit_table     dau_conv_forward_core.hpp:2025-2028,2135-2213
 *                             (floor, fractions, four bilinear weights)
 *   dau_oracle_offset_and_sum dau_conv_test.py:14-61
 *   dau_oracle_offset_and_dot dau_conv_test.py:95-175 (incl. the unit_testing edge rule :110-136)
 *   dau_oracle_forward        dau_conv_test.py:64-93
 *   dau_oracle_backward       dau_conv_test.py:222-295, base_dau_conv_layer.cu:335-355
 *                             (dmu,dsigma *= w ; ignored units zeroed ; NaN -> 0),
 *                             dau_conv_grad_op.cpp:297-303 (mu learning-rate factor)
 *
 * Layouts: activations NCHW contiguous; parameters [1,S,G,F] contiguous (f fastest).
 * Accumulation is in double and rounded to float once per output element, so the
 * oracle is at least as accurate as the float32-accumulating numpy original.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DAU_ORACLE_API __attribute__((visibility("default")))

DAU_ORACLE_API int dau_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Filter support of the C++ reference: 2*ceil(5*sigma)+1 (base_dau_conv_layer.cpp:146). */
DAU_ORACLE_API int dau_oracle_filter_support(float sigma) {
    return 2 * (int)ceilf(5.0f * sigma) + 1;
}

/*
 * Six k x k filters (row-major, [j][i] with i = x): Gn (blur), Dw (= Gn), Dmu1, Dmu2,
 * Dsigma and Gerr (Gn flipped in both axes).  unit_normalization = true.
 */
DAU_ORACLE_API void dau_oracle_filters(float sigma_f, int k, int single_dim_kernel,
                                       int forbid_positive_dim1, float *Gn, float *Dw,
                                       float *Dmu1, float *Dmu2, float *Dsigma, float *Gerr) {
    const double sigma = (double)sigma_f;
    const int c = (k - 1) / 2;
    const int n = k * k;
    double *g = (double *)malloc(sizeof(double) * n * 4);
    double *d1 = g + n, *d2 = g + 2 * n, *ds = g + 3 * n;
    double Z = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i) {
            const double u = i - c, v = j - c;
            double gv = exp(-(u * u + v * v) / (2.0 * sigma * sigma));
            if (single_dim_kernel && v != 0) gv = 0;      /* base_dau_conv_layer.cu:432-434 */
            if (forbid_positive_dim1 && u > 0) gv = 0;    /* :436-438 */
            g[j * k + i] = gv;
            d1[j * k + i] = u / (sigma * sigma) * gv;
            d2[j * k + i] = v / (sigma * sigma) * gv;
            ds[j * k + i] = (u * u + v * v) / (sigma * sigma * sigma) * gv;
            Z += gv; s1 += d1[j * k + i]; s2 += d2[j * k + i]; s3 += ds[j * k + i];
        }
    s1 /= Z; s2 /= Z; s3 /= Z;
    for (int t = 0; t < n; ++t) {
        const double gn = g[t] / Z;
        if (Gn) Gn[t] = (float)gn;
        if (Dw) Dw[t] = (float)gn;
        if (Dmu1) Dmu1[t] = (float)(d1[t] / Z - gn * s1);
        if (Dmu2) Dmu2[t] = (float)(d2[t] / Z - gn * s2);
        if (Dsigma) Dsigma[t] = (float)(ds[t] / Z - gn * s3);
    }
    if (Gerr)
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i) Gerr[j * k + i] = (float)(g[(k - 1 - j) * k + (k - 1 - i)] / Z);
    free(g);
}

/* Zero-padded correlation of `planes` HxW planes with one k x k filter. */
DAU_ORACLE_API void dau_oracle_blur(const float *x, long planes, int H, int W, const float *filt,
                                    int k, float *out) {
    const int c = (k - 1) / 2;
#pragma omp parallel
    {
        double *row = (double *)malloc(sizeof(double) * W);
#pragma omp for schedule(static)
        for (long p = 0; p < planes; ++p) {
            const float *xp = x + p * (long)H * W;
            float *op = out + p * (long)H * W;
            for (int y = 0; y < H; ++y) {
                for (int xx = 0; xx < W; ++xx) row[xx] = 0.0;
                for (int j = 0; j < k; ++j) {
                    const int yy = y + j - c;
                    if (yy < 0 || yy >= H) continue;
                    for (int i = 0; i < k; ++i) {
                        const double fv = filt[j * k + i];
                        if (fv == 0.0) continue;
                        const int sh = i - c; /* reads x[xx + sh] */
                        const int lo = sh < 0 ? -sh : 0;
                        const int hi = sh > 0 ? W - sh : W;
                        const float *src = xp + (long)yy * W + sh;
                        for (int xx = lo; xx < hi; ++xx) row[xx] += fv * (double)src[xx];
                    }
                }
                for (int xx = 0; xx < W; ++xx) op[(long)y * W + xx] = (float)row[xx];
            }
        }
        free(row);
    }
}

/*
 * Per-unit bookkeeping, float32 arithmetic exactly as the reference kernels do it:
 * integer offsets by floorf, fractions by float subtraction, four bilinear factors
 * b[dy][dx] by float products.  This is the "bit-exact" part of the parity contract.
 * out_off[2*u] = ox, out_off[2*u+1] = oy ; out_b[4*u + 2*dy + dx].
 */
DAU_ORACLE_API void dau_oracle_unit_table(const float *mu1, const float *mu2, long units,
                                          int use_interpolation, int32_t *out_off, float *out_b) {
    for (long u = 0; u < units; ++u) {
        const float m1 = mu1[u], m2 = mu2[u];
        const float fox = floorf(m1), foy = floorf(m2);
        float fx = m1 - fox, fy = m2 - foy;
        if (!use_interpolation) { fx = 0.0f; fy = 0.0f; }
        out_off[2 * u] = (int32_t)fox;
        out_off[2 * u + 1] = (int32_t)foy;
        out_b[4 * u + 0] = (1.0f - fx) * (1.0f - fy);
        out_b[4 * u + 1] = fx * (1.0f - fy);
        out_b[4 * u + 2] = (1.0f - fx) * fy;
        out_b[4 * u + 3] = fx * fy;
    }
}

static inline void axpy_shifted(double *acc, const float *plane, int H, int W, int oy, int ox,
                                double coef) {
    /* acc[y][x] += coef * plane[y+oy][x+ox], plane = 0 outside */
    const int y0 = oy < 0 ? -oy : 0, y1 = oy > 0 ? H - oy : H;
    const int x0 = ox < 0 ? -ox : 0, x1 = ox > 0 ? W - ox : W;
    if (coef == 0.0) return;
    for (int y = y0; y < y1; ++y) {
        const float *src = plane + (long)(y + oy) * W + ox;
        double *dst = acc + (long)y * W;
        for (int x = x0; x < x1; ++x) dst[x] += coef * (double)src[x];
    }
}

/*
 * y[n,f] = sum_{s, g < G-ignore} w[s,g,f] * bilinear(xb[n,s], . + (mu2,mu1)[s,g,f]).
 * Parameter strides are explicit so the same routine serves the input-gradient pass,
 * where the parameter tensors are read with s and f exchanged (np.swapaxes(.,1,3),
 * dau_conv_test.py:232-236).  ps/pg/pf = strides of the (in-channel, unit, out-channel)
 * indices in the parameter arrays; Sin/Fout = channel counts of xb / y.
 */
DAU_ORACLE_API void dau_oracle_offset_and_sum_strided(const float *xb, int N, int Sin, int Fout,
                                                      int G, int H, int W, const float *w,
                                                      const float *mu1, const float *mu2,
                                                      float mu_sign, long ps, long pg, long pf,
                                                      int ignore, int use_interpolation, float *y) {
    const long HW = (long)H * W;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * HW);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int f = 0; f < Fout; ++f) {
                for (long t = 0; t < HW; ++t) acc[t] = 0.0;
                for (int s = 0; s < Sin; ++s) {
                    const float *plane = xb + ((long)n * Sin + s) * HW;
                    for (int g = 0; g < G - ignore; ++g) {
                        const long pi = s * ps + g * pg + f * pf;
                        const float wv = w[pi];
                        const float m1 = mu_sign * mu1[pi], m2 = mu_sign * mu2[pi];
                        const float fox = floorf(m1), foy = floorf(m2);
                        float fx = m1 - fox, fy = m2 - foy;
                        if (!use_interpolation) { fx = 0.0f; fy = 0.0f; }
                        const int ox = (int)fox, oy = (int)foy;
                        /* premultiplied weights in float32, as the numpy oracle does */
                        const float w00 = wv * (1.0f - fx) * (1.0f - fy);
                        const float w01 = wv * fx * (1.0f - fy);
                        const float w10 = wv * (1.0f - fx) * fy;
                        const float w11 = wv * fx * fy;
                        axpy_shifted(acc, plane, H, W, oy, ox, w00);
                        if (use_interpolation) {
                            axpy_shifted(acc, plane, H, W, oy, ox + 1, w01);
                            axpy_shifted(acc, plane, H, W, oy + 1, ox, w10);
                            axpy_shifted(acc, plane, H, W, oy + 1, ox + 1, w11);
                        }
                    }
                }
                float *yp = y + ((long)n * Fout + f) * HW;
                for (long t = 0; t < HW; ++t) yp[t] = (float)acc[t];
            }
        free(acc);
    }
}

DAU_ORACLE_API void dau_oracle_offset_and_sum(const float *xb, int N, int S, int F, int G, int H,
                                              int W, const float *w, const float *mu1,
                                              const float *mu2, int ignore, int use_interpolation,
                                              float *y) {
    dau_oracle_offset_and_sum_strided(xb, N, S, F, G, H, W, w, mu1, mu2, 1.0f, (long)G * F, F, 1,
                                      ignore, use_interpolation, y);
}

static inline double dot_shifted(const float *plane, const float *err, int H, int W, int oy,
                                 int ox) {
    /* sum_{y,x} plane[y+oy][x+ox] * err[y][x] */
    const int y0 = oy < 0 ? -oy : 0, y1 = oy > 0 ? H - oy : H;
    const int x0 = ox < 0 ? -ox : 0, x1 = ox > 0 ? W - ox : W;
    double sum = 0.0;
    for (int y = y0; y < y1; ++y) {
        const float *src = plane + (long)(y + oy) * W + ox;
        const float *e = err + (long)y * W;
        double rs = 0.0;
        for (int x = x0; x < x1; ++x) rs += (double)src[x] * (double)e[x];
        sum += rs;
    }
    return sum;
}

/* The numpy oracle's rule for dropping the last error column/row (dau_conv_test.py:110-136). */
static int edge_disabled(int size) {
    if (size >= 64) return size % 64 == 0;
    if (size >= 32) return size % 32 == 0;
    if (size >= 16) return size % 16 == 0;
    if (size >= 8) return size % 8 == 0;
    return 0;
}

/*
 * out[s,g,f] = sum_{n,y,x} err[n,f,y,x] * bilinear(xk[n,s], (y,x) + (mu2,mu1)[s,g,f]),
 * ignored units -> 0.  err must already carry the unit_testing edge rule if wanted
 * (see dau_oracle_apply_edge_rule).
 */
DAU_ORACLE_API void dau_oracle_offset_and_dot(const float *xk, const float *err, int N, int S,
                                              int F, int G, int H, int W, const float *mu1,
                                              const float *mu2, int ignore, int use_interpolation,
                                              float *out) {
    const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int f = 0; f < F; ++f)
            for (int g = 0; g < G; ++g) {
                const long pi = ((long)s * G + g) * F + f;
                if (g >= G - ignore) { out[pi] = 0.0f; continue; }
                const float m1 = mu1[pi], m2 = mu2[pi];
                const float fox = floorf(m1), foy = floorf(m2);
                float fx = m1 - fox, fy = m2 - foy;
                if (!use_interpolation) { fx = 0.0f; fy = 0.0f; }
                const int ox = (int)fox, oy = (int)foy;
                const float b00 = (1.0f - fx) * (1.0f - fy), b01 = fx * (1.0f - fy);
                const float b10 = (1.0f - fx) * fy, b11 = fx * fy;
                double d00 = 0, d01 = 0, d10 = 0, d11 = 0;
                for (int n = 0; n < N; ++n) {
                    const float *plane = xk + ((long)n * S + s) * HW;
                    const float *e = err + ((long)n * F + f) * HW;
                    d00 += dot_shifted(plane, e, H, W, oy, ox);
                    if (use_interpolation) {
                        d01 += dot_shifted(plane, e, H, W, oy, ox + 1);
                        d10 += dot_shifted(plane, e, H, W, oy + 1, ox);
                        d11 += dot_shifted(plane, e, H, W, oy + 1, ox + 1);
                    }
                }
                out[pi] = (float)(d00 * b00 + d01 * b01 + d10 * b10 + d11 * b11);
            }
}

/* err_out = err with last column/row zeroed under the unit_testing rule. */
DAU_ORACLE_API void dau_oracle_apply_edge_rule(const float *err, long planes, int H, int W,
                                               float *err_out) {
    const int drop_col = edge_disabled(W), drop_row = edge_disabled(H);
    memcpy(err_out, err, sizeof(float) * planes * H * W);
    for (long p = 0; p < planes; ++p) {
        float *e = err_out + p * (long)H * W;
        if (drop_col) for (int y = 0; y < H; ++y) e[(long)y * W + W - 1] = 0.0f;
        if (drop_row) for (int x = 0; x < W; ++x) e[(long)(H - 1) * W + x] = 0.0f;
    }
}

/* Forward: blur with Gn (support k, or the support rule when k <= 0), then offset-and-sum. */
DAU_ORACLE_API void dau_oracle_forward(const float *x, int N, int S, int F, int G, int H, int W,
                                       const float *w, const float *mu1, const float *mu2,
                                       float sigma, int k, int ignore, int use_interpolation,
                                       int single_dim_kernel, int forbid_positive_dim1, float *y) {
    if (k <= 0) k = dau_oracle_filter_support(sigma);
    float *Gn = (float *)malloc(sizeof(float) * k * k);
    float *xb = (float *)malloc(sizeof(float) * (long)N * S * H * W);
    dau_oracle_filters(sigma, k, single_dim_kernel, forbid_positive_dim1, Gn, 0, 0, 0, 0, 0);
    dau_oracle_blur(x, (long)N * S, H, W, Gn, k, xb);
    dau_oracle_offset_and_sum(xb, N, S, F, G, H, W, w, mu1, mu2, ignore, use_interpolation, y);
    free(xb);
    free(Gn);
}

/*
 * Backward.  dx = forward(dy * Gerr ; params with s<->f exchanged, offsets negated) using
 * the UNMODIFIED dy; dw = r0 ; dmu1 = w*r1*lr ; dmu2 = w*r2*lr ; dsigma = w*r3, where
 * r_k = offset_and_dot(x * D_k, dy'), dy' = dy under the unit_testing edge rule.
 * Any output pointer may be NULL to skip it.
 */
DAU_ORACLE_API void dau_oracle_backward(const float *x, const float *dy, int N, int S, int F, int G,
                                        int H, int W, const float *w, const float *mu1,
                                        const float *mu2, float sigma, int k, int ignore,
                                        int use_interpolation, int single_dim_kernel,
                                        int forbid_positive_dim1, int unit_testing,
                                        float mu_learning_rate_factor, float *dx, float *dw,
                                        float *dmu1, float *dmu2, float *dsigma) {
    if (k <= 0) k = dau_oracle_filter_support(sigma);
    const long HW = (long)H * W, units = (long)S * G * F;
    float *filt = (float *)malloc(sizeof(float) * k * k * 6);
    float *Gn = filt, *Dw = filt + k * k, *D1 = filt + 2 * k * k, *D2 = filt + 3 * k * k,
          *Ds = filt + 4 * k * k, *Ge = filt + 5 * k * k;
    dau_oracle_filters(sigma, k, single_dim_kernel, forbid_positive_dim1, Gn, Dw, D1, D2, Ds, Ge);
    if (dx) {
        float *eb = (float *)malloc(sizeof(float) * (long)N * F * HW);
        dau_oracle_blur(dy, (long)N * F, H, W, Ge, k, eb);
        /* parameters [S,G,F] read as [F,G,S]: in-channel index (f) has stride 1,
           out-channel index (s) has stride G*F */
        dau_oracle_offset_and_sum_strided(eb, N, F, S, G, H, W, w, mu1, mu2, -1.0f, 1, F,
                                          (long)G * F, 0, use_interpolation, dx);
        free(eb);
    }
    if (dw || dmu1 || dmu2 || dsigma) {
        float *e2 = (float *)malloc(sizeof(float) * (long)N * F * HW);
        if (unit_testing) dau_oracle_apply_edge_rule(dy, (long)N * F, H, W, e2);
        else memcpy(e2, dy, sizeof(float) * (long)N * F * HW);
        float *xk = (float *)malloc(sizeof(float) * (long)N * S * HW);
        float *r = (float *)malloc(sizeof(float) * units);
        const float *D[4] = {Dw, D1, D2, Ds};
        float *outp[4] = {dw, dmu1, dmu2, dsigma};
        for (int kk = 0; kk < 4; ++kk) {
            if (!outp[kk]) continue;
            dau_oracle_blur(x, (long)N * S, H, W, D[kk], k, xk);
            dau_oracle_offset_and_dot(xk, e2, N, S, F, G, H, W, mu1, mu2, ignore, use_interpolation, r);
            for (long u = 0; u < units; ++u) {
                float v = r[u];
                if (kk > 0) v *= w[u];                                   /* base_dau_conv_layer.cu:335-337 */
                if (kk == 1 || kk == 2) v *= mu_learning_rate_factor;   /* dau_conv_grad_op.cpp:297-303 */
                if ((kk == 1 || kk == 2) && v != v) v = 0.0f;            /* base_dau_conv_layer.cu:354-355 */
                outp[kk][u] = v;
            }
        }
        free(r); free(xk); free(e2);
    }
    free(filt);
}
