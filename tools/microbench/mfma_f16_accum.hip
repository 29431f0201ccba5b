// Micro-benchmark: how does v_mfma_f32_32x32x16_f16 round?  (round 4: the two-limb dense gather-sum of k_dense_split.hip sums
// 2352 MFMAs into one fp32 accumulator; its error against the oracle was 4x that of the exact fp32 gather.)
// Part 1 -- single-step probes: C plus a few exactly known products, against what round-to-nearest / truncation would give.
// Part 2 -- chains: D = sum of M MFMAs of random f16 data (every row of A and every column of B identical, so every element of
//           D is the same 16 M-term dot product), chained through C, against a double reference; the same sum as an fp32 fmaf
//           chain and with the MFMA's C = 0 and the running sum kept by v_add_f32 (round-to-nearest).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f16_accum mfma_f16_accum.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a[t][16], b[t][16] (f16 bits as float for simplicity), c0; out[0] = chained result, out[1] = C=0 + v_add, out[2] = fmaf chain
__global__ void chain(const float* a, const float* b, int M, float c0, int flush, float* out) {
  const int h = threadIdx.x >> 5;
  f32x16 acc, run;
  for (int i = 0; i < 16; ++i) { acc[i] = c0; run[i] = c0; }
  f32x16 hier;                       // chained within groups of `flush` MFMAs (from zero), groups summed by v_add_f32
  f32x16 part;
  for (int i = 0; i < 16; ++i) { hier[i] = c0; part[i] = 0.0f; }
  float f = c0;
  for (int t = 0; t < M; ++t) {
    f16x8 av, bv;
    for (int k = 0; k < 8; ++k) { av[k] = (_Float16)a[t * 16 + 8 * h + k]; bv[k] = (_Float16)b[t * 16 + 8 * h + k]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, z, 0, 0, 0);
    for (int i = 0; i < 16; ++i) run[i] += z[i];
    part = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, part, 0, 0, 0);
    if ((t + 1) % flush == 0 || t + 1 == M) { for (int i = 0; i < 16; ++i) { hier[i] += part[i]; part[i] = 0.0f; } }
    for (int k = 0; k < 16; ++k) f = fmaf((float)(_Float16)a[t * 16 + k], (float)(_Float16)b[t * 16 + k], f);
  }
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = run[0]; out[2] = f; out[3] = hier[0]; }
}

static float* d_a; static float* d_b; static float* d_o;
static void run_chain(const std::vector<float>& a, const std::vector<float>& b, int M, float c0, int flush, float o[4]) {
  (void)hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  chain<<<1, 64>>>(d_a, d_b, M, c0, flush, d_o);
  (void)hipMemcpy(o, d_o, 16, hipMemcpyDeviceToHost);
}
static float f16r(float v) { return (float)(_Float16)v; }

int main() {
  const int MAXM = 8192;
  (void)hipMalloc(&d_a, MAXM * 16 * 4); (void)hipMalloc(&d_b, MAXM * 16 * 4); (void)hipMalloc(&d_o, 64);
  const float ulp = ldexpf(1.0f, -23);          // ulp of [1, 2)
  printf("== part 1: single MFMA, C = 1.5, products p_k (in ulps of C); round-to-nearest-even vs truncation\n");
  struct Probe { const char* what; std::vector<float> p; } probes[] = {
    {"one product +0.75 ulp", {0.75f}}, {"one product +0.5 ulp (tie)", {0.5f}}, {"one product +0.25 ulp", {0.25f}},
    {"one product -0.75 ulp", {-0.75f}}, {"one product -0.25 ulp", {-0.25f}},
    {"three products of +0.25 ulp (sum 0.75)", {0.25f, 0.25f, 0.25f}}, {"sixteen products of +1/16 ulp (sum 1.0)", std::vector<float>(16, 1.0f / 16)},
    {"+1.75 ulp", {1.75f}}, {"+1.25 ulp", {1.25f}}, {"eight of +0.125 and eight of -0.0625 (sum 0.5 tie)", {}},
  };
  for (auto& pr : probes) {
    std::vector<float> a(16, 0.0f), b(16, 0.0f);
    if (pr.p.empty()) { for (int k = 0; k < 16; ++k) pr.p.push_back(k < 8 ? 0.125f : -0.0625f); }
    double sum = 0;
    for (size_t k = 0; k < pr.p.size(); ++k) { a[k] = ldexpf(1.0f, -12); b[k] = pr.p[k] * ldexpf(1.0f, -11); sum += pr.p[k]; }   // a*b = p ulp
    float o[4]; run_chain(a, b, 1, 1.5f, 1, o);
    printf("  %-52s exact sum %+7.4f ulp -> MFMA %+5.2f ulp   (v_add of C=0 result %+5.2f, fmaf chain %+5.2f)\n", pr.what, sum, (o[0] - 1.5f) / ulp,
           (o[1] - 1.5f) / ulp, (o[2] - 1.5f) / ulp);
  }
  printf("== part 2: chains of M MFMAs, random f16 data; |error| / |exact sum| and / sum|a b| (median and max over 40 seeds)\n");
  for (int mode = 0; mode < 2; ++mode) {
    printf("  %s\n", mode == 0 ? "zero-mean data: a ~ U(-1,1), b ~ U(-1,1)" : "one-sided data: a ~ U(0,1), b ~ U(0,1) (the accumulator grows steadily)");
    for (int M : {16, 147, 588, 2352, 8192}) {
      std::vector<double> e0, e1, e2, e3;
      for (int seed = 0; seed < 40; ++seed) {
        srand(1000 * mode + 17 * seed + M);
        std::vector<float> a(M * 16), b(M * 16);
        double exact = 0, sabs = 0;
        for (int i = 0; i < M * 16; ++i) {
          float u = rand() / (float)RAND_MAX, v = rand() / (float)RAND_MAX;
          if (mode == 0) { u = 2 * u - 1; v = 2 * v - 1; }
          a[i] = f16r(u); b[i] = f16r(v);
          exact += (double)a[i] * b[i]; sabs += fabs((double)a[i] * b[i]);
        }
        float o[4]; run_chain(a, b, M, 0.0f, 49, o);
        e0.push_back(fabs(o[0] - exact) / sabs); e1.push_back(fabs(o[1] - exact) / sabs); e2.push_back(fabs(o[2] - exact) / sabs); e3.push_back(fabs(o[3] - exact) / sabs);
      }
      auto stat = [](std::vector<double>& v, double* med, double* mx) { std::sort(v.begin(), v.end()); *med = v[v.size() / 2]; *mx = v.back(); };
      double m0, x0, m1, x1, m2, x2, m3, x3; stat(e0, &m0, &x0); stat(e1, &m1, &x1); stat(e2, &m2, &x2); stat(e3, &m3, &x3);
      printf("    M = %5d (%6d products): err / sum|ab|  MFMA chain med %.2e max %.2e | C=0 + v_add med %.2e max %.2e | fmaf chain med %.2e max %.2e | chains of 49 + v_add med %.2e max %.2e\n",
             M, M * 16, m0, x0, m1, x1, m2, x2, m3, x3);
    }
  }
  return 0;
}
