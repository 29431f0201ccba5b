// Micro-benchmark: cost of alternating v_mfma_f32_4x4x1 with v_pk_fma_f32 at different group sizes.
// Same totals per iteration (32 MFMA + 64 pk_fma), grouped as G x (m MFMA, 2m pk).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int M>   // MFMAs per group (pk per group = 2*M)
__global__ void __launch_bounds__(1024) k(float* out, int iters, float wv) {
  f4 acc[16]; f2 pacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = f4{0,0,0,0}; pacc[i] = f2{1.f, 2.f}; }
  float a = wv + (threadIdx.x & 3); f2 x = {1.0f + threadIdx.x, 2.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 32 / M; ++g) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[(g * M + m) & 15] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x.x, acc[(g * M + m) & 15], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 2 * M; ++p) pacc[(g * 2 * M + p) & 15] = __builtin_elementwise_fma(pacc[(g * 2 * M + p) & 15], x, x);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + pacc[i].x + pacc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float time_ms(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize(); float best = 1e30f;
  for (int r = 0; r < 5; ++r) { (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4); const int iters = 2000;
#define RUN(M, TH) { float ms = time_ms([&]{ k<M><<<256, TH>>>(out, iters, 0.5f); }); double waves = 256.0 * TH / 64; \
    double cyc = ms * 1e-3 * 2.1e9 / (iters * (TH / 256.0)); \
    printf("group=%2d MFMA + %2d pk  waves/SIMD=%d : %.3f ms  %.1f cyc per (32 MFMA + 64 pk) per wave-slot @2.1GHz  (ideal 32*8+64*4=512)  total %.1f TFLOP/s\n", M, 2*M, TH/256, ms, cyc, (32*512.0 + 64*256.0) * iters * waves / ms * 1e-9); }
  RUN(1, 512) RUN(2, 512) RUN(4, 512) RUN(8, 512) RUN(16, 512) RUN(32, 512)
  RUN(1, 1024) RUN(2, 1024) RUN(4, 1024) RUN(8, 1024) RUN(16, 1024) RUN(32, 1024)
  return 0;
}
