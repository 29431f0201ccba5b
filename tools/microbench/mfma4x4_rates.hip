// Micro-benchmark: v_mfma_f32_4x4x1_16b_f32 on gfx950 -- lane layout, issue rate,
// co-issue with ds_read_b64, VALU and v_pk_fma_f32.  Sizes the DAU gather kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4x4_rates mfma4x4_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void layout_kernel(float* out) {
  int l = threadIdx.x;
  float a = 100.f + l;   // A operand
  float b = 1.f + l;     // B operand  (products identify (a-lane, b-lane))
  f4 c = {0, 0, 0, 0};
  f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = d[i];
}

// MODE bits: 1 = ds_read_b64 per 2 mfma ; VALU = extra v_add per mfma ; PK = v_pk_fma per 2 mfma
template<int NACC, bool LDS, int VALU, int PK>
__global__ void __launch_bounds__(512) rate_kernel(float* out, int iters, float wv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (i & 255) * 0.01f;
  __syncthreads();
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  f2 pacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pacc[i] = f2{1.f, 2.f};
  float a = wv + (threadIdx.x & 3);
  unsigned addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 512;
  unsigned vjunk = threadIdx.x;
  f2 x = {1.0f, 2.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; t += 2) {
      if (LDS) {
        asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(4)" : "=v"(x) : "v"(addr), "n"((t / 2) * 1024 % 32768));
      }
      acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x.x, acc[t], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vjunk) : "v"(addr));
      acc[t + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x.y, acc[t + 1], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vjunk) : "v"(addr));
#pragma unroll
      for (int p = 0; p < PK; ++p) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[(t / 2 * PK + p) & 7]) : "v"(x), "v"(x));
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = vjunk;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += pacc[i].x + pacc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<typename F> float time_ms(F f, int reps = 5) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
  layout_kernel<<<1, 64>>>(out);
  std::vector<float> h(256); (void)hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
  printf("== layout: D[reg r] at lane l = A(lane a)*B(lane b): a=(v/...)\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      float v = h[l * 4 + r]; int fa = -1, fb = -1;
      for (int a = 0; a < 64 && fa < 0; ++a) for (int b = 0; b < 64; ++b) if ((100.f + a) * (1.f + b) == v) { fa = a; fb = b; break; }
      printf("  r%d=A%02d*B%02d", r, fa, fb);
    }
    printf("\n");
    if (l == 7) { printf("  ...\n"); l = 55; }
  }
  const int iters = 2000;
#define RUN(NACC, LDS, VALU, PK, THREADS) { float ms = time_ms([&]{ rate_kernel<NACC, LDS, VALU, PK><<<256, THREADS, 65536>>>(out, iters, 0.5f); }); \
    double mf = (double)NACC * iters * 256 * (THREADS / 64); double flop = mf * 512; double pk = mf / 2 * PK * 64 * 4; \
    printf("nacc=%2d lds=%d valu/mfma=%d pk/2mfma=%d waves/SIMD=%d : %.3f ms  MFMA %.1f TFLOP/s (%.2f cyc/mfma/SIMD @2.4GHz)  +pk %.1f TFLOP/s\n", NACC, (int)LDS, VALU, PK, THREADS / 256, ms, flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * (THREADS / 256)), pk / ms * 1e-9); }
  RUN(32, false, 0, 0, 256) RUN(32, false, 0, 0, 512)
  RUN(32, true, 0, 0, 256)  RUN(32, true, 0, 0, 512)
  RUN(32, true, 1, 0, 256)  RUN(32, true, 1, 0, 512)
  RUN(32, true, 2, 0, 256)  RUN(32, true, 2, 0, 512)
  RUN(32, true, 0, 2, 256)  RUN(32, true, 0, 2, 512)
  RUN(32, true, 0, 4, 256)  RUN(32, true, 0, 4, 512)
  RUN(32, false, 0, 4, 256) RUN(32, false, 0, 4, 512)
  (void)hipFree(out);
  return 0;
}
