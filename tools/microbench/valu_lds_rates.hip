// Micro-benchmarks that size the DAU gather kernels on gfx950:
//  (1) v_fma_f32 / v_pk_fma_f32 issue rate vs waves per SIMD
//  (2) ds_read_b32 / b64 (aligned / 4-byte-misaligned) rate with FMAs interleaved
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_lds_rates valu_lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int MODE>
__global__ void valu_kernel(float* out, int iters, float a, float b) {
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 c = {acc[i], acc[i+1]};
        f2 x = {b, b};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(x));
        acc[i] = c.x; acc[i+1] = c.y;
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS read + FMA mix. WIDTH: 1=b32, 2=b64. MISALIGN: byte offset added. FMAS per read.
template<int WIDTH, int FMAS>
__global__ void lds_kernel(float* out, int iters, int misalign, float a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.5f;
  __syncthreads();
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  unsigned base = (threadIdx.x & 63) * (4 * WIDTH) + misalign + (threadIdx.x >> 6) * 1024;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (WIDTH == 1) {
        float v;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(r * 512));
        asm volatile("s_waitcnt lgkmcnt(6)");
#pragma unroll
        for (int k = 0; k < FMAS; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(r * FMAS + k) & 15]) : "s"(a), "v"(v));
      } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(r * 1024));
        asm volatile("s_waitcnt lgkmcnt(6)");
#pragma unroll
        for (int k = 0; k < FMAS; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[(r * FMAS + k) & 15]) : "s"(a), "v"((k & 1) ? v.y : v.x));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<typename F> float time_ms(F f, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8 * 1024 * sizeof(float)));
  const int iters = 4000;
  printf("== VALU: 32 FMA/lane/iter, %d iters\n", iters);
  for (int mode = 0; mode < 2; ++mode)
    for (int threads : {256, 512, 1024}) {
      int blocks = 256 * (threads == 256 ? 2 : 1);  // 2 blocks of 256 -> still 2 waves/SIMD; report both
      for (int bpc : {1, 2}) {
        if (threads * bpc > 2048) continue;
        int grid = 256 * bpc;
        float ms = time_ms([&] { if (mode == 0) valu_kernel<0><<<grid, threads>>>(out, iters, 1.0001f, 0.5f); else valu_kernel<1><<<grid, threads>>>(out, iters, 1.0001f, 0.5f); });
        double flop = 2.0 * 32 * iters * (double)grid * threads;
        printf("mode=%s threads=%4d blocks/CU=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", mode ? "pk_fma" : "fmac  ", threads, bpc, threads * bpc / 256, ms, flop / ms * 1e-9);
      }
      (void)blocks;
    }
  printf("== LDS: 8 reads/iter, FMAS per read varies; block=512 (2 waves/SIMD), 1 block/CU\n");
  const int li = 2000;
#define RUN_LDS(W, FM, MIS) { float ms = time_ms([&]{ lds_kernel<W, FM><<<256, 512, 65536>>>(out, li, MIS, 1.0001f); }); \
    double reads = 8.0 * li * 256 * 512; double bytes = reads * 4 * W; double fl = reads * FM * 2; \
    printf("width=b%d fmas/read=%2d misalign=%d : %.3f ms  LDS %.1f TB/s (%.1f B/clk/CU @2.4GHz)  FMA %.1f TFLOP/s\n", 32*W, FM, MIS, ms, bytes/ms*1e-9, bytes/ms*1e-6/256/2.4e3*1e-0/1e0, fl/ms*1e-9); }
  RUN_LDS(1, 0, 0) RUN_LDS(1, 2, 0) RUN_LDS(1, 4, 0) RUN_LDS(1, 6, 0) RUN_LDS(1, 8, 0)
  RUN_LDS(2, 0, 0) RUN_LDS(2, 2, 0) RUN_LDS(2, 4, 0) RUN_LDS(2, 6, 0) RUN_LDS(2, 8, 0) RUN_LDS(2, 12, 0)
  RUN_LDS(2, 0, 4) RUN_LDS(2, 4, 4)
  hipFree(out);
  return 0;
}
