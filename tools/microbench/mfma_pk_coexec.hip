// Micro-benchmark: do a v_mfma_f32_4x4x1-only wave and a v_pk_fma_f32-only wave that share ONE SIMD run concurrently?
// A 512-thread workgroup puts waves w and w+4 on the same SIMD (MI355X_MICROARCH.md "Two waves per SIMD"); the role is
// chosen by `wave >= 4`, never by parity.  Every wave times its own loop with s_memtime (shader clock) and the host
// times the launch, so the table shows each stream alone (partner absent), alone with an idle-spinning partner absent,
// both together, and one wave alternating the two streams (what the gather-dot does today).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_pk_coexec mfma_pk_coexec.hip
// Counters: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES -- ./mfma_pk_coexec pmc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kMfmaPerIter = 32;   // 32 x 8 issue cycles
constexpr int kPkPerIter = 64;     // 64 x 4 issue cycles: the same 256 cycles per iteration for either stream

__device__ __forceinline__ void mfma_iter(f4 (&acc)[16], float a, float b) {
#pragma unroll
  for (int m = 0; m < kMfmaPerIter; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m & 15], 0, 0, 0);
}
__device__ __forceinline__ void pk_iter(f2 (&pacc)[16], f2 x) {
#pragma unroll
  for (int p = 0; p < kPkPerIter; ++p) pacc[p & 15] = __builtin_elementwise_fma(pacc[p & 15], x, x);
}

// mode 0: waves 0-3 MFMA only, waves 4-7 leave at once        (MFMA alone, one wave per SIMD)
// mode 1: waves 0-3 leave, waves 4-7 pk only                  (pk alone, one wave per SIMD)
// mode 2: waves 0-3 MFMA only, waves 4-7 pk only              (the question)
// mode 3: all eight waves MFMA only                           (two MFMA waves per SIMD)
// mode 4: all eight waves pk only                             (two pk waves per SIMD)
// mode 5: all eight waves alternate groups of GM MFMA / 2 GM pk, partners in phase (today's gather-dot)
// mode 6: as 5, but waves 4-7 start with the pk group: partners half a period out of phase
// mode 7: roles by parity (wave & 1): both roles land on different SIMDs -> no sharing, the control
template <int MODE, int GM>
__global__ void __launch_bounds__(512) coexec(float* out, long long* cyc, int iters, float wv) {
  const int wave = threadIdx.x >> 6;
  f4 acc[16]; f2 pacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = f4{0, 0, 0, 0}; pacc[i] = f2{1.f, 2.f}; }
  const float a = wv + (threadIdx.x & 3); const f2 x = {1.0f + 1e-6f * threadIdx.x, 0.5f};
  bool do_m, do_p;
  if (MODE == 0) { do_m = wave < 4; do_p = false; }
  else if (MODE == 1) { do_m = false; do_p = wave >= 4; }
  else if (MODE == 2) { do_m = wave < 4; do_p = wave >= 4; }
  else if (MODE == 3) { do_m = true; do_p = false; }
  else if (MODE == 4) { do_m = false; do_p = true; }
  else if (MODE == 7) { do_m = !(wave & 1); do_p = wave & 1; }
  else { do_m = do_p = true; }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (MODE == 5 || MODE == 6) {
    if (MODE == 6 && wave >= 4) {   // half a period ahead: one pk group first
#pragma unroll
      for (int p = 0; p < 2 * GM; ++p) pacc[p & 15] = __builtin_elementwise_fma(pacc[p & 15], x, x);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < kMfmaPerIter / GM; ++g) {
#pragma unroll
        for (int m = 0; m < GM; ++m) acc[(g * GM + m) & 15] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x.x, acc[(g * GM + m) & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2 * GM; ++p) pacc[(g * 2 * GM + p) & 15] = __builtin_elementwise_fma(pacc[(g * 2 * GM + p) & 15], x, x);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) { mfma_iter(acc, a, x.x); __builtin_amdgcn_sched_barrier(0); }
  } else if (do_p) {
    for (int it = 0; it < iters; ++it) { pk_iter(pacc, x); __builtin_amdgcn_sched_barrier(0); }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + pacc[i].x + pacc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_out; static long long* g_cyc; static long long h_cyc[256 * 8];
template <int MODE, int GM> void run(const char* what, int iters, bool quiet) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  coexec<MODE, GM><<<256, 512>>>(g_out, g_cyc, iters, 0.5f); (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0); coexec<MODE, GM><<<256, 512>>>(g_out, g_cyc, iters, 0.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  (void)hipMemcpy(h_cyc, g_cyc, sizeof(h_cyc), hipMemcpyDeviceToHost);
  double lo = 0, hi = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h_cyc[b * 8 + w];
  lo /= 256.0 * 4 * iters; hi /= 256.0 * 4 * iters;
  if (quiet) return;
  // s_memtime ticks (constant-rate counter: ratios between rows matter) next to the launch time
  printf("%-66s launch %7.3f ms | ticks/iter waves0-3 %7.3f  waves4-7 %7.3f\n", what, best, lo, hi);
}

int main(int argc, char** argv) {
  (void)hipMalloc(&g_out, 256 * 512 * 4); (void)hipMalloc(&g_cyc, sizeof(h_cyc));
  const int iters = 4000; const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
  printf("per iteration: MFMA stream = %d v_mfma_f32_4x4x1 (x8 issue cycles), pk stream = %d v_pk_fma_f32 (x4) -> 256 cycles each\n", kMfmaPerIter, kPkPerIter);
  printf("iters %d, 256 workgroups x 512 threads (waves w and w+4 share a SIMD)\n", iters);
  run<0, 8>("0 MFMA-only waves 0-3, waves 4-7 absent", iters, false);
  run<1, 8>("1 pk-only waves 4-7, waves 0-3 absent", iters, false);
  run<2, 8>("2 MFMA-only waves 0-3 BESIDE pk-only waves 4-7 (same SIMDs)", iters, false);
  if (pmc) return 0;   // under the counters: only the three launches that answer the question, in this order
  run<7, 8>("7 MFMA-only even waves, pk-only odd waves (different SIMDs)", iters, false);
  run<3, 8>("3 eight MFMA-only waves (two per SIMD): each does the full stream", iters, false);
  run<4, 8>("4 eight pk-only waves (two per SIMD)", iters, false);
  run<5, 8>("5 every wave alternates 8 MFMA / 16 pk, partners in phase", iters, false);
  run<6, 8>("6 every wave alternates 8 MFMA / 16 pk, waves 4-7 half a period off", iters, false);
  run<5, 16>("5 every wave alternates 16 MFMA / 32 pk, in phase", iters, false);
  run<6, 16>("6 every wave alternates 16 MFMA / 32 pk, half a period off", iters, false);
  run<5, 32>("5 every wave alternates 32 MFMA / 64 pk, in phase", iters, false);
  run<6, 32>("6 every wave alternates 32 MFMA / 64 pk, half a period off", iters, false);
  printf("reading: if mode 2's launch time ~ max(mode 0, mode 1) the pipes co-execute; if ~ sum they share the issue port.\n");
  return 0;
}
