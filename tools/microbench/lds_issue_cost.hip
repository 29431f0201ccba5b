// What does an LDS read instruction cost the issuing SIMD when the matrix pipe is the intended bottleneck?
// Per iteration: 8 x v_mfma_f32_4x4x1 (64 cycles of matrix pipe) + 16 B/lane... variants of LDS reads delivering the
// same 32 B per lane: 4 x ds_read_b64 | 2 x ds_read2_b64 | 2 x ds_read_b128 | none.  16 waves per CU (4 per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_issue_cost.hip -o build/lds_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f2 v[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}};
    f4 q[2] = {{1, 2, 3, 4}, {5, 6, 7, 8}};
    const unsigned a8 = (unsigned)(wave * 2048 + lane * 8);     // conflict-free 8 B per lane
    const unsigned a16 = (unsigned)(wave * 2048 + lane * 16);   // conflict-free 16 B per lane
    const float w = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "v"(a8) : "memory");
        } else if (MODE == 2) {
            asm volatile("ds_read2_b64 %0, %2 offset0:0 offset1:64\n\tds_read2_b64 %1, %2 offset0:128 offset1:192"
                         : "=v"(q[0]), "=v"(q[1]) : "v"(a8) : "memory");
        } else if (MODE == 3) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(q[0]), "=v"(q[1]) : "v"(a16) : "memory");
        }
        if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float b[8];
        if (MODE == 1 || MODE == 0) { b[0] = v[0].x; b[1] = v[0].y; b[2] = v[1].x; b[3] = v[1].y; b[4] = v[2].x; b[5] = v[2].y; b[6] = v[3].x; b[7] = v[3].y; }
        else { b[0] = q[0][0]; b[1] = q[0][1]; b[2] = q[0][2]; b[3] = q[0][3]; b[4] = q[1][0]; b[5] = q[1][1]; b[6] = q[1][2]; b[7] = q[1][3]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, b[j], acc[j & 3], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(1024), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 1024 * sizeof(float));
    const int iters = 20000;
    const char* names[4] = {"8 MFMA only", "8 MFMA + 4 ds_read_b64", "8 MFMA + 2 ds_read2_b64", "8 MFMA + 2 ds_read_b128"};
    float ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
    for (int m = 0; m < 4; ++m) {
        // 4 blocks per CU in sequence, 16 waves each = 4 waves per SIMD: SIMD cycles per iteration of ONE wave at 2.4 GHz nominal
        const double clk = ms[m] * 1e-3 * 2.4e9 / (4.0 * iters) / 4.0;
        printf("%-28s %8.3f ms   %.1f cycles per wave-iteration (nominal 2.4 GHz), MFMA rate %.1f TFLOP/s\n", names[m], ms[m], clk,
               256.0 * 4 * 16 * iters * 8 * 512 / (ms[m] * 1e-3) * 1e-12);
    }
    return 0;
}
