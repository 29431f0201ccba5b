// v_mfma_f32_4x4x1_16b_f32 with CBSZ/ABID: does CBSZ=4, ABID=k feed every block with the A values of block k
// (lanes 4k..4k+3)?  Expected D[i] of every lane = A[lane 4k+i] * B[lane].
// hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_abid.hip -o /tmp/mfma_abid && /tmp/mfma_abid
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ABID>
__device__ void probe(float a, float b, float* out, int lane) {
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) out[(ABID * 64 + lane) * 4 + i] = d[i];
}

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = 100.0f + lane, b = 1.0f + lane;   // products are exact in fp32
    probe<0>(a, b, out, lane); probe<1>(a, b, out, lane); probe<5>(a, b, out, lane); probe<15>(a, b, out, lane);
}

int main() {
    float* d; hipMalloc(&d, 16 * 64 * 4 * sizeof(float)); hipMemset(d, 0, 16 * 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static float h[16 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int abid : {0, 1, 5, 15})
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) {
                const float want = (100.0f + 4 * abid + i) * (1.0f + lane);
                const float got = h[(abid * 64 + lane) * 4 + i];
                if (got != want) { if (bad < 8) printf("abid %d lane %d i %d: got %g want %g\n", abid, lane, i, got, want); ++bad; }
            }
    printf(bad ? "MISMATCH (%d)\n" : "OK: CBSZ=4 broadcasts block ABID's A to all 16 blocks (%d)\n", bad);
    return bad != 0;
}
