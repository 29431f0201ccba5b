// v_mfma_f32_4x4x1_16b_f32 with CBSZ/ABID.
//   CBSZ=4, ABID=k: does every block get the A values of block k (lanes 4k..4k+3)?  Expected D[i] of every lane =
//   A[lane 4k+i] * B[lane].                                                       (gather-dot, lane = unit, one input channel)
//   CBSZ=3, ABID=k (k < 8): the 16 blocks form two groups of 8; which blocks make a group, and does every block get the A values
//   of block k OF ITS GROUP?  Hypothesis H1: groups = lanes 0..31 / 32..63 (D[i] = A[32*(lane/32) + 4k + i] * B[lane]).
//   (binned gather-dot: the two half waves work on different input channels)        CBSZ=2 likewise with four groups of 16 lanes.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_abid.hip -o /tmp/mfma_abid && /tmp/mfma_abid
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__device__ void probe(float a, float b, float* out, int lane, int slot) {
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) out[(slot * 64 + lane) * 4 + i] = d[i];
}

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = 100.0f + lane, b = 1.0f + lane;   // products are exact in fp32
    probe<4, 0>(a, b, out, lane, 0); probe<4, 1>(a, b, out, lane, 1); probe<4, 5>(a, b, out, lane, 2); probe<4, 15>(a, b, out, lane, 3);
    probe<3, 0>(a, b, out, lane, 4); probe<3, 3>(a, b, out, lane, 5); probe<3, 7>(a, b, out, lane, 6);
    probe<2, 0>(a, b, out, lane, 7); probe<2, 3>(a, b, out, lane, 8);
}

int main() {
    const int slots = 9;
    float* d; hipMalloc(&d, slots * 64 * 4 * sizeof(float)); hipMemset(d, 0, slots * 64 * 4 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static float h[9 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int cbsz[9] = {4, 4, 4, 4, 3, 3, 3, 2, 2}, abid[9] = {0, 1, 5, 15, 0, 3, 7, 0, 3};
    int bad = 0;
    for (int s = 0; s < slots; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) {
                const int group_lanes = 4 << cbsz[s];                       // lanes that share one A source
                const int src = lane / group_lanes * group_lanes + 4 * abid[s] + i;
                const float want = (100.0f + src) * (1.0f + lane);
                const float got = h[(s * 64 + lane) * 4 + i];
                if (got != want) {
                    if (bad < 12) printf("cbsz %d abid %d lane %d i %d: got %g want %g (A source lane %g)\n", cbsz[s], abid[s], lane, i, got, want, got / (1.0f + lane) - 100.0f);
                    ++bad;
                }
            }
    printf(bad ? "MISMATCH (%d)\n" : "OK: CBSZ=c broadcasts block ABID's A within every group of 2^c consecutive blocks (c = 4, 3, 2) (%d)\n", bad);
    return bad != 0;
}
