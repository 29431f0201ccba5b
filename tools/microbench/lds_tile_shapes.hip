// ds_read_b64 rate for the lane -> position maps of the gather-sum tiles: 8 x 8 positions (rows 40 positions apart: the four rows of a
// half wave fall on four different 16-bank groups) against 32 x 2 (a half wave reads 32 consecutive positions), at several
// displacements.  16 waves per CU, reads only (lgkmcnt waited once per 8 reads).
// hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_tile_shapes.hip -o build/lds_tile_shapes && build/lds_tile_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(1024) k(float* out, int iters, int tw, int pitch, int disp) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 1024];
    for (int i = threadIdx.x; i < 32 * 1024; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int ly = lane / tw, lx = lane % tw;
    const unsigned a = (unsigned)(((ly + 4) * pitch + lx + 4 + disp) * 8);
    f2 v[8];
    float s = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:2560\n\tds_read_b64 %2, %8 offset:5120\n\tds_read_b64 %3, %8 offset:7680\n\t"
                     "ds_read_b64 %4, %8 offset:10240\n\tds_read_b64 %5, %8 offset:12800\n\tds_read_b64 %6, %8 offset:15360\n\tds_read_b64 %7, %8 offset:17920\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a) : "memory");
        s += v[0].x + v[7].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tw : {8, 32, 16})
        for (int pitch : {40, 72})
            for (int disp : {0, 1, 3, 4, 17}) {
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, 10, tw, pitch, disp);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, iters, tw, pitch, disp);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // one CU: 16 waves x iters x 8 reads; cycles per read at 2.4 GHz nominal
                printf("tile %2d x %d pitch %3d disp %2d: %.3f ms = %.2f clk per ds_read_b64 (ideal 4.0)\n", tw, 64 / tw, pitch, disp, ms,
                       ms * 1e-3 * 2.4e9 / (16.0 * iters * 8));
            }
    return 0;
}
