// A host program in C++ that drives the DAU operator through the C ABI alone (include/dau_conv.h + the HIP runtime):
// what the reference's TensorFlow op, or any other host with a C FFI, would do.  No Python, no torch.
//
//   hipcc -O2 -Iinclude examples/cabi_host.cpp -Ldau-convnet_amd/dau_conv -ldau_conv_hip -Wl,-rpath,'$ORIGIN/../dau-convnet_amd/dau_conv' -o build/cabi_host
//   build/cabi_host out.bin      -> writes [header | x w mu1 mu2 dy | y dx dw dmu1 dmu2 dsigma] as raw little-endian floats
//
// tests/test_gpu_cabi_host.py runs it on the GPU box and checks the outputs against the oracle on the inputs it wrote.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dau_conv.h"

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define DAU_OK_OR_DIE(x)                                                                       \
    do {                                                                                       \
        int rc_ = (x);                                                                         \
        if (rc_ != DAU_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, dau_conv_last_error()); return 3; } \
    } while (0)

static uint32_t lcg_state = 12345u;
static float uniform01() {   // 24-bit LCG sample in [0,1)
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

template <class T>
static T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 1; }
    const int N = 3, S = 5, F = 9, G = 4, H = 20, W = 24, K = 9;
    const size_t nx = (size_t)N * S * H * W, ny = (size_t)N * F * H * W, np = (size_t)S * G * F;
    std::vector<float> x(nx), dy(ny), w(np), mu1(np), mu2(np), sigma(np, 0.5f);
    for (auto& v : x) v = uniform01();
    for (auto& v : dy) v = uniform01() * 2.0f - 1.0f;
    for (auto& v : w) v = (uniform01() - 0.5f) * 0.4f;
    for (auto& v : mu1) v = (uniform01() * 2.0f - 1.0f) * 3.0f;
    for (auto& v : mu2) v = (uniform01() * 2.0f - 1.0f) * 3.0f;

    dau_conv_desc d{};
    d.struct_size = sizeof(d);
    d.batch = N; d.in_channels = S; d.out_channels = F; d.units_per_channel = G; d.height = H; d.width = W;
    d.max_kernel_size = K; d.number_units_ignore = 0; d.flags = DAU_FLAG_USE_INTERPOLATION; d.algo = DAU_ALGO_AUTO;
    d.sigma_hint = 0.5f; d.mu_learning_rate_factor = 1.0f;
    dau_conv_plan* plan = nullptr;
    DAU_OK_OR_DIE(dau_conv_plan_create(&d, &plan));
    size_t ws_f = 0, ws_b = 0;
    DAU_OK_OR_DIE(dau_conv_workspace_bytes(plan, DAU_PASS_FORWARD, &ws_f));
    DAU_OK_OR_DIE(dau_conv_workspace_bytes(plan, DAU_PASS_BACKWARD, &ws_b));
    const size_t ws_bytes = ws_f > ws_b ? ws_f : ws_b;

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    float *dx_ = to_device(x), *ddy = to_device(dy), *dw_ = to_device(w), *dm1 = to_device(mu1), *dm2 = to_device(mu2),
          *dsg = to_device(sigma);
    if (!dx_ || !ddy || !dw_ || !dm1 || !dm2 || !dsg) { fprintf(stderr, "device allocation failed\n"); return 2; }
    float *y, *gx, *gw, *gm1, *gm2, *gs;
    void* ws;
    HIP_OK(hipMalloc(&y, ny * 4)); HIP_OK(hipMalloc(&gx, nx * 4)); HIP_OK(hipMalloc(&gw, np * 4));
    HIP_OK(hipMalloc(&gm1, np * 4)); HIP_OK(hipMalloc(&gm2, np * 4)); HIP_OK(hipMalloc(&gs, np * 4));
    HIP_OK(hipMalloc(&ws, ws_bytes));

    DAU_OK_OR_DIE(dau_conv_forward(plan, st, dx_, dw_, dm1, dm2, dsg, y, ws, ws_bytes));
    float max_mu = 0.0f;
    DAU_OK_OR_DIE(dau_conv_check_status(plan, st, ws, &max_mu));
    DAU_OK_OR_DIE(dau_conv_backward(plan, st, dx_, ddy, dw_, dm1, dm2, dsg, gx, gw, gm1, gm2, gs, ws, ws_bytes, DAU_NEED_ALL));
    DAU_OK_OR_DIE(dau_conv_check_status(plan, st, ws, &max_mu));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> hy(ny), hgx(nx), hgw(np), hgm1(np), hgm2(np), hgs(np);
    HIP_OK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hgx.data(), gx, nx * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hgw.data(), gw, np * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hgm1.data(), gm1, np * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hgm2.data(), gm2, np * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hgs.data(), gs, np * 4, hipMemcpyDeviceToHost));

    FILE* f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 1; }
    const int32_t header[8] = {N, S, F, G, H, W, K, dau_conv_abi_version()};
    fwrite(header, sizeof(header), 1, f);
    for (const std::vector<float>* v : {&x, &w, &mu1, &mu2, &dy, &hy, &hgx, &hgw, &hgm1, &hgm2, &hgs})
        fwrite(v->data(), sizeof(float), v->size(), f);
    fclose(f);
    dau_conv_plan_destroy(plan);
    printf("cabi_host ok: max|mu| = %.3f, wrote %s\n", max_mu, argv[1]);
    return 0;
}
