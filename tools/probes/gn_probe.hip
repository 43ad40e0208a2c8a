// What bounds gn_apply (csrc/norm.hip compiled in with probe macros)?  Build variants:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGN_PROBE_SWAPGRID] [-DGN_PROBE_NOPRO] tools/probes/gn_probe.hip svd_xtend_amd/csrc/common.cpp -o tools/probes/gn_probe_X
#include "../../svd_xtend_amd/csrc/norm.hip"
#include <vector>

__global__ void copy_scale_kernel(const f16* __restrict__ in, f16* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float v[8];
        load8<f16>(in + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = siluf_(v[e] * 1.01f + 0.5f);
        store8<f16>(out + i * 8, v);
    }
}

int main() {
    const int n_s = 14, rows = 2560, C = 320;
    const long M = (long)n_s * rows;
    const int NSET = 6;
    f16 *x[NSET], *y[NSET]; float *gamma, *beta, *stats;
    std::vector<f16> h(M * C);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (f16)(((s >> 8) & 0xffff) / 32768.f - 1.f); }
    for (int i = 0; i < NSET; ++i) { hipMalloc(&x[i], M * C * 2); hipMalloc(&y[i], M * C * 2); hipMemcpy(x[i], h.data(), M * C * 2, hipMemcpyHostToDevice); }
    std::vector<float> ones(C, 1.f), zeros(C, 0.f);
    hipMalloc(&gamma, C * 4); hipMalloc(&beta, C * 4); hipMalloc(&stats, SVDX_GN_REPLICAS * n_s * 32 * SVDX_GN_STAT_FLOATS * 4);
    hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(beta, zeros.data(), C * 4, hipMemcpyHostToDevice);
    svdx_gn_stats(x[0], stats, n_s, rows, C, 32, 0, SVDX_F16, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto fn, double bytes) {
        for (int i = 0; i < 3; ++i) fn(i % NSET);
        hipEventRecord(e0);
        for (int i = 0; i < 30; ++i) fn(i % NSET);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.2f us  %6.0f GB/s\n", name, ms / 30 * 1e3, bytes / (ms / 30 * 1e-3) / 1e9);
    };
    run("gn_apply silu", [&](int i) { svdx_gn_apply(x[i], stats, gamma, beta, y[i], n_s, rows, C, 32, 1e-5f, 1, SVDX_F16, 0); }, 2.0 * M * C * 2);
    run("gn_apply no silu", [&](int i) { svdx_gn_apply(x[i], stats, gamma, beta, y[i], n_s, rows, C, 32, 1e-5f, 0, SVDX_F16, 0); }, 2.0 * M * C * 2);
    run("gn_stats", [&](int i) { svdx_gn_stats(x[i], stats, n_s, rows, C, 32, 1, SVDX_F16, 0); }, 1.0 * M * C * 2);
    for (int blocks : {1120, 2240, 4096}) {
        char nm[64]; snprintf(nm, 64, "copy+silu %d blocks x256", blocks);
        run(nm, [&](int i) { hipLaunchKernelGGL(copy_scale_kernel, dim3(blocks), dim3(256), 0, 0, x[i], y[i], M * C / 8); }, 2.0 * M * C * 2);
    }
    run("copy+silu 1120 blocks x240", [&](int i) { hipLaunchKernelGGL(copy_scale_kernel, dim3(1120), dim3(240), 0, 0, x[i], y[i], M * C / 8); }, 2.0 * M * C * 2);
    return 0;
}
