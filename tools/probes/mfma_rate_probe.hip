// How fast does ONE wave issue v_mfma_f32_16x16x32_f16, and how many waves per SIMD does the matrix pipe need to run at its 16-clock rate?
// (round-4 question behind tools/probes/regw_probe: a kernel that fits one wave per SIMD measured ~35 clocks per MFMA)
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe && tools/probes/mfma_rate_probe
// Each wave runs ITER x CH MFMAs on CH independent accumulator chains (operands in registers, no memory); one workgroup per CU with
// 4 x W waves (W per SIMD).  Prints clocks per MFMA per SIMD from s_memtime-free wall time: uses the shader clock counter.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(1024) void mfma_kernel(float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
void run(int waves_per_simd, float* out, unsigned long long* cyc, int cus) {
    const int iters = 20000 / CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_kernel<CH>, dim3(cus), dim3(256 * waves_per_simd), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_kernel<CH>, dim3(cus), dim3(256 * waves_per_simd), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * CH * waves_per_simd;
    const double tflops = (double)cus * 4 * n_per_simd * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%d wave(s) per SIMD, %d independent chains: %7.1f TFLOP/s chip-wide; counter ticks per MFMA per SIMD %.2f (s_memtime units); %.3f ms\n", waves_per_simd, CH, tflops,
           (double)h[0] / n_per_simd, ms);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* cyc; hipMalloc(&out, 64); hipMalloc(&cyc, 8 * 1024);
    printf("# %d CUs; dense f16 peak 2500 TFLOP/s = 16 shader clocks per 16x16x32 MFMA per SIMD at 2.4 GHz\n", cus);
    for (int w : {1, 2, 4}) { run<1>(w, out, cyc, cus); run<2>(w, out, cyc, cus); run<4>(w, out, cyc, cus); run<5>(w, out, cyc, cus); run<8>(w, out, cyc, cus); }
    return 0;
}
