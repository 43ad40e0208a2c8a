// Register-stationary weights for the short-K projections of the 64x40 level (round-4 probe for round 5).
//   C[M, N] = A[M, K] W[N, K]^T, fp16, K = 320, N a multiple of 320 (320: the square projections; 960: q/k/v; 2560: the feed-forward's first linear)
// The production tiles (csrc/gemm.hip) stage both operands through LDS and run 448 workgroups in lock-step: a 35840 x 320 x 320
// projection takes 18.5 us where its 46 MB are 8 us of HBM time.  Here a wave keeps ITS 80 columns of W (5 column tiles x 10 K-steps
// = 50 fragments = 200 registers) in registers for the whole launch and streams 16-row tiles of A straight from global memory into
// MFMA operands: no LDS, no barrier, loads of row tile i + 1 under the 50 MFMAs of tile i, results leave as 8-byte stores.
// Four waves (one per SIMD, 512 registers each) cover N = 320; a workgroup walks a contiguous range of row tiles.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/regw_probe.hip -Lsvd_xtend_amd/csrc -lsvdx -Wl,-rpath,'$ORIGIN/../../svd_xtend_amd/csrc' -o tools/probes/regw_probe
// Timing: `reps` launches captured in one hipGraph on a stream, replayed; both kernels the same way (a replayed launch carries ~1.5 us of boundary).
#include <hip/hip_runtime.h>
#include "../../include/svdx.h"      // the production kernel as the baseline of the same timing loop (link svd_xtend_amd/csrc/libsvdx.so)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 320, KS = K / 32;          // 10 MFMA K-steps
constexpr int CT = 5;                        // column tiles (of 16) per wave: 80 columns

// grid: (workgroups over row tiles, N / 320); block: 256 threads = 4 waves, wave w owns columns [nb * 320 + 80 w, + 80)
// SKIP (probe only): 1 no stores, 2 no A loads after the ring is primed, 4 one MFMA per column tile instead of ten, 8 no weight loads
template <int WPE, int DEPTH, int SKIP = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void regw_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * 320 + wave * 80;
    // weights: fragment (j, ks) = W[n0 + 16 j + r][32 ks + 8 g .. + 8]
    f16x8 wf[CT][KS];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (SKIP & 8) { const f16 v = (f16)(0.01f * (j + ks + lane)); wf[j][ks] = f16x8{v, v, v, v, v, v, v, v}; }
            else wf[j][ks] = *reinterpret_cast<const f16x8*>(W + (size_t)(n0 + 16 * j + r) * K + 32 * ks + 8 * g);
        }
    const int tiles = M / 16;
    const int t0 = (int)((long)blockIdx.x * tiles / gridDim.x), t1 = (int)((long)(blockIdx.x + 1) * tiles / gridDim.x);
    if (t0 >= t1) return;
    // ring of DEPTH row tiles in registers: tile t is multiplied while tiles t + 1 .. t + DEPTH - 1 travel (the loop is unrolled DEPTH times so
    // that the ring index is a compile-time constant: no register copies, and the wait in front of tile t's MFMAs leaves the younger loads in flight)
    f16x8 a[DEPTH][KS];
    auto load = [&](int t, f16x8 (&dst)[KS]) __attribute__((always_inline)) {
        const f16* p = A + (size_t)(min(t, t1 - 1) * 16 + r) * K + 8 * g;          // past the end: a harmless re-read of the last tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const f16x8*>(p + 32 * ks);
    };
    auto tile = [&](int t, const f16x8 (&x)[KS]) __attribute__((always_inline)) {
        f32x4 acc[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ((SKIP & 4) ? 1 : KS); ++ks)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][ks], x[ks], acc[j], 0, 0, 0);
        if (SKIP & 4) {              // keep every operand register alive
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) acc[ks % CT][0] += (float)x[ks][0] + (float)wf[ks % CT][ks][1];
        }
        // acc[j][i] = C[row 16 t + r][n0 + 16 j + 4 g + i]
        f16* c = C + (size_t)(t * 16 + r) * N + n0 + 4 * g;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            f16x4 o = {(f16)acc[j][0], (f16)acc[j][1], (f16)acc[j][2], (f16)acc[j][3]};
            if (!(SKIP & 1) || acc[j][0] == 123.456f) *reinterpret_cast<f16x4*>(c + 16 * j) = o;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load(t0 + d, a[d]);
    for (int t = t0; t < t1; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (t + d < t1) {
                if (!(SKIP & 2)) load(t + d + DEPTH - 1, a[(d + DEPTH - 1) % DEPTH]);
                tile(t + d, a[d]);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int M = 35840;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int N : {320, 960, 2560}) {
        const int NSET = N == 320 ? 8 : 3;                       // rotate operands past the 256 MB Infinity Cache
        std::vector<f16*> As(NSET), Cs(NSET);
        f16* W; hipMalloc(&W, (size_t)N * K * 2);
        std::vector<f16> ha((size_t)M * K), hw((size_t)N * K);
        srand(1);
        for (auto& v : ha) v = (f16)((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : hw) v = (f16)((rand() % 2001 - 1000) / 1000.f * 0.056f);
        hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        for (int s = 0; s < NSET; ++s) {
            hipMalloc(&As[s], (size_t)M * K * 2); hipMalloc(&Cs[s], (size_t)M * N * 2);
            hipMemcpy(As[s], ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
            hipMemset(Cs[s], 0xff, (size_t)M * N * 2);
        }
        hipStream_t st; hipStreamCreate(&st);
        const int reps = 40;
        auto time_graph = [&](auto&& launch) {
            for (int i = 0; i < NSET; ++i) launch(i);
            hipStreamSynchronize(st);
            hipGraph_t gr; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < reps; ++i) launch(i % NSET);
            hipStreamEndCapture(st, &gr);
            hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipGraphExecDestroy(ge); hipGraphDestroy(gr);
            return ms * 1e3 / reps;
        };
        auto check = [&]() {
            std::vector<f16> hc((size_t)16 * N);
            double worst = 0;
            for (int blk : {0, 1117, 2239}) {
                hipMemcpy(hc.data(), Cs[0] + (size_t)blk * 16 * N, (size_t)16 * N * 2, hipMemcpyDeviceToHost);
                for (int rr = 0; rr < 16; ++rr)
                    for (int n = 0; n < N; n += 7) {
                        double ref = 0;
                        for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)(blk * 16 + rr) * K + k] * (double)(float)hw[(size_t)n * K + k];
                        worst = fmax(worst, fabs(ref - (double)(float)hc[(size_t)rr * N + n]));
                    }
            }
            return worst;
        };
        const double bytes = (double)M * K * 2 + (double)M * N * 2 + (double)N * K * 2;
        for (int variant : {6, 26}) {
            const double us = time_graph([&](int s) {
                svdx_gemm(As[s], W, Cs[s], M, N, K, K, K, N, nullptr, nullptr, 0, 0, 0, nullptr, 0, nullptr, As[s], SVDX_OUT_ACT, 1.f, 1, variant, 0, nullptr, nullptr, 0,
                          SVDX_F16, st);
            });
            printf("N=%4d  svdx_gemm variant %2d              %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, variant, us, 2.0 * M * N * K / us / 1e6,
                   bytes / us / 1e6, check());
            for (int s = 0; s < NSET; ++s) hipMemsetAsync(Cs[s], 0xff, (size_t)M * N * 2, st);
        }
        for (int depth : {3}) {
            dim3 grid(cus, N / 320);
            const double us = time_graph([&](int s) {
                if (depth == 2) hipLaunchKernelGGL((regw_kernel<1, 2>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N);
                else if (depth == 3) hipLaunchKernelGGL((regw_kernel<1, 3>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N);
                else if (depth == 4) hipLaunchKernelGGL((regw_kernel<1, 4>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N);
                else if (depth == 5) hipLaunchKernelGGL((regw_kernel<1, 5>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N);
                else hipLaunchKernelGGL((regw_kernel<1, 6>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N);
            });
            printf("N=%4d  register-stationary W, ring of %d tiles, grid %4d x %d   %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, depth, cus,
                   N / 320, us, 2.0 * M * N * K / us / 1e6, bytes / us / 1e6, check());
            for (int s = 0; s < NSET; ++s) hipMemsetAsync(Cs[s], 0xff, (size_t)M * N * 2, st);
        }
        if (N == 320) {
            dim3 grid(cus, 1);
            const char* what[] = {"everything", "no stores", "no A loads", "no stores, no A loads", "1 MFMA per column tile", "1 MFMA, no stores", "1 MFMA, no A loads", "1 MFMA, no stores, no A loads",
                                  "no weight loads", "no weight loads, no stores", "no weight loads, no A loads", "no weight loads, no stores, no A loads"};
            for (int sk = 0; sk < 12; ++sk) {
                const double us = time_graph([&](int s) {
                    switch (sk) {
                    case 0: hipLaunchKernelGGL((regw_kernel<1, 3, 0>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 1: hipLaunchKernelGGL((regw_kernel<1, 3, 1>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 2: hipLaunchKernelGGL((regw_kernel<1, 3, 2>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 3: hipLaunchKernelGGL((regw_kernel<1, 3, 3>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 4: hipLaunchKernelGGL((regw_kernel<1, 3, 4>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 5: hipLaunchKernelGGL((regw_kernel<1, 3, 5>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 6: hipLaunchKernelGGL((regw_kernel<1, 3, 6>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 7: hipLaunchKernelGGL((regw_kernel<1, 3, 7>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 8: hipLaunchKernelGGL((regw_kernel<1, 3, 8>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 9: hipLaunchKernelGGL((regw_kernel<1, 3, 9>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    case 10: hipLaunchKernelGGL((regw_kernel<1, 3, 10>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    default: hipLaunchKernelGGL((regw_kernel<1, 3, 11>), grid, dim3(256), 0, st, As[s], W, Cs[s], M, N); break;
                    }
                });
                printf("N= 320  breakdown (ring of 3): %-40s %7.2f us\n", what[sk], us);
            }
        }
        hipStreamDestroy(st);
        for (int s = 0; s < NSET; ++s) { hipFree(As[s]); hipFree(Cs[s]); }
        hipFree(W);
    }
    return 0;
}
