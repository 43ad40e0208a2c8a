// pp_probe.hip -- feasibility probe for a "ping-pong" NT GEMM main loop on gfx950 (not product code).
// 512 threads = two groups of 4 waves; each SIMD hosts one wave of each group.  The groups own the upper / lower half of a
// (64*MB) x (32*NB) tile and alternate: while group P runs its MFMAs on fragments already in registers, group Q fetches its
// fragments from LDS (and vice versa), separated by s_barrier; LDS-DMA staging runs 3 K-steps (of 32) ahead in a 4-slot ring.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/pp_probe.hip -o tools/probes/pp_probe
// run:   pp_probe M N K
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Params { const f16* A; const f16* B; f16* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n; int a_bytes, b_bytes; };

template <int NB, int MB, int FL>   // FL: timing experiments -- 1 no staging in the loop, 2 no fragment reads in the loop, 4 no barriers
__global__ __launch_bounds__(512) void pp_kernel(Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM4 = 32 * MB, WM4 = 16 * MB, BMT = 2 * BM4, BN3 = 32 * NB, WN3 = 16 * NB;
    constexpr int KT = 32, NSTG = 4, LOOK = 3;
    constexpr int STAGE = (BMT + BN3) * KT * 2;
    constexpr int NLA = (BMT + 127) / 128, NLB = (BN3 + 127) / 128, PP = NLA + NLB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int bid = blockIdx.x;
    const int pid_m = bid / p.tiles_n, pid_n = bid - pid_m * p.tiles_n;
    const int m0 = (FL & 8) ? 0 : pid_m * BMT, n0 = (FL & 8) ? 0 : pid_n * BN3;      // FL 8: every block reads the same tiles (L2-resident)
    const int nt = p.K / KT;
    const int ld_row = tid >> 2, pc = tid & 3;
    const int lc = pc ^ (((ld_row >> 2) & 1) * 3);
    int voa[NLA], vob[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int r = i * 128 + ld_row;
        voa[i] = (r < BMT && m0 + r < p.M) ? ((m0 + r) * p.lda + lc * 8) * 2 : (int)0x80000000;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int r = i * 128 + ld_row;
        vob[i] = (r < BN3 && n0 + r < p.N) ? ((n0 + r) * p.ldb + lc * 8) * 2 : (int)0x80000000;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.B), 0, p.b_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* scratch = smem + NSTG * STAGE + wave_u * 1024;
    auto issue_piece = [&](int kt, int pi) __attribute__((always_inline)) {
        char* As = smem + (kt & (NSTG - 1)) * STAGE;
        char* Bs = As + BMT * KT * 2;
        const int so = kt * KT * 2;
        if (pi < NLA) {
            char* dst = (pi * 128 + 128 <= BMT || pi * 128 + wave_u * 16 < BMT) ? As + (pi * 512 + wave_u * 64) * 16 : scratch;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voa[pi], so, 0, 0);
        } else {
            const int i = pi - NLA;
            char* dst = (i * 128 + 128 <= BN3 || i * 128 + wave_u * 16 < BN3) ? Bs + (i * 512 + wave_u * 64) * 16 : scratch;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, vob[i], so, 0, 0);
        }
    };
    f32x4 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int chunk = (fg ^ (((fr >> 2) & 1) * 3)) * 16;
    f16x8 af[MB], bf[NB];
#define LOAD_FRAGS(kt)                                                                                        \
    {                                                                                                         \
        const char* As_ = smem + ((kt) & (NSTG - 1)) * STAGE;                                                 \
        const char* Bs_ = As_ + BMT * KT * 2;                                                                 \
        _Pragma("unroll") for (int i = 0; i < MB; ++i)                                                        \
            af[i] = *reinterpret_cast<const f16x8*>(As_ + (grp * BM4 + wm * WM4 + i * 16 + fr) * 64 + chunk); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                        \
            bf[i] = *reinterpret_cast<const f16x8*>(Bs_ + (wn * WN3 + i * 16 + fr) * 64 + chunk);             \
    }
#define COMPUTE(kt_issue, ISSUE)                                                                              \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                      \
            _Pragma("unroll") for (int j = 0; j < MB; ++j)                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);         \
            if (ISSUE && i < PP) {                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                issue_piece(kt_issue, i);                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
        }                                                                                                     \
    }
    static_assert(PP <= NB, "pieces must fit between the MFMA rows");
    // prologue: LOOK stages in flight
#pragma unroll
    for (int s = 0; s < LOOK; ++s)
#pragma unroll
        for (int pi = 0; pi < PP; ++pi) issue_piece(s, pi);
    wait_vmcnt<(LOOK - 1) * PP>();
    __builtin_amdgcn_s_barrier();
#define ISSUE_STAGE(kt) { if (!(FL & 1)) { _Pragma("unroll") for (int pi = 0; pi < PP; ++pi) issue_piece(kt, pi); } }
    // staging pieces are issued in a group's LOAD segment (an LDS-DMA piece costs its wave ~60-180 issue cycles: inside the
    // compute segment that starved the MFMA pipe)
    if (grp == 0) {
        LOAD_FRAGS(0);
        int it = 0;
        for (; it < nt - LOOK; ++it) {
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();                       // B1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            COMPUTE(0, false);
            if (!(FL & 1)) wait_vmcnt<PP>();                                   // own pieces of stage it+1 landed (it+2 may fly)
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();                       // B2
            if (!(FL & 2)) LOAD_FRAGS(it + 1);
            ISSUE_STAGE(it + LOOK);
        }
        for (; it < nt; ++it) {
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            COMPUTE(0, false);
            wait_vmcnt<0>();
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();
            if (it + 1 < nt) LOAD_FRAGS(it + 1);
        }
    } else {
        int it = 0;
        for (; it < nt - LOOK; ++it) {
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();                       // B1
            if (!(FL & 2) || it == 0) LOAD_FRAGS(it);
            ISSUE_STAGE(it + LOOK);
            if (!(FL & 1)) wait_vmcnt<2 * PP>();                               // own pieces of stage it+1 landed (it+2, it+3 may fly)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();                       // B2
            COMPUTE(0, false);
        }
        for (; it < nt; ++it) {
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();
            LOAD_FRAGS(it);
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(FL & 4)) __builtin_amdgcn_s_barrier();
            COMPUTE(0, false);
        }
    }
    // direct epilogue: lane (fr, fg) owns row fr and columns fg*4..+3 of every 16x16 block (acc is transposed: mfma(B, A))
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = m0 + grp * BM4 + wm * WM4 + j * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + wn * WN3 + i * 16 + fg * 4;
            if (n >= p.N) continue;
            f16 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)acc[i][j][e];
            *reinterpret_cast<uint64_t*>(p.C + (size_t)m * p.ldc + n) = *reinterpret_cast<uint64_t*>(o);
        }
    }
#endif
}

__global__ void ref_kernel(const f16* A, const f16* B, float* out, int M, int N, int K, int lda, int ldb, const int* ms, const int* ns, int cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)ms[i] * lda + k] * (float)B[(size_t)ns[i] * ldb + k];
    out[i] = s;
}

template <int NB, int MB, int FL>
float run(Params p, int iters) {
    constexpr int LDS = 4 * (64 * MB + 32 * NB) * 32 * 2 + 8192;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_kernel<NB, MB, FL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    p.tiles_m = (p.M + 64 * MB - 1) / (64 * MB);
    p.tiles_n = (p.N + 32 * NB - 1) / (32 * NB);
    dim3 grid(p.tiles_m * p.tiles_n);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pp_kernel<NB, MB, FL>), grid, dim3(512), LDS, 0, p);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pp_kernel<NB, MB, FL>), grid, dim3(512), LDS, 0, p);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    std::vector<f16> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = (f16)rnd();
    for (auto& v : hB) v = (f16)(rnd() * 0.25f);
    f16 *A, *B, *C;
    CHECK(hipMalloc(&A, hA.size() * 2)); CHECK(hipMalloc(&B, hB.size() * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 2));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    Params p{A, B, C, M, N, K, K, K, N, 0, 0, (int)((size_t)M * K * 2), (int)((size_t)N * K * 2)};
    const int cnt = 4096;
    std::vector<int> ms(cnt), ns(cnt);
    for (int i = 0; i < cnt; ++i) { s = s * 1664525u + 1013904223u; ms[i] = (s >> 8) % M; s = s * 1664525u + 1013904223u; ns[i] = (s >> 8) % N; }
    ms[0] = M - 1; ns[0] = N - 1; ms[1] = 0; ns[1] = 0;
    int *dms, *dns; float* dref;
    CHECK(hipMalloc(&dms, cnt * 4)); CHECK(hipMalloc(&dns, cnt * 4)); CHECK(hipMalloc(&dref, cnt * 4));
    CHECK(hipMemcpy(dms, ms.data(), cnt * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dns, ns.data(), cnt * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3(cnt / 256), dim3(256), 0, 0, A, B, dref, M, N, K, K, K, dms, dns, cnt);
    std::vector<float> ref(cnt);
    CHECK(hipMemcpy(ref.data(), dref, cnt * 4, hipMemcpyDeviceToHost));
    for (int cfg = 0; cfg < 3; ++cfg) {
        if ((cfg < 2 && N % 160) || (cfg == 2 && N % 128)) continue;
        CHECK(hipMemset(C, 0, (size_t)M * N * 2));
        float t = cfg == 0 ? run<5, 4, 0>(p, 20) : cfg == 1 ? run<5, 5, 0>(p, 20) : run<4, 4, 0>(p, 20);
        std::vector<f16> hC((size_t)M * N);
        CHECK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int i = 0; i < cnt; ++i) {
            double e = fabs((double)(float)hC[(size_t)ms[i] * N + ns[i]] - ref[i]) / (fabs(ref[i]) + 1.0);
            if (e > maxerr) maxerr = e;
        }
        const char* nm = cfg == 0 ? "256x160" : cfg == 1 ? "320x160" : "256x128";
        printf("pp %s  M=%d N=%d K=%d  %.3f ms  %.1f TF/s  maxrelerr %.2e\n", nm, M, N, K, t, 2.0 * M * N * K / t / 1e9, maxerr);
    }
    if (N % 128 == 0) {
        printf("experiments (256x128; results wrong by construction): ");
        printf("no-staging %.0f  ", 2.0 * M * N * K / run<4, 4, 1>(p, 20) / 1e9);
        printf("no-fragreads %.0f  ", 2.0 * M * N * K / run<4, 4, 2>(p, 20) / 1e9);
        printf("neither %.0f  ", 2.0 * M * N * K / run<4, 4, 3>(p, 20) / 1e9);
        printf("neither+nobarrier %.0f  ", 2.0 * M * N * K / run<4, 4, 7>(p, 20) / 1e9);
        printf("all-but-barriers %.0f  ", 2.0 * M * N * K / run<4, 4, 4>(p, 20) / 1e9);
        printf("same-tile(L2 hits) %.0f  same-tile+no-fragreads %.0f TF/s\n", 2.0 * M * N * K / run<4, 4, 8>(p, 20) / 1e9,
               2.0 * M * N * K / run<4, 4, 10>(p, 20) / 1e9);
    }
    return 0;
}
