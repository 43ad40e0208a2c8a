// Operand-ingest microbenchmark: bytes per second a CU pulls from L2 / the fabric with the access shape of the GEMM staging
// (16 B per lane, 8 lanes per 128-byte row segment, 8 rows per wave instruction), by destination:
//   vgpr   global_load_dwordx4 into registers
//   lds    global_load_lds_dwordx4 (LDS-DMA), nothing reads the LDS
//   mixed  waves 0-3 lds, waves 4-7 vgpr (is the limit per path or shared?)
// and by footprint: every workgroup streams its own rows (A-operand like), all workgroups stream the same rows (B-operand like: L2 /
// L1 hits), or a region small enough to sit in the CU's 32 KB L1.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/ingest_probe.hip -o tools/probes/ingest_probe && tools/probes/ingest_probe
// Round 4 question: the GEMM K-loops sit at 21-24 B/clk/CU of staged operand bytes (profiles/r1_gemm_ingest_probe.txt) while the L2
// delivers ~64 B/clk/CU in aggregate -- is that a property of the LDS-DMA path that a register-path operand would not share?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROW_BYTES = 128;        // one K-step of a row: 64 fp16
constexpr int UNROLL = 4;             // loads in flight per wave before the wait (x 16 waves x 1 KB = 64 KB per CU)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// mode 0 vgpr, 1 lds, 2 mixed.  Each workgroup (8 waves) walks `rows` rows x `ksteps` K-steps: K-step kt reads bytes [kt*128, kt*128+128)
// of every row (row pitch ld); wave w takes rows w*8 + 64*j.
template <int MODE>
__global__ __launch_bounds__(512) void ingest_kernel(const char* __restrict__ src, long block_stride, int rows, int ld, int ksteps, int reps,
                                                     unsigned* sink) {
    extern __shared__ char smem[];                 // 8 waves x UNROLL x 1 KB landing slots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane >> 3, c = lane & 7;
    const char* base = src + (long)blockIdx.x * block_stride + (long)(wave * 8 + r) * ld + c * 16;
    const bool to_lds = MODE == 1 || (MODE == 2 && wave < 4);
    unsigned acc = 0;
    const int passes = rows / 64;                  // wave instructions per K-step and wave: a multiple of UNROLL, or 1 (then ksteps is one)
    auto burst = [&](const char* p0, long step) __attribute__((always_inline)) {
        if (to_lds) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p0 + u * step),
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * UNROLL + u) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p0 + u * step) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");     // the values exist from here on
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].w;
        }
    };
    for (int rep = 0; rep < reps; ++rep) {
        if (passes >= UNROLL) {
            for (int kt = 0; kt < ksteps; ++kt)
                for (int j0 = 0; j0 < passes; j0 += UNROLL) burst(base + (long)j0 * 64 * ld + kt * ROW_BYTES, 64L * ld);
        } else {
            for (int kt = 0; kt < ksteps; kt += UNROLL) burst(base + kt * ROW_BYTES, ROW_BYTES);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;          // keeps the register loads alive
}

struct Case { const char* name; int rows, ld, ksteps; bool shared; };

int main() {
    const int blocks_per_cu[] = {1, 2};
    int dev_cus = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); dev_cus = prop.multiProcessorCount;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const Case cases[] = {
        {"private rows, 512 rows x 40 K-steps (A-like: 2.6 MB per workgroup, streams from the fabric / MALL)", 512, 5120, 40, false},
        {"shared rows, 512 rows x 40 K-steps (B-like: every workgroup reads the same 2.6 MB: L2 hits)", 512, 5120, 40, true},
        {"private rows, 64 rows x 4 K-steps re-read (32 KB per workgroup: L1-resident)", 64, 512, 4, false},
        {"private rows, 512 rows x 4 K-steps re-read (256 KB per workgroup: L2-resident)", 512, 512, 4, false},
    };
    const size_t total = (size_t)512 * 512 * 5120 + (1 << 20);
    char* src; unsigned* sink;
    hipMalloc(&src, total); hipMalloc(&sink, 64);
    hipMemset(src, 1, total);
    printf("# %s, %d CUs, nominal %d MHz; GB/s per CU and chip-wide TB/s; B/clk at the nominal clock (the sustained clock is lower under load)\n",
           prop.name, dev_cus, clk_khz / 1000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Case& cs : cases) {
        for (int bpc : blocks_per_cu) {
            const int blocks = dev_cus * bpc;
            const long stride = cs.shared ? 0 : (long)cs.rows * cs.ld;
            if (!cs.shared && (size_t)blocks * stride > total) continue;
            const long bytes_per_rep = (long)cs.rows * cs.ksteps * ROW_BYTES;
            const int reps = (int)((64L << 20) / bytes_per_rep) + 1;          // ~64 MB per workgroup
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e30f;
                for (int t = 0; t < 4; ++t) {
                    hipEventRecord(e0);
                    const size_t sh = 8 * UNROLL * 1024;
                    if (mode == 0) hipLaunchKernelGGL(ingest_kernel<0>, dim3(blocks), dim3(512), sh, 0, src, stride, cs.rows, cs.ld, cs.ksteps, reps, sink);
                    else if (mode == 1) hipLaunchKernelGGL(ingest_kernel<1>, dim3(blocks), dim3(512), sh, 0, src, stride, cs.rows, cs.ld, cs.ksteps, reps, sink);
                    else hipLaunchKernelGGL(ingest_kernel<2>, dim3(blocks), dim3(512), sh, 0, src, stride, cs.rows, cs.ld, cs.ksteps, reps, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (t > 0 && ms < best) best = ms;
                }
                const double bytes = (double)blocks * reps * bytes_per_rep;
                const double gbs_cu = bytes / (best * 1e-3) / 1e9 / dev_cus;
                printf("%-100s  wg/CU %d  %-5s  %8.1f GB/s/CU  %6.2f TB/s  %5.1f B/clk\n", cs.name, bpc, mode == 0 ? "vgpr" : mode == 1 ? "lds" : "mixed",
                       gbs_cu, gbs_cu * dev_cus / 1e3, gbs_cu * 1e9 / (clk_khz * 1e3));
            }
        }
    }
    return 0;
}
