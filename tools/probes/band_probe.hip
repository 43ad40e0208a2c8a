// A-stationary band GEMM for the short-K projections of the 64x40 level (round-4 probe for round 5; see DESIGN.md section 8 "what comes next").
//   C[M, N] = A[M, K] W[N, K]^T, fp16, K = 320, N a multiple of 320, M = 35840
// One workgroup of 8 waves per CU owns a band of 140 rows (computed as ten 16-row tiles): the band's A rows are loaded ONCE into LDS (100 KB),
// the weights of a 320-column chunk stream through two 20 KB stages (32 of K each, from L2), waves are tiled 2 (rows) x 4 (columns):
// 5 x 5 accumulator tiles per wave.  256 workgroups = one round; with N > 320 a workgroup walks the column chunks with its band resident.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/band_probe.hip -Lsvd_xtend_amd/csrc -lsvdx -Wl,-rpath,'$ORIGIN/../../svd_xtend_amd/csrc' -o tools/probes/band_probe
// Timing as in regw_probe: 40 launches in one hipGraph, replayed; the production svdx_gemm tile in the same loop.
#include <hip/hip_runtime.h>
#include "../../include/svdx.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 320, KC = K / 8;            // 40 16-byte chunks per A row
constexpr int BAND = 140, BROWS = 160;        // rows owned / rows computed (ten 16-row tiles)
constexpr int NC = 320;                       // columns per chunk
constexpr int A_BYTES = BROWS * K * 2;        // 102,400
constexpr int B_STAGE = NC * 32 * 2;          // 20,480: [320 rows n][4 chunks of 16 B]
constexpr int LDS_BYTES = A_BYTES + 2 * B_STAGE;

// LDS layouts (16-byte chunks; the LDS-DMA writes lane L of an instruction to base + 16 L, so the swizzle is applied to the SOURCE chunk a lane fetches):
//   A band:   row-major, 40 chunks per row, physical chunk pc of row r holds logical chunk (pc & ~7) | ((pc ^ (r >> 1)) & 7)
//   B stage:  row-major, 4 chunks per row n,  physical chunk pc of row n holds logical chunk pc ^ ((n >> 2) & 3)
__global__ __launch_bounds__(512) void band_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves: rows [80 wm, +80), columns [80 wn, +80) of the chunk
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * BAND;
    // ---- the band: 160 rows x 40 chunks = 6400 chunks = 100 wave instructions; wave w issues instructions w, w + 8, ... (12 or 13 each)
    for (int i = wave; i < BROWS * KC / 64; i += 8) {
        const int q = i * 64 + lane, row = q / KC, pc = q - row * KC;
        const int c = (pc & ~7) | ((pc ^ (row >> 1)) & 7);
        const int grow = min(row0 + row, M - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)grow * K + c * 8),
                                         (__attribute__((address_space(3))) void*)(As + (size_t)i * 1024), 16, 0, 0);
    }
    const int nchunks = N / NC;
    // weight stage: 320 rows x 4 chunks = 1280 chunks = 20 wave instructions; wave w issues w, w + 8, w + 16 (< 20)
    auto issue_b = [&](int chunk, int ks, int stage) __attribute__((always_inline)) {
        char* Bs = smem + A_BYTES + stage * B_STAGE;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = wave + 8 * t;
            if (i < 20) {
                const int q = i * 64 + lane, n = q >> 2, pc = q & 3;
                const int c = pc ^ ((n >> 2) & 3);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (size_t)(chunk * NC + n) * K + ks * 32 + c * 8),
                                                 (__attribute__((address_space(3))) void*)(Bs + (size_t)i * 1024), 16, 0, 0);
            }
        }
    };
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        f32x4 acc[5][5];                                       // [column tile][row tile], transposed: a lane holds 4 consecutive columns of one row
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        issue_b(chunk, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ks = 0; ks < K / 32; ++ks) {
            if (ks + 1 < K / 32) issue_b(chunk, ks + 1, (ks + 1) & 1);
            const char* Bs = smem + A_BYTES + (ks & 1) * B_STAGE;
            f16x8 af[5], bf[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = wm * 80 + i * 16 + r;
                const int lc = ks * 4 + g, pc = (lc & ~7) | ((lc ^ (row >> 1)) & 7);
                af[i] = *reinterpret_cast<const f16x8*>(As + (size_t)row * (K * 2) + pc * 16);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = wn * 80 + j * 16 + r;
                bf[j] = *reinterpret_cast<const f16x8*>(Bs + (size_t)n * 64 + ((g ^ ((n >> 2) & 3)) * 16));
            }
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // results: lane (r, g) of tile (j, i) holds C[row0 + 80 wm + 16 i + r][chunk * 320 + 80 wn + 16 j + 4 g .. + 4]
        if (mode == 0) {                                       // 8-byte stores straight from the accumulators
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int lrow = wm * 80 + i * 16 + r;
                if (lrow < BAND && row0 + lrow < M) {
                    f16* c = C + (size_t)(row0 + lrow) * N + chunk * NC + wn * 80 + 4 * g;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        f16x4 o = {(f16)acc[j][i][0], (f16)acc[j][i][1], (f16)acc[j][i][2], (f16)acc[j][i][3]};
                        *reinterpret_cast<f16x4*>(c + 16 * j) = o;
                    }
                }
            }
        } else {                                               // parked in the two weight stages (free now, 40,960 B) and written as full 16-byte row pieces:
            // two row tiles (32 rows x 656 B) at a time, the two wave-row groups taking turns
            char* park = smem + A_BYTES;                        // 40,960 B
            constexpr int PITCH = (NC + 8) * 2;                 // 656 B
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {              // row tiles {0, 1}, {2, 3}, {4} of a wave-row group
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    if (wm == grp) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int i = pass * 2 + ii;
                            if (i < 5) {
#pragma unroll
                                for (int j = 0; j < 5; ++j) {
                                    const f32x4 v = acc[j][i < 5 ? i : 4];
                                    f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                                    *reinterpret_cast<f16x4*>(park + (size_t)(ii * 16 + r) * PITCH + (wn * 80 + j * 16 + 4 * g) * 2) = o;
                                }
                            }
                        }
                    }
                    __syncthreads();
                    const int nrows = pass == 2 ? 16 : 32;
                    for (int q = tid; q < nrows * (NC / 8); q += 512) {          // 40 16-byte pieces per row
                        const int lr = q / (NC / 8), pc = q - lr * (NC / 8);
                        const int lrow = grp * 80 + pass * 32 + lr;
                        if (lrow < BAND && row0 + lrow < M)
                            *reinterpret_cast<f16x8*>(C + (size_t)(row0 + lrow) * N + chunk * NC + pc * 8) = *reinterpret_cast<const f16x8*>(park + (size_t)lr * PITCH + pc * 16);
                    }
                    __syncthreads();
                }
            }
        }
    }
}

// ---- second form: a ninth PRODUCER wave issues every LDS-DMA load (and is the only wave that waits for them), the eight consumer waves never wait on vmcnt,
// so their result stores drain under the next chunk's MFMAs; results leave through a 20 KB park buffer of their own (32 rows x 640 B, 16-byte chunks swizzled
// like the band) as full 16-byte row pieces, and the producer stages the next chunk's first weight tiles during that epilogue.
constexpr int PARK = 32 * NC * 2;             // 20,480
constexpr int LDS2_BYTES = A_BYTES + 2 * B_STAGE + PARK;     // 163,840 = all of the CU's LDS
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int NPROD = 4;                      // producer waves: an LDS-DMA piece costs its wave 60-180 issue cycles, a K-step has 20 of them and 800 clocks of MFMAs
__global__ __launch_bounds__(512 + 64 * NPROD) void band2_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* park = smem + A_BYTES + 2 * B_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * BAND;
    const int nchunks = N / NC, nsteps = nchunks * (K / 32);           // global K-step index s = chunk * 10 + ks, stage s & 1
    // the band: all waves fetch it (no stores are outstanding yet, so everybody may wait for it)
    for (int i = wave; i < BROWS * KC / 64; i += 8 + NPROD) {
        const int q = i * 64 + lane, row = q / KC, pc = q - row * KC;
        const int c = (pc & ~7) | ((pc ^ (row >> 1)) & 7);
        const int grow = min(row0 + row, M - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)grow * K + c * 8),
                                         (__attribute__((address_space(3))) void*)(As + (size_t)i * 1024), 16, 0, 0);
    }
    if (wave >= 8) {                                                    // ---- producers
        const int pw = wave - 8;
        auto issue_b = [&](int s) __attribute__((always_inline)) {
            const int chunk = s / (K / 32), ks = s - chunk * (K / 32);
            char* Bs = smem + A_BYTES + (s & 1) * B_STAGE;
            for (int i = pw; i < 20; i += NPROD) {
                const int q = i * 64 + lane, n = q >> 2, pc = q & 3;
                const int c = pc ^ ((n >> 2) & 3);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (size_t)(chunk * NC + n) * K + ks * 32 + c * 8),
                                                 (__attribute__((address_space(3))) void*)(Bs + (size_t)i * 1024), 16, 0, 0);
            }
        };
        issue_b(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();                                                   // B0: band + step 0 have landed
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) issue_b(s + 1);                          // into the stage the consumers left at the previous barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raw_barrier();                                               // end of step s
            if ((s + 1) % (K / 32) == 0) {                               // the consumers' epilogue barriers of this chunk (5 passes x 2)
                for (int b = 0; b < 10; ++b) raw_barrier();
            }
        }
        return;
    }
    // ---- consumers
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, g = lane >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's pieces of the band
    raw_barrier();                                                       // B0
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        f32x4 acc[5][5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < K / 32; ++ks) {
            const int s = chunk * (K / 32) + ks;
            const char* Bs = smem + A_BYTES + (s & 1) * B_STAGE;
            f16x8 af[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = wm * 80 + i * 16 + r;
                const int lc = ks * 4 + g, pc = (lc & ~7) | ((lc ^ (row >> 1)) & 7);
                af[i] = *reinterpret_cast<const f16x8*>(As + (size_t)row * (K * 2) + pc * 16);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {                                // one weight fragment live at a time (170 registers per wave with nine waves per CU)
                const int n = wn * 80 + j * 16 + r;
                const f16x8 bf = *reinterpret_cast<const f16x8*>(Bs + (size_t)n * 64 + ((g ^ ((n >> 2) & 3)) * 16));
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[i], acc[j][i], 0, 0, 0);
            }
            raw_barrier();                                               // end of step s
        }
        // epilogue: row tile i of both wave-row groups (32 rows) per pass through the park buffer; physical 16-byte chunk of (row lr, logical chunk c): (c & ~7) | ((c ^ (lr >> 1)) & 7)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int lr = wm * 16 + r;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int col = wn * 80 + j * 16 + 4 * g;               // 8-byte half of logical chunk col / 8
                const int c = col >> 3, pc = (c & ~7) | ((c ^ (lr >> 1)) & 7);
                f16x4 o = {(f16)acc[j][i][0], (f16)acc[j][i][1], (f16)acc[j][i][2], (f16)acc[j][i][3]};
                *reinterpret_cast<f16x4*>(park + (size_t)lr * (NC * 2) + pc * 16 + (col & 4) * 2) = o;
            }
            raw_barrier();
            for (int q = tid; q < 32 * (NC / 8); q += 512) {
                const int lr2 = q / (NC / 8), c = q - lr2 * (NC / 8);
                const int lrow = (lr2 >> 4) * 80 + i * 16 + (lr2 & 15);
                const int pc = (c & ~7) | ((c ^ (lr2 >> 1)) & 7);
                if (lrow < BAND && row0 + lrow < M)
                    *reinterpret_cast<f16x8*>(C + (size_t)(row0 + lrow) * N + chunk * NC + c * 8) = *reinterpret_cast<const f16x8*>(park + (size_t)lr2 * (NC * 2) + pc * 16);
            }
            raw_barrier();
        }
    }
}

// ---- third form: as band2, but 16-wide K-steps through FOUR 10 KB weight stages, the producers two steps ahead (counted vmcnt): a stage's load latency is no longer
// exposed in every step
__global__ __launch_bounds__(512 + 64 * NPROD) void band3_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    constexpr int ST3 = NC * 16 * 2;                                    // 10,240: [320 rows n][16 of K] = 32 B per row, FOUR stages (same 40 KB)
    constexpr int KS = K / 16;                                          // 20 K-steps of 16 (v_mfma_f32_16x16x16_f16)
    char* park = smem + A_BYTES + 4 * ST3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * BAND;
    const int nchunks = N / NC, nsteps = nchunks * KS;                 // global K-step index s = chunk * 20 + ks, stage s & 3
    // the band: all waves fetch it (no stores are outstanding yet, so everybody may wait for it)
    for (int i = wave; i < BROWS * KC / 64; i += 8 + NPROD) {
        const int q = i * 64 + lane, row = q / KC, pc = q - row * KC;
        const int c = (pc & ~7) | ((pc ^ (row >> 1)) & 7);
        const int grow = min(row0 + row, M - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)grow * K + c * 8),
                                         (__attribute__((address_space(3))) void*)(As + (size_t)i * 1024), 16, 0, 0);
    }
    if (wave >= 8) {                                                    // ---- producers
        const int pw = wave - 8;
        auto issue_b = [&](int s) __attribute__((always_inline)) {      // 320 rows x 2 chunks = 640 chunks = 10 wave instructions: producers 0 / 1 issue 3, 2 / 3 issue 2
            const int chunk = s / KS, ks = s - chunk * KS;
            char* Bs = smem + A_BYTES + (s & 3) * ST3;
            for (int i = pw; i < 10; i += NPROD) {
                const int q = i * 64 + lane, n = q >> 1, pc = q & 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (size_t)(chunk * NC + n) * K + ks * 16 + pc * 8),
                                                 (__attribute__((address_space(3))) void*)(Bs + (size_t)i * 1024), 16, 0, 0);
            }
        };
        issue_b(0);
        if (nsteps > 1) issue_b(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();                                                   // B0: band + steps 0, 1 have landed
        for (int s = 0; s < nsteps; ++s) {
            if (s + 2 < nsteps) {
                issue_b(s + 2);                                          // two steps ahead, into the stage the consumers left two barriers ago
                if (pw < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // step s + 1 has landed, step s + 2 (3 or 2 pieces of this wave) stays in flight
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            raw_barrier();                                               // end of step s
            if ((s + 1) % KS == 0) {                                     // the consumers' epilogue barriers of this chunk (5 passes x 2)
                for (int b = 0; b < 10; ++b) raw_barrier();
            }
        }
        return;
    }
    // ---- consumers
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, g = lane >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's pieces of the band
    raw_barrier();                                                       // B0
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        f32x4 acc[5][5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            const int s = chunk * KS + ks;
            const char* Bs = smem + A_BYTES + (s & 3) * ST3;
            f16x4 af[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {                                // lane (r, g): A[row][16 ks + 4 g .. + 4] = half (g & 1) of logical chunk 2 ks + (g >> 1)
                const int row = wm * 80 + i * 16 + r;
                const int lc = ks * 2 + (g >> 1), pc = (lc & ~7) | ((lc ^ (row >> 1)) & 7);
                af[i] = *reinterpret_cast<const f16x4*>(As + (size_t)row * (K * 2) + pc * 16 + (g & 1) * 8);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = wn * 80 + j * 16 + r;
                const f16x4 bf = *reinterpret_cast<const f16x4*>(Bs + (size_t)n * 32 + g * 8);
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(bf, af[i], acc[j][i], 0, 0, 0);
            }
            raw_barrier();                                               // end of step s
        }
        // epilogue: row tile i of both wave-row groups (32 rows) per pass through the park buffer; physical 16-byte chunk of (row lr, logical chunk c): (c & ~7) | ((c ^ (lr >> 1)) & 7)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int lr = wm * 16 + r;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int col = wn * 80 + j * 16 + 4 * g;               // 8-byte half of logical chunk col / 8
                const int c = col >> 3, pc = (c & ~7) | ((c ^ (lr >> 1)) & 7);
                f16x4 o = {(f16)acc[j][i][0], (f16)acc[j][i][1], (f16)acc[j][i][2], (f16)acc[j][i][3]};
                *reinterpret_cast<f16x4*>(park + (size_t)lr * (NC * 2) + pc * 16 + (col & 4) * 2) = o;
            }
            raw_barrier();
            for (int q = tid; q < 32 * (NC / 8); q += 512) {
                const int lr2 = q / (NC / 8), c = q - lr2 * (NC / 8);
                const int lrow = (lr2 >> 4) * 80 + i * 16 + (lr2 & 15);
                const int pc = (c & ~7) | ((c ^ (lr2 >> 1)) & 7);
                if (lrow < BAND && row0 + lrow < M)
                    *reinterpret_cast<f16x8*>(C + (size_t)(row0 + lrow) * N + chunk * NC + c * 8) = *reinterpret_cast<const f16x8*>(park + (size_t)lr2 * (NC * 2) + pc * 16);
            }
            raw_barrier();
        }
    }
}

int main() {
    const int M = 35840;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&band2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&band3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    for (int N : {320, 960, 2560}) {
        const int NSET = N == 320 ? 8 : 3;
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
            for (int blk : {0, 8, 9, 1117, 2239}) {             // incl. the tiles around the first band boundary (row 140 = tile 8.75)
                hipMemcpy(hc.data(), Cs[0] + (size_t)blk * 16 * N, (size_t)16 * N * 2, hipMemcpyDeviceToHost);
                for (int rr = 0; rr < 16; ++rr)
                    for (int n = 0; n < N; n += 7) {
                        double ref = 0;
                        for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)(blk * 16 + rr) * K + k] * (double)(float)hw[(size_t)n * K + k];
                        const double got = (double)(float)hc[(size_t)rr * N + n];
                        worst = fmax(worst, std::isfinite(got) ? fabs(ref - got) : 1e30);
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
            printf("N=%4d  svdx_gemm variant %2d                    %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, variant, us, 2.0 * M * N * K / us / 1e6,
                   bytes / us / 1e6, check());
            for (int s = 0; s < NSET; ++s) hipMemsetAsync(Cs[s], 0xff, (size_t)M * N * 2, st);
        }
        for (int mode : {0, 1}) {
            const double us = time_graph([&](int s) { hipLaunchKernelGGL(band_kernel, dim3(M / BAND), dim3(512), LDS_BYTES, st, As[s], W, Cs[s], M, N, mode); });
            printf("N=%4d  A-stationary band, %-22s %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, mode ? "stores through LDS" : "8-byte stores", us,
                   2.0 * M * N * K / us / 1e6, bytes / us / 1e6, check());
            for (int s = 0; s < NSET; ++s) hipMemsetAsync(Cs[s], 0xff, (size_t)M * N * 2, st);
        }
        {
            const double us = time_graph([&](int s) { hipLaunchKernelGGL(band2_kernel, dim3(M / BAND), dim3(512 + 64 * NPROD), LDS2_BYTES, st, As[s], W, Cs[s], M, N); });
            printf("N=%4d  A-stationary band, producer waves + park buffer %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, us, 2.0 * M * N * K / us / 1e6,
                   bytes / us / 1e6, check());
        }
        {
            for (int s = 0; s < NSET; ++s) hipMemsetAsync(Cs[s], 0xff, (size_t)M * N * 2, st);
            const double us = time_graph([&](int s) { hipLaunchKernelGGL(band3_kernel, dim3(M / BAND), dim3(512 + 64 * NPROD), LDS2_BYTES, st, As[s], W, Cs[s], M, N); });
            printf("N=%4d  A-stationary band, 4 stages of K = 16, producers 2 ahead %7.2f us   %6.1f TFLOP/s   %5.2f TB/s (A + C once)   max |err| %.2e\n", N, us, 2.0 * M * N * K / us / 1e6,
                   bytes / us / 1e6, check());
        }
        hipStreamDestroy(st);
        for (int s = 0; s < NSET; ++s) { hipFree(As[s]); hipFree(Cs[s]); }
        hipFree(W);
    }
    return 0;
}
