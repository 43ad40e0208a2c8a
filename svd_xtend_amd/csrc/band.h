// band.h -- the pieces shared by the kernels that keep a BAND of activation rows resident in LDS and stream weights past it
// (tsa.hip: the fused temporal self-attention; round 2's LayerNorm + GEGLU projection kernel was removed in round 3): the swizzled LDS image [C/64][144 rows][128 B],
// its load and in-place LayerNorm, and the streaming GEMM whose A operand the image is.
#pragma once
#include "common.h"

namespace {

constexpr float TSA_LOG2E = 1.4426950408889634f;
constexpr int TSA_MB = 9;                      // 16-row blocks per band (up to 144 rows)
constexpr int TSA_RP = TSA_MB * 16;
constexpr int TSA_WAVES = 8;
constexpr int TSA_MBW = 5;                     // row blocks per wave: waves 0-3 own blocks 0-4, waves 4-7 blocks 5-8
constexpr int TSA_NG = 2;                      // 32-column groups per wave per pass: groups cw and cw + 4 of the pass's eight
constexpr int TSA_PW = 4 * TSA_NG * 32;        // 256 columns per pass
constexpr int TSA_BST = TSA_PW * 128;          // bytes of one B stage (K extent 64): 32 KiB
constexpr int TSA_NPC = TSA_BST / 1024 / TSA_WAVES;   // DMA pieces per wave per stage (4)
constexpr int TSA_NSTG = 2;
constexpr int TSA_MAXC = 320;

#ifndef TSA_STAMP
#define TSA_STAMP(i) do { } while (0)
#endif

typedef unsigned int tsa_u4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ tsa_u4 tsa_pack8(const float (&f)[8]) {
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = from_f<T>(f[e]);
    return __builtin_bit_cast(tsa_u4, o);
}

constexpr int TSA_OOB = (int)0x80000000;       // a buffer offset no descriptor of this kernel covers: loads give 0, stores are dropped

// 16-byte buffer store followed by two wait states.  A gfx950 vector store of more than 8 bytes reads its data VGPRs a few clocks
// AFTER it issues: a VALU / MFMA / LDS-return write to one of them in the next two issue slots lands in the stored data
// (tools/probes/storewar_probe.hip: 23 % of the stores with no wait state, 0.4 % with one, none with two).  LLVM's hazard recognizer
// inserts ONE wait state, and none at all when the store takes its scalar offset from an SGPR (GCNHazardRecognizer treats that
// form as hazard-free) -- which is the form the image side copy uses: its first version put a v_xor of the data register right
// behind the store and the saved `o` came out with a lane offset in place of a value in ~1 element per million.  So every wide store of
// this kernel is issued through this wrapper (the nop sits among MFMAs: it costs nothing), and tests/test_store_hazard.py scans the
// ISA of every kernel in the library for the pattern.
__device__ __forceinline__ void tsa_store16(tsa_u4 v, __amdgpu_buffer_rsrc_t rs, int voffset, int soffset) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voffset), "s"(rs), "s"(soffset) : "memory");
}

// out[rows of the band, N] = IMG[rows, K] * Bmat[N, K]^T.  A pass covers 256 columns = eight groups of 32; wave (h, cw) = (wave / 4,
// wave % 4) owns row blocks h*5 .. h*5+4 (block 9 does not exist: skipped) x groups cw and cw + 4 (interleaved, so that a short last
// pass still spreads over the waves).  A group is two MFMA column blocks whose weight rows are PERMUTED when they are DMA-ed into the
// stage -- LDS row g*32 + b*16 + c holds weight row g*32 + (c/4)*8 + b*4 + c%4 -- so that the lane that owns tile columns 4 fg .. 4 fg+3
// of both blocks owns the 8 CONSECUTIVE output columns g*32 + 8 fg .. +7: one 16-byte store instead of two 8-byte ones.  (A CU issues
// one vector store per ~17 clocks whatever its width -- tools/probes/store_probe.hip: 26 B/clk with 8-byte lanes, 58 with 16.)
//
// A K-step is 64 deep: two 32-deep halves of 36 MFMAs per SIMD.  Everything that is not an MFMA is placed INSIDE the MFMA stream,
// where its issue slot is cheap (an LDS-DMA piece costs ~60 clocks among MFMAs, 100-185 in a burst behind a barrier):
//   - the four DMA pieces of the next weight stage go out one per row block of the first half;
//   - the first half's A fragments were fetched during the previous step (the image never changes), so only the four B fragments are
//     read behind the barrier; the second half's fragments travel under the first half's MFMAs;
//   - the side job -- the A image itself (n in phase 2, o in phase 4: both saved for the backward) goes to `side`, one band row per
//     unit (40 of the 64 lanes at C = 320; scalar row arithmetic, three vector instructions) -- sits between the halves.
// Variants that were measured and dropped (tools/probes/tsa_probe): output stores parked in registers and drained under the next
// pass (no gain once they were 16 bytes wide), 32-deep stages in a four-deep ring with counted vmcnt waits (twice the barriers, no
// shorter waits: the step is issue-bound, not latency-bound).
// `pre(pass)` runs at the head of a pass's last K-step (the out-projection starts its residual loads there), `epi(pass, acc)` after it,
// `mid(pass, ks)` between the MFMA halves of every K-step (ffn.hip computes and stores the previous pass's GEGLU there).
// Ends with every wave past a barrier and all its stores complete.
template <typename T, int KS, int MIDV = 0, bool ALL_LIVE = false, typename Pre, typename Mid, typename Epi, typename Grow>
__device__ __forceinline__ void tsa_band_gemm(const char* IMG, char* BST, const void* Bmat, int b_bytes, int N, int tid, Pre&& pre, Mid&& mid, Epi&& epi,
                                              void* side, int side_bytes, int R, Grow&& grow, unsigned long long* wait_cycles = nullptr,
                                              int half_rows = 128, int pass_rows = TSA_PW) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename TT<T>::v8 v8;
    constexpr int Kd = KS * 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int h = wave >> 2, cw = wave & 3;
    // N: output columns in units of the pass (plain: the matrix has N rows; GEGLU pairs: N = 2 x the columns of one half)
    const int npass = (N + TSA_PW - 1) / TSA_PW, total = npass * KS;
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Bmat), 0, b_bytes, 0x00020000);
    int vob[TSA_NPC];
#pragma unroll
    for (int i = 0; i < TSA_NPC; ++i) {
        const int id = (i * TSA_WAVES + wave) * 64 + lane;       // 16-byte unit of the stage: LDS row = id / 8, physical chunk = id % 8
        const int r = id >> 3, pc = id & 7, lc = pc ^ (r & 7);
        // the weight row this LDS row holds: groups 0-3 of the pass are weight rows 0..127 of the pass, groups 4-7 start `half_rows`
        // further on (128 for a plain projection: the next 128 columns; F for a GEGLU pair: the gate rows of the same columns)
        const int n = ((r >> 5) & 3) * 32 + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3) + (r >> 7) * half_rows;
        vob[i] = (n * Kd + lc * 8) * 2;                          // rows beyond the matrix lie beyond b_bytes: the descriptor returns zeros
    }
    int i_ks = 0, i_stage = 0, i_shift = 0;                      // the DMA's position: K-step, stage, byte shift of its pass
    auto issue_piece = [&](int i) __attribute__((always_inline)) {       // literal i
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(BST + i_stage * TSA_BST + (i * TSA_WAVES + wave) * 1024), 16,
                                                 vob[i] + i_shift, i_ks * 128, 0, 0);
    };
    auto issue_done = [&]() __attribute__((always_inline)) {
        i_stage ^= 1;
        if (++i_ks == KS) { i_ks = 0; i_shift += pass_rows * Kd * 2; }
    };
#ifdef TSA_NO_SIDE
    side = nullptr;
#endif
    __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(side ? side : const_cast<void*>(Bmat), 0, side ? side_bytes : 0, 0x00020000);
    // side job: unit u = band row u (wave-uniform), lane l < C/8 moves 16-byte chunk l
    const int side_per = side ? (TSA_RP + TSA_WAVES * total - 1) / (TSA_WAVES * total) : 0;     // rows per wave per K-step
    const int s_c = min(lane, KS * 8 - 1);
    const int s_lds = (s_c >> 3) * (TSA_RP * 128), s_x = (s_c & 7) * 16;
    const int s_off = lane < KS * 8 ? lane * 16 : TSA_OOB;
    auto side_row = [&](int u) __attribute__((always_inline)) {  // u wave-uniform
        const int row = min(u, TSA_RP - 1);
        const tsa_u4 v = *reinterpret_cast<const tsa_u4*>(IMG + s_lds + row * 128 + (s_x ^ ((row & 7) * 16)));
        tsa_store16(v, rsS, u < R ? s_off : TSA_OOB, u < R ? grow(u) * (Kd * 2) : 0);
    };
    const char* Ah = IMG + h * (TSA_MBW * 16 * 128) + fr * 128;
    const int ch0 = (fg ^ (fr & 7)) * 16, ch1 = ((4 + fg) ^ (fr & 7)) * 16;       // the lane's 16-byte chunk of the two 32-deep halves
    const int offB = (cw * 32 + fr) * 128;
    auto rows_ok = [&](int i) __attribute__((always_inline)) { return i < TSA_MB - TSA_MBW || h == 0; };
    v8 a0[TSA_MBW], a1[TSA_MBW], b0[TSA_NG][2], b1[TSA_NG][2];
#pragma unroll
    for (int i = 0; i < TSA_MBW; ++i) a0[i] = *reinterpret_cast<const v8*>(Ah + i * (16 * 128) + ch0);
#pragma unroll
    for (int i = 0; i < TSA_NPC; ++i) issue_piece(i);
    issue_done();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int q = 0, cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        f32x4 acc[TSA_NG][2][TSA_MBW];
#pragma unroll
        for (int g = 0; g < TSA_NG; ++g)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < TSA_MBW; ++i) acc[g][b][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bool live[TSA_NG];                                       // a group beyond N is skipped (wave-uniform)
#pragma unroll
        for (int g = 0; g < TSA_NG; ++g) live[g] = ALL_LIVE || pass * TSA_PW + (g * 4 + cw) * 32 < N;     // ALL_LIVE: N is a multiple of 256
#pragma unroll
        for (int ks = 0; ks < KS; ++ks, ++q) {
            const bool more = q + 1 < total;
            if (ks == KS - 1) pre(pass);
            const char* As = Ah + ks * (TSA_RP * 128);
            const char* An = Ah + (ks + 1 < KS ? ks + 1 : 0) * (TSA_RP * 128);
            const char* Bs = BST + cur * TSA_BST + offB;
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g)
#pragma unroll
                for (int b = 0; b < 2; ++b) b0[g][b] = *reinterpret_cast<const v8*>(Bs + (g * 128 + b * 16) * 128 + ch0);
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g)
#pragma unroll
                for (int b = 0; b < 2; ++b) b1[g][b] = *reinterpret_cast<const v8*>(Bs + (g * 128 + b * 16) * 128 + ch1);
#pragma unroll
            for (int i = 0; i < TSA_MBW; ++i) a1[i] = *reinterpret_cast<const v8*>(As + i * (16 * 128) + ch1);
#pragma unroll
            for (int i = 0; i < TSA_MBW; ++i) {
                if (live[0] && rows_ok(i)) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[0][b][i] = TT<T>::mfma(b0[0][b], a0[i], acc[0][b][i]);
                    if (live[1]) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[1][b][i] = TT<T>::mfma(b0[1][b], a0[i], acc[1][b][i]);
                    }
                }
                if (i < TSA_NPC && more) issue_piece(i);         // the next stage: its slot was last read in step q - 1
            }
            if (more) issue_done();
#pragma unroll
            for (int i = 0; i < TSA_MBW; ++i) a0[i] = *reinterpret_cast<const v8*>(An + i * (16 * 128) + ch0);
            for (int k = 0; k < side_per; ++k) side_row((q * side_per + k) * TSA_WAVES + wave);
            // the second half's MFMAs with the caller's `mid` work (ffn.hip: ~250 VALU instructions + 3 stores) in the SAME scheduling
            // region, interleaved one MFMA : a few VALU by sched_group_barrier -- a wave's VALU issue slots between its own MFMAs are
            // free, a VALU block in front of the MFMAs is not
            if (live[0]) {
                mid(pass, ks);
#pragma unroll
                for (int i = 0; i < TSA_MB - TSA_MBW; ++i) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[0][b][i] = TT<T>::mfma(b1[0][b], a1[i], acc[0][b][i]);
                    if (live[1]) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[1][b][i] = TT<T>::mfma(b1[1][b], a1[i], acc[1][b][i]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4 * (TSA_MB - TSA_MBW); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, MIDV, 0);   // MIDV VALU
                }
                if (h == 0) {
#pragma unroll
                    for (int i = TSA_MB - TSA_MBW; i < TSA_MBW; ++i) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[0][b][i] = TT<T>::mfma(b1[0][b], a1[i], acc[0][b][i]);
                        if (live[1]) {
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc[1][b][i] = TT<T>::mfma(b1[1][b], a1[i], acc[1][b][i]);
                        }
                    }
                }
            } else {
                mid(pass, ks);
            }
#ifdef TSA_STAMPS
            const unsigned long long w0 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#ifdef TSA_STAMPS
            if (wait_cycles) *wait_cycles += __builtin_readcyclecounter() - w0;
#endif
            cur ^= 1;
        }
#ifdef TSA_STAMPS
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        if (live[0]) epi(pass, acc);
#ifdef TSA_STAMPS
        if (wait_cycles) wait_cycles[1] += __builtin_readcyclecounter() - c0;
#endif
    }
#ifdef TSA_STAMPS
    const unsigned long long d0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef TSA_STAMPS
    if (wait_cycles) wait_cycles[2] += __builtin_readcyclecounter() - d0;
#endif
#endif
}

// band rows -> LDS image (one DMA piece = 8 rows x 128 B of one 64-channel block); rows beyond R read zeros.  Ends behind a barrier.
template <int KB, typename Grow>
__device__ __forceinline__ void band_load(char* IMG, const void* x, int x_bytes, int R, Grow&& grow, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int C = KB * 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, x_bytes, 0x00020000);
    const int npieces = KB * (TSA_RP / 8);
    for (int pq = wave; pq < npieces; pq += TSA_WAVES) {
        const int kblk = pq / (TSA_RP / 8), rg = pq - kblk * (TSA_RP / 8);
        const int row = rg * 8 + (lane >> 3), lc = (lane & 7) ^ (lane >> 3);
        const int voff = row < R ? (grow(row) * C + kblk * 64 + lc * 8) * 2 : (int)0x80000000;     // padding rows read zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(IMG + pq * 1024), 16, voff, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

// LayerNorm of the image in place; 16 lanes per row, 4 rows per wave per step, 3 independent steps in flight; row quad
// (step * 8 + wave) of the band's 36.  The arithmetic is on float pairs (v_pk_add/mul/fma_f32): with two waves per SIMD this phase is
// VALU-bound.  (mean, rstd) of the real rows go to `stats`; padding rows become zero.  Ends behind a barrier.
template <typename T, int KB, typename Grow>
__device__ __forceinline__ void band_layernorm(char* IMG, const float* gamma_p, const float* beta_p, float eps, float* stats, int R,
                                               Grow&& grow, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int C = KB * 64, C8 = C / 8;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    struct { const float* gamma; const float* beta; float eps; float* stats; } p = {gamma_p, beta_p, eps, stats};
    const int l16 = lane & 15;
    constexpr int NCH = (C8 + 15) / 16;                 // 16-byte chunks per lane (3 at C = 320)
    constexpr int G = 3;                                // row groups in flight
    f32x2v gm[NCH][4], bt[NCH][4];
    bool cv[NCH];
    int cl[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = l16 + 16 * j;
        cv[j] = 16 * (j + 1) <= C8 || c < C8;           // a literal `true` for all but the last chunk
        cl[j] = min(c, C8 - 1);                         // invalid chunks read a valid address and are masked to zero
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gm[j][e] = f32x2v{p.gamma[cl[j] * 8 + 2 * e], p.gamma[cl[j] * 8 + 2 * e + 1]};
            bt[j][e] = f32x2v{p.beta[cl[j] * 8 + 2 * e], p.beta[cl[j] * 8 + 2 * e + 1]};
        }
    }
    const float invC = 1.f / (float)C;
    constexpr int NQ = (TSA_RP / 4 + TSA_WAVES - 1) / TSA_WAVES;      // steps per wave (5; the last one only for waves 0-3)
    for (int it0 = 0; it0 < NQ; it0 += G) {
        f32x2v v[G][NCH][4];
        float mean[G], rstd[G];
        int row[G];
        bool inb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int quad = (it0 + g) * TSA_WAVES + wave;
            inb[g] = quad < TSA_RP / 4;                  // wave-uniform
            row[g] = min(quad, TSA_RP / 4 - 1) * 4 + (lane >> 4);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const Vec8<T> t = *reinterpret_cast<const Vec8<T>*>(IMG + (cl[j] >> 3) * (TSA_RP * 128) + row[g] * 128 + (((cl[j] & 7) ^ (row[g] & 7)) * 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[g][j][e] = f32x2v{to_f<T>(t.v[2 * e]), to_f<T>(t.v[2 * e + 1])};
                    if (!cv[j]) v[g][j][e] = f32x2v{0.f, 0.f};
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x2v s2 = f32x2v{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s2 += v[g][j][e];
            mean[g] = row16_sum(s2[0] + s2[1]) * invC;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x2v ss = f32x2v{0.f, 0.f};
            const f32x2v m2 = f32x2v{mean[g], mean[g]};
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[g][j][e] -= m2;                   // kept: the normalised value is d * (rstd * gamma) + beta
                    if (!cv[j]) v[g][j][e] = f32x2v{0.f, 0.f};
                    ss = __builtin_elementwise_fma(v[g][j][e], v[g][j][e], ss);
                }
            rstd[g] = rsqrtf(row16_sum(ss[0] + ss[1]) * invC + p.eps);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!inb[g]) continue;
            const bool real = row[g] < R;
            if (real && l16 == 0) *reinterpret_cast<float2*>(p.stats + (size_t)grow(row[g]) * 2) = float2{mean[g], rstd[g]};
            const f32x2v r2 = real ? f32x2v{rstd[g], rstd[g]} : f32x2v{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                Vec8<T> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2v y = __builtin_elementwise_fma(v[g][j][e], r2 * gm[j][e], real ? bt[j][e] : f32x2v{0.f, 0.f});
                    o.v[2 * e] = from_f<T>(y[0]);
                    o.v[2 * e + 1] = from_f<T>(y[1]);
                }
                if (cv[j]) *reinterpret_cast<Vec8<T>*>(IMG + (cl[j] >> 3) * (TSA_RP * 128) + row[g] * 128 + (((cl[j] & 7) ^ (row[g] & 7)) * 16)) = o;
            }
        }
    }
    __syncthreads();
#endif
}

}  // namespace
