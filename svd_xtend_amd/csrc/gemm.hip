// gemm.hip -- NT GEMM on MFMA for gfx950 with an implicit-convolution A operand.
//
//   acc[m,n] = sum_k Aeff[m,k] * B[n,k]      (f16/bf16 in, f32 accumulate, v_mfma_f32_16x16x32_*)
//
// One kernel serves every matmul-shaped op of the SVD UNet step: nn.Linear fwd /
// data-grad / weight-grad, conv2d 3x3 (stride 1/2, nearest-x2 source) and its data-grads, Conv3d (3,1,1).
// Convolutions never materialise im2col: each K-tile of 64 channels belongs to one filter tap, and the A
// rows of that tile are fetched from the tap's shifted pixel (or from a zero page outside the image).
//
// Kernels in this file (variant argument of svdx_gemm):
//   variant 0 / 1  gemm_kernel: 128x128x64, global_load_lds_dwordx4 (LDS-DMA) with 64-bit pointers, swizzle on the per-lane SOURCE
//              address -- the fallback when a buffer exceeds the 2 GiB range of variant 4's 32-bit offsets (round 1's register-staged
//              variant 0 was removed in round 4: it now runs this kernel too)
//   variant 4  gemm_v4_kernel         : production (see its banner): 128x160 tiles, lean buffer_load...lds loop, coalesced epilogue
//   gemm_tn_kernel                    : weight gradients straight from row-major dY / X via ds_read_b64_tr_b16
// Common: 256 threads = 4 waves (2x2); LDS rows of 128 B with the 16-byte chunk index XOR-swizzled by (row & 7) so the
// ds_read_b128 fragment reads are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0 measured); XCD-aware block -> tile mapping
// (consecutive tiles of one A row-panel stay on one XCD's L2); split-K into float slabs + svdx_gemm_finalize.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int CS_LD = 132;                          // padded f32 row of the epilogue staging tile

struct GemmParams {
    const void* A; const void* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; const float* rowvec; int rv_ld, rv_rpg, rv_mod;
    const void* res; int ldres;
    svdx_gather g; const void* zero_page;
    int out_mode; float alpha; int split_k; int tiles_m, tiles_n; int vec_ok; long slab_stride; int a_bytes, b_bytes;
    int xcd_n, sub_m, sub_n;   // variant 4: the 8 XCDs form an (8/xcd_n) x xcd_n grid, each owning sub_m x sub_n tiles (0: balanced row-major split)
    int z_xcd;                 // variant 4, split_k in {2, 4, 8}: K slice z owns 8 / split_k XCDs, arranged (8/split_k/xcd_n) x xcd_n over the tiles (1-D grid)
    int epi; const void* aux_in; void* aux_out; int aux_dim;   // fused GEGLU epilogues (variant 4)
    float* a_colsum;                                           // TN form: += column sums of A (the bias gradient), or null
    float* found_inf;                                          // TN form, float store / += modes: *found_inf = 1 when a value written is not finite (or null)
    // variant 4, second operand pair: acc += A2 [M, K2] B2^T [N, K2] after the main reduction (the LoRA term of a projection)
    const void* A2; const void* B2; int K2, lda2, ldb2, a2_bytes, b2_bytes;
    int a2_seg;   // > 0: output columns [j * a2_seg, (j + 1) * a2_seg) read A2 columns [j * K2, (j + 1) * K2) (fused q/k/v adapters)
    // variant 4, activation output: GroupNorm statistics of the tensor this launch writes (sum, sum of squares per (sample, group) of the
    // ROUNDED values, in the fixed-point replica slots svdx_gn_apply reads) -- the norm that consumes C needs no pass of its own over it
    unsigned long long* gn_stats; int gn_rows, gn_cg; float gn_m0, gn_m1;
    // variant 4 / 5: rows a row tile OWNS (= its stride over M).  Equal to the tile height except for the 144-row tile of variant 36,
    // which steps 140 rows (35840 = 256 x 140, 8960 = 64 x 140: the grid fills every workgroup slot of the chip exactly); rows of a
    // tile beyond its step are computed and not stored (they are the next tile's).
    int m_step;
};
constexpr int GN_MAX_S = 8, GN_MAX_G = 36;               // samples / groups one output tile may touch (else the host runs svdx_gn_stats)
constexpr int GN_LDS = GN_MAX_S * GN_MAX_G * 2 * 8;

struct RowInfo { int a, b, base; };   // per gathered A row (meaning depends on gather mode)

template <typename T>
__device__ __forceinline__ const T* a_row_ptr(const GemmParams& p, const RowInfo& ri, int m_clamped, int k0,
                                              int tap, int ci0, bool& valid) {
    const T* A = reinterpret_cast<const T*>(p.A);
    const svdx_gather& g = p.g;
    valid = true;
    if (g.mode == SVDX_GATHER_PLAIN) return A + (size_t)m_clamped * p.lda + k0;
    int src;
    if (g.mode == SVDX_GATHER_CONV3X3 || g.mode == SVDX_GATHER_CONV3X3_PAD0) {
        int dy = tap / 3, dx = tap - dy * 3;
        int ys = ri.a + dy, xs = ri.b + dx;
        valid = (ys >= 0) & (ys < g.hi) & (xs >= 0) & (xs < g.wi);
        int wsrc = g.wi >> g.ups;
        src = ri.base + (ys >> g.ups) * wsrc + (xs >> g.ups);
    } else if (g.mode == SVDX_GATHER_CONV3X3_DGRAD2) {
        int dy = tap / 3, dx = tap - dy * 3;
        int y2 = ri.a - dy, x2 = ri.b - dx;
        valid = (y2 >= 0) & ((y2 & 1) == 0) & ((y2 >> 1) < g.hi) & (x2 >= 0) & ((x2 & 1) == 0) & ((x2 >> 1) < g.wi);
        src = ri.base + (y2 >> 1) * g.wi + (x2 >> 1);
    } else {   // TEMPORAL3
        int ts = ri.a + tap - 1;
        valid = (ts >= 0) & (ts < g.t);
        src = ri.base + ts * g.hw;
    }
    return A + (size_t)(valid ? src : 0) * g.lda + ci0;
}

__device__ __forceinline__ RowInfo decode_row(const svdx_gather& g, int m) {
    RowInfo ri{0, 0, 0};
    if (g.mode == SVDX_GATHER_CONV3X3 || g.mode == SVDX_GATHER_CONV3X3_PAD0) {
        int x = m % g.wo, t = m / g.wo;
        int y = t % g.ho, n = t / g.ho;
        const int pad = g.mode == SVDX_GATHER_CONV3X3 ? 1 : 0;      // PAD0: the zero row / column sit below / right of the image only
        ri.a = y * g.stride - pad;
        ri.b = x * g.stride - pad;
        ri.base = n * (g.hi >> g.ups) * (g.wi >> g.ups);
    } else if (g.mode == SVDX_GATHER_CONV3X3_DGRAD2) {
        int x = m % g.wo, t = m / g.wo;
        int y = t % g.ho, n = t / g.ho;
        ri.a = y + 1;
        ri.b = x + 1;
        ri.base = n * g.hi * g.wi;
    } else if (g.mode == SVDX_GATHER_TEMPORAL3) {
        int pp = m % g.hw, t = m / g.hw;
        int tt = t % g.t, b = t / g.t;
        ri.a = tt;
        ri.base = b * g.t * g.hw + pp;
    }
    return ri;
}

// GradScaler's inf check folded into the kernels that WRITE the gradients (round 5): a weight gradient that is stored once per step is
// tested where it is stored, and svdx_check_finite_spans covers the few slots that are accumulated -- the 1.59 GB pass of
// svdx_check_finite over the flat buffer leaves the single-rank step.  Any number of threads may raise the flag (same value, no ordering).
__device__ __forceinline__ bool not_finite(float v) { return (__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ void raise_found_inf(float* found, bool bad) {
    if (found && __any(bad) && (threadIdx.x & 63) == 0) *found = 1.f;
}

template <typename T>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], char* smem, int m0, int n0, int z, int tid,
                                              int lane, int wm, int wn) {
    // ---- epilogue: stage 64 rows at a time through LDS as f32, then vectorised fused store ----
    float* Cs = reinterpret_cast<float*>(smem);
    bool bad_any = false;
    const bool lead = (z == 0) && p.out_mode != SVDX_OUT_F32_SLAB;
    T* Ct = reinterpret_cast<T*>(p.C);
    float* Cf = reinterpret_cast<float*>(p.C) + (p.out_mode == SVDX_OUT_F32_SLAB ? (size_t)z * p.slab_stride : 0);
    const T* R = reinterpret_cast<const T*>(p.res);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(i * 16 + (lane >> 4) * 4 + r) * CS_LD + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int id = i * 256 + tid;
            const int row = id >> 4, c8 = id & 15;
            const int m = m0 + pass * 64 + row, nc = n0 + c8 * 8;
            if (m >= p.M || nc >= p.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + row * CS_LD + c8 * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + row * CS_LD + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j] * p.alpha; v[4 + j] = hi[j] * p.alpha; }
            const bool full = p.vec_ok && (nc + 8 <= p.N);
            const int nvalid = min(8, p.N - nc);
            if (lead) {
                if (p.bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += p.bias[nc + j];
                }
                if (p.rowvec) {
                    const int gi = p.rv_mod ? (m % p.rv_mod) : (m / p.rv_rpg);
                    const float* rv = p.rowvec + (size_t)gi * p.rv_ld + nc;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += rv[j];
                }
                if (R) {
                    const T* rp = R + (size_t)m * p.ldres + nc;
                    if (full) {
                        float rr[8];
                        load8<T>(rp, rr);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += rr[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += to_f<T>(rp[j]);
                    }
                }
            }
            const size_t co = (size_t)m * p.ldc + nc;
            if (p.out_mode == SVDX_OUT_ACT) {
                if (full) {
                    store8<T>(Ct + co, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) Ct[co + j] = from_f<T>(v[j]);
                }
            } else if (p.out_mode == SVDX_OUT_F32 || p.out_mode == SVDX_OUT_F32_SLAB) {
                if (full) {
                    *reinterpret_cast<f32x4*>(Cf + co) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(Cf + co + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) Cf[co + j] = v[j];
                }
                if (p.found_inf && p.out_mode == SVDX_OUT_F32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bad_any |= j < nvalid && not_finite(v[j]);
                }
            } else if (p.out_mode == SVDX_OUT_F32_ADD) {      // this block owns the element: plain read-modify-write
                if (full) {
                    f32x4 c0 = *reinterpret_cast<const f32x4*>(Cf + co), c1 = *reinterpret_cast<const f32x4*>(Cf + co + 4);
                    c0 += f32x4{v[0], v[1], v[2], v[3]};
                    c1 += f32x4{v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(Cf + co) = c0;
                    *reinterpret_cast<f32x4*>(Cf + co + 4) = c1;
                    if (p.found_inf) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bad_any |= not_finite(c0[j]) || not_finite(c1[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) { const float t = Cf[co + j] + v[j]; Cf[co + j] = t; bad_any |= p.found_inf && not_finite(t); }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < nvalid) atomicAdd(Cf + co + j, v[j]);
            }
        }
        __syncthreads();
    }
    raise_found_inf(p.found_inf, bad_any);
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping (block b runs on XCD b % 8; give each XCD a contiguous run of tiles) ----
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int pid_m = swz / p.tiles_n, pid_n = swz - pid_m * p.tiles_n;
    const int m0 = pid_m * BM, n0 = pid_n * BN;

    // ---- split-K range ----
    const int kt_total = p.K / BK;
    const int z = blockIdx.y;
    const int kt_per = (kt_total + p.split_k - 1) / p.split_k;
    const int kt_begin = z * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;

    // ---- per-thread staging assignment: 4 A rows + 4 B rows, one 16-byte chunk each ----
    const int ld_row = tid >> 3;                 // 0..31
    const int pc = tid & 7;                      // physical chunk inside the 128-byte LDS row
    const int lc = pc ^ (ld_row & 7);            // logical chunk (k offset lc*8) that lives there
    int a_m[4];
    RowInfo a_ri[4];
    const T* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = min(m0 + i * 32 + ld_row, p.M - 1);
        a_m[i] = m;
        a_ri[i] = decode_row(p.g, m);
        int n = min(n0 + i * 32 + ld_row, p.N - 1);
        b_ptr[i] = reinterpret_cast<const T*>(p.B) + (size_t)n * p.ldb + lc * 8;
    }
    const int cin = p.g.mode == SVDX_GATHER_PLAIN ? p.K : p.g.cin;
    const T* zero = reinterpret_cast<const T*>(p.zero_page);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage_ptrs = [&](int kt, const T* (&pa)[4], const T* (&pb)[4]) __attribute__((always_inline)) {
        const int k0 = kt * BK;
        int tap = 0, ci0 = k0;
        if (p.g.mode != SVDX_GATHER_PLAIN) { tap = k0 / cin; ci0 = k0 - tap * cin; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool valid;
            const T* ptr = a_row_ptr<T>(p, a_ri[i], a_m[i], k0, tap, ci0, valid);
            pa[i] = valid ? ptr + lc * 8 : zero;
            pb[i] = b_ptr[i] + k0;
        }
    };
    auto issue_glds = [&](int kt, int stage) __attribute__((always_inline)) {
        const T* pa[4]; const T* pb[4];
        stage_ptrs(kt, pa, pb);
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * BK * 2;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // LDS destination = wave-uniform base + lane*16 (chunk id = i*256 + tid)
            const int base = (i * 256 + wave_u * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[i],
                                             (__attribute__((address_space(3))) void*)(As + base), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[i],
                                             (__attribute__((address_space(3))) void*)(Bs + base), 16, 0, 0);
        }
    };
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const char* As = smem + stage * STAGE_BYTES;
        const char* Bs = As + BM * BK * 2;
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = ((kk * 4 + fg) ^ (fr & 7)) * 16;
            v8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const v8*>(As + (wm * 64 + i * 16 + fr) * 128 + chunk);
                bf[i] = *reinterpret_cast<const v8*>(Bs + (wn * 64 + i * 16 + fr) * 128 + chunk);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = TT<T>::mfma(af[i], bf[j], acc[i][j]);
        }
    };

    issue_glds(kt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end - 1; ++kt) {
        issue_glds(kt + 1, cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    __syncthreads();

    gemm_epilogue<T>(p, acc, smem, m0, n0, z, tid, lane, wm, wn);
}


// ================================================================================================================
// TN form for weight gradients:  C[n,k] (+)= sum_r A[r,n] * B[r,k]   (A = dY [R, lda], B = X [R, ldb]; float output)
// Both operands are stored with the reduction index r as the ROW, so tiles are staged row-major ([64 r][128 cols],
// 256-byte rows, coalesced 16-byte loads along the columns) and the MFMA fragments -- 8 consecutive r for one column --
// are fetched with ds_read_b64_tr_b16, gfx950's transposing LDS read: within a 16-lane group lane i supplies the
// address of row (i>>2), column piece (i&3)*4, and receives column i of the 4x16 block (probed on hardware, see
// tools/probes/tr_probe.hip).  No transposed copies of dY / X are ever materialised and every output element has
// one owner, so no atomics either.  8-byte units of a row are XOR-swizzled with a 3-bit row id so the 32 lanes of
// a half-wave hit 32 distinct bank pairs.
// ================================================================================================================
// ---- LDS-DMA that the compiler does not see (the TN kernels, round 5) ----------------------------------------------------------------------
// hipcc (ROCm 7.2) orders every ds_read_b64_tr_b16 INTRINSIC behind all pending LDS-DMA with an `s_waitcnt vmcnt(0)` -- for the plain
// ds_read_b128 of gemm_v4_kernel it proves the stage being read distinct from the stage being filled, for the intrinsic it has no alias
// information -- so in every TN kernel of rounds 1-4 the next tile's loads and this tile's MFMAs ran strictly one after the other
// (profiles/r5_tn_isa_waits.txt: the wait sits between the staging pieces and the first fragment read of every K-step; standalone counters:
// waves parked 57-63 % of their cycles, MFMA busy 0.18 in the step).  Issued from an `asm` statement the DMA is invisible to that pass; the
// kernels count their vmcnt themselves (they always did: counted waits + s_barrier).  No VGPR destination, so nothing for the register
// allocator to mis-track (cdna_hip_programming.md 5.7); M0 -- the LDS base of the instruction -- is saved and restored inside the statement.
typedef int desc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ desc4 hidden_desc(const void* p, unsigned bytes) {      // raw buffer descriptor: base, stride 0, num_records, gfx950 flags
    const unsigned long a = (unsigned long)p;
    return desc4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <int SIZE, bool STREAM = false>
__device__ __forceinline__ void hidden_dma(desc4 d, char* lds, unsigned voff) {     // lane l: SIZE bytes from d.base + voff -> lds + l * SIZE (0 beyond num_records)
    static_assert(SIZE == 16 || SIZE == 4, "dwordx4 or dword");
    /*SIM-BEGIN*/
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds);
    unsigned keep;
    if (SIZE == 16 && STREAM)          // nt (streaming policy).  Round 6 tried it on the X operand of the weight-gradient kernels (a saved activation, dead after the
                                       // launch): +0.3 ms / step -- the tiles of a row slice share those lines through the L2 (profiles/r6q_ab_tn_nt_operand.txt); unused
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(la), "v"(voff), "s"(d) : "memory");
    else if (SIZE == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(la), "v"(voff), "s"(d) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(la), "v"(voff), "s"(d) : "memory");
    /*SIM-END sim_hidden_dma<SIZE>(d, lds, voff); */
}

__device__ __forceinline__ int tn_rid(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T>
__device__ __forceinline__ typename TT<T>::v8 tn_frag(const char* tile, int colblk16, int ks, int fr, int fg) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    const int r0 = ks * 32 + fg * 8 + (fr >> 2);
    const int unit = colblk16 * 4 + (fr & 3);
    const int a0 = r0 * 256 + ((unit ^ (tn_rid(r0) << 2)) * 8);
    const int a1 = (r0 + 4) * 256 + ((unit ^ (tn_rid(r0 + 4) << 2)) * 8);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(tile + a0));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(tile + a1));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename TT<T>::v8, r);
}

// ---- which (row slice z, output tile) a TN workgroup computes.  Workgroup ids go round-robin to the 8 XCDs (id & 7), each with a private
// 4 MiB L2, and a TN workgroup streams ROWS of dY / X: everything that shares rows wants to sit on one XCD at the same time.
//   * split_k a multiple of 8 (the 64x40-level gradients: 16-32 slices of 35840 rows): XCD x runs slices x, x + 8, ... -- ALL output
//     tiles of a slice side by side, walking the slice's rows in step, so each row of dY and X comes over the fabric once (round 3 put
//     the slices of one TILE on an XCD: they share nothing, and every operand row was fetched by up to 8 XCDs -- 344 MB where 115 MB are
//     algorithmic on the 320 x 1280 gradient, profiles/r4_pmc_traffic_by_shape.txt);
//   * split_k in {1, 2, 4}: each slice owns 8 / split_k XCDs, arranged xm x xn over its tile grid so that the cheaper operand is the one
//     re-fetched (host: tn_arrange -- the rule of launch_gemm_v4).
// Grid: 8 * zsets * sub_m * sub_n workgroups in x; out-of-range ids exit.
struct TnWho { int pid_m, pid_n, z; bool live; };
__device__ __forceinline__ TnWho tn_who(const GemmParams& p) {
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;
    const int per = p.sub_m * p.sub_n;
    const int zs = l / per, r = l - zs * per;
    TnWho w;
    int xi_m = 0, xi_n = 0;
    if (p.split_k >= 8 || 8 % p.split_k) {           // (a slice count that neither divides 8 nor is a multiple of it leaves XCDs idle: legal, slow)
        w.z = zs * 8 + xcd;
    } else {
        const int xps = 8 / p.split_k, xi = xcd % xps;
        w.z = xcd / xps;
        xi_m = xi / p.xcd_n;
        xi_n = xi - xi_m * p.xcd_n;
    }
    const int lm = r / p.sub_n;
    w.pid_m = xi_m * p.sub_m + lm;
    w.pid_n = xi_n * p.sub_n + (r - lm * p.sub_n);
    w.live = w.z < p.split_k && w.pid_m < p.tiles_m && w.pid_n < p.tiles_n;
    return w;
}

// BUF (round 5): the staging as buffer_load ... lds through two raw descriptors (operands below 2 GiB), issued by hidden_dma above: per-thread
// byte offsets computed once, the row bound from num_records instead of a zero page (a column beyond the operand gets an offset beyond it), and
// -- the point -- no compiler-inserted vmcnt(0) between the staging pieces and the fragment reads: loads and MFMAs overlap at last.
template <typename T, int NSTG, bool BUF = false>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const TnWho who = tn_who(p);
    if (!who.live) return;
    const int pid_m = who.pid_m, pid_n = who.pid_n;
    const int m0 = pid_m * BM, n0 = pid_n * BN;          // m0: first output row (a column of A), n0: first output col (a column of B)
    const int R = p.K;                                   // reduction length (rows of A and B)
    const int kt_total = (R + BK - 1) / BK;
    const int z = who.z;
    const int kt_per = (kt_total + p.split_k - 1) / p.split_k;
    const int kt_begin = z * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) {               // a slice without rows (the host never asks for one): its column-sum row is all zeros
        if (p.a_colsum && p.out_mode == SVDX_OUT_F32_SLAB && pid_n == 0 && tid < BM && m0 + tid < p.M) p.a_colsum[(size_t)z * p.M + m0 + tid] = 0.f;
        return;
    }

    // staging: tile = 64 rows x 16 chunks (16 B); thread handles 4 rows, one fixed physical chunk
    const int pc = tid & 15, ld_row = tid >> 4;          // ld_row 0..15; rows ld_row + 16*i
    const T* zero = reinterpret_cast<const T*>(p.zero_page);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    // BUF: byte offsets of this thread's four pieces inside K-tile 0 (the K-tile adds kt * BK rows); a column beyond the operand gets an
    // offset beyond num_records, so the piece reads zeros like a row beyond R does (the descriptor checks voffset, not soffset)
    const desc4 rsA = hidden_desc(p.A, BUF ? (unsigned)p.a_bytes : 0u), rsB = hidden_desc(p.B, BUF ? (unsigned)p.b_bytes : 0u);
    constexpr unsigned OOB = 0x7ffffff0u;
    unsigned voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + ld_row;
        const int lc = pc ^ (tn_rid(row) << 1);
        const int ca = m0 + lc * 8, cb = n0 + lc * 8;
        voa[i] = ca < p.M ? (unsigned)(row * p.lda + ca) * 2u : OOB;
        vob[i] = cb < p.N ? (unsigned)(row * p.ldb + cb) * 2u : OOB;
    }
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * BK * 2;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        if (BUF) {
            const unsigned ka = (unsigned)kt * (unsigned)(BK * 2) * (unsigned)p.lda, kb = (unsigned)kt * (unsigned)(BK * 2) * (unsigned)p.ldb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int base = (i * 256 + wave_u * 64) * 16;
                hidden_dma<16>(rsA, As + base, voa[i] + ka);
                hidden_dma<16>(rsB, Bs + base, vob[i] + kb);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + ld_row;
            const int lc = pc ^ (tn_rid(row) << 1);       // logical 16-byte chunk stored at physical chunk pc
            const int r = kt * BK + row;
            const int ca = m0 + lc * 8, cb = n0 + lc * 8;
            const T* pa = (r < R && ca < p.M) ? A + (size_t)r * p.lda + ca : zero;
            const T* pb = (r < R && cb < p.N) ? B + (size_t)r * p.ldb + cb : zero;
            const int base = (i * 256 + wave_u * 64) * 16;      // chunk id = i*256 + tid = row*16 + pc
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                             (__attribute__((address_space(3))) void*)(As + base), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb,
                                             (__attribute__((address_space(3))) void*)(Bs + base), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of A = A^T * ones: the first column tile's wn == 0 waves spend 4 extra MFMAs per k-half on an
    // all-ones B fragment (no LDS traffic, float accumulation) instead of a separate pass over dY
    const bool do_cs = p.a_colsum != nullptr && pid_n == 0 && wn == 0;
    f32x4 accb[4];
    v8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a 320-wide output ends in the middle of its third 128-wide tile: the waves whose 64 x 64 quadrant lies entirely outside the
    // output (wave-uniform) skip their fragment reads and MFMAs and only take part in the staging and the barriers
    const bool quad_live = m0 + wm * 64 < p.M && n0 + wn * 64 < p.N;
    auto compute = [&](int stage) __attribute__((always_inline)) {
        if (!quad_live) return;
        const char* As = smem + stage * STAGE_BYTES;
        const char* Bs = As + BM * BK * 2;
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = tn_frag<T>(As, wm * 4 + i, ks, fr, fg);
                bf[i] = tn_frag<T>(Bs, wn * 4 + i, ks, fr, fg);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = TT<T>::mfma(af[i], bf[j], acc[i][j]);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = TT<T>::mfma(af[i], ones, accb[i]);
            }
        }
    };
    // the same stage ring as gemm_v4_kernel (see its K-loop banner): NSTG - 1 tiles staged ahead, counted vmcnt + raw s_barrier when
    // NSTG > 2.  Every wave issues 8 pieces (4 of A, 4 of B) per tile.
    static_assert(NSTG >= 2 && NSTG <= 4, "2 to 4 LDS stages");
    constexpr int LA = NSTG - 1;
    auto wait_tiles = [&](int t) __attribute__((always_inline)) {
        if (NSTG == 2 || t <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (t == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    };
    auto stage_barrier = [&]() __attribute__((always_inline)) {
        if (NSTG == 2) { __syncthreads(); return; }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_tiles = kt_end - kt_begin;
#pragma unroll
    for (int t = 0; t < LA; ++t)
        if (t < n_tiles) issue(kt_begin + t, t);
    wait_tiles(min(LA, n_tiles) - 1);
    stage_barrier();
    {
        int cur = 0, nxt = LA, kt = 0;
        for (; kt + LA < n_tiles; ++kt) {
            issue(kt_begin + kt + LA, nxt);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            wait_tiles(LA - 1);
            stage_barrier();
            cur = cur + 1 == NSTG ? 0 : cur + 1;
            nxt = nxt + 1 == NSTG ? 0 : nxt + 1;
        }
        for (; kt < n_tiles; ++kt) {
            compute(cur);
            if (kt + 1 < n_tiles) {
                wait_tiles(n_tiles - 2 - kt);
                stage_barrier();
            }
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    }
    __syncthreads();
    if (do_cs && (lane & 15) == 0) {        // every column of the 16x16 result holds the sums: lanes 0/16/32/48 own rows 4*(lane>>4)..+3
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + e;
                if (n >= p.M) continue;
                // one owner per (row slice z, column n): a split reduction leaves its partial in slab z of a_colsum (summed in a fixed
                // order by svdx_gemm_finalize), an unsplit one adds to the running bias gradient -- no atomics, reproducible bits
                if (p.out_mode == SVDX_OUT_F32_SLAB) p.a_colsum[(size_t)z * p.M + n] = accb[i][e];
                else p.a_colsum[n] += accb[i][e];
            }
    }
    gemm_epilogue<T>(p, acc, smem, m0, n0, z, tid, lane, wm, wn);
}


// ----------------------------------------------------------------------------------------------------------------
// Eight-wave TN tiles: 128*PA output rows x 128*PB output columns (instantiated: 256 x 256), one workgroup per CU.
// The weight gradients of the step are [2560..10240] x [320..1280] outputs over 560..35840 rows: with 128 x 128 tiles the grids are
// 60-800 tiles that need up to 32 row slices to fill the chip; the larger tiles stage 2/3 (256 x 128) or 1/2 (256 x 256) of the
// operand bytes per flop and need a quarter of the slices.  Same operand layout as gemm_tn_kernel: an operand tile is PA (PB)
// panels of [64 r][128 cols] with that kernel's 8-byte-unit swizzle, fragments by ds_read_b64_tr_b16; accumulators are kept
// transposed (acc = mfma(B_frag, A_frag)) so that a lane owns 4 consecutive output columns: results leave as 16-byte stores
// straight from the registers.  Stage ring as in gemm_v4_kernel.
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int PA, int PB, int NSTG, bool BUF = false>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    constexpr int NT = 512;
    constexpr int TM = 128 * PA, TK = 128 * PB;                    // output tile
    constexpr int WMN = 2 * PA, WNN = 8 / WMN;                     // waves along the output rows / columns (64 rows each)
    constexpr int NJ = TK / WNN / 16;                              // 16-column blocks per wave (4 | 8)
    constexpr int PANEL = 64 * 256;                                // bytes of one [64 r][128 cols] panel
    constexpr int STAGE = (PA + PB) * PANEL;
    constexpr int NPIECE = 2 * (PA + PB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNN, wn = wave % WNN;
    const TnWho who = tn_who(p);
    if (!who.live) return;
    const int pid_m = who.pid_m, pid_n = who.pid_n;
    const int m0 = pid_m * TM, n0 = pid_n * TK;
    const int R = p.K;
    const int kt_total = (R + BK - 1) / BK;
    const int z = who.z;
    const int kt_per = (kt_total + p.split_k - 1) / p.split_k;
    const int kt_begin = z * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) {
        if (p.a_colsum && p.out_mode == SVDX_OUT_F32_SLAB && pid_n == 0 && tid < TM && m0 + tid < p.M) p.a_colsum[(size_t)z * p.M + m0 + tid] = 0.f;
        return;
    }
    const int pc = tid & 15, ld_row = tid >> 4;                   // ld_row 0..31; rows ld_row + 32 * i of a panel
    const T* zero = reinterpret_cast<const T*>(p.zero_page);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // BUF: see gemm_tn_kernel -- buffer_load ... lds through descriptors; per-thread byte offsets of its pieces inside K-tile 0
    const desc4 rsA = hidden_desc(p.A, BUF ? (unsigned)p.a_bytes : 0u), rsB = hidden_desc(p.B, BUF ? (unsigned)p.b_bytes : 0u);
    constexpr unsigned OOB = 0x7ffffff0u;
    unsigned voa[2][PA], vob[2][PB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + ld_row;
        const int lc = pc ^ (tn_rid(row) << 1);
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int ca = m0 + pa * 128 + lc * 8;
            voa[i][pa] = ca < p.M ? (unsigned)(row * p.lda + ca) * 2u : OOB;
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int cb = n0 + pb * 128 + lc * 8;
            vob[i][pb] = cb < p.N ? (unsigned)(row * p.ldb + cb) * 2u : OOB;
        }
    }
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        char* As = smem + stage * STAGE;
        char* Bs = As + PA * PANEL;
        if (BUF) {
            const unsigned ka = (unsigned)kt * (unsigned)(BK * 2) * (unsigned)p.lda, kb = (unsigned)kt * (unsigned)(BK * 2) * (unsigned)p.ldb;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int base = (i * NT + wave_u * 64) * 16;
#pragma unroll
                for (int pa = 0; pa < PA; ++pa)
                    hidden_dma<16>(rsA, As + pa * PANEL + base, voa[i][pa] + ka);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    hidden_dma<16>(rsB, Bs + pb * PANEL + base, vob[i][pb] + kb);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 32 + ld_row;
            const int lc = pc ^ (tn_rid(row) << 1);
            const int r = kt * BK + row;
            const int base = (i * NT + wave_u * 64) * 16;            // chunk id inside the panel = i * 512 + tid = row * 16 + pc
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                const int ca = m0 + pa * 128 + lc * 8;
                const T* src = (r < R && ca < p.M) ? A + (size_t)r * p.lda + ca : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(As + pa * PANEL + base), 16, 0, 0);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int cb = n0 + pb * 128 + lc * 8;
                const T* src = (r < R && cb < p.N) ? B + (size_t)r * p.ldb + cb : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Bs + pb * PANEL + base), 16, 0, 0);
            }
        }
    };
    f32x4 acc[4][NJ];                                             // [16-row block i][16-column block j], transposed inside a block
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = p.a_colsum != nullptr && pid_n == 0 && wn == 0;      // bias gradient = A^T * ones (see gemm_tn_kernel)
    f32x4 accb[4];
    v8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const char* As = smem + stage * STAGE;
        const char* Bs = As + PA * PANEL;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int blk = wm * 4 + i;
                af[i] = tn_frag<T>(As + (blk >> 3) * PANEL, blk & 7, ks, fr, fg);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int blk = wn * NJ + j;
                bf[j] = tn_frag<T>(Bs + (blk >> 3) * PANEL, blk & 7, ks, fr, fg);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = TT<T>::mfma(bf[j], af[i], acc[i][j]);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = TT<T>::mfma(af[i], ones, accb[i]);
            }
        }
    };
    static_assert(NSTG >= 2 && NSTG <= 3, "2 or 3 LDS stages");
    constexpr int LA = NSTG - 1;
    auto wait_tiles = [&](int t) __attribute__((always_inline)) {
        if (NSTG == 2 || t <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
    };
    auto stage_barrier = [&]() __attribute__((always_inline)) {
        if (NSTG == 2) { __syncthreads(); return; }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const int n_tiles = kt_end - kt_begin;
#pragma unroll
    for (int t = 0; t < LA; ++t)
        if (t < n_tiles) issue(kt_begin + t, t);
    wait_tiles(min(LA, n_tiles) - 1);
    stage_barrier();
    {
        int cur = 0, nxt = LA, kt = 0;
        for (; kt + LA < n_tiles; ++kt) {
            issue(kt_begin + kt + LA, nxt);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            wait_tiles(LA - 1);
            stage_barrier();
            cur = cur + 1 == NSTG ? 0 : cur + 1;
            nxt = nxt + 1 == NSTG ? 0 : nxt + 1;
        }
        for (; kt < n_tiles; ++kt) {
            compute(cur);
            if (kt + 1 < n_tiles) {
                wait_tiles(n_tiles - 2 - kt);
                stage_barrier();
            }
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    }
    if (do_cs && fr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = m0 + wm * 64 + i * 16 + fg * 4 + e;
                if (n >= p.M) continue;
                if (p.out_mode == SVDX_OUT_F32_SLAB) p.a_colsum[(size_t)z * p.M + n] = accb[i][e];
                else p.a_colsum[n] += accb[i][e];
            }
    }
    float* Cf = reinterpret_cast<float*>(p.C) + (p.out_mode == SVDX_OUT_F32_SLAB ? (size_t)z * p.slab_stride : 0);
    bool bad_any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = m0 + wm * 64 + i * 16 + fr;
        if (n >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = n0 + wn * (16 * NJ) + j * 16 + fg * 4;
            if (k >= p.N) continue;
            float* o = Cf + (size_t)n * p.ldc + k;
            const f32x4 v = acc[i][j] * p.alpha;
            if (p.vec_ok && k + 4 <= p.N) {
                if (p.out_mode == SVDX_OUT_F32) {
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));      // a weight gradient: read next by the optimizer
                    if (p.found_inf) bad_any |= not_finite(v[0]) || not_finite(v[1]) || not_finite(v[2]) || not_finite(v[3]);
                } else if (p.out_mode == SVDX_OUT_F32_SLAB) *reinterpret_cast<f32x4*>(o) = v;
                else if (p.out_mode == SVDX_OUT_F32_ADD) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(o) + v;
                    *reinterpret_cast<f32x4*>(o) = t;
                    if (p.found_inf) bad_any |= not_finite(t[0]) || not_finite(t[1]) || not_finite(t[2]) || not_finite(t[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(o + e, v[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (k + e >= p.N) continue;
                    if (p.out_mode == SVDX_OUT_F32 || p.out_mode == SVDX_OUT_F32_SLAB) {
                        o[e] = v[e];
                        bad_any |= p.found_inf && p.out_mode == SVDX_OUT_F32 && not_finite(v[e]);
                    } else if (p.out_mode == SVDX_OUT_F32_ADD) {
                        const float t = o[e] + v[e];
                        o[e] = t;
                        bad_any |= p.found_inf && not_finite(t);
                    } else atomicAdd(o + e, v[e]);
                }
            }
        }
    }
    raise_found_inf(p.found_inf, bad_any);
}


// Which (row tile, column tile, K slice) a workgroup of the gemm_v4 / gemm_v5 kernels computes (false: none -- the grid is rounded up).
// Workgroup ids go round-robin to the 8 XCDs (id & 7), each with a private 4 MiB L2.  Every XCD re-fetches whatever operand
// rows its tiles touch, so the host picks an (8/xcd_n) x xcd_n arrangement of XCDs over the tile grid that minimises
// A_bytes * xcd_n + B_bytes * (8 / xcd_n): row bands when A dominates (the 64x40 level: A = 23 MB x taps, B < 2 MB), column
// bands when the weights dominate (10x16 / 5x8 levels: B = 30-60 MB, A = 1-6 MB; measured 8x weight re-fetch before).
// Split-K (round 4): the K slices of one tile share nothing, so a slice count of 2 / 4 / 8 gives every slice its OWN 8 / split_k XCDs
// (z_xcd): an XCD then streams half / a quarter / an eighth of K for its tiles instead of all of it -- 2240 x 1280 x 10240 in two slices
// fetched 197 MB where 72 MB are algorithmic with every XCD holding both slices of a 3 x 5 tile block; 144 MB with 2 x 2 XCDs per slice.
__device__ __forceinline__ bool v4_tile_of_block(const GemmParams& p, int& pid_m, int& pid_n, int& z) {
    const int bid = blockIdx.x;
    z = blockIdx.y;
    if (p.z_xcd) {
        const int xcd = bid & 7, l = bid >> 3, xps = 8 / p.split_k;
        z = xcd / xps;
        const int xi = xcd - z * xps;
        const int xi_m = xi / p.xcd_n, xi_n = xi - xi_m * p.xcd_n;
        const int lm = l / p.sub_n;
        pid_m = xi_m * p.sub_m + lm;
        pid_n = xi_n * p.sub_n + (l - lm * p.sub_n);
        if (lm >= p.sub_m || pid_m >= p.tiles_m || pid_n >= p.tiles_n) return false;
    } else if (p.xcd_n > 0) {
        const int xcd = bid & 7, l = bid >> 3;
        const int xi_m = xcd / p.xcd_n, xi_n = xcd - xi_m * p.xcd_n;
        const int lm = l / p.sub_n;
        pid_m = xi_m * p.sub_m + lm;
        pid_n = xi_n * p.sub_n + (l - lm * p.sub_n);
        if (lm >= p.sub_m || pid_m >= p.tiles_m || pid_n >= p.tiles_n) return false;
    } else {
        const int nwg = p.tiles_m * p.tiles_n;
        const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
        const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
        pid_m = swz / p.tiles_n;
        pid_n = swz - pid_m * p.tiles_n;
    }
    return true;
}

// GEGLU-forward tiles pair 16 value columns with their 16 gate columns: local n-block 2q is rows F*0 + c, block 2q+1 is rows
// F + c of the [2F, K] projection, so a lane ends up holding a value and its gate (no weight re-packing needed).
template <int BN3>
__device__ __forceinline__ int v4_brow(const GemmParams& p, int pid_n, int n0, int nl) {
    return p.epi == SVDX_EPI_GEGLU_FWD ? (((nl >> 4) & 1) ? p.aux_dim : 0) + pid_n * (BN3 / 2) + (nl >> 5) * 16 + (nl & 15)
                                       : min(n0 + nl, p.N - 1);
}

// ---- the store side shared by gemm_v4_kernel and gemm_v5_kernel.  acc[i][j] is the TRANSPOSED 16x16 block (n-block i, m-block j) of a wave that owns
// rows wm * WM4 + j * 16 .. and columns wn * WN3 + i * 16 .. of a BM4 x BN3 tile computed by NT threads: lane (fr, fg) holds row fr, columns fg * 4 .. + 3.
template <typename T, int NB, int MB, int NT, int BM4, int BN3, int WM4, int WN3>
__device__ __forceinline__ void v4_epilogue(const GemmParams& p, char* smem, f32x4 (&acc)[NB][MB], int z, int m0, int n0, int pid_m, int pid_n,
                                            int wm, int wn, int tid) {
    const int lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int Fdim = p.aux_dim;
    const int m_lim = min(p.M, m0 + p.m_step);          // first row this tile does not own
    auto brow = [&](int nl) __attribute__((always_inline)) { return v4_brow<BN3>(p, pid_n, n0, nl); };
    // ---- epilogue A (activation output): coalesced.  Each lane adds bias / row vector to its 4-column groups, rounds to the
    //      activation dtype and parks them in LDS (the stage buffers are free now); then every thread moves 16-byte row
    //      pieces: residual add + store with full-line coalescing (20 lanes cover one 320-byte output row of the tile).
    if (p.out_mode == SVDX_OUT_ACT && p.vec_ok && (p.ldc % 8 == 0) && (!p.res || p.ldres % 8 == 0) &&
        (n0 + BN3 <= p.N || p.epi == SVDX_EPI_GEGLU_FWD)) {
        constexpr int PITCH = (BN3 + 8) * 2;              // bytes per staged row (multiple of 16)
        __syncthreads();
        if (p.gn_stats) {                                  // the statistics table sits behind the parked tile; every wave has left the K-loop
            unsigned long long* gacc = reinterpret_cast<unsigned long long*>(smem + ((BM4 * PITCH + 15) & ~15));
            for (int i = tid; i < GN_MAX_S * GN_MAX_G * 2; i += NT) gacc[i] = 0ull;
        }
        {
            const bool lead0 = (z == 0);
            const int nb0 = n0 + wn * WN3 + fg * 4;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float bb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bb[e] = (lead0 && p.bias) ? p.bias[p.epi == SVDX_EPI_GEGLU_FWD ? brow(wn * WN3 + i * 16 + fg * 4 + e) : nb0 + i * 16 + e] : 0.f;
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int ml = wm * WM4 + j * 16 + fr;
                    const int m = min(m0 + ml, p.M - 1);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha + bb[e];
                    if (lead0 && p.rowvec) {
                        const float* rv = p.rowvec + (size_t)(p.rv_mod ? (m % p.rv_mod) : (m / p.rv_rpg)) * p.rv_ld + nb0 + i * 16;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    Vec4<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(v[e]);
                    *reinterpret_cast<Vec4<T>*>(smem + ml * PITCH + (wn * WN3 + i * 16 + fg * 4) * 2) = o;
                }
            }
        }
        __syncthreads();
        if (p.epi == SVDX_EPI_GEGLU_FWD) {
            // tile = 128 rows x (BN3/32) pairs of [16 values | 16 gates]; emit pre (both halves) and h = value * gelu(gate)
            T* pre = reinterpret_cast<T*>(p.C);
            T* hh = reinterpret_cast<T*>(p.aux_out);
            constexpr int NPAIR = BN3 / 32;
            for (int id = tid; id < BM4 * NPAIR * 2; id += NT) {
                const int row = id / (NPAIR * 2), r2 = id - row * (NPAIR * 2);
                const int pr = r2 >> 1, c2 = r2 & 1;
                const int m = m0 + row;
                if (m >= m_lim) continue;
                const Vec8<T> a8 = *reinterpret_cast<const Vec8<T>*>(smem + row * PITCH + (pr * 32 + c2 * 8) * 2);
                const Vec8<T> g8 = *reinterpret_cast<const Vec8<T>*>(smem + row * PITCH + (pr * 32 + 16 + c2 * 8) * 2);
                const int fc = pid_n * (BN3 / 2) + pr * 16 + c2 * 8;
                if (fc >= Fdim) continue;
                Vec8<T> h8;
#pragma unroll
                for (int e = 0; e < 8; ++e) h8.v[e] = from_f<T>(to_f<T>(a8.v[e]) * gelu_erf(to_f<T>(g8.v[e])));
                // `pre` is not read again before the backward sweep: streaming (non-temporal) stores keep it out of the L2 / Infinity
                // Cache lines the next kernels want
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, a8), reinterpret_cast<u32x4*>(pre + (size_t)m * p.ldc + fc));
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, g8), reinterpret_cast<u32x4*>(pre + (size_t)m * p.ldc + Fdim + fc));
                *reinterpret_cast<Vec8<T>*>(hh + (size_t)m * Fdim + fc) = h8;
            }
            return;
        }
        constexpr int CPR = BN3 / 8;                      // 16-byte chunks per row
        if (p.epi == SVDX_EPI_GEGLU_BWD) {
            // the tile holds d(h) for columns n0..; read (value, gate) from pre and emit d(pre) = [dh*gelu(g) | dh*a*gelu'(g)]
            const T* pre = reinterpret_cast<const T*>(p.aux_in);
            T* dpre = reinterpret_cast<T*>(p.C);
            for (int id = tid; id < BM4 * CPR; id += NT) {
                const int row = id / CPR, c = id - row * CPR;
                const int m = m0 + row;
                if (m >= m_lim) continue;
                const Vec8<T> d8 = *reinterpret_cast<const Vec8<T>*>(smem + row * PITCH + c * 16);
                const size_t po = (size_t)m * (2 * Fdim) + n0 + c * 8;
                // `pre` was written a whole forward sweep ago and is dead after this read: streaming loads, so that its 183 MB (64x40 level) do not
                // push the d(pre) lines this launch writes -- the A operand of the next GEMM -- out of the Infinity Cache
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const Vec8<T> a8 = __builtin_bit_cast(Vec8<T>, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pre + po)));
                const Vec8<T> g8 = __builtin_bit_cast(Vec8<T>, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pre + po + Fdim)));
                Vec8<T> da, dg;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = to_f<T>(d8.v[e]), gv = to_f<T>(g8.v[e]);
                    const GeluParts gp = gelu_parts(gv);
                    da.v[e] = from_f<T>(d * gv * gp.cdf);
                    dg.v[e] = from_f<T>(d * to_f<T>(a8.v[e]) * (gp.cdf + gp.pdf_x));
                }
                *reinterpret_cast<Vec8<T>*>(dpre + po) = da;
                *reinterpret_cast<Vec8<T>*>(dpre + po + Fdim) = dg;
            }
            return;
        }
        T* Ct2 = reinterpret_cast<T*>(p.C);
        const T* R2 = (z == 0) ? reinterpret_cast<const T*>(p.res) : nullptr;
        if (p.gn_stats) {
            // Same stores, with every thread on ONE fixed 16-byte column chunk (threads beyond the last whole row of a pass idle) so that the
            // statistics of its 8 channels stay in registers while it walks down the tile's rows.  Rows ascend, so the GroupNorm sample
            // (frame or clip) of a thread changes monotonically: on a change, and at the end, the 8 channel sums are merged into their
            // groups and added to the tile's [sample][group] table in LDS as 64-bit fixed-point integers (integer addition is
            // associative: the statistics do not depend on the order of the atomics -- run-to-run identical, like svdx_gn_stats);
            // the table's nonzero entries then go to the replica slots in HBM, one atomic each.
            constexpr int RPS = NT / CPR;
            unsigned long long* gacc = reinterpret_cast<unsigned long long*>(smem + ((BM4 * PITCH + 15) & ~15));
            const int c = tid % CPR;
            const int s_first = m0 / p.gn_rows, g_first = n0 / p.gn_cg;
            float a0[8], a1[8];
            int cur_s = -1;
            auto flush = [&](int sl) __attribute__((always_inline)) {
                int cur = (n0 + c * 8) / p.gn_cg;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int g = (n0 + c * 8 + e) / p.gn_cg;
                    if (g != cur) {
                        atomicAdd(&gacc[(sl * GN_MAX_G + cur - g_first) * 2], (unsigned long long)__float2ll_rn(s0 * p.gn_m0));
                        atomicAdd(&gacc[(sl * GN_MAX_G + cur - g_first) * 2 + 1], (unsigned long long)__float2ll_rn(s1 * p.gn_m1));
                        cur = g; s0 = 0.f; s1 = 0.f;
                    }
                    s0 += a0[e]; s1 += a1[e];
                }
                atomicAdd(&gacc[(sl * GN_MAX_G + cur - g_first) * 2], (unsigned long long)__float2ll_rn(s0 * p.gn_m0));
                atomicAdd(&gacc[(sl * GN_MAX_G + cur - g_first) * 2 + 1], (unsigned long long)__float2ll_rn(s1 * p.gn_m1));
            };
            if (tid < RPS * CPR) {
                for (int row = tid / CPR; row < BM4; row += RPS) {
                    const int m = m0 + row;
                    if (m >= m_lim) break;
                    Vec8<T> o8 = *reinterpret_cast<const Vec8<T>*>(smem + row * PITCH + c * 16);
                    const size_t co = (size_t)m * p.ldc + n0 + c * 8;
                    if (R2) {
                        const Vec8<T> r8 = *reinterpret_cast<const Vec8<T>*>(R2 + (size_t)m * p.ldres + n0 + c * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8.v[e] = from_f<T>(to_f<T>(o8.v[e]) + to_f<T>(r8.v[e]));
                    }
                    *reinterpret_cast<Vec8<T>*>(Ct2 + co) = o8;
                    const int sl = m / p.gn_rows - s_first;
                    if (sl != cur_s) {
                        if (cur_s >= 0) flush(cur_s);
                        cur_s = sl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
                    }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float v = to_f<T>(o8.v[e]); a0[e] += v; a1[e] += v * v; }
                }
                if (cur_s >= 0) flush(cur_s);
            }
            __syncthreads();
            const int ns_t = (m_lim - 1) / p.gn_rows - s_first + 1, ng_t = (min(n0 + BN3, p.N) - 1) / p.gn_cg - g_first + 1;
            const int n_s = p.M / p.gn_rows, G = p.N / p.gn_cg;
            unsigned long long* out = p.gn_stats + (size_t)((pid_m + pid_n) % SVDX_GN_REPLICAS) * n_s * G * 2;
            for (int i = tid; i < ns_t * ng_t * 2; i += NT) {
                const int w = i & 1, gl = (i >> 1) % ng_t, sl = (i >> 1) / ng_t;
                const unsigned long long v = gacc[(sl * GN_MAX_G + gl) * 2 + w];
                if (v) atomicAdd(out + ((size_t)(s_first + sl) * G + g_first + gl) * 2 + w, v);
            }
            return;
        }
#pragma unroll 2
        for (int id = tid; id < BM4 * CPR; id += NT) {
            const int row = id / CPR, c = id - row * CPR;
            const int m = m0 + row;
            if (m >= m_lim) continue;
            const Vec8<T> t8 = *reinterpret_cast<const Vec8<T>*>(smem + row * PITCH + c * 16);
            const size_t co = (size_t)m * p.ldc + n0 + c * 8;
            if (R2) {
                const Vec8<T> r8 = *reinterpret_cast<const Vec8<T>*>(R2 + (size_t)m * p.ldres + n0 + c * 8);
                Vec8<T> o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8.v[e] = from_f<T>(to_f<T>(t8.v[e]) + to_f<T>(r8.v[e]));
                *reinterpret_cast<Vec8<T>*>(Ct2 + co) = o8;
            } else {
                *reinterpret_cast<Vec8<T>*>(Ct2 + co) = t8;
            }
        }
        return;
    }
    // ---- direct epilogue: lane (fr, fg) owns row m = .. + fr and columns n = .. + fg*4 + {0..3} of every 16x16 block ----
    const bool lead = (z == 0) && p.out_mode != SVDX_OUT_F32_SLAB;
    T* Ct = reinterpret_cast<T*>(p.C);
    float* Cf = reinterpret_cast<float*>(p.C) + (p.out_mode == SVDX_OUT_F32_SLAB ? (size_t)z * p.slab_stride : 0);
    const T* R = reinterpret_cast<const T*>(p.res);
    const int nbase = n0 + wn * WN3 + fg * 4;
    float bv[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nbase + i * 16 + e;
            bv[i][e] = (lead && p.bias && n < p.N) ? p.bias[n] : 0.f;
        }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = m0 + wm * WM4 + j * 16 + fr;
        if (m >= m_lim) continue;
        const float* rv = nullptr;
        if (lead && p.rowvec) rv = p.rowvec + (size_t)(p.rv_mod ? (m % p.rv_mod) : (m / p.rv_rpg)) * p.rv_ld;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = nbase + i * 16;
            if (n >= p.N) continue;
            const int nvalid = min(4, p.N - n);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha + bv[i][e];
            if (rv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nvalid) v[e] += rv[n + e];
            }
            const bool full = p.vec_ok && nvalid == 4;
            if (lead && R) {
                const T* rp = R + (size_t)m * p.ldres + n;
                if (full) {
                    const Vec4<T> r4 = *reinterpret_cast<const Vec4<T>*>(rp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += to_f<T>(r4.v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nvalid) v[e] += to_f<T>(rp[e]);
                }
            }
            const size_t co = (size_t)m * p.ldc + n;
            if (p.out_mode == SVDX_OUT_ACT) {
                if (full) {
                    Vec4<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(v[e]);
                    *reinterpret_cast<Vec4<T>*>(Ct + co) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nvalid) Ct[co + e] = from_f<T>(v[e]);
                }
            } else if (p.out_mode == SVDX_OUT_F32 || p.out_mode == SVDX_OUT_F32_SLAB) {
                if (full) *reinterpret_cast<f32x4*>(Cf + co) = f32x4{v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nvalid) Cf[co + e] = v[e];
                }
            } else if (p.out_mode == SVDX_OUT_F32_ADD) {
                if (full) {
                    f32x4 c = *reinterpret_cast<const f32x4*>(Cf + co);
                    c += f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(Cf + co) = c;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nvalid) Cf[co + e] += v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nvalid) atomicAdd(Cf + co + e, v[e]);
            }
        }
    }
}

// ================================================================================================================
// variant 4 (production): 128 x (32*NB) x 64 tile, NB = 4 or 5, lean K-loop.
//  * All channel counts of the SVD UNet are multiples of 320, so BN = 160 tiles N exactly (BN = 128 wastes 17 % of the MFMA
//    work at N = 320) and raises the MFMA : ds_read ratio (40 : 18 per wave per K-tile).
//  * Accumulators are kept TRANSPOSED (acc = mfma(B_frag, A_frag)): a lane owns 4 consecutive output columns of one row.
//    Activation outputs are rounded, parked in LDS and written with full-line coalesced 16-byte stores (bias / row vector /
//    residual / GEGLU forward+backward fused there); float outputs (slabs, += for weight-grad style uses) go straight from
//    registers with 16-byte stores.
//  * Staging: buffer_load_dwordx4 ... lds with per-lane 32-bit byte offsets computed once per filter tap, the K position as a
//    single scalar soffset, and zero padding from the buffer bounds check (offset >= num_records reads 0) -- no zero page,
//    no select, no 64-bit pointer arithmetic in the loop.  PMC counters that motivated this (profiles/r1_gemm_pmc.txt): the
//    pointer-arithmetic loop issued 2.4 VALU + 2.9 SALU instructions per MFMA (issue bound, MFMA busy ~33 %); this loop issues
//    0.3 VALU + 0.95 SALU.
//  * Tried and dropped: a 5-stage / 4-stage LDS ring with counted vmcnt + raw s_barrier (no gain in
//    situ, slower in isolation: the loop is not latency bound), direct 8-byte epilogue stores (worse DRAM efficiency).
// ================================================================================================================
template <typename T, int NB, int MB, bool DUAL, int WGM, int NSTG>
__global__ __launch_bounds__(128 * WGM, 2) void gemm_v4_kernel(GemmParams p) {     // (two waves per SIMD: the four-wave tiles share a CU in pairs -- 256 registers, whatever the allocator would like)
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and LDS-DMA builtin only exist in the device pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    constexpr int NT = 128 * WGM;                     // threads: WGM x 2 waves (WGM = 2: four waves, WGM = 4: eight = two per SIMD)
    constexpr int BM4 = 16 * MB * WGM;                // tile rows: 128 / 160 (four waves) or 256 / 320 (eight waves)
    constexpr int WM4 = 16 * MB;                      // rows per wave
    constexpr int BN3 = 32 * NB;                      // 2 waves along N, NB/2... each wave owns NB*16 columns
    constexpr int WN3 = 16 * NB;                      // columns per wave
    constexpr int KT = BK;                             // K extent of one stage
    constexpr int STAGE3 = (BM4 + BN3) * KT * 2;
    constexpr int CPRW = KT / 8;                       // 16-byte chunks per staged row (8 | 4)
    constexpr int RPP = NT / CPRW;                     // rows covered by one load pass (32 | 64)
    constexpr int NLA = BM4 / RPP;                      // A loads per thread per stage (4 | 2)
    constexpr int NLB = (BN3 + RPP - 1) / RPP;         // B loads per thread per stage (NB | 3 or 2)
    constexpr int ROWB = KT * 2;                       // bytes per staged row (128 | 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int pid_m, pid_n, z;
    if (!v4_tile_of_block(p, pid_m, pid_n, z)) return;
    const int m0 = pid_m * p.m_step, n0 = pid_n * BN3;
    const int kt_total = p.K / KT;
    const int kt_per = (kt_total + p.split_k - 1) / p.split_k;
    const int kt_begin = z * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;

    // ---- lean staging: buffer_load ... lds with per-lane byte offsets (fixed per filter tap) + one scalar K offset ----
    const int ld_row = tid / CPRW, pc = tid % CPRW;
    const int lc = pc ^ (ld_row & 7);
    RowInfo a_ri[NLA];
    int a_m[NLA], voa[NLA], vob[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        a_m[i] = min(m0 + i * RPP + ld_row, p.M - 1);
        a_ri[i] = decode_row(p.g, a_m[i]);
    }
    // GEGLU-forward tiles pair 16 value columns with their 16 gate columns: local n-block 2q is rows F*0 + c, block 2q+1 is rows
    // F + c of the [2F, K] projection, so a lane ends up holding a value and its gate (no weight re-packing needed).
    const int Fdim = p.aux_dim;
    auto brow = [&](int nl) __attribute__((always_inline)) {
        return p.epi == SVDX_EPI_GEGLU_FWD ? (((nl >> 4) & 1) ? Fdim : 0) + pid_n * (BN3 / 2) + (nl >> 5) * 16 + (nl & 15)
                                           : min(n0 + nl, p.N - 1);
    };
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int nl = i * RPP + ld_row;               // tile-local B row; rows >= BN3 (ring, NB = 5) are dummy loads that keep vmcnt uniform
        vob[i] = nl < BN3 ? (brow(nl) * p.ldb + lc * 8) * 2 : (int)0x80000000;
    }
    const bool plain = p.g.mode == SVDX_GATHER_PLAIN;
    const int cin = plain ? p.K : p.g.cin;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // DUAL: after the K-tiles of (A, B) the same pipeline runs the K2 / KT tiles of the plain pair (A2, B2) into the same
    // accumulators -- y = x W^T + (s x A^T) B^T in one launch instead of a second GEMM that re-reads and re-writes y
    bool seg2 = false;
    int tap = plain ? 0 : (kt_begin * KT) / cin;
    int ci0 = kt_begin * KT - tap * cin;                // channel offset inside the tap (plain: k offset)
    auto set_tap = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            bool valid;
            const T* ptr = a_row_ptr<T>(p, a_ri[i], a_m[i], 0, tap, 0, valid);
            // invalid (zero-padding) rows: an offset beyond num_records makes the buffer load return zeros
            voa[i] = valid ? (int)((ptr - reinterpret_cast<const T*>(p.A)) + lc * 8) * 2 : (int)0x80000000;
        }
    };
    set_tap();
    // One staging piece = one buffer_load ... lds per wave (1 KiB).  Issuing an LDS-DMA piece costs the wave ~60-180 issue cycles
    // (MI355X_MICROARCH.md), so the pieces of the NEXT stage are spread between the MFMA rows of the current one instead of being
    // issued as a burst in front of them (which left the wave's MFMA pipe idle for ~1000 cycles per K-step).
    constexpr int NPIECE = NLA + NLB;
    // Eight waves cover 64 tile rows per pass, so a 160-row B tile ends in the middle of its third pass: the waves whose 8 rows lie
    // beyond the tile have nothing to fetch there and skip that piece (their vmcnt arithmetic below counts one piece less per tile).
    constexpr bool B_PARTIAL = NLB * RPP > BN3;
    const bool skip_last = B_PARTIAL && (NLB - 1) * RPP + wave_u * (64 / CPRW) >= BN3;
    auto issue_piece = [&](int stage, int pi) __attribute__((always_inline)) {
        char* As = smem + stage * STAGE3;
        char* Bs = As + BM4 * KT * 2;
        if (pi < NLA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(As + (pi * NT + wave_u * 64) * 16), 16,
                                                     voa[pi], ci0 * 2, 0, 0);
        } else {
            const int i = pi - NLA;
            if (B_PARTIAL && i == NLB - 1 && skip_last) return;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(Bs + (i * NT + wave_u * 64) * 16), 16, vob[i],
                                                     (tap * cin + ci0) * 2, 0, 0);
        }
    };
    const int n_main = kt_end - kt_begin;                          // K-tiles of the main pair handled by this block
    const int n_tiles = n_main + (DUAL ? p.K2 / KT : 0);
    int issued = 0;
    // called after a tile was issued: make the addressing state describe the next one
    auto advance_k = [&]() __attribute__((always_inline)) {
        ++issued;
        if (DUAL && issued == n_main) {                            // next tile is the first one of (A2, B2): plain addressing
            seg2 = true;
            ci0 = 0;
            tap = 0;                                                   // B offset (tap * cin + ci0) becomes ci0
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A2), 0, p.a2_bytes, 0x00020000);
            rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B2), 0, p.b2_bytes, 0x00020000);
            const int a2_col = p.a2_seg > 0 ? (n0 / p.a2_seg) * p.K2 : 0;
#pragma unroll
            for (int i = 0; i < NLA; ++i) voa[i] = (min(m0 + i * RPP + ld_row, p.M - 1) * p.lda2 + a2_col + lc * 8) * 2;
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const int nl = i * RPP + ld_row;
                vob[i] = nl < BN3 ? (brow(nl) * p.ldb2 + lc * 8) * 2 : (int)0x80000000;
            }
            return;
        }
        ci0 += KT;
        if (!plain && !(DUAL && seg2) && ci0 == cin) { ci0 = 0; ++tap; set_tap(); }
    };
    f32x4 acc[NB][MB];                                   // [n-block][m-block], transposed: rows = n, cols = m
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    // compute stage `stage`; when ISSUE, stage a later K-tile into `nstage` piece by piece between the MFMA rows
#define SVDX_V4_COMPUTE(stage, nstage, ISSUE)                                                                                \
    {                                                                                                                        \
        const char* As_ = smem + (stage) * STAGE3;                                                                           \
        const char* Bs_ = As_ + BM4 * KT * 2;                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < KT / 32; ++kk) {                                                             \
            const int chunk = ((kk * 4 + fg) ^ (fr & 7)) * 16;                                                               \
            v8 af[MB], bf[NB];                                                                                               \
            _Pragma("unroll") for (int i = 0; i < MB; ++i)                                                                   \
                af[i] = *reinterpret_cast<const v8*>(As_ + (wm * WM4 + i * 16 + fr) * ROWB + chunk);                         \
            _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                                   \
                bf[i] = *reinterpret_cast<const v8*>(Bs_ + (wn * WN3 + i * 16 + fr) * ROWB + chunk);                         \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                                 \
                __builtin_amdgcn_s_setprio(1);  /* the MFMA row outranks the other waves' loads and LDS reads */            \
                _Pragma("unroll") for (int j = 0; j < MB; ++j) acc[i][j] = TT<T>::mfma(bf[i], af[j], acc[i][j]);             \
                __builtin_amdgcn_s_setprio(0);                                                                               \
                if (ISSUE && kk * NB + i < NPIECE) {                                                                         \
                    __builtin_amdgcn_sched_barrier(0);                                                                       \
                    issue_piece((nstage), kk * NB + i);                                                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                       \
                }                                                                                                            \
            }                                                                                                                \
        }                                                                                                                    \
    }
    static_assert(NPIECE <= (KT / 32) * NB, "not enough MFMA rows to hide the staging pieces");
    static_assert(NSTG >= 2 && NSTG <= 4, "2 to 4 LDS stages");
    // The stages form a ring: while tile kt is computed, tiles kt+1 .. kt+NSTG-2 are in flight and the pieces of tile kt+NSTG-1 are
    // issued into the stage tile kt-1 just left.  One barrier per K-step.  With more than two stages the wait in front of it is a
    // COUNTED vmcnt -- it retires this wave's pieces of tile kt+1 and leaves the younger tiles in flight across the barrier, which
    // must then be the raw s_barrier (__syncthreads() fences with vmcnt(0) while an LDS-DMA is pending); every wave issues the same
    // number of pieces per tile, so the count is exact.  The two-stage form (wait for everything, __syncthreads) is the round-1 loop:
    // right when two workgroups share a CU and hide each other's drain; the ring is for grids of <= 1 workgroup per CU (the 10x16 /
    // 5x8 levels, where a drained K-step is one exposed L2 / HBM round trip) and for the eight-wave tiles.
    auto wait_tiles = [&](int t) __attribute__((always_inline)) {          // leave at most the t youngest tiles in flight (wave-uniform)
        if (NSTG == 2 || t <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        if (B_PARTIAL && skip_last) {
            if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NPIECE - 1)) : "memory");
        } else {
            if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPIECE) : "memory");
        }
    };
    auto stage_barrier = [&]() __attribute__((always_inline)) {
        if (NSTG == 2) { __syncthreads(); return; }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        constexpr int LA = NSTG - 1;                                       // tiles staged ahead of the one being computed
#pragma unroll
        for (int t = 0; t < LA; ++t)
            if (t < n_tiles) {
#pragma unroll
                for (int pi = 0; pi < NPIECE; ++pi) issue_piece(t, pi);
                advance_k();
            }
        wait_tiles(min(LA, n_tiles) - 1);
        stage_barrier();
        int cur = 0, nxt = LA, kt = 0;
        for (; kt + LA < n_tiles; ++kt) {
            SVDX_V4_COMPUTE(cur, nxt, true);
            advance_k();
            wait_tiles(LA - 1);
            stage_barrier();
            cur = cur + 1 == NSTG ? 0 : cur + 1;
            nxt = nxt + 1 == NSTG ? 0 : nxt + 1;
        }
        for (; kt < n_tiles; ++kt) {
            SVDX_V4_COMPUTE(cur, nxt, false);
            if (kt + 1 < n_tiles) {
                wait_tiles(n_tiles - 2 - kt);
                stage_barrier();
            }
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    }
#undef SVDX_V4_COMPUTE

    v4_epilogue<T, NB, MB, NT, BM4, BN3, WM4, WN3>(p, smem, acc, z, m0, n0, pid_m, pid_n, wm, wn, tid);
#endif
}

// ================================================================================================================
// variant 5 family (round 6): two-role eight-wave tiles, (32 MF) x (64 NF) x 64, ONE workgroup per CU.
//
// Why: gemm_v4_kernel runs all waves of a workgroup through the same sequence -- fragment reads, then MFMA rows -- so the two waves that
// share a SIMD's matrix pipe read LDS together and multiply together; counters (profiles/r5_stall_counters.txt) show its waves parked at
// waits / barriers 35-55 % of their cycles.  Here the eight waves are 2 (M) x 4 (N); waves 0-3 (one per SIMD) and waves 4-7 (their SIMD
// partners) run the SAME program ONE BARRIER APART: while one group is in the MFMA segment of a phase its partners are in the load
// segment of theirs (ds_read_b128 of the next fragment group + their share of the staging DMA), and they swap at every barrier -- the
// matrix pipe of each SIMD always has exactly one wave feeding it (cdna_hip_programming.md, "256^2 8-phase template").
//
// One K-tile (64) = four phases, one quadrant of the wave's (16 MF) x (16 NF) output each:
//     phase     fragments read                 MFMA quadrant      staging issued (LDS-DMA pieces of 64 rows x 128 B by all 512 threads)
//     0         A sub 0, B sub 0               (m0, n0)           A rows that hold sub 1, K-tile t+1   (last read: phase 2 of t-1)
//     1         B sub 1                        (m0, n1)           --                                    [wait A]
//     2         A sub 1                        (m1, n1)           A rows of sub 0 only + B rows that hold sub 0, K-tile t+2   (last read: phase 0 of t)
//     3         -- (B sub 0 stays in registers) (m1, n0)          B rows of sub 1 only, K-tile t+2     (last read: phase 1 of t)   [wait B]
// LDS holds TWO K-tiles; inside a tile the rows are grouped by sub-tile ([sub 0: wave row 0, wave row 1][sub 1: ...]), so that a region is
// free for K-tile t+2 as soon as its last phase of K-tile t is over.  A region is re-staged two phases after its last read (the partner
// group is one barrier behind: its reads of that phase retire one barrier interval later).  Two counted waits per K-tile, each leaving ONE
// WHOLE K-TILE of pieces in flight across the barriers: [wait B] in phase 3 retires what phases 0 and 1 of K-tile t+1 read (issued in
// phases 2 and 3 of t-1), [wait A] in phase 1 retires the sub-1 rows of A that phase 2 reads (issued in phase 0 of t-1) -- every piece
// flies for four to five phases, and the first read of a region comes a barrier after its wait (both groups), as the LDS-DMA ordering
// rule demands.  (The first cut of this kernel re-read B sub 0 in phase 3, which held its region until then: the weights' last pieces had
// two phases to land.  Isolated, L2-warm runs did not care; inside the step, where every weight tile comes from HBM, the tile lost 5-13 %
// to the two-per-CU four-wave tiles -- profiles/r6d_tune_dump.txt.)  Beyond the last K-tile the pieces are issued with an out-of-range
// offset (no memory access, zeros into a free region): every wave's vmcnt arithmetic stays the same to the end.
// Epilogues: v4_epilogue (same accumulator layout: a wave owns rows wr * 16 MF .. and columns wc * 16 NF ..).
// ================================================================================================================
template <typename T, int MF, int NF>
__global__ __launch_bounds__(512) void gemm_v5_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    constexpr int NT = 512;
    constexpr int BM5 = 32 * MF, BN5 = 64 * NF, WM5 = 16 * MF, WN5 = 16 * NF;
    constexpr int MF0 = (MF + 1) / 2, MF1 = MF - MF0, NF0 = (NF + 1) / 2, NF1 = NF - NF0;
    constexpr int A0R = 2 * MF0 * 16, B0R = 4 * NF0 * 16;          // LDS rows of sub-tile 0 of A / B (all wave rows / wave columns)
    constexpr int KT = BK, ROWB = KT * 2;
    constexpr int BUFB = (BM5 + BN5) * ROWB;                         // one K-tile: A rows, then B rows
    constexpr int PA = (BM5 + 63) / 64, PB = BN5 / 64;               // staging passes (64 LDS rows each) per operand and K-tile
    static_assert(BM5 % 32 == 0 && BN5 % 64 == 0 && 2 * BUFB <= 160 * 1024, "tile does not fit");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave_u >> 2, wc = wave_u & 3;
    int pid_m, pid_n, z;
    if (!v4_tile_of_block(p, pid_m, pid_n, z)) return;
    const int m0 = pid_m * p.m_step, n0 = pid_n * BN5;
    const int kt_total = p.K / KT;
    const int kt_per = (kt_total + p.split_k - 1) / p.split_k;
    const int kt_begin = z * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;
    const int n_tiles = kt_end - kt_begin;

    // ---- staging addresses: pass i moves LDS rows 64 i + tid / 8, 16-byte chunk tid % 8 (XOR-swizzled by the row, as in gemm_v4_kernel)
    const int ld_row = tid >> 3, pc = tid & 7;
    const int lc = pc ^ (ld_row & 7);
    RowInfo a_ri[PA];
    int a_m[PA], voa[PA], vob[PB];
    constexpr int OOB = (int)0x80000000;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = 64 * i + ld_row;                                // LDS row -> row of the tile
        const int s = r >= A0R, q = s ? r - A0R : r, per = (s ? MF1 : MF0) * 16;
        const int tr = (q / per) * WM5 + (s ? MF0 * 16 : 0) + q % per;
        a_m[i] = min(m0 + tr, p.M - 1);
        a_ri[i] = decode_row(p.g, a_m[i]);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int r = 64 * i + ld_row;
        const int s = r >= B0R, q = s ? r - B0R : r, per = (s ? NF1 : NF0) * 16;
        const int nl = (q / per) * WN5 + (s ? NF0 * 16 : 0) + q % per;
        vob[i] = (v4_brow<BN5>(p, pid_n, n0, nl) * p.ldb + lc * 8) * 2;
    }
    const bool plain = p.g.mode == SVDX_GATHER_PLAIN;
    const int cin = plain ? p.K : p.g.cin;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);
    // the A cursor (K-tile a_t: filter tap + channel offset inside it) and the B cursor (K-tile b_t) advance independently: an operand's
    // pieces are issued in K-tile order, but A's and B's pieces of one K-tile are two phases apart
    int a_t = 0, b_t = 0;
    int tap = plain ? 0 : (kt_begin * KT) / cin;
    int ci0 = kt_begin * KT - tap * cin;
    int kb = kt_begin * KT * 2;                                      // byte offset of B's K-tile
    auto set_tap = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bool valid;
            const T* ptr = a_row_ptr<T>(p, a_ri[i], a_m[i], 0, tap, 0, valid);
            voa[i] = valid ? (int)((ptr - reinterpret_cast<const T*>(p.A)) + lc * 8) * 2 : OOB;
        }
    };
    set_tap();
    auto advance_a = [&]() __attribute__((always_inline)) {
        ++a_t;
        ci0 += KT;
        if (a_t >= n_tiles) {
#pragma unroll
            for (int i = 0; i < PA; ++i) voa[i] = OOB;
        } else if (!plain && ci0 == cin) { ci0 = 0; ++tap; set_tap(); }
    };
    auto advance_b = [&]() __attribute__((always_inline)) {
        ++b_t;
        kb += KT * 2;
        if (b_t >= n_tiles) {
#pragma unroll
            for (int i = 0; i < PB; ++i) vob[i] = OOB;
        }
    };
    // pieces of one slot.  SLOT 0: A passes that touch sub 1; 2: A passes inside sub 0 and B passes that touch sub 0; 3: B passes inside sub 1.
    // An A pass whose rows lie beyond the tile for this wave (160-row tiles: pass 2, waves 4-7) is skipped: that wave's counts are one lower.
#define SVDX_V5_ISSUE(SLOT, buf)                                                                                                            \
    {                                                                                                                                       \
        char* base_ = smem + (buf) * BUFB + wave_u * 1024;                                                                                  \
        if ((SLOT) == 0 || (SLOT) == 2) {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < PA; ++i) {          /* ascending: a pass that straddles sub 0 / sub 1 leads slot 0 */       \
                const bool in0 = 64 * i + 64 <= A0R;                                                                                        \
                if (((SLOT) == 2) != in0) continue;                                                                                         \
                if (64 * i + 64 > BM5 && 64 * i + wave_u * 8 >= BM5) continue;                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base_ + i * 8192), 16, voa[i],      \
                                                         ci0 * 2, 0, 0);                                                                    \
            }                                                                                                                               \
        }                                                                                                                                   \
        if ((SLOT) == 2 || (SLOT) == 3) {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                                                \
                const bool in1 = 64 * i >= B0R;                                                                                             \
                if (((SLOT) == 3) != in1) continue;                                                                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base_ + BM5 * ROWB + i * 8192), 16, \
                                                         vob[i], kb, 0, 0);                                                                 \
            }                                                                                                                               \
        }                                                                                                                                   \
    }
    // [wait A] leaves one K-tile's worth of this wave's pieces in flight.  [wait B] must also retire the A pass that straddles the two
    // sub-tiles (160-row tiles: rows 64-127 of 96 + 64) -- it holds sub-0 rows that phase 0 reads, but could only be issued with the sub-1
    // group in phase 0; it is the OLDEST piece of that group (ascending pass order), so [wait B] simply leaves one piece fewer in flight.
    static_assert(B0R % 64 == 0, "a B pass must not straddle the sub-tiles (it would be restaged one phase after its sub-1 rows were read)");
    constexpr int N_STRADDLE = (A0R % 64) ? 1 : 0;
    constexpr bool A_PARTIAL = 64 * PA > BM5;
    const bool skips_one = A_PARTIAL && 64 * (PA - 1) + wave_u * 8 >= BM5;
#define SVDX_V5_WAIT(LESS)                                                                                                                  \
    {                                                                                                                                       \
        if (A_PARTIAL && skips_one) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB - 1 - (LESS)) : "memory");                             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB - (LESS)) : "memory");                                                        \
    }

    f32x4 acc[NF][MF];                                                // [n-block][m-block], transposed like gemm_v4_kernel's
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int frag0 = fr * ROWB + ((fg ^ (fr & 7)) * 16), frag1 = fr * ROWB + (((4 + fg) ^ (fr & 7)) * 16);     // k-halves 0 / 1 of a fragment row
    const int a_lds0 = wr * (MF0 * 16) * ROWB, a_lds1 = (A0R + wr * (MF1 * 16)) * ROWB;
    const int b_lds0 = (BM5 + wc * (NF0 * 16)) * ROWB, b_lds1 = (BM5 + B0R + wc * (NF1 * 16)) * ROWB;
    v8 a0f[MF0][2], a1f[MF1 > 0 ? MF1 : 1][2], b0f[NF0][2], b1f[NF1 > 0 ? NF1 : 1][2];
#define SVDX_V5_READ(dst, n, ldsoff, buf)                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                                                    \
        dst[i_][0] = *reinterpret_cast<const v8*>(smem + (buf) * BUFB + (ldsoff) + i_ * 16 * ROWB + frag0);                                 \
        dst[i_][1] = *reinterpret_cast<const v8*>(smem + (buf) * BUFB + (ldsoff) + i_ * 16 * ROWB + frag1);                                 \
    }
#define SVDX_V5_MMA(bf_, nb_, ioff, af_, mb_, joff)                                                                                         \
    {                                                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                                      \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < (nb_); ++i_)                                                                           \
                _Pragma("unroll") for (int j_ = 0; j_ < (mb_); ++j_)                                                                       \
                    acc[(ioff) + i_][(joff) + j_] = TT<T>::mfma(bf_[i_][kk_], af_[j_][kk_], acc[(ioff) + i_][(joff) + j_]);                 \
        __builtin_amdgcn_s_setprio(0);                                                                                                      \
    }
#define SVDX_V5_BARRIER()                                                                                                                   \
    {                                                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                                       \
        asm volatile("" ::: "memory");                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                                  \
    }
    // one K-tile out of buffer `buf`; its slot-0 pieces go to the other buffer (K-tile t+1), slots 2 / 3 into this one (K-tile t+2)
#define SVDX_V5_TILE(buf)                                                                                                                   \
    {                                                                                                                                       \
        SVDX_V5_READ(b0f, NF0, b_lds0, buf);                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                                  \
        SVDX_V5_READ(a0f, MF0, a_lds0, buf);                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                                  \
        SVDX_V5_ISSUE(0, (buf) ^ 1);                                                                                                        \
        advance_a();                                                                                                                        \
        SVDX_V5_BARRIER();                                                                                                                  \
        SVDX_V5_MMA(b0f, NF0, 0, a0f, MF0, 0);                                                                                              \
        SVDX_V5_BARRIER();                                                                                                                  \
        if (NF1 > 0) {                                                                                                                      \
            SVDX_V5_READ(b1f, NF1, b_lds1, buf);                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                              \
        }                                                                                                                                   \
        SVDX_V5_WAIT(0);                                                                                                                    \
        SVDX_V5_BARRIER();                                                                                                                  \
        if (NF1 > 0) SVDX_V5_MMA(b1f, NF1, NF0, a0f, MF0, 0);                                                                               \
        SVDX_V5_BARRIER();                                                                                                                  \
        if (MF1 > 0) {                                                                                                                      \
            SVDX_V5_READ(a1f, MF1, a_lds1, buf);                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                              \
        }                                                                                                                                   \
        SVDX_V5_ISSUE(2, buf);                                                                                                              \
        SVDX_V5_BARRIER();                                                                                                                  \
        if (MF1 > 0 && NF1 > 0) SVDX_V5_MMA(b1f, NF1, NF0, a1f, MF1, MF0);                                                                  \
        SVDX_V5_BARRIER();                                                                                                                  \
        SVDX_V5_ISSUE(3, buf);                                                                                                              \
        advance_b();                                                                                                                        \
        SVDX_V5_WAIT(N_STRADDLE);                                                                                                                   \
        SVDX_V5_BARRIER();                                                                                                                  \
        if (MF1 > 0) SVDX_V5_MMA(b0f, NF0, 0, a1f, MF1, MF0);                                                                               \
        SVDX_V5_BARRIER();                                                                                                                  \
    }
    // ---- prologue: what the steady state would have issued before K-tile 0 (phases 2, 3 of "tile -2"; 0, 2, 3 of "tile -1") ----
    SVDX_V5_ISSUE(2, 0);
    SVDX_V5_ISSUE(3, 0);
    advance_b();
    SVDX_V5_ISSUE(0, 0);
    advance_a();
    SVDX_V5_ISSUE(2, 1);
    SVDX_V5_ISSUE(3, 1);
    advance_b();
    SVDX_V5_WAIT(N_STRADDLE);
    SVDX_V5_BARRIER();
    if (wr == 1) SVDX_V5_BARRIER();                                  // waves 4-7 run one barrier behind their SIMD partners from here on
    for (int t = 0; t < n_tiles; t += 2) {
        SVDX_V5_TILE(0);
        if (t + 1 < n_tiles) SVDX_V5_TILE(1);
    }
    if (wr == 0) SVDX_V5_BARRIER();                                  // ... and meet them again
#undef SVDX_V5_TILE
#undef SVDX_V5_WAIT
#undef SVDX_V5_BARRIER
#undef SVDX_V5_MMA
#undef SVDX_V5_READ
#undef SVDX_V5_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the out-of-range tail pieces: the epilogue parks its tile in these buffers
    __syncthreads();
    v4_epilogue<T, NF, MF, NT, BM5, BN5, WM5, WN5>(p, smem, acc, z, m0, n0, pid_m, pid_n, wr, wc, tid);
#endif
}

// ---- skinny linear: one wave per output column, lanes split K (trans = 0); MT rows of X per pass ------------------
template <typename T, int MT>
__device__ __forceinline__ void small_linear_nt_body(const float* __restrict__ X, const T* __restrict__ W, const float* __restrict__ bias,
                                                     float* Y, int M, int N, int K, int ldw, int silu_in, int accumulate, int blk) {
    const int lane = threadIdx.x & 63;
    const int n = blk * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T* w = W + (size_t)n * ldw;
    for (int mb = 0; mb < M; mb += MT) {
        float acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi] = 0.f;
        for (int k8 = lane; k8 * 8 < K; k8 += 64) {
            float wv[8];
            load8<T>(w + k8 * 8, wv);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                if (mb + mi < M) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(X + (size_t)(mb + mi) * K + k8 * 8);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(X + (size_t)(mb + mi) * K + k8 * 8 + 4);
                    float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += (silu_in ? siluf_(xv[j]) : xv[j]) * wv[j];
                    acc[mi] += s;
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const float s = wave_sum(acc[mi]);
            if (lane == 0 && mb + mi < M) {
                float* y = Y + (size_t)(mb + mi) * N + n;
                const float r = s + (bias ? bias[n] : 0.f);
                *y = accumulate ? *y + r : r;
            }
        }
    }
}

template <typename T, int MT>
__global__ __launch_bounds__(256) void small_linear_nt(const float* __restrict__ X, const T* __restrict__ W, const float* __restrict__ bias,
                                                       float* Y, int M, int N, int K, int ldw, int silu_in, int accumulate) {
    small_linear_nt_body<T, MT>(X, W, bias, Y, M, N, K, ldw, silu_in, accumulate, blockIdx.x);
}

// trans = 1: Y[m,k] (+)= sum_n X[m,n] W[n,k].  The op is a GEMV over a matrix of a few MB: latency-shaped.  A block owns 64
// output columns (8 chunks of 8) x 32 row lanes; a thread walks the rows n = nl, nl + 32, ... with four 16-byte loads in flight and
// the 32 partial sums of a column are then added in a fixed order -- no atomics, so the result is run-to-run identical (the
// gradient of the cross-attention value path depends on it).
template <typename T>
__device__ __forceinline__ void small_linear_nn_body(const float* X, const T* W, float* Y, int M, int N, int K, int ldw, int accumulate,
                                                     int blk, float (*part)[65]) {
    const int kc = threadIdx.x & 7, nl = threadIdx.x >> 3;
    const int k8 = blk * 8 + kc;
    const int m = blockIdx.y;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k8 * 8 < K) {
        const float* x = X + (size_t)m * N;
        const T* wp = W + k8 * 8;
        int n = nl;
        for (; n + 96 < N; n += 128) {
            const Vec8<T> w0 = *reinterpret_cast<const Vec8<T>*>(wp + (size_t)n * ldw);
            const Vec8<T> w1 = *reinterpret_cast<const Vec8<T>*>(wp + (size_t)(n + 32) * ldw);
            const Vec8<T> w2 = *reinterpret_cast<const Vec8<T>*>(wp + (size_t)(n + 64) * ldw);
            const Vec8<T> w3 = *reinterpret_cast<const Vec8<T>*>(wp + (size_t)(n + 96) * ldw);
            const float x0 = x[n], x1 = x[n + 32], x2 = x[n + 64], x3 = x[n + 96];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] += x0 * to_f<T>(w0.v[j]);
                acc[j] += x1 * to_f<T>(w1.v[j]);
                acc[j] += x2 * to_f<T>(w2.v[j]);
                acc[j] += x3 * to_f<T>(w3.v[j]);
            }
        }
        for (; n < N; n += 32) {
            const Vec8<T> w0 = *reinterpret_cast<const Vec8<T>*>(wp + (size_t)n * ldw);
            const float x0 = x[n];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += x0 * to_f<T>(w0.v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[nl][kc * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int k = blk * 64 + threadIdx.x;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) sum += part[r][threadIdx.x];
        if (k < K) {
            float* y = Y + (size_t)m * K + k;
            *y = accumulate ? *y + sum : sum;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void small_linear_nn(const float* X, const T* W, float* Y, int M, int N, int K, int ldw, int accumulate) {
    __shared__ float part[32][65];
    small_linear_nn_body<T>(X, W, Y, M, N, K, ldw, accumulate, blockIdx.x, part);
}

// ---- the same two kernels over a PACK of jobs in one launch: the per-clip vector chain of the KV-length-1 cross-attention is ~5 us of
// launch latency per linear and there are 64 of them per forward sweep (v and out of 32 blocks).  The job table travels in the kernel
// arguments (graph-capturable, no device-side table to keep alive); a workgroup finds its job by scanning the block prefix.
struct LinPack {
    svdx_lin_job job[SVDX_BATCH_MAX_JOBS];
    int start[SVDX_BATCH_MAX_JOBS + 1];        // first workgroup of job j; start[n_jobs] = grid size
    int n_jobs;
};

__device__ __forceinline__ int pack_find(const int* start, int n_jobs, int blk) {
    int j = 0;
    while (j + 1 < n_jobs && blk >= start[j + 1]) ++j;
    return j;
}

template <typename T, int MT>
__global__ __launch_bounds__(256) void small_linear_nt_batch(const LinPack pk, int M) {
    const int j = pack_find(pk.start, pk.n_jobs, blockIdx.x);
    const svdx_lin_job& q = pk.job[j];
    small_linear_nt_body<T, MT>(q.X, (const T*)q.W, q.bias, q.Y, M, q.N, q.K, q.ldw, q.flags & 1, (q.flags >> 1) & 1, blockIdx.x - pk.start[j]);
}

template <typename T>
__global__ __launch_bounds__(256) void small_linear_nn_batch(const LinPack pk, int M) {
    __shared__ float part[32][65];
    const int j = pack_find(pk.start, pk.n_jobs, blockIdx.x);
    const svdx_lin_job& q = pk.job[j];
    small_linear_nn_body<T>(q.X, (const T*)q.W, q.Y, M, q.N, q.K, q.ldw, (q.flags >> 1) & 1, blockIdx.x - pk.start[j], part);
}

// split-K epilogue: v = sum over `nsplit` float slabs (+ bias + rowvec + res);  C = (dtype)v, or Cf += v (weight grads)
// GN: also the GroupNorm statistics of the activation tensor it writes (see svdx_gemm_gn): a block's 256 pieces of 4 consecutive
// channels are 1024 consecutive elements of C -- at most two samples -- whose (sample, group) sums meet in a small LDS table of 64-bit
// fixed-point integers before they go to the replica slots, one atomic per touched entry.
struct FinGn { unsigned long long* stats; int rows, cg; float m0, m1; };
constexpr int FIN_GN_G = 64;
template <typename T, bool GN>
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const float* __restrict__ acc, int nsplit, long slab_stride, T* __restrict__ C,
                                                            float* __restrict__ Cf, int f32_store, int M, int N, int ldc,
                                                            const float* __restrict__ bias, const float* __restrict__ rowvec,
                                                            int rv_ld, int rv_rpg, int rv_mod, const T* __restrict__ res, int ldres,
                                                            const float* __restrict__ cs_slabs, float* cs_out, int cs_n, FinGn gn) {
    __shared__ unsigned long long gacc[GN ? 2 * FIN_GN_G * 2 : 1];
    if (cs_slabs) {                            // bias gradient: the row slices' column sums, added in slice order (spread over the grid)
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < cs_n; n += gridDim.x * blockDim.x) {
            float t = 0.f;
            for (int z = 0; z < nsplit; ++z) t += cs_slabs[(size_t)z * cs_n + n];
            cs_out[n] += t;
        }
    }
    const long total4 = (long)M * N / 4;       // N % 4 == 0
    for (long b4 = (long)blockIdx.x * blockDim.x; b4 < total4; b4 += (long)gridDim.x * blockDim.x) {
        const long i4 = b4 + threadIdx.x;
        if (GN) {
            for (int t = threadIdx.x; t < 2 * FIN_GN_G * 2; t += 256) gacc[t] = 0ull;
            __syncthreads();
        }
        if (i4 < total4) {
            const long i = i4 * 4;
            const int m = (int)(i / N), n = (int)(i - (long)m * N);
            f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(acc + i));       // slabs: written once, read once
            for (int z = 1; z < nsplit; ++z) v += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(acc + (size_t)z * slab_stride + i));
            if (bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias[n + e];
            }
            if (rowvec) {
                const float* rv = rowvec + (size_t)(rv_mod ? m % rv_mod : m / rv_rpg) * rv_ld + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
            if (res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += to_f<T>(res[(size_t)m * ldres + n + e]);
            }
            if (Cf) {
                float* o = Cf + (size_t)m * ldc + n;
                if (f32_store) {             // a weight gradient: next read by the optimizer, a whole backward sweep later
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += v[e];
                }
            } else {
                Vec4<T> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(v[e]);
                *reinterpret_cast<Vec4<T>*>(C + (size_t)m * ldc + n) = o;
                if (GN) {
                    const int sl = m / gn.rows - (int)((b4 * 4) / N) / gn.rows;      // 0 or 1 (host: N * rows >= 1024)
                    int cur = n / gn.cg;
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int g = (n + e) / gn.cg;
                        if (g != cur) {
                            atomicAdd(&gacc[(sl * FIN_GN_G + cur) * 2], (unsigned long long)__float2ll_rn(s0 * gn.m0));
                            atomicAdd(&gacc[(sl * FIN_GN_G + cur) * 2 + 1], (unsigned long long)__float2ll_rn(s1 * gn.m1));
                            cur = g; s0 = 0.f; s1 = 0.f;
                        }
                        const float x = to_f<T>(o.v[e]);
                        s0 += x; s1 += x * x;
                    }
                    atomicAdd(&gacc[(sl * FIN_GN_G + cur) * 2], (unsigned long long)__float2ll_rn(s0 * gn.m0));
                    atomicAdd(&gacc[(sl * FIN_GN_G + cur) * 2 + 1], (unsigned long long)__float2ll_rn(s1 * gn.m1));
                }
            }
        }
        if (GN) {
            __syncthreads();
            const int G = N / gn.cg, n_s = M / gn.rows;
            const int s_first = (int)((b4 * 4) / N) / gn.rows;              // sample of the block's first element
            unsigned long long* out = gn.stats + (size_t)(blockIdx.x % SVDX_GN_REPLICAS) * n_s * G * 2;
            for (int t = threadIdx.x; t < 2 * G * 2; t += 256) {
                const int w = t & 1, g = (t >> 1) % G, s2 = (t >> 1) / G;
                const unsigned long long a = gacc[(s2 * FIN_GN_G + g) * 2 + w];
                if (a && s_first + s2 < n_s) atomicAdd(out + ((size_t)(s_first + s2) * G + g) * 2 + w, a);
            }
            __syncthreads();
        }
    }
}

// ---- the float form of that epilogue over a PACK of jobs: the weight gradients of a backward sweep are not read before the optimizer,
// so the reducing launches of their row-sliced TN GEMMs (96 per step at c2, a few us of launch latency each for a few hundred KB) wait
// and run as one launch per 48 -- the same arithmetic in the same order per element (slice 0 first), so the bits do not change.
struct GradFinPack {
    svdx_gradfin_job job[SVDX_BATCH_MAX_JOBS];
    int start[SVDX_BATCH_MAX_JOBS + 1];
    int n_jobs;
};
__global__ __launch_bounds__(256) void grad_finalize_batch_kernel(const GradFinPack pk) {
    const int j = pack_find(pk.start, pk.n_jobs, blockIdx.x);
    const svdx_gradfin_job& q = pk.job[j];
    const int blk = blockIdx.x - pk.start[j], nblk = pk.start[j + 1] - pk.start[j];
    if (q.colsum_slabs) {                      // bias gradient: the row slices' column sums, added in slice order
        for (int n = blk * 256 + threadIdx.x; n < q.colsum_n; n += nblk * 256) {
            float t = 0.f;
            for (int z = 0; z < q.nsplit; ++z) t += q.colsum_slabs[(size_t)z * q.colsum_n + n];
            q.colsum_out[n] += t;
        }
    }
    const long total4 = q.count / 4;
    bool bad_any = false;
    for (long i4 = (long)blk * 256 + threadIdx.x; i4 < total4; i4 += (long)nblk * 256) {
        const long i = i4 * 4;
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q.acc + i));
        for (int z = 1; z < q.nsplit; ++z) v += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q.acc + (size_t)z * q.slab_stride + i));
        f32x4* o = reinterpret_cast<f32x4*>(q.dst + i);
        if (q.store) __builtin_nontemporal_store(v, o);
        else { v = *o + v; *o = v; }
        bad_any |= not_finite(v[0]) || not_finite(v[1]) || not_finite(v[2]) || not_finite(v[3]);
    }
    raise_found_inf(q.found_inf, bad_any);
}

// X == nullptr: a column of ones (the bias gradient, K = 1)
__device__ __forceinline__ void outer_acc_body(const float* dY, const float* X, float* dW, int M, int N, int K, float scale, size_t blk) {
    const size_t idx = blk * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (size_t)n * K);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dY[(size_t)m * N + n] * (X ? X[(size_t)m * K + k] : 1.f);
    dW[idx] += scale * s;
}

__global__ void outer_acc_kernel(const float* dY, const float* X, float* dW, int M, int N, int K, float scale) {
    outer_acc_body(dY, X, dW, M, N, K, scale, blockIdx.x);
}

struct OuterPack {
    svdx_outer_job job[SVDX_BATCH_MAX_JOBS];
    int start[SVDX_BATCH_MAX_JOBS + 1];
    int n_jobs;
};

__global__ void outer_acc_batch_kernel(const OuterPack pk, int M) {
    const int j = pack_find(pk.start, pk.n_jobs, blockIdx.x);
    const svdx_outer_job& q = pk.job[j];
    outer_acc_body(q.dY, q.X, q.dW, M, q.N, q.K, q.scale, (size_t)(blockIdx.x - pk.start[j]));
}

__global__ void timestep_embed_kernel(const float* t, float* out, int n, int dim) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int i = idx / half, j = idx - i * half;
    const float f = expf(-9.210340371976184f * (float)j / (float)half);   // ln(10000)
    const float a = t[i] * f;
    out[(size_t)i * dim + j] = cosf(a);
    out[(size_t)i * dim + half + j] = sinf(a);
}

template <typename T>
int launch_gemm(const GemmParams& p, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        attr_set = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, p.split_k);
    hipLaunchKernelGGL((gemm_kernel<T>), grid, dim3(NTHREADS), 2 * STAGE_BYTES, st, p);
    SVDX_LAUNCH_CHECK("svdx_gemm");
    return 0;
}

// Tile counts, vector-store eligibility and the XCD arrangement of a BMT x BNT tile grid (see v4_tile_of_block): the arrangement with the
// least operand re-fetch among those that keep >= 90 % of the best tile balance; -> the launch grid.  Shared by the v4 and v5 launchers.
static int arrange_nt_grid(GemmParams& p, int BMT, int BNT, dim3& grid, int rows_computed = 0) {      // BMT: rows a tile owns; rows_computed: its height when larger
    if (p.gn_stats) {
        // the statistics are taken in the coalesced store loop, which handles whole column tiles of 16-byte-aligned rows only
        const bool ok = p.N % BNT == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && (!p.res || (p.ldres % 8 == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                        ((rows_computed ? rows_computed : BMT) - 1) / p.gn_rows + 2 <= GN_MAX_S && (BNT - 1) / p.gn_cg + 2 <= GN_MAX_G;
        if (!ok) { svdx_set_error("svdx_gemm_gn: N=%d / ldc=%d / rows=%d / cg=%d do not fit the %dx%d tile's statistics path", p.N, p.ldc, p.gn_rows, p.gn_cg, BMT, BNT); return -2; }
    }
    p.tiles_m = cdiv(p.M, BMT);
    p.tiles_n = p.epi == SVDX_EPI_GEGLU_FWD ? cdiv(p.aux_dim, BNT / 2) : cdiv(p.N, BNT);
    p.vec_ok = (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && (!p.res || (p.ldres % 4 == 0 && ((uintptr_t)p.res & 15) == 0));
    const double a_bytes = 2.0 * p.M * (p.g.mode == SVDX_GATHER_PLAIN ? p.K : 2 * p.g.cin);   // conv: unique rows + halo
    const double b_bytes = 2.0 * (p.epi == SVDX_EPI_GEGLU_FWD ? 2 * p.aux_dim : p.N) * p.K;
    int best_xn = 0;
    double best_cost = 0, best_eff = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int xn = 1; xn <= 8; xn *= 2) {
            const int xm = 8 / xn, sm = cdiv(p.tiles_m, xm), sn = cdiv(p.tiles_n, xn);
            const double eff = (double)p.tiles_m * p.tiles_n / (8.0 * sm * sn);
            if (pass == 0) { best_eff = eff > best_eff ? eff : best_eff; continue; }
            const double cost = a_bytes * xn + b_bytes * xm;
            if (eff >= 0.9 * best_eff && (best_xn == 0 || cost < best_cost)) { best_xn = xn; best_cost = cost; }
        }
    p.xcd_n = best_xn;
    p.z_xcd = 0;
    int gx = p.tiles_m * p.tiles_n, gy = p.split_k;
    if (best_xn > 0) {
        p.sub_m = cdiv(p.tiles_m, 8 / best_xn);
        p.sub_n = cdiv(p.tiles_n, best_xn);
        gx = 8 * p.sub_m * p.sub_n;
    }
    // K slices on XCDs of their own (see v4_tile_of_block): among the arrangements of the 8 / split_k XCDs of a slice, the least re-fetch that keeps
    // >= 90 % of the best balance; taken when it fetches less than the slice-agnostic arrangement and leaves no more workgroup slots empty
    if (best_xn > 0 && (p.split_k == 2 || p.split_k == 4 || p.split_k == 8)) {
        const int xps = 8 / p.split_k;
        int zb_xn = 0; double zb_cost = 0, zb_eff = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int xn = 1; xn <= xps; xn *= 2) {
                const int xm = xps / xn, sm = cdiv(p.tiles_m, xm), sn = cdiv(p.tiles_n, xn);
                const double eff = (double)p.tiles_m * p.tiles_n / ((double)xps * sm * sn);
                if (pass == 0) { zb_eff = eff > zb_eff ? eff : zb_eff; continue; }
                const double cost = a_bytes * xn + b_bytes * xm;
                if (eff >= 0.9 * zb_eff && (zb_xn == 0 || cost < zb_cost)) { zb_xn = xn; zb_cost = cost; }
            }
        const int zsm = cdiv(p.tiles_m, xps / zb_xn), zsn = cdiv(p.tiles_n, zb_xn);
        if (zb_cost < best_cost && zsm * zsn <= p.sub_m * p.sub_n * p.split_k) {
            p.z_xcd = 1; p.xcd_n = zb_xn; p.sub_m = zsm; p.sub_n = zsn;
            gx = 8 * zsm * zsn; gy = 1;
        }
    }
    grid = dim3(gx, gy);
    return 0;
}

template <typename T, int NB, int MB, int WGM = 2, int NSTG = 2, int MSTEP = 0>
int launch_gemm_v4(GemmParams p, hipStream_t st) {
    constexpr int BMT = 16 * MB * WGM, BNT = 32 * NB;
    constexpr int STEP = MSTEP ? MSTEP : BMT;                    // rows a tile owns (variant 36: 140 of its 144)
    static_assert(STEP <= BMT && STEP > 0, "a tile cannot own more rows than it computes");
    constexpr int LDS_STG = NSTG * (BMT + BNT) * BK * 2, LDS_EPI = ((BMT * (BNT + 8) * 2 + 15) & ~15) + GN_LDS;      // K-loop stages | the rounded output tile parked for the coalesced stores (+ the GroupNorm statistics table behind it)
    constexpr int LDS = LDS_STG > LDS_EPI ? LDS_STG : LDS_EPI;
    static_assert(LDS <= 160 * 1024, "stages (and the epilogue tile parked in them) must fit the 160 KiB LDS");
    constexpr bool HAS_DUAL = WGM == 2 && NSTG == 2;            // the LoRA second-operand loop is only instantiated for the round-1 tiles
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v4_kernel<T, NB, MB, false, WGM, NSTG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (HAS_DUAL)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v4_kernel<T, NB, MB, HAS_DUAL, WGM, NSTG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if (p.K2 > 0 && !HAS_DUAL) { svdx_set_error("svdx_gemm_dual: this tile variant has no second-operand loop"); return -2; }
    dim3 grid;
    p.m_step = STEP;
    if (int rc = arrange_nt_grid(p, STEP, BNT, grid, BMT)) return rc;
    if (p.K2 > 0) hipLaunchKernelGGL((gemm_v4_kernel<T, NB, MB, HAS_DUAL, WGM, NSTG>), grid, dim3(128 * WGM), LDS, st, p);
    else hipLaunchKernelGGL((gemm_v4_kernel<T, NB, MB, false, WGM, NSTG>), grid, dim3(128 * WGM), LDS, st, p);
    SVDX_LAUNCH_CHECK("svdx_gemm");
    return 0;
}

// the two-role eight-wave tiles (gemm_v5_kernel): (32 MF) x (64 NF), two K-tiles of LDS
template <typename T, int MF, int NF>
int launch_gemm_v5(GemmParams p, hipStream_t st) {
    constexpr int BMT = 32 * MF, BNT = 64 * NF;
    constexpr int LDS_STG = 2 * (BMT + BNT) * BK * 2, LDS_EPI = ((BMT * (BNT + 8) * 2 + 15) & ~15) + GN_LDS;
    constexpr int LDS = LDS_STG > LDS_EPI ? LDS_STG : LDS_EPI;
    static_assert(LDS <= 160 * 1024, "K-tiles (and the epilogue tile parked in them) must fit the 160 KiB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v5_kernel<T, MF, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    if (p.K2 > 0) { svdx_set_error("svdx_gemm_dual: this tile variant has no second-operand loop"); return -2; }
    dim3 grid;
    p.m_step = BMT;
    if (int rc = arrange_nt_grid(p, BMT, BNT, grid)) return rc;
    hipLaunchKernelGGL((gemm_v5_kernel<T, MF, NF>), grid, dim3(512), LDS, st, p);
    SVDX_LAUNCH_CHECK("svdx_gemm");
    return 0;
}

// grid of the TN kernels (tn_who): fills p.xcd_n / sub_m / sub_n, returns the number of workgroups.  The host side asks for 1, 2, 4 or a
// multiple of 8 row slices (ops._tn_formula); any other count runs with some XCDs idle.
static long tn_arrange(GemmParams& p) {
    if (p.split_k >= 8 || 8 % p.split_k) {
        p.xcd_n = 1; p.sub_m = p.tiles_m; p.sub_n = p.tiles_n;
        return 8L * cdiv(p.split_k, 8) * p.sub_m * p.sub_n;
    }
    const int xps = 8 / p.split_k;
    const double a_bytes = 2.0 * p.K * p.M, b_bytes = 2.0 * p.K * p.N;          // dY [R, N_out] and X [R, K_out] of one call (p.M / p.N = output rows / columns)
    int best_xn = 1;
    double best_cost = 0, best_eff = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int xn = 1; xn <= xps; xn *= 2) {
            const int xm = xps / xn, sm = cdiv(p.tiles_m, xm), sn = cdiv(p.tiles_n, xn);
            const double eff = (double)p.tiles_m * p.tiles_n / ((double)xps * sm * sn);
            if (pass == 0) { best_eff = eff > best_eff ? eff : best_eff; continue; }
            const double cost = a_bytes * xn + b_bytes * xm;                     // every XCD column re-fetches dY, every XCD row re-fetches X
            if (eff >= 0.9 * best_eff && (best_cost == 0 || cost < best_cost)) { best_xn = xn; best_cost = cost; }
        }
    p.xcd_n = best_xn;
    p.sub_m = cdiv(p.tiles_m, xps / best_xn);
    p.sub_n = cdiv(p.tiles_n, best_xn);
    return 8L * p.sub_m * p.sub_n;
}

template <typename T, int NSTG, bool BUF = false>
int launch_gemm_tn(GemmParams p, hipStream_t st) {
    constexpr int LDS = NSTG * STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T, NSTG, BUF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    dim3 grid((unsigned)tn_arrange(p));
    hipLaunchKernelGGL((gemm_tn_kernel<T, NSTG, BUF>), grid, dim3(NTHREADS), LDS, st, p);
    SVDX_LAUNCH_CHECK("svdx_gemm_tn");
    return 0;
}

template <typename T, int PA, int PB, int NSTG, bool BUF = false>
int launch_gemm_tn8(GemmParams p, hipStream_t st) {
    constexpr int LDS = NSTG * (PA + PB) * 64 * 256;
    static_assert(LDS <= 160 * 1024, "stages must fit the 160 KiB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn8_kernel<T, PA, PB, NSTG, BUF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    p.tiles_m = cdiv(p.M, 128 * PA);
    p.tiles_n = cdiv(p.N, 128 * PB);
    dim3 grid((unsigned)tn_arrange(p));
    hipLaunchKernelGGL((gemm_tn8_kernel<T, PA, PB, NSTG, BUF>), grid, dim3(512), LDS, st, p);
    SVDX_LAUNCH_CHECK("svdx_gemm_tn");
    return 0;
}

}  // namespace

extern "C" int svdx_gemm_tn(const void* A, const void* B, float* C, int R, int N, int K, int lda, int ldb, int ldc,
                            float* a_colsum, const void* zero_page, int out_mode, int split_k, int stages, float* found_inf, int dtype,
                            void* stream) {
    SVDX_CHECK_ARG(A && B && C && zero_page && R > 0 && N > 0 && K > 0, "svdx_gemm_tn: bad args");
    SVDX_CHECK_ARG(!found_inf || out_mode == SVDX_OUT_F32 || out_mode == SVDX_OUT_F32_ADD, "svdx_gemm_tn: found_inf goes with the store / += modes");
    // operands below 2 GiB are staged through buffer descriptors (SVDX_TN_FLAT: a test asks for the large-operand path on a small operand)
    const long a_span = ((long)(R - 1) * lda + N) * 2, b_span = ((long)(R - 1) * ldb + K) * 2;
    // (the kernels mark columns beyond the operand with the offset 0x7ffffff0, which must itself lie beyond num_records: spans up to that value only)
    const bool buf = a_span <= 0x7ffffff0L && b_span <= 0x7ffffff0L && !(stages & SVDX_TN_FLAT);
    stages &= ~SVDX_TN_FLAT;
    SVDX_CHECK_ARG(stages == 0 || (stages >= 2 && stages <= 4) || stages == 18,
                   "svdx_gemm_tn: stages=%d (0 = default, 2..4 stages of the 128x128 tile, 18 = the 256x256 eight-wave tile)", stages);
    // the unsplit / ADD modes read-modify-write a_colsum from every z slice: one slice only (SVDX_OUT_F32_SLAB keeps a row per slice)
    SVDX_CHECK_ARG(!a_colsum || split_k == 1 || out_mode == SVDX_OUT_F32_SLAB, "svdx_gemm_tn: a_colsum with split_k > 1 needs slab output");
    SVDX_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
                       ((uintptr_t)zero_page & 15) == 0, "svdx_gemm_tn: operands must be 16-byte aligned, N/K multiples of 8");
    SVDX_CHECK_ARG(out_mode == SVDX_OUT_F32 || out_mode == SVDX_OUT_F32_ADD || out_mode == SVDX_OUT_F32_SLAB ||
                       out_mode == SVDX_OUT_F32_ATOMIC, "svdx_gemm_tn: float output modes only");
    SVDX_CHECK_ARG(split_k >= 1 && (split_k == 1 || out_mode == SVDX_OUT_F32_SLAB || out_mode == SVDX_OUT_F32_ATOMIC),
                   "svdx_gemm_tn: split_k needs slab or atomic output");
    SVDX_CHECK_ARG(out_mode != SVDX_OUT_F32_SLAB || (split_k - 1) * cdiv(cdiv(R, BK), split_k) < cdiv(R, BK),
                   "svdx_gemm_tn: split_k=%d leaves slices without rows (R=%d): their slabs would stay unwritten", split_k, R);
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.M = N; p.N = K; p.K = R; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = nullptr; p.rowvec = nullptr; p.rv_ld = 0; p.rv_rpg = 0; p.rv_mod = 0; p.res = nullptr; p.ldres = 0;
    p.g = svdx_gather{}; p.zero_page = zero_page; p.out_mode = out_mode; p.alpha = 1.f; p.split_k = split_k;
    p.epi = 0; p.aux_in = nullptr; p.aux_out = nullptr; p.aux_dim = 0; p.a_colsum = a_colsum; p.found_inf = found_inf;
    p.a_bytes = buf ? (int)a_span : 0; p.b_bytes = buf ? (int)b_span : 0;
    p.gn_stats = nullptr; p.gn_rows = p.gn_cg = 0; p.gn_m0 = p.gn_m1 = 0.f;
    p.tiles_m = cdiv(N, BM); p.tiles_n = cdiv(K, BN);
    p.vec_ok = (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0) && (K % 8 == 0);
    p.slab_stride = (long)N * ldc;
    DISPATCH_DTYPE(dtype, {
        hipStream_t st = (hipStream_t)stream;
        if (stages == 18) return buf ? launch_gemm_tn8<T, 2, 2, 2, true>(p, st) : launch_gemm_tn8<T, 2, 2, 2>(p, st);
        if (stages == 3) return buf ? launch_gemm_tn<T, 3, true>(p, st) : launch_gemm_tn<T, 3>(p, st);
        if (stages == 4) return buf ? launch_gemm_tn<T, 4, true>(p, st) : launch_gemm_tn<T, 4>(p, st);
        return buf ? launch_gemm_tn<T, 2, true>(p, st) : launch_gemm_tn<T, 2>(p, st);
    });
}

static int gemm_entry(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                      const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                      int out_mode, float alpha, int split_k, int variant, int epilogue, const void* aux_in, void* aux_out,
                      int aux_dim, const void* A2, const void* B2, int K2, int lda2, int ldb2, int a2_seg, float* gn_stats, int gn_rows,
                      int gn_cg, int dtype, void* stream) {
    SVDX_CHECK_ARG(A && B && C, "svdx_gemm: null operand");
    if (gn_stats) {
        SVDX_CHECK_ARG(variant >= 2 && out_mode == SVDX_OUT_ACT && split_k == 1 && epilogue == SVDX_EPI_NONE && K2 <= 0,
                       "svdx_gemm_gn: statistics ride on an unsplit variant-4 launch with activation output and no fused epilogue");
        SVDX_CHECK_ARG(gn_rows > 0 && gn_cg > 0 && M % gn_rows == 0 && N % gn_cg == 0 && ((uintptr_t)gn_stats & 7) == 0,
                       "svdx_gemm_gn: M=%d must be whole samples of %d rows, N=%d whole groups of %d channels", M, gn_rows, N, gn_cg);
    }
    if (K2 > 0) {
        SVDX_CHECK_ARG(A2 && B2 && K2 % BK == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && lda2 >= K2 && ldb2 >= K2 &&
                           (((uintptr_t)A2 | (uintptr_t)B2) & 15) == 0, "svdx_gemm_dual: second operand pair misaligned (K2=%d)", K2);
        SVDX_CHECK_ARG(variant >= 2 && split_k == 1, "svdx_gemm_dual: needs variant 4 and split_k == 1");
        const int bn = (variant != 8 && N % 160 == 0) ? 160 : 128;           // tile width the dispatch below will pick
        SVDX_CHECK_ARG(a2_seg == 0 || (a2_seg > 0 && N % a2_seg == 0 && a2_seg % bn == 0 && lda2 >= (N / a2_seg) * K2),
                       "svdx_gemm_dual: a2_seg_n=%d must divide N=%d, be a multiple of the %d-column tile, and fit lda2", a2_seg, N, bn);
    }
    if (epilogue != SVDX_EPI_NONE) {
        SVDX_CHECK_ARG(variant >= 2 && out_mode == SVDX_OUT_ACT && split_k == 1 && !res && !rowvec && aux_dim > 0 && aux_dim % 64 == 0 &&
                           (!gather || gather->mode == SVDX_GATHER_PLAIN), "svdx_gemm: fused GEGLU epilogue needs variant 4, plain A, no split-K");
        SVDX_CHECK_ARG(((uintptr_t)C & 15) == 0, "svdx_gemm: fused epilogue output must be 16-byte aligned");
        if (epilogue == SVDX_EPI_GEGLU_FWD)
            SVDX_CHECK_ARG(N == 2 * aux_dim && ldc == N && aux_out && ((uintptr_t)aux_out & 15) == 0, "svdx_gemm: GEGLU fwd wants C=[M,2F], aux_out=[M,F]");
        else
            SVDX_CHECK_ARG(epilogue == SVDX_EPI_GEGLU_BWD && N == aux_dim && ldc == 2 * aux_dim && aux_in && ((uintptr_t)aux_in & 15) == 0 &&
                               (N % 160 == 0 || N % 128 == 0), "svdx_gemm: GEGLU bwd wants N=F (multiple of 128 or 160), C=dpre [M,2F], aux_in=pre [M,2F]");
    }
    SVDX_CHECK_ARG(M > 0 && N > 0 && K > 0, "svdx_gemm: bad sizes M=%d N=%d K=%d", M, N, K);
    SVDX_CHECK_ARG(K % BK == 0, "svdx_gemm: K=%d must be a multiple of %d", K, BK);
    SVDX_CHECK_ARG(ldb % 8 == 0 && ((uintptr_t)B & 15) == 0, "svdx_gemm: B must be 16-byte aligned (ldb=%d)", ldb);
    SVDX_CHECK_ARG(split_k >= 1 && (split_k == 1 || out_mode == SVDX_OUT_F32_ATOMIC || out_mode == SVDX_OUT_F32_SLAB),
                   "svdx_gemm: split_k=%d needs atomic or slab output", split_k);
    SVDX_CHECK_ARG(!rowvec || rv_mod > 0 || rv_rows_per_group > 0, "svdx_gemm: rowvec needs a grouping");
    // a slice without K-tiles would leave its slab unwritten (found by tests/sim/fuzz.py with NaN-filled slabs; ops.choose_cfg never asks for one)
    SVDX_CHECK_ARG(out_mode != SVDX_OUT_F32_SLAB || (split_k - 1) * cdiv(K / BK, split_k) < K / BK,
                   "svdx_gemm: split_k=%d leaves slices without K-tiles (K=%d): their slabs would stay unwritten", split_k, K);
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.rowvec = rowvec; p.rv_ld = rv_ld; p.rv_rpg = rv_rows_per_group; p.rv_mod = rv_mod;
    p.res = res; p.ldres = ldres; p.zero_page = zero_page;
    p.out_mode = out_mode; p.alpha = alpha; p.split_k = split_k; p.slab_stride = (long)M * ldc;
    p.epi = epilogue; p.aux_in = aux_in; p.aux_out = aux_out; p.aux_dim = aux_dim; p.a_colsum = nullptr; p.found_inf = nullptr;
    p.A2 = A2; p.B2 = B2; p.K2 = K2 > 0 ? K2 : 0; p.lda2 = lda2; p.ldb2 = ldb2; p.a2_bytes = p.b2_bytes = 0; p.a2_seg = K2 > 0 ? a2_seg : 0;
    p.gn_stats = reinterpret_cast<unsigned long long*>(gn_stats); p.gn_rows = gn_rows; p.gn_cg = gn_cg; p.gn_m0 = p.gn_m1 = 0.f;
    if (gn_stats) {
        int k0, k1;
        gn_fixed_scales((long)gn_rows * gn_cg, 0, k0, k1);
        p.gn_m0 = exp2f((float)k0); p.gn_m1 = exp2f((float)k1);
    }
    if (gather && gather->mode != SVDX_GATHER_PLAIN) {
        p.g = *gather;
        SVDX_CHECK_ARG(p.g.cin % BK == 0, "svdx_gemm: gather cin=%d must be a multiple of %d", p.g.cin, BK);
        SVDX_CHECK_ARG(p.g.lda % 8 == 0 && ((uintptr_t)A & 15) == 0, "svdx_gemm: gather source misaligned");
        SVDX_CHECK_ARG(zero_page && ((uintptr_t)zero_page & 15) == 0, "svdx_gemm: gather needs an aligned zero page");
        const int taps = p.g.mode == SVDX_GATHER_TEMPORAL3 ? 3 : 9;
        SVDX_CHECK_ARG(K == taps * p.g.cin, "svdx_gemm: K=%d != taps*cin=%d", K, taps * p.g.cin);
        if (p.g.mode == SVDX_GATHER_CONV3X3 || p.g.mode == SVDX_GATHER_CONV3X3_PAD0) {
            SVDX_CHECK_ARG(M == p.g.n_img * p.g.ho * p.g.wo, "svdx_gemm: conv rows mismatch");
            SVDX_CHECK_ARG(p.g.mode == SVDX_GATHER_CONV3X3 || (p.g.stride == 2 && !p.g.ups), "svdx_gemm: the pad-0 gather is the stride-2 downsample");
            SVDX_CHECK_ARG(p.g.stride == 1 || p.g.stride == 2, "svdx_gemm: conv stride");
            SVDX_CHECK_ARG(!p.g.ups || (p.g.hi % 2 == 0 && p.g.wi % 2 == 0), "svdx_gemm: upsampled dims must be even");
        } else if (p.g.mode == SVDX_GATHER_CONV3X3_DGRAD2) {
            SVDX_CHECK_ARG(M == p.g.n_img * p.g.ho * p.g.wo, "svdx_gemm: dgrad rows mismatch");
        } else if (p.g.mode == SVDX_GATHER_TEMPORAL3) {
            SVDX_CHECK_ARG(M == p.g.n_img * p.g.t * p.g.hw, "svdx_gemm: temporal rows mismatch");
        } else {
            svdx_set_error("svdx_gemm: unknown gather mode %d", p.g.mode);
            return -2;
        }
    } else {
        p.g = svdx_gather{};
        p.g.mode = SVDX_GATHER_PLAIN;
        SVDX_CHECK_ARG(lda % 8 == 0 && ((uintptr_t)A & 15) == 0, "svdx_gemm: A must be 16-byte aligned (lda=%d)", lda);
    }
    p.tiles_m = cdiv(M, BM);
    p.tiles_n = cdiv(N, BN);
    const int esz = out_mode == SVDX_OUT_ACT ? 2 : 4;
    p.vec_ok = (ldc % 8 == 0) && (((uintptr_t)C % (esz == 2 ? 16 : 4)) == 0) &&
               (!res || (ldres % 8 == 0 && ((uintptr_t)res & 15) == 0));
    hipStream_t st = (hipStream_t)stream;
    // extents of the A / B buffers for the bounds-checked buffer loads of variant 4 (must stay below 2 GiB)
    long a_rows = M;
    if (p.g.mode == SVDX_GATHER_CONV3X3 || p.g.mode == SVDX_GATHER_CONV3X3_PAD0) a_rows = (long)p.g.n_img * (p.g.hi >> p.g.ups) * (p.g.wi >> p.g.ups);
    else if (p.g.mode == SVDX_GATHER_CONV3X3_DGRAD2) a_rows = (long)p.g.n_img * p.g.hi * p.g.wi;
    else if (p.g.mode == SVDX_GATHER_TEMPORAL3) a_rows = (long)p.g.n_img * p.g.t * p.g.hw;
    const int a_ld = p.g.mode == SVDX_GATHER_PLAIN ? lda : p.g.lda;
    const int a_w = p.g.mode == SVDX_GATHER_PLAIN ? K : p.g.cin;
    long a_bytes = ((a_rows - 1) * a_ld + a_w) * 2, b_bytes = ((long)(N - 1) * ldb + K) * 2;
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) { a_bytes = 0; b_bytes = 0; }   // falls back to variant 1 (64-bit pointers)
    if (p.K2 > 0) {
        const long b_rows = epilogue == SVDX_EPI_GEGLU_FWD ? 2L * aux_dim : N;
        const long a2 = ((long)(M - 1) * lda2 + (long)(a2_seg > 0 ? N / a2_seg : 1) * K2) * 2, b2 = ((b_rows - 1) * ldb2 + K2) * 2;
        if (a_bytes == 0 || a2 >= (1L << 31) || b2 >= (1L << 31)) { svdx_set_error("svdx_gemm_dual: operands too large for the buffer-addressed kernel"); return -2; }
        p.a2_bytes = (int)a2; p.b2_bytes = (int)b2;
    }
    DISPATCH_DTYPE(dtype, {
        if (variant >= 2 && a_bytes > 0 && b_bytes > 0) {
            p.a_bytes = (int)a_bytes; p.b_bytes = (int)b_bytes;
            // tile choice: variant 4 = heuristic; 6 / 7 / 8 force 160x160 / 128x160 / 128x128 (the host autotuner times them)
            const bool nb5 = epilogue == SVDX_EPI_GEGLU_FWD || variant == 8 || variant == 17 || variant == 18 || variant == 21 || variant == 22 || variant == 24 || variant == 26 || variant == 27 ? false : (N % 160 == 0);
            // the GEGLU-backward epilogue only exists in the coalesced store path, which takes whole column tiles: d(pre) of a partial last
            // tile would be written as plain d(h) (found by tests/sim/fuzz.py; F = 4C of the UNet is always a multiple of 128)
            SVDX_CHECK_ARG(epilogue != SVDX_EPI_GEGLU_BWD || N % (nb5 ? 160 : 128) == 0,
                           "svdx_gemm: GEGLU bwd with tile variant %d needs N=%d to be a multiple of its %d-column tile", variant, N, nb5 ? 160 : 128);
            // 160-row tiles when they turn a 1.1-wave grid (512 resident blocks) into a single wave, e.g. M = 35840, N = 320:
            // 280 x 2 = 560 tiles of 128 rows vs 224 x 2 = 448 tiles of 160 rows
            const long t128 = (long)cdiv(M, 128) * cdiv(N, 160), t160 = (long)cdiv(M, 160) * cdiv(N, 160);
            const bool mb5 = nb5 && ((variant == 6) ||
                                     (variant == 4 && split_k == 1 && (cdiv(t128, 512) * 4 > cdiv(t160, 512) * 5) && t160 >= 384));
            if (variant >= 16) {
                // ring-staged tiles (see the K-loop banner), one workgroup per CU:
                //   16: 256x160, 3 stages, eight waves   17: 256x128, 3 stages, eight waves   18: 256x256, 2 stages, eight waves
                //   20: 128x160, 4 stages, four waves    21: 128x128, 4 stages, four waves
                //   23: 192x160, 3 stages, eight waves   22: 192x128, 3 stages, eight waves  (M = 8960: 47 row tiles x 5 = 235 of 256 CUs
                //                                                                             where 256-row tiles give 175)
                //   25:  96x160, 4 stages, four waves    24:  96x128, 4 stages, four waves   (M = 2240, N = 1280, short K: 240 tiles, not 180)
                //   26: 192x128, TWO stages, eight waves: 80 KB of LDS and 114 VGPRs, so TWO workgroups share a CU -- the tile under the GEGLU
                //       epilogues, where main loop, GELU polynomial and 275-366 MB of stores run one after the other inside a workgroup
                //   28: 128x160, 27: 128x128, TWO stages, eight waves (72 / 64 KB of LDS: two workgroups per CU): candidates of the in-situ
                //       tuner for the short-K linears, not yet in the cost model (no measured rate)
                // A 160-wide request on an N that 160 does not divide (or with the GEGLU-forward epilogue) takes the 128-wide sibling;
                // 256-wide tiles need N % 256 == 0 (their GEGLU epilogues have no partial column tile).
                const int n_cols = epilogue == SVDX_EPI_GEGLU_FWD ? 2 * aux_dim : N;
                // Two-role tiles (gemm_v5_kernel, round 6), one workgroup per CU:   32: 256x256   34: 160x320.  Their GEGLU epilogues take whole
                // column tiles; a request that does not divide runs the ring tiles' 160- / 128-wide siblings instead (like 18 -> 17 below).
                if (variant == 32 && (epilogue == SVDX_EPI_NONE || n_cols % 256 == 0)) return launch_gemm_v5<T, 8, 4>(p, st);
                if (variant == 34 && (epilogue == SVDX_EPI_NONE || n_cols % 320 == 0)) return launch_gemm_v5<T, 5, 5>(p, st);
                if (variant == 32 || variant == 34) variant = 16;
                // 36 (round 6): 144 x 160, SIX waves (3 x 2), two stages, two workgroups per CU, row tiles 140 apart -- the 64x40 level's
                //     35840 rows are 256 x 140 and the 32x20 level's 8960 are 64 x 140, so N = 320 / 1280 give exactly 512 workgroups: every
                //     slot of the chip, where the 160-row tile of variant 6 fills 448 of them.  160-wide only; no GEGLU-forward epilogue.
                if (variant == 36) {
                    if (nb5 && epilogue != SVDX_EPI_GEGLU_FWD) return launch_gemm_v4<T, 5, 3, 3, 2, 140>(p, st);
                    variant = 28;
                }
                switch (variant) {
                    case 18: if (n_cols % 256 == 0) return launch_gemm_v4<T, 8, 4, 4, 2>(p, st);   // else: fall through to 256 x 128
                    case 16: case 17: return nb5 ? launch_gemm_v4<T, 5, 4, 4, 3>(p, st) : launch_gemm_v4<T, 4, 4, 4, 3>(p, st);
                    case 20: case 21: return nb5 ? launch_gemm_v4<T, 5, 4, 2, 4>(p, st) : launch_gemm_v4<T, 4, 4, 2, 4>(p, st);
                    case 22: case 23: return nb5 ? launch_gemm_v4<T, 5, 3, 4, 3>(p, st) : launch_gemm_v4<T, 4, 3, 4, 3>(p, st);
                    case 26: return launch_gemm_v4<T, 4, 3, 4, 2>(p, st);
                    case 27: case 28: return nb5 ? launch_gemm_v4<T, 5, 2, 4, 2>(p, st) : launch_gemm_v4<T, 4, 2, 4, 2>(p, st);
                    case 24: case 25: return nb5 ? launch_gemm_v4<T, 5, 3, 2, 4>(p, st) : launch_gemm_v4<T, 4, 3, 2, 4>(p, st);
                    default: svdx_set_error("svdx_gemm: unknown variant %d", variant); return -2;
                }
            }
            if (mb5) return launch_gemm_v4<T, 5, 5>(p, st);
            return nb5 ? launch_gemm_v4<T, 5, 4>(p, st) : launch_gemm_v4<T, 4, 4>(p, st);
        }
        if (epilogue != SVDX_EPI_NONE) { svdx_set_error("svdx_gemm: fused epilogue unavailable (buffer too large for variant 4)"); return -2; }
        if (gn_stats) { svdx_set_error("svdx_gemm_gn: statistics unavailable (buffer too large for variant 4)"); return -2; }
        return launch_gemm<T>(p, st);
    });
}

extern "C" int svdx_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                         const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                         int out_mode, float alpha, int split_k, int variant, int epilogue, const void* aux_in, void* aux_out,
                         int aux_dim, int dtype, void* stream) {
    return gemm_entry(A, B, C, M, N, K, lda, ldb, ldc, bias, rowvec, rv_ld, rv_rows_per_group, rv_mod, res, ldres, gather, zero_page,
                      out_mode, alpha, split_k, variant, epilogue, aux_in, aux_out, aux_dim, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0, 0, dtype, stream);
}

extern "C" int svdx_gemm_gn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                            const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                            const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                            float alpha, int variant, float* gn_stats, int gn_rows, int gn_cg, int dtype, void* stream) {
    SVDX_CHECK_ARG(gn_stats, "svdx_gemm_gn: gn_stats must not be NULL");
    return gemm_entry(A, B, C, M, N, K, lda, ldb, ldc, bias, rowvec, rv_ld, rv_rows_per_group, rv_mod, res, ldres, gather, zero_page,
                      SVDX_OUT_ACT, alpha, 1, variant, SVDX_EPI_NONE, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, gn_stats, gn_rows, gn_cg, dtype, stream);
}

extern "C" int svdx_gemm_dual(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                              const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                              const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                              int out_mode, float alpha, int variant, const void* A2, const void* B2, int K2, int lda2, int ldb2,
                              int a2_seg_n, int dtype, void* stream) {
    SVDX_CHECK_ARG(K2 > 0, "svdx_gemm_dual: K2 must be positive");
    return gemm_entry(A, B, C, M, N, K, lda, ldb, ldc, bias, rowvec, rv_ld, rv_rows_per_group, rv_mod, res, ldres, gather, zero_page,
                      out_mode, alpha, 1, variant, SVDX_EPI_NONE, nullptr, nullptr, 0, A2, B2, K2, lda2, ldb2, a2_seg_n, nullptr, 0, 0, dtype, stream);
}

extern "C" int svdx_small_linear(const float* X, const void* W, const float* bias, float* Y, int M, int N, int K,
                                 int ldw, int trans, int silu_in, int accumulate, int dtype, void* stream) {
    SVDX_CHECK_ARG(X && W && Y && M > 0 && N > 0 && K > 0, "svdx_small_linear: bad args");
    SVDX_CHECK_ARG(K % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)W & 15) == 0, "svdx_small_linear: K/ldw must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        if (trans == 0) {
            SVDX_CHECK_ARG(((uintptr_t)X & 15) == 0 && K % 4 == 0, "svdx_small_linear: X must be 16-byte aligned");
            if (M == 1)
                hipLaunchKernelGGL((small_linear_nt<T, 1>), dim3(cdiv(N, 4)), dim3(256), 0, st, X, (const T*)W, bias, Y, M, N, K, ldw,
                                   silu_in, accumulate);
            else if (M <= 4)
                hipLaunchKernelGGL((small_linear_nt<T, 4>), dim3(cdiv(N, 4)), dim3(256), 0, st, X, (const T*)W, bias, Y, M, N, K, ldw,
                                   silu_in, accumulate);
            else
                hipLaunchKernelGGL((small_linear_nt<T, 8>), dim3(cdiv(N, 4)), dim3(256), 0, st, X, (const T*)W, bias, Y, M, N, K, ldw,
                                   silu_in, accumulate);
        } else {
            SVDX_CHECK_ARG(!bias && !silu_in, "svdx_small_linear: trans=1 takes no bias/activation");
            hipLaunchKernelGGL((small_linear_nn<T>), dim3(cdiv(K / 8, 8), M), dim3(256), 0, st, X, (const T*)W, Y, M, N, K, ldw, accumulate);
        }
    });
    SVDX_LAUNCH_CHECK("svdx_small_linear");
    return 0;
}

extern "C" int svdx_gemm_finalize(const float* acc, int nsplit, int64_t slab_stride, void* C, int c_is_f32_accumulate, int M, int N,
                                  int ldc, const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                                  const void* res, int ldres, const float* colsum_slabs, float* colsum_out, int colsum_n, int dtype,
                                  void* stream) {
    SVDX_CHECK_ARG(!colsum_slabs || (colsum_out && colsum_n > 0), "svdx_gemm_finalize: colsum_slabs needs colsum_out / colsum_n");
    SVDX_CHECK_ARG(acc && C && M > 0 && N > 0 && nsplit >= 1, "svdx_gemm_finalize: bad args");
    SVDX_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && slab_stride % 4 == 0 && (!res || ldres % 4 == 0) && (!rowvec || rv_ld % 4 == 0),
                   "svdx_gemm_finalize: N/ld must be multiples of 4");
    SVDX_CHECK_ARG(!rowvec || rv_mod > 0 || rv_rows_per_group > 0, "svdx_gemm_finalize: rowvec needs a grouping");
    const int blocks = (int)std::min<long>(((long)M * N / 4 + 255) / 256, 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gemm_finalize_kernel<T, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, nsplit,
                                             (long)slab_stride, c_is_f32_accumulate ? (T*)nullptr : (T*)C,
                                             c_is_f32_accumulate ? (float*)C : (float*)nullptr, c_is_f32_accumulate == 2, M, N, ldc, bias, rowvec, rv_ld,
                                             rv_rows_per_group, rv_mod, (const T*)res, ldres, colsum_slabs, colsum_out, colsum_n, FinGn{}));
    SVDX_LAUNCH_CHECK("svdx_gemm_finalize");
    return 0;
}

extern "C" int svdx_gemm_finalize_gn(const float* acc, int nsplit, int64_t slab_stride, void* C, int M, int N, int ldc, const float* bias,
                                     const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod, const void* res, int ldres,
                                     float* gn_stats, int gn_rows, int gn_cg, int dtype, void* stream) {
    SVDX_CHECK_ARG(acc && C && gn_stats && M > 0 && N > 0 && nsplit >= 1, "svdx_gemm_finalize_gn: bad args");
    SVDX_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && slab_stride % 4 == 0 && (!res || ldres % 4 == 0) && (!rowvec || rv_ld % 4 == 0),
                   "svdx_gemm_finalize_gn: N/ld must be multiples of 4");
    SVDX_CHECK_ARG(!rowvec || rv_mod > 0 || rv_rows_per_group > 0, "svdx_gemm_finalize_gn: rowvec needs a grouping");
    SVDX_CHECK_ARG(gn_rows > 0 && gn_cg > 0 && M % gn_rows == 0 && N % gn_cg == 0 && N / gn_cg <= FIN_GN_G && (long)N * gn_rows >= 1024 &&
                       ((uintptr_t)gn_stats & 7) == 0,
                   "svdx_gemm_finalize_gn: M=%d whole samples of %d rows (N x rows >= 1024), N=%d whole groups of %d channels (<= %d groups)", M, gn_rows, N, gn_cg, FIN_GN_G);
    FinGn gn;
    gn.stats = reinterpret_cast<unsigned long long*>(gn_stats); gn.rows = gn_rows; gn.cg = gn_cg;
    int k0, k1;
    gn_fixed_scales((long)gn_rows * gn_cg, 0, k0, k1);
    gn.m0 = exp2f((float)k0); gn.m1 = exp2f((float)k1);
    const int blocks = (int)std::min<long>(((long)M * N / 4 + 255) / 256, 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gemm_finalize_kernel<T, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, nsplit,
                                             (long)slab_stride, (T*)C, (float*)nullptr, 0, M, N, ldc, bias, rowvec, rv_ld,
                                             rv_rows_per_group, rv_mod, (const T*)res, ldres, (const float*)nullptr, (float*)nullptr, 0, gn));
    SVDX_LAUNCH_CHECK("svdx_gemm_finalize_gn");
    return 0;
}

extern "C" int svdx_outer_acc(const float* dY, const float* X, float* dW, int M, int N, int K, float scale, void* stream) {
    SVDX_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0, "svdx_outer_acc: bad args");
    const size_t n = (size_t)N * K;
    hipLaunchKernelGGL(outer_acc_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dY, X, dW, M, N, K, scale);
    SVDX_LAUNCH_CHECK("svdx_outer_acc");
    return 0;
}

extern "C" int svdx_small_linear_batch(const svdx_lin_job* jobs, int n_jobs, int M, int trans, int dtype, void* stream) {
    SVDX_CHECK_ARG(jobs && n_jobs > 0 && M > 0, "svdx_small_linear_batch: bad args");
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n_jobs; j0 += SVDX_BATCH_MAX_JOBS) {
        LinPack pk;
        pk.n_jobs = std::min(n_jobs - j0, SVDX_BATCH_MAX_JOBS);
        int blocks = 0;
        for (int j = 0; j < pk.n_jobs; ++j) {
            const svdx_lin_job& q = jobs[j0 + j];
            SVDX_CHECK_ARG(q.X && q.W && q.Y && q.N > 0 && q.K > 0, "svdx_small_linear_batch: job %d: bad args", j0 + j);
            SVDX_CHECK_ARG(q.K % 8 == 0 && q.ldw % 8 == 0 && ((uintptr_t)q.W & 15) == 0,
                           "svdx_small_linear_batch: job %d: K/ldw must be multiples of 8", j0 + j);
            if (trans == 0)
                SVDX_CHECK_ARG(((uintptr_t)q.X & 15) == 0, "svdx_small_linear_batch: job %d: X must be 16-byte aligned", j0 + j);
            else
                SVDX_CHECK_ARG(!q.bias && !(q.flags & 1), "svdx_small_linear_batch: trans=1 takes no bias/activation");
            pk.job[j] = q;
            pk.start[j] = blocks;
            blocks += trans == 0 ? cdiv(q.N, 4) : cdiv(q.K / 8, 8);
        }
        for (int j = pk.n_jobs; j <= SVDX_BATCH_MAX_JOBS; ++j) pk.start[j] = blocks;
        DISPATCH_DTYPE(dtype, {
            if (trans == 0) {
                if (M == 1) hipLaunchKernelGGL((small_linear_nt_batch<T, 1>), dim3(blocks), dim3(256), 0, st, pk, M);
                else if (M <= 4) hipLaunchKernelGGL((small_linear_nt_batch<T, 4>), dim3(blocks), dim3(256), 0, st, pk, M);
                else hipLaunchKernelGGL((small_linear_nt_batch<T, 8>), dim3(blocks), dim3(256), 0, st, pk, M);
            } else {
                hipLaunchKernelGGL((small_linear_nn_batch<T>), dim3(blocks, M), dim3(256), 0, st, pk, M);
            }
        });
        SVDX_LAUNCH_CHECK("svdx_small_linear_batch");
    }
    return 0;
}

extern "C" int svdx_outer_acc_batch(const svdx_outer_job* jobs, int n_jobs, int M, void* stream) {
    SVDX_CHECK_ARG(jobs && n_jobs > 0 && M > 0, "svdx_outer_acc_batch: bad args");
    for (int j0 = 0; j0 < n_jobs; j0 += SVDX_BATCH_MAX_JOBS) {
        OuterPack pk;
        pk.n_jobs = std::min(n_jobs - j0, SVDX_BATCH_MAX_JOBS);
        long blocks = 0;
        for (int j = 0; j < pk.n_jobs; ++j) {
            const svdx_outer_job& q = jobs[j0 + j];
            SVDX_CHECK_ARG(q.dY && q.dW && q.N > 0 && q.K > 0 && (q.X || q.K == 1), "svdx_outer_acc_batch: job %d: bad args", j0 + j);
            pk.job[j] = q;
            pk.start[j] = (int)blocks;
            blocks += cdiv((size_t)q.N * q.K, 256);
            SVDX_CHECK_ARG(blocks < (1l << 31), "svdx_outer_acc_batch: too many workgroups");
        }
        for (int j = pk.n_jobs; j <= SVDX_BATCH_MAX_JOBS; ++j) pk.start[j] = (int)blocks;
        hipLaunchKernelGGL(outer_acc_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk, M);
        SVDX_LAUNCH_CHECK("svdx_outer_acc_batch");
    }
    return 0;
}

extern "C" int svdx_grad_finalize_batch(const svdx_gradfin_job* jobs, int n_jobs, void* stream) {
    SVDX_CHECK_ARG(jobs && n_jobs > 0, "svdx_grad_finalize_batch: bad args");
    for (int j0 = 0; j0 < n_jobs; j0 += SVDX_BATCH_MAX_JOBS) {
        GradFinPack pk;
        pk.n_jobs = std::min(n_jobs - j0, SVDX_BATCH_MAX_JOBS);
        long blocks = 0;
        for (int j = 0; j < pk.n_jobs; ++j) {
            const svdx_gradfin_job& q = jobs[j0 + j];
            SVDX_CHECK_ARG(q.acc && q.dst && q.nsplit >= 1 && q.count > 0 && q.count % 4 == 0 && q.slab_stride % 4 == 0 &&
                               (((uintptr_t)q.acc | (uintptr_t)q.dst) & 15) == 0 && (!q.colsum_slabs || (q.colsum_out && q.colsum_n > 0)),
                           "svdx_grad_finalize_batch: job %d: bad args", j0 + j);
            pk.job[j] = q;
            pk.start[j] = (int)blocks;
            blocks += std::min<long>(cdiv(q.count / 4, 256), 1024);
        }
        for (int j = pk.n_jobs; j <= SVDX_BATCH_MAX_JOBS; ++j) pk.start[j] = (int)blocks;
        hipLaunchKernelGGL(grad_finalize_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk);
        SVDX_LAUNCH_CHECK("svdx_grad_finalize_batch");
    }
    return 0;
}

extern "C" int svdx_timestep_embed(const float* t, float* out, int n, int dim, void* stream) {
    SVDX_CHECK_ARG(t && out && n > 0 && dim > 0 && dim % 2 == 0, "svdx_timestep_embed: bad args");
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(cdiv((long)n * dim / 2, 256)), dim3(256), 0, (hipStream_t)stream, t, out,
                       n, dim);
    SVDX_LAUNCH_CHECK("svdx_timestep_embed");
    return 0;
}
