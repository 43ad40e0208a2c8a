// norm.hip -- GroupNorm(32)(+SiLU) and LayerNorm, forward and backward, for channels-last rows (gfx950).
//
// HBM-bound kernels.  Every thread owns one fixed 16-byte column chunk (8 channels)
// and walks rows, so per-channel scale/shift (or partial sums) live in registers; loads/stores are 16 B per
// lane, coalesced across the row.  Group statistics are reduced thread -> LDS -> one atomicAdd per group per
// block.  GroupNorm "sample" = `rows` consecutive rows (a frame for the 2-D norms, a whole clip for the
// TemporalResnetBlock norms whose groups span all frames).
#include "common.h"
#include <stdlib.h>

namespace {


struct GnGeom {
    int n_s, rows, C, G, cg, cc, rpi;   // cg channels/group, cc 16-byte chunks per row, rpi rows per iteration
    int slab;                           // rows per block: sized so that even the 5x8 latent level launches >= ~1000 blocks
};

// Group sums live in SVDX_GN_REPLICAS partial copies ([rep][n_s][G][2]): blocks spread their atomics over the copies (a clip-wide
// temporal norm has only 64 distinct addresses and >1000 blocks: same-address atomics serialise), readers add them up.
// The sums are 64-bit FIXED-POINT integers: integer addition is associative, so the result does not depend on the order in which
// the blocks' atomics land -- the statistics (and with them the whole step) are run-to-run identical, which float atomics are not.
// Scale 2^k per (kind, count): k = 62 - (bits of the largest term) - ceil(log2(count)).  The plain sum is safe over the whole fp16
// range (|x| <= 2^16); the sum of squares is scaled for a group whose ROOT-MEAN-SQUARE stays below 2^12 -- 2^24 per term instead of
// the worst case 2^32: eight more fraction bits, which clip-wide norms over 8 x 576 x 1024 rows need for groups of small activations
// (with the worst-case scale a block partial was rounded to 1/32 there; a group with an rms of 4096 has left fp16 training long
// before).  Backward: |dz gamma| <= 2^18, |xhat| <= 2^8.  A block's float partial converts exactly (24 significant bits).
__device__ __forceinline__ void group_sums(const float* stats, int n_s, int n, int G, int g, float i0, float i1, float& s, float& ss) {
    long long a = 0, b = 0;
    const long long* st = reinterpret_cast<const long long*>(stats);
#pragma unroll
    for (int r = 0; r < SVDX_GN_REPLICAS; ++r) {
        const size_t o = (((size_t)r * n_s + n) * G + g) * 2;
        a += st[o]; b += st[o + 1];
    }
    s = (float)a * i0; ss = (float)b * i1;
}
__device__ __forceinline__ void group_mean_rstd(const float* stats, int n_s, int n, int G, int g, float cnt, float eps, float i0, float i1,
                                                float& mean, float& rstd) {
    float s, ss;
    group_sums(stats, n_s, n, G, g, i0, i1, s, ss);
    mean = s / cnt;
    const float var = fmaxf(ss / cnt - mean * mean, 0.f);
    rstd = rsqrtf(var + eps);
}

// Per-block statistics: thread g < G adds the replicas of group g once and leaves (mean, rstd) [and the backward sums / cnt] in LDS.
// (Every thread doing this for its own 2-3 groups was 48 tiny loads per thread against 5 rows of data: the prologue, not the
// streaming, set the kernel time -- 19 us where the rows alone take 12.)
template <bool BWD>
__device__ __forceinline__ void gn_block_stats(const float* stats, const float* bstats, const GnGeom& q, int n, float eps, float* sm) {
    const int g = threadIdx.x;
    if (g < q.G) {
        int fk0, fk1, bk0, bk1;
        gn_fixed_scales((long)q.rows * q.cg, 0, fk0, fk1);
        const float cnt = (float)q.rows * q.cg;
        float mean, rstd;
        group_mean_rstd(stats, q.n_s, n, q.G, g, cnt, eps, exp2f((float)-fk0), exp2f((float)-fk1), mean, rstd);
        sm[g * 4] = mean; sm[g * 4 + 1] = rstd;
        if (BWD) {
            gn_fixed_scales((long)q.rows * q.cg, 1, bk0, bk1);
            float s1, s2;
            group_sums(bstats, q.n_s, n, q.G, g, exp2f((float)-bk0), exp2f((float)-bk1), s1, s2);
            sm[g * 4 + 2] = s1 / cnt; sm[g * 4 + 3] = s2 / cnt;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void gn_load8f(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}

// MODE 0: stats of x.  MODE 1: backward stats (sum dz*gamma, sum dz*gamma*xhat).
template <typename T, int MODE>
__global__ void gn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* out,
                                 GnGeom q, float eps, int silu) {
    __shared__ unsigned long long gacc[64];
    __shared__ float sm[32 * 4];
    const int t = threadIdx.x;
    if (t < 64) gacc[t] = 0ull;
    int sk0, sk1;
    gn_fixed_scales((long)q.rows * q.cg, MODE, sk0, sk1);            // scales of the sums this launch produces
    const float m0 = exp2f((float)sk0), m1 = exp2f((float)sk1);
    const int n = blockIdx.x, slab = blockIdx.y;
    const int j = t % q.cc, ry = t / q.cc;
    const bool worker = ry < q.rpi;
    const int r0 = slab * q.slab, r1 = min(q.rows, r0 + q.slab);
    // U rows per batch, the next batch already travelling while this one is accumulated
    constexpr int U = MODE == 0 ? 4 : 2;
    const size_t base = (size_t)n * q.rows * q.C + j * 8;
    Vec8<T> xb[U], db[U], xn[U], dn[U];
    auto fetch = [&](int r, Vec8<T> (&xo)[U], Vec8<T> (&dd)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = min(r + u * q.rpi, q.rows - 1);          // rows past the slab are loaded (in range) and ignored
            xo[u] = *reinterpret_cast<const Vec8<T>*>(x + base + (size_t)rr * q.C);
            if (MODE == 1) dd[u] = *reinterpret_cast<const Vec8<T>*>(dy + base + (size_t)rr * q.C);
        }
    };
    int r = r0 + ry;
    if (worker && r < r1) fetch(r, xb, db);                         // the first rows travel while the statistics are put together
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
    float sc[8], sh[8], gm[8], mu[8], rs[8];
    if (MODE == 1) {
        gn_block_stats<false>(stats, nullptr, q, n, eps, sm);
        float bt[8];
        gn_load8f(gamma + j * 8, gm);
        gn_load8f(beta + j * 8, bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (j * 8 + e) / q.cg;
            mu[e] = sm[g * 4]; rs[e] = sm[g * 4 + 1];
            sc[e] = rs[e] * gm[e];
            sh[e] = bt[e] - mu[e] * sc[e];
        }
    } else {
        __syncthreads();                                             // gacc is zero
    }
    if (worker) {
        for (; r < r1; r += U * q.rpi) {
            const int rn = r + U * q.rpi;
            if (rn < r1) fetch(rn, xn, dn);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * q.rpi < r1) {
                    if (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float xv = to_f<T>(xb[u].v[e]); a0[e] += xv; a1[e] += xv * xv; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xv = to_f<T>(xb[u].v[e]);
                            float dz = to_f<T>(db[u].v[e]);
                            if (silu) dz *= silu_gradf_(xv * sc[e] + sh[e]);
                            const float dzg = dz * gm[e];
                            a0[e] += dzg;
                            a1[e] += dzg * (xv - mu[e]) * rs[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { xb[u] = xn[u]; if (MODE == 1) db[u] = dn[u]; }
        }
        // merge the 8 channels into their groups (runs of equal group id), then one LDS atomic per run
        int cur = (j * 8) / q.cg;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (j * 8 + e) / q.cg;
            if (g != cur) {
                atomicAdd(&gacc[cur * 2], (unsigned long long)__float2ll_rn(s0 * m0));
                atomicAdd(&gacc[cur * 2 + 1], (unsigned long long)__float2ll_rn(s1 * m1));
                cur = g; s0 = 0.f; s1 = 0.f;
            }
            s0 += a0[e]; s1 += a1[e];
        }
        atomicAdd(&gacc[cur * 2], (unsigned long long)__float2ll_rn(s0 * m0));
        atomicAdd(&gacc[cur * 2 + 1], (unsigned long long)__float2ll_rn(s1 * m1));
    }
    __syncthreads();
    if (t < 2 * q.G)
        atomicAdd(reinterpret_cast<unsigned long long*>(out) + ((size_t)(slab % SVDX_GN_REPLICAS) * q.n_s + n) * q.G * 2 + t, gacc[t]);
}

// MODE 0: y = act(xhat*gamma+beta).  MODE 1: dx = rstd*(dz*gamma - (s1 + xhat*s2)/cnt) (+ add).
template <typename T, int MODE>
__global__ void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                const float* __restrict__ bstats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ add, T* __restrict__ out,
                                GnGeom q, float eps, int silu) {
    __shared__ float sm[32 * 4];
    const int t = threadIdx.x;
    const int n = blockIdx.x, slab = blockIdx.y;
    const int j = t % q.cc, ry = t / q.cc;
    const bool worker = ry < q.rpi;
    const int r0 = slab * q.slab, r1 = min(q.rows, r0 + q.slab);
    // batches of U rows, next batch prefetched (see gn_reduce_kernel)
    constexpr int U = MODE == 0 ? 4 : 2;
    const size_t base = (size_t)n * q.rows * q.C + j * 8;
    Vec8<T> xb[U], db[U], ab[U], xn[U], dn[U], an[U];
    auto fetch = [&](int r, Vec8<T> (&xo)[U], Vec8<T> (&dd)[U], Vec8<T> (&aa)[U]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = base + (size_t)min(r + u * q.rpi, q.rows - 1) * q.C;
            xo[u] = *reinterpret_cast<const Vec8<T>*>(x + off);
            if (MODE == 1) {
                dd[u] = *reinterpret_cast<const Vec8<T>*>(dy + off);
                if (add) aa[u] = *reinterpret_cast<const Vec8<T>*>(add + off);
            }
        }
    };
    int r = r0 + ry;
    if (worker && r < r1) fetch(r, xb, db, ab);                     // the first rows travel while the statistics are put together
    gn_block_stats<MODE == 1>(stats, bstats, q, n, eps, sm);
    if (!worker) return;
    float sc[8], sh[8], gm[8], mu[8], rs[8], b1[8], b2[8], bt[8];
    gn_load8f(gamma + j * 8, gm);
    gn_load8f(beta + j * 8, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (j * 8 + e) / q.cg;
        mu[e] = sm[g * 4]; rs[e] = sm[g * 4 + 1];
        sc[e] = rs[e] * gm[e];
        sh[e] = bt[e] - mu[e] * sc[e];
        if (MODE == 1) { b1[e] = sm[g * 4 + 2]; b2[e] = sm[g * 4 + 3]; }
    }
    for (; r < r1; r += U * q.rpi) {
        const int rn = r + U * q.rpi;
        // forward: the next batch travels while this one is normalised; backward (three operands per row): one batch at a time --
        // a second register set halved the occupancy and cost more than the prefetch gave (39 vs 31 us at the 64x40 level)
        if (MODE == 0 && rn < r1) fetch(rn, xn, dn, an);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * q.rpi;
            if (rr < r1) {
                float o[8];
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = to_f<T>(xb[u].v[e]) * sc[e] + sh[e];
                        o[e] = silu ? siluf_(z) : z;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xv = to_f<T>(xb[u].v[e]);
                        float dz = to_f<T>(db[u].v[e]);
                        if (silu) dz *= silu_gradf_(xv * sc[e] + sh[e]);
                        const float xhat = (xv - mu[e]) * rs[e];
                        o[e] = rs[e] * (dz * gm[e] - (b1[e] + xhat * b2[e]));
                        if (add) o[e] += to_f<T>(ab[u].v[e]);
                    }
                }
                store8<T>(out + base + (size_t)rr * q.C, o);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) xb[u] = xn[u];
        } else if (rn < r1) {
            fetch(rn, xb, db, ab);
        }
    }
}

int gn_geom(GnGeom& q, int n_s, int rows, int C, int G, int& threads, bool reduce = false) {
    SVDX_CHECK_ARG(n_s > 0 && rows > 0 && C > 0 && G > 0 && G <= 32, "groupnorm: bad sizes");
    SVDX_CHECK_ARG(C % G == 0 && C % 8 == 0 && C / 8 <= 1024, "groupnorm: C=%d must be a multiple of 8 and of G", C);
    q.n_s = n_s; q.rows = rows; q.C = C; q.G = G; q.cg = C / G; q.cc = C / 8;
    q.rpi = q.cc >= 256 ? 1 : 256 / q.cc;
    threads = q.cc * q.rpi;
    // a statistics launch ends with 2G 64-bit atomics per block (the chip retires ~15 per ns): 64-row slabs while that still leaves
    // >= 512 blocks; the apply kernels want >= 1024 blocks of 32 rows
    int slab = reduce ? 64 : 32;
    const long want = reduce ? 512 : 1024;
    while (slab > 2 * q.rpi && slab > 2 && (long)n_s * ((rows + slab - 1) / slab) < want) slab >>= 1;
    q.slab = slab;
    return 0;
}

// ---- LayerNorm: one wave per row, whole row in registers (C <= 2048) ------------------------------------------
constexpr int LN_MAXCH_LIMIT = 4;

// R rows per wave in flight: a C = 320 row is ONE 640-byte load, and a wave with a single load outstanding cannot keep HBM busy
// (one row at a time ran at 3.1 TB/s where a plain copy reaches 5).
template <typename T, int LN_MAXCH, int R>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int cc = C / 8;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const float invC = 1.f / (float)C;
    for (int row0 = wid * R; row0 < rows; row0 += nw * R) {
        Vec8<T> raw[R][LN_MAXCH];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = min(row0 + k, rows - 1);
#pragma unroll
            for (int i = 0; i < LN_MAXCH; ++i) {
                const int j = lane + i * 64;
                if (j < cc) raw[k][i] = *reinterpret_cast<const Vec8<T>*>(x + (size_t)row * C + j * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = row0 + k;
            float v[LN_MAXCH][8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXCH; ++i) {
                if (lane + i * 64 < cc) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[i][e] = to_f<T>(raw[k][i].v[e]); s += v[i][e]; }
                }
            }
            const float mean = wave_sum(s) * invC;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXCH; ++i) {
                if (lane + i * 64 < cc) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; ss += d * d; }
                }
            }
            const float rstd = rsqrtf(wave_sum(ss) * invC + eps);
            if (row < rows) {
                if (lane == 0) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }
#pragma unroll
                for (int i = 0; i < LN_MAXCH; ++i) {
                    const int j = lane + i * 64;
                    if (j < cc) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gamma[j * 8 + e] + beta[j * 8 + e];
                        store8<T>(y + (size_t)row * C + j * 8, o);
                    }
                }
            }
        }
    }
}

// ---- LayerNorm, 16 lanes per row (C <= 640): a wave normalises FOUR rows per step, the row sums are DPP reductions inside the
// 16-lane group (no LDS crossbar), every lane keeps the affine parameters of its NCH chunks.  One wave per row spent a full wave of
// instruction issue on 640 bytes: the kernels were issue / latency bound at 2.9-3.1 TB/s.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, float* __restrict__ stats, int rows, int C, float eps) {
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int cc = C / 8;
    float gm[NCH][8], bt[NCH][8];
    bool cv[NCH];
    int cl[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = l16 + 16 * j;
        cv[j] = c < cc;
        cl[j] = min(c, cc - 1);
        gn_load8f(gamma + cl[j] * 8, gm[j]);
        gn_load8f(beta + cl[j] * 8, bt[j]);
    }
    const float invC = 1.f / (float)C;
    constexpr int R = 2;                                  // rows in flight per 16-lane group
    for (int row0 = (blockIdx.x * 16 + grp) * R; row0 < rows; row0 += gridDim.x * 16 * R) {
        Vec8<T> raw[R][NCH];
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
            for (int j = 0; j < NCH; ++j) raw[k][j] = *reinterpret_cast<const Vec8<T>*>(x + (size_t)min(row0 + k, rows - 1) * C + cl[j] * 8);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = row0 + k;
            float v[NCH][8];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[j][e] = cv[j] ? to_f<T>(raw[k][j].v[e]) : 0.f; s += v[j][e]; }
            const float mean = row16_sum(s) * invC;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = cv[j] ? v[j][e] - mean : 0.f; ss += d * d; }
            const float rstd = rsqrtf(row16_sum(ss) * invC + eps);
            if (row < rows) {
                if (l16 == 0) *reinterpret_cast<float2*>(stats + (size_t)row * 2) = float2{mean, rstd};
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    if (cv[j]) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * gm[j][e] + bt[j][e];
                        store8<T>(y + (size_t)row * C + cl[j] * 8, o);
                    }
                }
            }
        }
    }
}

// ---- LayerNorm backward: LANES lanes per row, every lane owns the 16-byte chunks ll, ll + LANES, ... (NCH of them) of its row.
// Round 3 rewrite.  The round-2 kernels kept the per-column affine-gradient accumulators of up to five chunks per lane in registers
// (2 x 5 x 8 floats) beside four operand rows: 213 VGPRs at C = 320 and 355 at C = 640 -- one or two waves per SIMD on a kernel that
// only has memory latency to hide (25-28 us per launch where the operands stream in 8 us).  Here the lane count follows the row
// width so that NCH <= 3 at every width of the UNet (16 / 32 / 64 lanes for C = 320 / 640 / 1280: 2.5 chunks per lane), the frozen
// LayerNorms compile without the accumulators (AFFINE = false), and nothing but the operand chunks themselves stays live across the
// row reduction.  Affine gradients: block partials through a [row groups][C] LDS slab added in a fixed order, then either one
// `partial[block][2C]` row (reduced by ln_param_reduce_kernel -- deterministic) or, without scratch, float atomics.
template <int LANES>
__device__ __forceinline__ float ln_group_sum(float v) {
    v = row16_sum(v);
    if (LANES >= 32) v += __shfl_xor(v, 16, 64);
    if (LANES >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <typename T, int LANES, int NCH, bool AFFINE>
__global__ __launch_bounds__(256, 2) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const T* __restrict__ add, const T* __restrict__ add2,
                                                        float add2_scale, T* __restrict__ dx, float* dgamma, float* dbeta, float* partial, int rows,
                                                        int C) {
    extern __shared__ float red[];   // AFFINE: [256 / LANES row groups][C]
    constexpr int GROUPS = 256 / LANES;
    const int ll = threadIdx.x % LANES, grp = threadIdx.x / LANES;
    const int cc = C / 8;
    // (four waves per SIMD -- launch bound 4, gamma re-read per row -- spills at three chunks per lane and ran 2-3x slower:
    // profiles/r4_ln_bwd_sweep.txt)
    float gm[NCH][8], pg[AFFINE ? NCH : 1][8], pb[AFFINE ? NCH : 1][8];
    bool cv[NCH];
    int cl[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = ll + LANES * j;
        cv[j] = c < cc;
        cl[j] = min(c, cc - 1);
        gn_load8f(gamma + cl[j] * 8, gm[j]);
        if (AFFINE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pg[j][e] = pb[j][e] = 0.f;
        }
    }
    const float invC = 1.f / (float)C;
    for (int row = blockIdx.x * GROUPS + grp; row < rows; row += gridDim.x * GROUPS) {
        // every operand of the row is requested before the first use (four chunks per lane with the affine accumulators would spill:
        // that instantiation -- rows wider than 1536, none in the UNet -- fetches the two addends late instead)
        constexpr bool PRE = !(AFFINE && NCH > 3);
        Vec8<T> xr[NCH], dr[NCH], ar[PRE ? NCH : 1], a2r[PRE ? NCH : 1];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (cv[j]) {
                const size_t off = (size_t)row * C + cl[j] * 8;
                xr[j] = *reinterpret_cast<const Vec8<T>*>(x + off);
                dr[j] = *reinterpret_cast<const Vec8<T>*>(dy + off);
                if (PRE && add) ar[j] = *reinterpret_cast<const Vec8<T>*>(add + off);
                if (PRE && add2) a2r[j] = *reinterpret_cast<const Vec8<T>*>(add2 + off);
            }
        }
        const float2 mr = *reinterpret_cast<const float2*>(stats + (size_t)row * 2);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (cv[j]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = to_f<T>(dr[j].v[e]);
                    const float xh = (to_f<T>(xr[j].v[e]) - mr.x) * mr.y;
                    const float dg = d * gm[j][e];
                    s1 += dg;
                    s2 += dg * xh;
                    if (AFFINE) {
                        pg[j][e] += d * xh;
                        pb[j][e] += d;
                    }
                }
            }
        }
        const float m1 = ln_group_sum<LANES>(s1) * invC, m2 = ln_group_sum<LANES>(s2) * invC;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (cv[j]) {
                const size_t off = (size_t)row * C + cl[j] * 8;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (to_f<T>(xr[j].v[e]) - mr.x) * mr.y;
                    o[e] = mr.y * (to_f<T>(dr[j].v[e]) * gm[j][e] - m1 - xh * m2);
                }
                if (add) {
                    const Vec8<T> a8 = PRE ? ar[PRE ? j : 0] : *reinterpret_cast<const Vec8<T>*>(add + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += to_f<T>(a8.v[e]);
                }
                if (add2) {
                    const Vec8<T> a8 = PRE ? a2r[PRE ? j : 0] : *reinterpret_cast<const Vec8<T>*>(add2 + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += add2_scale * to_f<T>(a8.v[e]);
                }
                store8<T>(dx + off, o);
            }
        }
    }
    if (AFFINE) {
        // column partials: every row group parks its own in LDS, then each thread adds the groups of its columns in order; gamma and
        // beta gradients take turns in the same [GROUPS][C] slab (20 KiB at the three widths of the UNet)
        float* mine = red + (size_t)grp * C;
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (cv[j]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) mine[cl[j] * 8 + e] = which == 0 ? pg[j][e] : pb[j][e];
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < C; i += 256) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) t += red[(size_t)g * C + i];
                if (partial) partial[(size_t)blockIdx.x * 2 * C + which * C + i] = t;
                else atomicAdd((which == 0 ? dgamma : dbeta) + i, t);
            }
            __syncthreads();
        }
    }
}

// dgamma[c] += sum_b partial[b][c], dbeta[c] += sum_b partial[b][C + c]: 64 columns x 16 row groups per block.
__device__ __forceinline__ void ln_param_reduce_body(const float* __restrict__ partial, int nblk, int C, float* dgamma, float* dbeta,
                                                     int blk, float (*sm)[64]) {
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blk * 64 + cl;
    float s = 0.f;
    if (col < 2 * C) {
#pragma unroll 8
        for (int b = rg; b < nblk; b += 16) s += partial[(size_t)b * 2 * C + col];
    }
    sm[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sm[i][cl];
        if (col < C) dgamma[col] += t; else dbeta[col - C] += t;
    }
}

__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, int nblk, int C,
                                                               float* dgamma, float* dbeta) {
    __shared__ float sm[16][64];
    ln_param_reduce_body(partial, nblk, C, dgamma, dbeta, blockIdx.x, sm);
}

// the reductions of MANY LayerNorm backwards in one launch (svdx_ln_bwd with defer_reduce leaves the partial rows; the 48 trainable
// LayerNorms of a backward sweep otherwise pay 48 launches of ~5 us for a few hundred KB each); job table in the kernel arguments
struct LnRedPack {
    svdx_lnred_job job[SVDX_BATCH_MAX_JOBS];
    int start[SVDX_BATCH_MAX_JOBS + 1];
    int n_jobs;
};

__global__ __launch_bounds__(1024) void ln_param_reduce_batch_kernel(const LnRedPack pk) {
    __shared__ float sm[16][64];
    int j = 0;
    while (j + 1 < pk.n_jobs && (int)blockIdx.x >= pk.start[j + 1]) ++j;
    const svdx_lnred_job& q = pk.job[j];
    ln_param_reduce_body(q.partial, q.nblk, q.C, q.dgamma, q.dbeta, blockIdx.x - pk.start[j], sm);
}

// workgroups (= partial rows left in scratch) of the affine-gradient form of svdx_ln_bwd
static int ln_bwd_affine_blocks(int rows, int C) {
    const int cc = C / 8;
    const int lanes = cc <= 48 ? 16 : (cc <= 96 ? 32 : 64);
    const int groups = 256 / lanes;
    // Two workgroups per CU (208 VGPRs) x 256 CUs: ONE resident round that strides over the rows, so that the per-block epilogue (two
    // LDS reductions + the partial row) is paid 512 times and no round runs part-empty -- 35.1 -> 27.7 us at 35840 x 320, 22.3 -> 17.5
    // at 8960 x 640 against the former rows / (2 groups) blocks (profiles/r4_ln_bwd_sweep.txt; kernels.ln_bwd_blocks mirrors the rule).
    const int per = 1;
    const int cap = min(512, SVDX_LN_PARTIAL_ROWS);
    return max(1, min(cdiv(rows, per * groups), cap));
}

}  // namespace

extern "C" int svdx_ln_bwd_blocks(int rows, int C) { return rows > 0 && C > 0 && C % 8 == 0 ? ln_bwd_affine_blocks(rows, C) : 0; }

extern "C" int svdx_ln_param_reduce_batch(const svdx_lnred_job* jobs, int n_jobs, void* stream) {
    SVDX_CHECK_ARG(jobs && n_jobs > 0, "svdx_ln_param_reduce_batch: bad args");
    for (int j0 = 0; j0 < n_jobs; j0 += SVDX_BATCH_MAX_JOBS) {
        LnRedPack pk;
        pk.n_jobs = min(n_jobs - j0, SVDX_BATCH_MAX_JOBS);
        int blocks = 0;
        for (int j = 0; j < pk.n_jobs; ++j) {
            const svdx_lnred_job& q = jobs[j0 + j];
            SVDX_CHECK_ARG(q.partial && q.dgamma && q.dbeta && q.nblk > 0 && q.C > 0, "svdx_ln_param_reduce_batch: job %d: bad args", j0 + j);
            pk.job[j] = q;
            pk.start[j] = blocks;
            blocks += cdiv(2 * q.C, 64);
        }
        for (int j = pk.n_jobs; j <= SVDX_BATCH_MAX_JOBS; ++j) pk.start[j] = blocks;
        hipLaunchKernelGGL(ln_param_reduce_batch_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, pk);
        SVDX_LAUNCH_CHECK("svdx_ln_param_reduce_batch");
    }
    return 0;
}

extern "C" int svdx_gn_stats(const void* x, float* stats, int n_s, int rows, int C, int G, int prezeroed, int dtype, void* stream) {
    GnGeom q; int threads;
    if (int rc = gn_geom(q, n_s, rows, C, G, threads, true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!prezeroed) { if (int rc = svdx_zero(stats, sizeof(long long) * 2 * n_s * G * SVDX_GN_REPLICAS, stream)) return rc; }      // a kernel, not a memset node: see svdx_zero
    dim3 grid(n_s, cdiv(rows, q.slab));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_reduce_kernel<T, 0>), grid, dim3(threads), 0, st, (const T*)x,
                                             (const T*)nullptr, (const float*)nullptr, (const float*)nullptr,
                                             (const float*)nullptr, stats, q, 0.f, 0));
    SVDX_LAUNCH_CHECK("svdx_gn_stats");
    return 0;
}

extern "C" int svdx_gn_apply(const void* x, const float* stats, const float* gamma, const float* beta, void* y,
                             int n_s, int rows, int C, int G, float eps, int silu, int dtype, void* stream) {
    GnGeom q; int threads;
    if (int rc = gn_geom(q, n_s, rows, C, G, threads)) return rc;
    dim3 grid(n_s, cdiv(rows, q.slab));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_apply_kernel<T, 0>), grid, dim3(threads), 0, (hipStream_t)stream,
                                             (const T*)x, (const T*)nullptr, stats, (const float*)nullptr, gamma, beta,
                                             (const T*)nullptr, (T*)y, q, eps, silu));
    SVDX_LAUNCH_CHECK("svdx_gn_apply");
    return 0;
}

extern "C" int svdx_gn_bwd_stats(const void* dy, const void* x, const float* stats, const float* gamma,
                                 const float* beta, float* bstats, int n_s, int rows, int C, int G, float eps,
                                 int silu, int prezeroed, int dtype, void* stream) {
    GnGeom q; int threads;
    if (int rc = gn_geom(q, n_s, rows, C, G, threads, true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!prezeroed) { if (int rc = svdx_zero(bstats, sizeof(long long) * 2 * n_s * G * SVDX_GN_REPLICAS, stream)) return rc; }
    dim3 grid(n_s, cdiv(rows, q.slab));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_reduce_kernel<T, 1>), grid, dim3(threads), 0, st, (const T*)x,
                                             (const T*)dy, stats, gamma, beta, bstats, q, eps, silu));
    SVDX_LAUNCH_CHECK("svdx_gn_bwd_stats");
    return 0;
}

extern "C" int svdx_gn_bwd_apply(const void* dy, const void* x, const float* stats, const float* bstats,
                                 const float* gamma, const float* beta, const void* add, void* dx, int n_s, int rows,
                                 int C, int G, float eps, int silu, int dtype, void* stream) {
    GnGeom q; int threads;
    if (int rc = gn_geom(q, n_s, rows, C, G, threads)) return rc;
    dim3 grid(n_s, cdiv(rows, q.slab));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_apply_kernel<T, 1>), grid, dim3(threads), 0, (hipStream_t)stream,
                                             (const T*)x, (const T*)dy, stats, bstats, gamma, beta, (const T*)add,
                                             (T*)dx, q, eps, silu));
    SVDX_LAUNCH_CHECK("svdx_gn_bwd_apply");
    return 0;
}

extern "C" int svdx_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows,
                           int C, float eps, int dtype, void* stream) {
    SVDX_CHECK_ARG(rows > 0 && C % 8 == 0 && C / 8 <= 64 * LN_MAXCH_LIMIT, "svdx_ln_fwd: C=%d unsupported", C);
    const int nch16 = (C / 8 + 15) / 16;
    if (nch16 <= 5) {                       // C <= 640: 16 lanes per row
        const int blocks16 = min(cdiv(rows, 32), 2048);
#define LN_FWD16(NCH) hipLaunchKernelGGL((ln_fwd16_kernel<T, NCH>), dim3(blocks16), dim3(256), 0, (hipStream_t)stream, (const T*)x, gamma, beta, \
                                         (T*)y, stats, rows, C, eps)
        DISPATCH_DTYPE(dtype, { if (nch16 == 1) LN_FWD16(1); else if (nch16 == 2) LN_FWD16(2); else if (nch16 == 3) LN_FWD16(3); else LN_FWD16(5); });
#undef LN_FWD16
        SVDX_LAUNCH_CHECK("svdx_ln_fwd");
        return 0;
    }
    const int nch = (C / 8 + 63) / 64;
    const int R = nch == 1 ? 4 : (nch == 2 ? 2 : 1);
    const int blocks = min(cdiv(rows, 4 * R), 4096);
#define LN_FWD(NCH, RR) hipLaunchKernelGGL((ln_fwd_kernel<T, NCH, RR>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, gamma, \
                                           beta, (T*)y, stats, rows, C, eps)
    DISPATCH_DTYPE(dtype, { if (nch == 1) LN_FWD(1, 4); else if (nch == 2) LN_FWD(2, 2); else if (nch == 3) LN_FWD(3, 1); else LN_FWD(4, 1); });
#undef LN_FWD
    SVDX_LAUNCH_CHECK("svdx_ln_fwd");
    return 0;
}

extern "C" int svdx_ln_bwd(const void* dy, const void* x, const float* stats, const float* gamma, const void* add, const void* add2,
                           float add2_scale, void* dx, float* dgamma, float* dbeta, float* scratch, int rows, int C, int defer_reduce,
                           int dtype, void* stream) {
    SVDX_CHECK_ARG(rows > 0 && C % 8 == 0 && C / 8 <= 64 * LN_MAXCH_LIMIT, "svdx_ln_bwd: C=%d unsupported", C);
    SVDX_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "svdx_ln_bwd: dgamma/dbeta must come together");
    hipStream_t st = (hipStream_t)stream;
    const int cc = C / 8;
    // lanes per row: the fewest of 16 / 32 / 64 that cover the row with at most three chunks per lane (C <= 384 / 768 / 1536); wider
    // rows take four chunks on 64 lanes
    const int lanes = cc <= 48 ? 16 : (cc <= 96 ? 32 : 64);
    const int nch = (cc + lanes - 1) / lanes;
    const int groups = 256 / lanes;
    // affine grads: with scratch, up to SVDX_LN_PARTIAL_ROWS blocks each leave one [2C] partial row; without, every block ends with
    // 2*C float atomics, so keep one block per CU there
    SVDX_CHECK_ARG(!defer_reduce || (dgamma && scratch), "svdx_ln_bwd: defer_reduce needs dgamma/dbeta and scratch");
    int blocks = min(cdiv(rows, groups), 2048);
    if (dgamma) blocks = scratch ? ln_bwd_affine_blocks(rows, C) : min(blocks, 256);
    const size_t sh = dgamma ? sizeof(float) * groups * C : 0;
    SVDX_CHECK_ARG(sh <= 64 * 1024, "svdx_ln_bwd: C=%d too wide for the affine-gradient slab", C);
#define LN_BWD_A(L, N, A)                                                                                                          \
    hipLaunchKernelGGL((ln_bwd_kernel<T, L, N, A>), dim3(blocks), dim3(256), sh, st, (const T*)dy, (const T*)x, stats, gamma,           \
                       (const T*)add, (const T*)add2, add2_scale, (T*)dx, dgamma, dbeta, scratch, rows, C)
#define LN_BWD(L, N)                                                                                                                  \
    {                                                                                                                                 \
        if (dgamma) LN_BWD_A(L, N, true); else LN_BWD_A(L, N, false);                                                                   \
    }
    DISPATCH_DTYPE(dtype, {
        if (lanes == 16) { if (nch == 1) LN_BWD(16, 1) else if (nch == 2) LN_BWD(16, 2) else LN_BWD(16, 3) }
        else if (lanes == 32) LN_BWD(32, 3)
        else { if (nch <= 3) LN_BWD(64, 3) else LN_BWD(64, 4) }
    });
#undef LN_BWD
#undef LN_BWD_A
    SVDX_LAUNCH_CHECK("svdx_ln_bwd");
    if (dgamma && scratch && !defer_reduce) {
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(2 * C, 64)), dim3(1024), 0, st, scratch, blocks, C, dgamma, dbeta);
        SVDX_LAUNCH_CHECK("svdx_ln_bwd(param reduce)");
    }
    return 0;
}
