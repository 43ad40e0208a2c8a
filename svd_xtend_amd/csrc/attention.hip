// attention.hip -- self-attention kernels of the SVD UNet for gfx950 (head_dim 64).
//
// Spatial attention (sequence = HW): flash-style, MFMA 16x16x32, fp32 online softmax.
//   The score tile is computed TRANSPOSED (S^T = K Q^T) so that every lane owns one query column: row max /
//   sum are in-lane reductions plus two shuffles, the O rescale is a per-lane scalar, and the probabilities
//   feed the second MFMA (O^T = V^T P^T) straight from their accumulator registers -- no LDS round trip.
//   The k-slot <-> key mapping of that MFMA is arbitrary as long as both operands agree, and the V^T operand (8 keys of one
//   head dimension per lane) is read from the ROW-major V tile with ds_read_b64_tr_b16, gfx950's transposing LDS read
//   (frag_tr).  The backward uses the same trick in both orientations (dQ kernel: lanes own queries, K^T from the K tile;
//   dK/dV kernel: lanes own keys, Q^T / dO^T from the Q / dO tiles): no head-transposed copies exist in HBM or LDS.
//   LDS tiles are row-major (128-byte rows), XOR-swizzled on the 16-byte chunk index: plain and transposing fragment reads
//   are both bank-conflict free.
//
// Temporal attention (sequence = frames): T <= 32, so each (clip, pixel, head) problem is one
//   wave of plain VALU work on data addressed in place with stride HW*ld -- the (B*T,HW,C)<->(B*HW,T,C)
//   transposes of the reference never happen.  The op is HBM-bound (AI = T/2 flop/B).
#include "common.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
// raw v_exp_f32: inputs here are <= ~8 and underflow to 0 is exactly what masked / far-below-max scores want
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float RESCALE_THR = 8.f;   // defer the O/l rescale while the running max grows by less than this (log2 units)

// ---- LDS tile staging -------------------------------------------------------------------------------------------
// 64 rows x 64 halfs, row r taken from base + row_index(r)*ld (rows clamped to s_max-1), chunk-swizzled.
template <typename T>
__device__ __forceinline__ void stage_rows(char* lds, const T* base, size_t ld, int row0, int s_max, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * 256 + tid;
        const int r = id >> 3, pc = id & 7, lc = pc ^ (r & 7);
        const int row = min(row0 + r, s_max - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + lc * 8);
        *reinterpret_cast<uint4*>(lds + r * 128 + pc * 16) = v;
    }
}
// two-phase staging (global -> registers early, registers -> LDS after the barrier): the loads of tile t+1 are in flight while
// tile t is being multiplied (cdna_hip_programming.md T14)
struct Tile2 { uint4 a, b; };       // returned / passed BY VALUE so the staging registers stay in SSA form (no scratch)
template <typename T>
__device__ __forceinline__ Tile2 load_rows(const T* base, size_t ld, int row0, int s_max, int tid) {
    Tile2 r;
    {
        const int id = tid, rr = id >> 3, pc = id & 7, lc = pc ^ (rr & 7);
        r.a = *reinterpret_cast<const uint4*>(base + (size_t)min(row0 + rr, s_max - 1) * ld + lc * 8);
    }
    {
        const int id = 256 + tid, rr = id >> 3, pc = id & 7, lc = pc ^ (rr & 7);
        r.b = *reinterpret_cast<const uint4*>(base + (size_t)min(row0 + rr, s_max - 1) * ld + lc * 8);
    }
    return r;
}
__device__ __forceinline__ void store_rows(char* lds, const Tile2 r, int tid) {
    *reinterpret_cast<uint4*>(lds + (tid >> 3) * 128 + (tid & 7) * 16) = r.a;
    *reinterpret_cast<uint4*>(lds + ((256 + tid) >> 3) * 128 + (tid & 7) * 16) = r.b;
}
// row-major fragment: row (blk*16 + fr), logical 16-byte chunk (ks*4 + fg)
template <typename T>
__device__ __forceinline__ typename TT<T>::v8 frag_rows(const char* lds, int blk, int ks, int fr, int fg) {
    return *reinterpret_cast<const typename TT<T>::v8*>(lds + (blk * 16 + fr) * 128 + (((ks * 4 + fg) ^ (fr & 7)) * 16));
}
// transposed fragment (row d = db*16 + fr of X^T; k-slots 0..3 <-> positions pb0*16 + fg*4 + e, slots 4..7 <-> pb1*16 + fg*4 + e)
// read straight from the ROW-major tile X (rows = positions, the swizzle of store_rows): gfx950's transposing LDS read
// ds_read_b64_tr_b16 hands lane (g, i) of a 16-lane group the (i&3)-th half of the 8-byte unit addressed by lane (g, 4j + (i>>2))
// for j = 0..3 (probed: tools/probes/tr_probe.hip).  Lane (g, i') therefore points at row pb*16 + g*4 + (i'>>2), unit
// db*4 + (i'&3), and lane (fg, fr) receives X[pb*16 + fg*4 + j][db*16 + fr] -- with no transposed
// copy in HBM or LDS.  Rows r..r+7 of one read land in distinct 16-byte chunks per bank half: conflict-free with this swizzle.
template <typename T>
__device__ __forceinline__ typename TT<T>::v8 frag_tr(const char* lds, int db, int pb0, int pb1, int fr, int fg) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const int u = db * 4 + (fr & 3);
    const int r0 = pb0 * 16 + fg * 4 + (fr >> 2), r1 = pb1 * 16 + fg * 4 + (fr >> 2);
    const int a0 = r0 * 128 + (((u >> 1) ^ (r0 & 7)) * 16) + (u & 1) * 8;
    const int a1 = r1 * 128 + (((u >> 1) ^ (r1 & 7)) * 16) + (u & 1) * 8;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + a0));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + a1));
    const v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename TT<T>::v8, r);
}
template <typename T>
__device__ __forceinline__ typename TT<T>::v8 pack8(const f32x4& a, const f32x4& b) {
    typename TT<T>::v8 r;
    r[0] = from_f<T>(a[0]); r[1] = from_f<T>(a[1]); r[2] = from_f<T>(a[2]); r[3] = from_f<T>(a[3]);
    r[4] = from_f<T>(b[0]); r[5] = from_f<T>(b[1]); r[6] = from_f<T>(b[2]); r[7] = from_f<T>(b[3]);
    return r;
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const f32x4& v, float mul) {
    Vec4<T> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(v[e] * mul);
    *reinterpret_cast<Vec4<T>*>(p) = o;
}

// ---- which (128-row tile, head, sample) a workgroup computes.  Workgroup ids go round-robin to the 8 XCDs (id & 7), each with a private
// 4 MiB L2, and every tile of one (head, sample) pair streams that pair's WHOLE K / V (forward, dQ) or Q / dO (dK/dV): a (tile, head, sample)
// grid in dispatch order spread the 20 tiles of a pair over all eight XCDs, each of which then pulled the pair's operands over the fabric
// for itself -- 403 MB read per forward launch at the 64x40 level where q, k, v are 69 MB (profiles/r5_pmc_traffic_by_grid.txt).  Round 6: a
// 1-D grid of 8 * ceil(pairs / 8) * tiles workgroups; the j-th workgroup of XCD c is tile j % tiles of pair (j / tiles) * 8 + c, so the
// tiles of a pair run side by side on ONE XCD (three pairs at a time fill its 64 workgroup slots: 2 MB of K / V in its L2).
struct AttnWho { int tile, h, n; bool live; };
__device__ __forceinline__ AttnWho attn_who(int tiles, int heads, int pairs) {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int pl = slot / tiles, pair = pl * 8 + xcd;
    AttnWho w;
    w.tile = slot - pl * tiles;
    w.live = pair < pairs;
    w.n = pair / heads;
    w.h = pair - w.n * heads;
    return w;
}
static inline unsigned attn_grid(int tiles, int pairs) { return 8u * (unsigned)((pairs + 7) / 8) * (unsigned)tiles; }

// ================================================================================================================
// forward: block = (128 queries, head, frame); 4 waves x 32 queries; KV tiles of 64 keys
// ================================================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, T* __restrict__ o, float* __restrict__ lse,
                                                       int heads, int S, int ld, int ld_o, float sl2, int pairs) {
    typedef typename TT<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
    char* Ks = smem;
    char* Vs = smem + 64 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const AttnWho who = attn_who((S + 127) >> 7, heads, pairs);
    if (!who.live) return;
    const int h = who.h, n = who.n;
    const int q0 = who.tile * 128 + wave * 32;
    const T* qb_ = q + (size_t)n * S * ld + h * 64;
    const T* kb_ = k + (size_t)n * S * ld + h * 64;
    const T* vb_ = v + (size_t)n * S * ld + h * 64;

    v8 qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int qi = min(q0 + qb * 16 + fr, S - 1);
            qf[qb][ks] = *reinterpret_cast<const v8*>(qb_ + (size_t)qi * ld + ks * 32 + fg * 8);
        }
    f32x4 oacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[2] = {-1e30f, -1e30f}, lrow[2] = {0.f, 0.f};

    const int ntiles = (S + 63) / 64;
    Tile2 rk = load_rows<T>(kb_, ld, 0, S, tid);
    Tile2 rv = load_rows<T>(vb_, ld, 0, S, tid);
    // one KV/Q tile; the out-of-range mask exists only in the instance that runs the last, partial tile
    auto tile = [&](const int t, auto tail_c) __attribute__((always_inline)) {
        constexpr bool tail = decltype(tail_c)::value;
        __syncthreads();
        store_rows(Ks, rk, tid);
        store_rows(Vs, rv, tid);
        __syncthreads();
        {   // unconditional (the last iteration re-reads its own tile): keeps the staging registers out of scratch
            const int tn = min(t + 1, ntiles - 1);
            rk = load_rows<T>(kb_, ld, tn * 64, S, tid);
            rv = load_rows<T>(vb_, ld, tn * 64, S, tid);
        }
        f32x4 s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const v8 kf = frag_rows<T>(Ks, kb, ks, fr, fg);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) s[kb][qb] = TT<T>::mfma(kf, qf[qb][ks], s[kb][qb]);
            }
        if constexpr (tail) {                                               // compiled into the partial-tile instance only
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 64 + kb * 16 + fg * 4 + r >= S) s[kb][qb][r] = -1e30f;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mx = fmaxf(mx * sl2, -1e30f);                           // scaled (log2) domain; keeps -1e30 finite
            // deferred rescale: keep the old reference max while the new one is within RESCALE_THR (P stays <= 2^8)
            if (!__all(mx - mrow[qb] <= RESCALE_THR)) {
                const float mn = fmaxf(mrow[qb], mx);
                const float alpha = fexp2(mrow[qb] - mn);
                mrow[qb] = mn;
                lrow[qb] *= alpha;
#pragma unroll
                for (int db = 0; db < 4; ++db) oacc[db][qb] *= alpha;
            }
            const float nmref = -mrow[qb];
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {                        // whole-vector arithmetic: packed fp32 FMA / add
                const f32x4 e = s[kb][qb] * sl2 + nmref;
                const f32x4 pr = {fexp2(e[0]), fexp2(e[1]), fexp2(e[2]), fexp2(e[3])};
                s[kb][qb] = pr;
                ps += pr;
            }
            lrow[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            v8 pf[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) pf[qb] = pack8<T>(s[2 * k2][qb], s[2 * k2 + 1][qb]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const v8 vf = frag_tr<T>(Vs, db, 2 * k2, 2 * k2 + 1, fr, fg);     // V^T fragment from the row-major V tile
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) oacc[db][qb] = TT<T>::mfma(vf, pf[qb], oacc[db][qb]);
            }
        }
    };
    for (int t = 0; t < (S >> 6); ++t) tile(t, std::false_type{});
    if (S & 63) tile(ntiles - 1, std::true_type{});
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float lt = lrow[qb];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const int qi = q0 + qb * 16 + fr;
        if (qi < S) {
            const float inv = 1.f / lt;
            T* op = o + (size_t)(n * S + qi) * ld_o + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) store4<T>(op + db * 16, oacc[db][qb], inv);
            if (fg == 0) lse[(size_t)(n * heads + h) * S + qi] = (mrow[qb] + log2f(lt)) * LN2;
        }
    }
}

template <typename T>
__global__ void attn_bwd_prep_kernel(const T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ D, int nb,
                                     int heads, int S, int ld_o) {
    const long n = (long)nb * S * heads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % heads);
        const long row = i / heads;
        const T* op = o + row * ld_o + h * 64;
        const T* dp = d_o + row * ld_o + h * 64;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a[8], b[8];
            load8<T>(op + c * 8, a);
            load8<T>(dp + c * 8, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += a[e] * b[e];
        }
        const long nn = row / S;
        const int s = (int)(row - nn * S);
        D[(nn * heads + h) * S + s] = acc;
    }
}

// ================================================================================================================
// backward dQ: same orientation as the forward (lanes own queries), loops over KV tiles
// ================================================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                          const T* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ Dv,
                                                          T* __restrict__ dq, int heads, int S, int ld, int ld_o, int ld_d,
                                                          float scale, int pairs) {
    typedef typename TT<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
    char* Ks = smem;
    char* Vs = smem + 64 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const AttnWho who = attn_who((S + 127) >> 7, heads, pairs);
    if (!who.live) return;
    const int h = who.h, n = who.n;
    const int q0 = who.tile * 128 + wave * 32;
    const float sl2 = scale * LOG2E;
    const T* qb_ = q + (size_t)n * S * ld + h * 64;
    const T* kb_ = k + (size_t)n * S * ld + h * 64;
    const T* vb_ = v + (size_t)n * S * ld + h * 64;
    const T* dob = d_o + (size_t)n * S * ld_o + h * 64;

    v8 qf[2][2], dof[2][2];
    float lse2[2], dd[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = min(q0 + qb * 16 + fr, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qb][ks] = *reinterpret_cast<const v8*>(qb_ + (size_t)qi * ld + ks * 32 + fg * 8);
            dof[qb][ks] = *reinterpret_cast<const v8*>(dob + (size_t)qi * ld_o + ks * 32 + fg * 8);
        }
        lse2[qb] = -lse[(size_t)(n * heads + h) * S + qi] * LOG2E;       // both kept NEGATED: the loop adds them (packed fp32 add / FMA)
        dd[qb] = -Dv[(size_t)(n * heads + h) * S + qi];
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (S + 63) / 64;
    Tile2 rk = load_rows<T>(kb_, ld, 0, S, tid);
    Tile2 rv = load_rows<T>(vb_, ld, 0, S, tid);
    // one KV/Q tile; the out-of-range mask exists only in the instance that runs the last, partial tile
    auto tile = [&](const int t, auto tail_c) __attribute__((always_inline)) {
        constexpr bool tail = decltype(tail_c)::value;
        __syncthreads();
        store_rows(Ks, rk, tid);
        store_rows(Vs, rv, tid);
        __syncthreads();
        {
            const int tn = min(t + 1, ntiles - 1);
            rk = load_rows<T>(kb_, ld, tn * 64, S, tid);
            rv = load_rows<T>(vb_, ld, tn * 64, S, tid);
        }
        f32x4 s[4][2], dp[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const v8 kf = frag_rows<T>(Ks, kb, ks, fr, fg);
                const v8 vf = frag_rows<T>(Vs, kb, ks, fr, fg);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    s[kb][qb] = TT<T>::mfma(kf, qf[qb][ks], s[kb][qb]);
                    dp[kb][qb] = TT<T>::mfma(vf, dof[qb][ks], dp[kb][qb]);
                }
            }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                // whole-vector arithmetic: packed fp32 FMA / add / mul (two elements per VALU issue)
                const f32x4 e = s[kb][qb] * sl2 + lse2[qb];
                const f32x4 g = dp[kb][qb] + dd[qb];
                const f32x4 pr = {fexp2(e[0]), fexp2(e[1]), fexp2(e[2]), fexp2(e[3])};
                s[kb][qb] = pr * g;                                   // dS^T
                if (kb & 1) __builtin_amdgcn_sched_barrier(0);        // bounds how many blocks the scheduler keeps in flight (144 vs 196 VGPRs)
            }
        if constexpr (tail) {                                               // compiled into the partial-tile instance only
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 64 + kb * 16 + fg * 4 + r >= S) s[kb][qb][r] = 0.f;
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            v8 df[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) df[qb] = pack8<T>(s[2 * k2][qb], s[2 * k2 + 1][qb]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const v8 ktf = frag_tr<T>(Ks, db, 2 * k2, 2 * k2 + 1, fr, fg);     // K^T fragment from the K tile already staged
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) acc[db][qb] = TT<T>::mfma(ktf, df[qb], acc[db][qb]);
            }
        }
    };
    for (int t = 0; t < (S >> 6); ++t) tile(t, std::false_type{});
    if (S & 63) tile(ntiles - 1, std::true_type{});
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        if (qi < S) {
            T* op = dq + (size_t)(n * S + qi) * ld_d + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) store4<T>(op + db * 16, acc[db][qb], scale);
        }
    }
}

// ================================================================================================================
// backward dK/dV: lanes own keys (block = 128 keys, 4 waves x 32 keys), loops over query tiles of 64
// ================================================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                           const T* __restrict__ d_o, const float* __restrict__ lse,
                                                           const float* __restrict__ Dv, T* __restrict__ dk, T* __restrict__ dv,
                                                           int heads, int S, int ld, int ld_o, int ld_d, float scale, int pairs) {
    typedef typename TT<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128 + 512];
    char* Qs = smem;
    char* dOs = smem + 64 * 128;
    float* Ls = reinterpret_cast<float*>(smem + 2 * 64 * 128);   // [64] -lse*log2e, then [64] -D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const AttnWho who = attn_who((S + 127) >> 7, heads, pairs);
    if (!who.live) return;
    const int h = who.h, n = who.n;
    const int k0 = who.tile * 128 + wave * 32;
    const float sl2 = scale * LOG2E;
    const T* qb_ = q + (size_t)n * S * ld + h * 64;
    const T* kb_ = k + (size_t)n * S * ld + h * 64;
    const T* vb_ = v + (size_t)n * S * ld + h * 64;
    const T* dob = d_o + (size_t)n * S * ld_o + h * 64;
    const float* lsb = lse + (size_t)(n * heads + h) * S;
    const float* dvb = Dv + (size_t)(n * heads + h) * S;

    v8 kf[2][2], vf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int ki = min(k0 + kb * 16 + fr, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kb][ks] = *reinterpret_cast<const v8*>(kb_ + (size_t)ki * ld + ks * 32 + fg * 8);
            vf[kb][ks] = *reinterpret_cast<const v8*>(vb_ + (size_t)ki * ld + ks * 32 + fg * 8);
        }
    }
    f32x4 dka[4][2], dva[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dka[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int ntiles = (S + 63) / 64;
    Tile2 rq, rdo;
    float rls = 0.f;
#define SVDX_DKV_PREFETCH(t_)                                               \
    {                                                                       \
        const int tt_ = (t_);                                               \
        rq = load_rows<T>(qb_, ld, tt_ * 64, S, tid);                       \
        rdo = load_rows<T>(dob, ld_o, tt_ * 64, S, tid);                    \
        const int qi_ = min(tt_ * 64 + (tid & 63), S - 1);                  \
        rls = tid < 64 ? -lsb[qi_] * LOG2E : -dvb[qi_];                     \
    }
    SVDX_DKV_PREFETCH(0);
    // one KV/Q tile; the out-of-range mask exists only in the instance that runs the last, partial tile
    auto tile = [&](const int t, auto tail_c) __attribute__((always_inline)) {
        constexpr bool tail = decltype(tail_c)::value;
        __syncthreads();
        store_rows(Qs, rq, tid);
        store_rows(dOs, rdo, tid);
        if (tid < 128) Ls[tid] = rls;
        __syncthreads();
        SVDX_DKV_PREFETCH(min(t + 1, ntiles - 1));
        // S[q,key] and dP[q,key]: rows = queries (A from LDS), cols = this lane's key (B resident)
        f32x4 s[4][2], dp[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const v8 qf = frag_rows<T>(Qs, qb, ks, fr, fg);
                const v8 df = frag_rows<T>(dOs, qb, ks, fr, fg);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[qb][kb] = TT<T>::mfma(qf, kf[kb][ks], s[qb][kb]);
                    dp[qb][kb] = TT<T>::mfma(df, vf[kb][ks], dp[qb][kb]);
                }
            }
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qb * 16 + fg * 4);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ls + 64 + qb * 16 + fg * 4);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                // whole-vector arithmetic: packed fp32 FMA / add / mul (two elements per VALU issue)
                const f32x4 e = s[qb][kb] * sl2 + l4;
                const f32x4 pr = {fexp2(e[0]), fexp2(e[1]), fexp2(e[2]), fexp2(e[3])};
                s[qb][kb] = pr;
                dp[qb][kb] = pr * (dp[qb][kb] + d4);               // dS
            }
        }
        if constexpr (tail) {                                               // compiled into the partial-tile instance only
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 64 + qb * 16 + fg * 4 + r >= S) { s[qb][kb][r] = 0.f; dp[qb][kb][r] = 0.f; }
        }
        // dV^T[d,key] += dO^T[d,q] P[q,key] ;  dK^T[d,key] += Q^T[d,q] dS[q,key]
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            v8 pf[2], sf[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                pf[kb] = pack8<T>(s[2 * k2][kb], s[2 * k2 + 1][kb]);
                sf[kb] = pack8<T>(dp[2 * k2][kb], dp[2 * k2 + 1][kb]);
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const v8 dotf = frag_tr<T>(dOs, db, 2 * k2, 2 * k2 + 1, fr, fg);    // dO^T / Q^T fragments from the row-major tiles
                const v8 qtf = frag_tr<T>(Qs, db, 2 * k2, 2 * k2 + 1, fr, fg);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    dva[db][kb] = TT<T>::mfma(dotf, pf[kb], dva[db][kb]);
                    dka[db][kb] = TT<T>::mfma(qtf, sf[kb], dka[db][kb]);
                }
            }
        }
    };
    for (int t = 0; t < (S >> 6); ++t) tile(t, std::false_type{});
    if (S & 63) tile(ntiles - 1, std::true_type{});
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int ki = k0 + kb * 16 + fr;
        if (ki < S) {
            T* kp = dk + (size_t)(n * S + ki) * ld_d + h * 64 + fg * 4;
            T* vp = dv + (size_t)(n * S + ki) * ld_d + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                store4<T>(kp + db * 16, dka[db][kb], scale);
                store4<T>(vp + db * 16, dva[db][kb], 1.f);
            }
        }
    }
}

}  // namespace

#define ATTN_ARGS_OK(ld_, ptr_) ((ld_) % 8 == 0 && (((uintptr_t)(ptr_)) & 15) == 0)

extern "C" int svdx_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int nb, int heads, int S, int ld,
                             int ld_o, float scale, int dtype, void* stream) {
    SVDX_CHECK_ARG(q && k && v && o && lse && nb > 0 && heads > 0 && S > 0, "svdx_attn_fwd: bad args");
    SVDX_CHECK_ARG(ATTN_ARGS_OK(ld, q) && ATTN_ARGS_OK(ld, k) && ATTN_ARGS_OK(ld, v) && ld_o % 4 == 0 && (((uintptr_t)o) & 7) == 0,
                   "svdx_attn_fwd: alignment (ld=%d ld_o=%d)", ld, ld_o);
    dim3 grid(attn_grid(cdiv(S, 128), heads * nb));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q, (const T*)k,
                                             (const T*)v, (T*)o, lse, heads, S, ld, ld_o, scale * LOG2E, heads * nb));
    SVDX_LAUNCH_CHECK("svdx_attn_fwd");
    return 0;
}

extern "C" int svdx_attn_bwd_prep(const void* o, const void* d_o, float* D, int nb, int heads, int S, int ld_o, int dtype, void* stream) {
    SVDX_CHECK_ARG(o && d_o && D && ATTN_ARGS_OK(ld_o, o) && ATTN_ARGS_OK(ld_o, d_o), "svdx_attn_bwd_prep: bad args");
    const long n = (long)nb * S * heads;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_bwd_prep_kernel<T>), dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)o, (const T*)d_o, D, nb, heads, S, ld_o));
    SVDX_LAUNCH_CHECK("svdx_attn_bwd_prep");
    return 0;
}

extern "C" int svdx_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* D,
                                 void* dk, void* dv, int nb, int heads, int S, int ld, int ld_o, int ld_d, float scale, int dtype,
                                 void* stream) {
    SVDX_CHECK_ARG(q && k && v && d_o && lse && D && dk && dv, "svdx_attn_bwd_dkv: null argument");
    SVDX_CHECK_ARG(ATTN_ARGS_OK(ld, q) && ATTN_ARGS_OK(ld, k) && ATTN_ARGS_OK(ld, v) && ATTN_ARGS_OK(ld_o, d_o) && ld_d % 4 == 0,
                   "svdx_attn_bwd_dkv: alignment");
    dim3 grid(attn_grid(cdiv(S, 128), heads * nb));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q,
                                             (const T*)k, (const T*)v, (const T*)d_o, lse, D, (T*)dk, (T*)dv, heads, S, ld, ld_o,
                                             ld_d, scale, heads * nb));
    SVDX_LAUNCH_CHECK("svdx_attn_bwd_dkv");
    return 0;
}

extern "C" int svdx_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* D,
                                void* dq, int nb, int heads, int S, int ld, int ld_o, int ld_d, float scale, int dtype,
                                void* stream) {
    SVDX_CHECK_ARG(q && k && v && d_o && lse && D && dq, "svdx_attn_bwd_dq: null argument");
    SVDX_CHECK_ARG(ATTN_ARGS_OK(ld, q) && ATTN_ARGS_OK(ld, k) && ATTN_ARGS_OK(ld, v) && ATTN_ARGS_OK(ld_o, d_o) && ld_d % 4 == 0,
                   "svdx_attn_bwd_dq: alignment");
    dim3 grid(attn_grid(cdiv(S, 128), heads * nb));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q,
                                             (const T*)k, (const T*)v, (const T*)d_o, lse, D, (T*)dq, heads, S, ld, ld_o, ld_d,
                                             scale, heads * nb));
    SVDX_LAUNCH_CHECK("svdx_attn_bwd_dq");
    return 0;
}
