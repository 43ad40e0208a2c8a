// tattn.hip -- temporal self-attention core on the matrix pipe (gfx950): attention over the T <= 32 frames of one pixel, per head.
//
// Replaces the SDPA call of diffusers' AttnProcessor2_0 inside TemporalBasicTransformerBlock.attn1
// (/root/reference/src/unet_spatio_temporal_condition.py:170-192 instantiates the blocks; the trainable set, train_svd.py:761-766, is
// exactly these blocks, so this op runs forward AND backward in all 16 of them) wherever the whole-op kernel of csrc/tsa.hip does not
// apply -- every backward, and the forward at C > 320, T > 16 or with LoRA adapters on attn1.
//
// One wave owns one (clip, pixel, head) problem: Q, K, V are [T, 64] slices of the fused q/k/v rows, T*HW rows apart per frame --
// nothing is transposed or gathered in HBM.  A row of a slice is 128 contiguous bytes, which is exactly the operand layout of
// v_mfma_f32_16x16x32 when the contraction runs over the head dimension: lane (r = lane & 15, g = lane >> 4) of a fragment holds
// elements [32 kk + 8 g, +8) of row r, ONE 16-byte load straight from HBM, usable as the A or the B operand alike.  So
//     S^T = K Q^T        (lane: column t = r, rows s = 4 g + e)      and      S = Q K^T      (lane: column s = r, rows t = 4 g + e)
// cost two MFMAs each and give the scores in BOTH register layouts; the softmax runs on the four values a lane holds plus two
// cross-row-group shuffles (layout 1) or a 16-lane DPP reduction (layout 2).  The second matmuls contract over frames -- K = 16,
// v_mfma_f32_16x16x16 -- with the probabilities (or dS) as the B operand straight from the registers they were computed in, and the
// [T, 64] operand transposed on the way out of LDS by ds_read_b64_tr_b16 (a lane receives 4 consecutive frames of one column).
// Results come out as O^T / dQ^T / dK^T / dV^T blocks: a lane owns 4 consecutive head-dim columns of one frame -> 8-byte stores.
//
// Forward: 2 + 4 MFMAs per problem (T <= 16) against ~1.1 k VALU FMAs per lane in the round-1 kernel; backward: 8 + 12 against ~5 k.
// The op moves 4 (forward) / 7 (backward) slices of T*128 bytes per problem and is HBM-bound by a wide margin (7 flop/B).
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int PITCH = 144;                       // bytes per LDS row of a [T, 64] slice: 16-byte aligned, rows 36 banks apart

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma16;
template <> struct Mfma16<f16> {
    static __device__ __forceinline__ f32x4 run(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<bf16> {
    static __device__ __forceinline__ f32x4 run(bf16x4 a, bf16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
    return v;
}

// fragments of a [T, 64] slice: f[tb][kk] = elements [32 kk + 8 g, +8) of row 16 tb + r (zeros beyond frame Tn)
template <typename T, int TB>
__device__ __forceinline__ void load_frags(typename TT<T>::v8 (&f)[TB][2], const T* base, size_t tstride, int Tn, int r, int g) {
    typedef typename TT<T>::v8 v8;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        const int row = tb * 16 + r;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (T)0.f;
            f[tb][kk] = row < Tn ? *reinterpret_cast<const v8*>(base + (size_t)row * tstride + kk * 32 + g * 8) : z;
        }
    }
}

// park the fragments of a slice as a row-major [16 TB][64] LDS tile (for the transposing reads below)
template <typename T, int TB>
__device__ __forceinline__ void park_frags(char* tile, const typename TT<T>::v8 (&f)[TB][2], int r, int g) {
    typedef typename TT<T>::v8 v8;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) *reinterpret_cast<v8*>(tile + (tb * 16 + r) * PITCH + (kk * 32 + g * 8) * 2) = f[tb][kk];
}

// X^T operand of v_mfma_f32_16x16x16 out of a row-major tile: rows d = 16 j + r of X^T, k = frames 16 fb + 4 g + {0..3}
template <typename T>
__device__ __forceinline__ typename TT<T>::v4 tr_frag(const char* tile, int fb, int j, int r, int g) {
    const int a = (fb * 16 + g * 4 + (r >> 2)) * PITCH + (j * 16 + (r & 3) * 4) * 2;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + a));
    return __builtin_bit_cast(typename TT<T>::v4, v);
}

template <typename T>
__device__ __forceinline__ typename TT<T>::v4 pack4(const f32x4& v) {
    typename TT<T>::v4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
    return o;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, const f32x4& v, float mul) {
    Vec4<T> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(v[e] * mul);
    *reinterpret_cast<Vec4<T>*>(p) = o;
}

// softmax of the score blocks in layout 1 (S^T: lane column t = 16 tb + r, rows s = 16 sb + 4 g + e): in place -> probabilities
template <int TB>
__device__ __forceinline__ void softmax_cols(f32x4 (&st)[TB][TB], int Tn, int g, float sl2) {
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        float mx = -1e30f;
#pragma unroll
        for (int sb = 0; sb < TB; ++sb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = (sb * 16 + g * 4 + e) < Tn ? st[sb][tb][e] * sl2 : -1e30f;
                st[sb][tb][e] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int sb = 0; sb < TB; ++sb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { st[sb][tb][e] = __builtin_amdgcn_exp2f(st[sb][tb][e] - mx); sum += st[sb][tb][e]; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int sb = 0; sb < TB; ++sb) st[sb][tb] *= inv;
    }
}

// the same in layout 2 (S: lane column s = 16 sb + r, rows t = 16 tb + 4 g + e)
template <int TB>
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[TB][TB], int Tn, int r, float sl2) {
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mx = -1e30f;
#pragma unroll
            for (int sb = 0; sb < TB; ++sb) {
                const float x = (sb * 16 + r) < Tn ? s[tb][sb][e] * sl2 : -1e30f;
                s[tb][sb][e] = x;
                mx = fmaxf(mx, x);
            }
            mx = row16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int sb = 0; sb < TB; ++sb) { s[tb][sb][e] = __builtin_amdgcn_exp2f(s[tb][sb][e] - mx); sum += s[tb][sb][e]; }
            const float inv = __builtin_amdgcn_rcpf(row16_sum(sum));
#pragma unroll
            for (int sb = 0; sb < TB; ++sb) s[tb][sb][e] *= inv;
        }
}

template <typename T, int TB>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ o, int Tn, int HW, int heads, int ld, int ld_o, float sl2, long nprob) {
    typedef typename TT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    char* Vl = smem + wave * (TB * 16 * PITCH);
    const size_t ts = (size_t)HW * ld;
    for (long prob = (long)blockIdx.x * 4 + wave; prob < nprob; prob += (long)gridDim.x * 4) {
        const int h = (int)(prob % heads);
        const long bp = prob / heads;
        const int p = (int)(bp % HW);
        const size_t row0 = (size_t)(bp / HW) * Tn * HW + p;
        v8 qf[TB][2], kf[TB][2], vf[TB][2];
        load_frags<T, TB>(qf, q + row0 * ld + h * 64, ts, Tn, r, g);
        load_frags<T, TB>(kf, k + row0 * ld + h * 64, ts, Tn, r, g);
        load_frags<T, TB>(vf, v + row0 * ld + h * 64, ts, Tn, r, g);
        park_frags<T, TB>(Vl, vf, r, g);
        f32x4 st[TB][TB];                                      // S^T blocks [key block][query block]
#pragma unroll
        for (int sb = 0; sb < TB; ++sb)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                st[sb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) st[sb][tb] = TT<T>::mfma(kf[sb][kk], qf[tb][kk], st[sb][tb]);
            }
        softmax_cols<TB>(st, Tn, g, sl2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS writes have landed before it reads them back
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int t = tb * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sb = 0; sb < TB; ++sb) acc = Mfma16<T>::run(tr_frag<T>(Vl, sb, j, r, g), pack4<T>(st[sb][tb]), acc);
                if (t < Tn) store4<T>(o + (row0 + (size_t)t * HW) * ld_o + h * 64 + j * 16 + g * 4, acc, 1.f);
            }
        }
        __builtin_amdgcn_wave_barrier();                        // the tile is rewritten by the next problem
    }
}

template <typename T, int TB>
__global__ __launch_bounds__(256) void tattn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        const T* __restrict__ d_o, T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv,
                                                        int Tn, int HW, int heads, int ld, int ld_o, int ld_d, float scale, long nprob) {
    typedef typename TT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = TB * 16 * PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    char* Ql = smem + wave * (3 * TILE);
    char* Kl = Ql + TILE;
    char* Dl = Kl + TILE;
    const float sl2 = scale * LOG2E;
    for (long prob = (long)blockIdx.x * 4 + wave; prob < nprob; prob += (long)gridDim.x * 4) {
        const int h = (int)(prob % heads);
        const long bp = prob / heads;
        const int p = (int)(bp % HW);
        const size_t row0 = (size_t)(bp / HW) * Tn * HW + p;
        v8 qf[TB][2], kf[TB][2], vf[TB][2], df[TB][2];
        load_frags<T, TB>(qf, q + row0 * ld + h * 64, (size_t)HW * ld, Tn, r, g);
        load_frags<T, TB>(kf, k + row0 * ld + h * 64, (size_t)HW * ld, Tn, r, g);
        load_frags<T, TB>(vf, v + row0 * ld + h * 64, (size_t)HW * ld, Tn, r, g);
        load_frags<T, TB>(df, d_o + row0 * ld_o + h * 64, (size_t)HW * ld_o, Tn, r, g);
        park_frags<T, TB>(Ql, qf, r, g);
        park_frags<T, TB>(Kl, kf, r, g);
        park_frags<T, TB>(Dl, df, r, g);
        // scores and dP = dO V^T in both layouts: 1 = [key block][query block] (lane: column t, rows s), 2 = [query block][key block]
        f32x4 p1[TB][TB], p2[TB][TB], g1[TB][TB], g2[TB][TB];
#pragma unroll
        for (int a = 0; a < TB; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                p1[a][b] = p2[a][b] = g1[a][b] = g2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    p1[a][b] = TT<T>::mfma(kf[a][kk], qf[b][kk], p1[a][b]);       // S^T  [s block a][t block b]
                    p2[a][b] = TT<T>::mfma(qf[a][kk], kf[b][kk], p2[a][b]);       // S    [t block a][s block b]
                    g1[a][b] = TT<T>::mfma(vf[a][kk], df[b][kk], g1[a][b]);       // dP^T [s block a][t block b]
                    g2[a][b] = TT<T>::mfma(df[a][kk], vf[b][kk], g2[a][b]);       // dP   [t block a][s block b]
                }
            }
        softmax_cols<TB>(p1, Tn, g, sl2);
        softmax_rows<TB>(p2, Tn, r, sl2);
        // dS = P (dP - sum_s P dP) * scale, in both layouts (g1 / g2 become dS)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            float dl = 0.f;
#pragma unroll
            for (int sb = 0; sb < TB; ++sb)
#pragma unroll
                for (int e = 0; e < 4; ++e) dl += p1[sb][tb][e] * g1[sb][tb][e];
            dl += __shfl_xor(dl, 16, 64);
            dl += __shfl_xor(dl, 32, 64);
#pragma unroll
            for (int sb = 0; sb < TB; ++sb)
#pragma unroll
                for (int e = 0; e < 4; ++e) g1[sb][tb][e] = p1[sb][tb][e] * (g1[sb][tb][e] - dl) * scale;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d2 = 0.f;
#pragma unroll
                for (int sb = 0; sb < TB; ++sb) d2 += p2[tb][sb][e] * g2[tb][sb][e];
                d2 = row16_sum(d2);
#pragma unroll
                for (int sb = 0; sb < TB; ++sb) g2[tb][sb][e] = p2[tb][sb][e] * (g2[tb][sb][e] - d2) * scale;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // dQ^T[d, t] = sum_s K[s, d] dS[t, s]          (contraction over keys: dS in layout 1 is the B operand as it stands)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int t = tb * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sb = 0; sb < TB; ++sb) acc = Mfma16<T>::run(tr_frag<T>(Kl, sb, j, r, g), pack4<T>(g1[sb][tb]), acc);
                if (t < Tn) store4<T>(dq + (row0 + (size_t)t * HW) * ld_d + h * 64 + j * 16 + g * 4, acc, 1.f);
            }
        }
        // dV^T[d, s] = sum_t dO[t, d] P[t, s]   and   dK^T[d, s] = sum_t Q[t, d] dS[t, s]      (contraction over queries: layout 2)
#pragma unroll
        for (int sb = 0; sb < TB; ++sb) {
            const int s = sb * 16 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f}, ak = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    av = Mfma16<T>::run(tr_frag<T>(Dl, tb, j, r, g), pack4<T>(p2[tb][sb]), av);
                    ak = Mfma16<T>::run(tr_frag<T>(Ql, tb, j, r, g), pack4<T>(g2[tb][sb]), ak);
                }
                if (s < Tn) {
                    store4<T>(dv + (row0 + (size_t)s * HW) * ld_d + h * 64 + j * 16 + g * 4, av, 1.f);
                    store4<T>(dk + (row0 + (size_t)s * HW) * ld_d + h * 64 + j * 16 + g * 4, ak, 1.f);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#define TATTN_ALIGNED(ld, ptr) ((ld) % 8 == 0 && ((uintptr_t)(ptr) & 15) == 0)

template <typename T, int TB>
void launch_fwd(const void* q, const void* k, const void* v, void* o, int Tn, int HW, int heads, int ld, int ld_o, float scale, long nprob,
                hipStream_t st) {
    const int blocks = (int)std::min<long>((nprob + 3) / 4, 256L * 16);
    hipLaunchKernelGGL((tattn_fwd_kernel<T, TB>), dim3(blocks), dim3(256), 4 * TB * 16 * PITCH, st, (const T*)q, (const T*)k, (const T*)v, (T*)o,
                       Tn, HW, heads, ld, ld_o, scale * LOG2E, nprob);
}

template <typename T, int TB>
void launch_bwd(const void* q, const void* k, const void* v, const void* d_o, void* dq, void* dk, void* dv, int Tn, int HW, int heads, int ld,
                int ld_o, int ld_d, float scale, long nprob, hipStream_t st) {
    const int blocks = (int)std::min<long>((nprob + 3) / 4, 256L * 16);
    hipLaunchKernelGGL((tattn_bwd_kernel<T, TB>), dim3(blocks), dim3(256), 4 * 3 * TB * 16 * PITCH, st, (const T*)q, (const T*)k, (const T*)v,
                       (const T*)d_o, (T*)dq, (T*)dk, (T*)dv, Tn, HW, heads, ld, ld_o, ld_d, scale, nprob);
}

}  // namespace

extern "C" int svdx_tattn_fwd(const void* q, const void* k, const void* v, void* o, int B, int Tn, int HW, int heads, int ld,
                              int ld_o, float scale, int dtype, void* stream) {
    SVDX_CHECK_ARG(q && k && v && o && B > 0 && Tn > 0 && Tn <= 32 && HW > 0 && heads > 0, "svdx_tattn_fwd: bad args (T<=32)");
    SVDX_CHECK_ARG(TATTN_ALIGNED(ld, q) && TATTN_ALIGNED(ld, k) && TATTN_ALIGNED(ld, v) && ld_o % 4 == 0 && ((uintptr_t)o & 7) == 0,
                   "svdx_tattn_fwd: alignment");
    const long nprob = (long)B * HW * heads;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        if (Tn <= 16) launch_fwd<T, 1>(q, k, v, o, Tn, HW, heads, ld, ld_o, scale, nprob, st);
        else launch_fwd<T, 2>(q, k, v, o, Tn, HW, heads, ld, ld_o, scale, nprob, st);
    });
    SVDX_LAUNCH_CHECK("svdx_tattn_fwd");
    return 0;
}

extern "C" int svdx_tattn_bwd(const void* q, const void* k, const void* v, const void* d_o, void* dq, void* dk, void* dv, int B,
                              int Tn, int HW, int heads, int ld, int ld_o, int ld_d, float scale, int dtype, void* stream) {
    SVDX_CHECK_ARG(q && k && v && d_o && dq && dk && dv && B > 0 && Tn > 0 && Tn <= 32, "svdx_tattn_bwd: bad args (T<=32)");
    SVDX_CHECK_ARG(TATTN_ALIGNED(ld, q) && TATTN_ALIGNED(ld, k) && TATTN_ALIGNED(ld, v) && TATTN_ALIGNED(ld_o, d_o) && ld_d % 4 == 0 &&
                       (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) == 0, "svdx_tattn_bwd: alignment");
    const long nprob = (long)B * HW * heads;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        if (Tn <= 16) launch_bwd<T, 1>(q, k, v, d_o, dq, dk, dv, Tn, HW, heads, ld, ld_o, ld_d, scale, nprob, st);
        else launch_bwd<T, 2>(q, k, v, d_o, dq, dk, dv, Tn, HW, heads, ld, ld_o, ld_d, scale, nprob, st);
    });
    SVDX_LAUNCH_CHECK("svdx_tattn_bwd");
    return 0;
}
