// tsa.hip -- the temporal self-attention op of diffusers' TemporalBasicTransformerBlock as ONE kernel for gfx950
//   h1 = h + to_out( softmax_T( to_q(n) to_k(n)^T / 8 ) to_v(n) ) + bias + cross-attention row vector,   n = LayerNorm(h)
// (norm1 -> attn1 -> residual of the block instantiated at /root/reference/src/unet_spatio_temporal_condition.py:170-192; the
// trainable set of /root/reference/train_svd.py:761-766).  Unfused this is four launches and four HBM round trips of [M, C]-sized
// tensors (svdx_ln_fwd, the q/k/v GEMM, svdx_tattn_fwd, the out-projection GEMM).
//
// One workgroup owns a BAND: P neighbouring pixels of one clip x all T frames = P*T rows of the (b, t, y, x)-ordered activation
// (the rows of one pixel sit HW rows apart -- they are gathered by address, nothing is transposed).  With P*T = 140 rows at the
// 64x40 level of the benched shape, 35840 rows are exactly 256 bands = one per CU.
//   phase 1  the band's rows are DMA-ed (buffer_load ... lds) into an XOR-swizzled LDS image [C/64][144 rows][128 B] and
//            layer-normalised IN PLACE (16 lanes per row, fp32 statistics on float pairs); (mean, rstd) go to HBM here, n during
//            phase 2 -- the weight-gradient GEMM dW_qkv = dqkv^T n and the LayerNorm backward need them
//   phase 2  qkv = n W_qkv^T on v_mfma_f32_16x16x32: the image is the resident A operand, W_qkv streams through a double-buffered
//            LDS stage (lean buffer_load ... lds pieces, as gemm_v4); 8 waves = 2 row halves x 4 column groups, so every SIMD holds
//            two waves and one's LDS latency hides under the other's MFMAs; passes of 256 columns with the weight rows permuted so
//            that q, k, v leave as 16-byte stores (saved for the backward; read back by phase 3 from L2); the image is copied to n
//            one band row per K-step between the MFMA halves
//   phase 3  per (pixel, head): S^T = K Q^T (2 MFMAs, T padded to 16 and masked), softmax over the 4 lanes that share a query,
//            O^T = V^T P^T (4 MFMAs, V^T fragments by ds_read_b64_tr_b16 from a per-wave 2 KiB tile); O overwrites the image (the
//            out-projection's A operand)
//   phase 4  h1 = o W_o^T + b_o + cvec + h with the same streaming GEMM loop; the residual rows are fetched during the last K-step
//            of a pass; the image is copied to o (dW_o = dh1^T o needs it) as in phase 2
// HBM traffic per row: read h once (+ once more from L2 for the residual); write n, q, k, v, o, h1 once (all but h1 are needed by
// the backward).  Measured budget, the variants that were dropped and the wide-store data hazard: DESIGN.md section 6.
#include "common.h"

namespace {
struct TsaParams {
    const void* x; const float* gamma; const float* beta; float eps;
    const void* wqkv; const void* wo; const float* bo;
    const float* cvec; int rv_ld, rv_rpg, rv_mod;
    void* n1; float* stats; void* qkv; void* o; void* h1;
    int B, T, HW, C, heads, P; float sl2;
    int x_bytes, wqkv_bytes, wo_bytes;
#ifdef TSA_STAMPS
    unsigned long long* stamps;      // tools/probes/tsa_probe.hip: [block][16] shader-clock stamps of wave 0
#endif
};

#ifdef TSA_STAMPS
#define TSA_STAMP(i) do { if (tid == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
// phase 3 of wave 0: cycles between consecutive marks of the problem loop, summed into stamps[8 + i]
#define TSA_ACC(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if ((i) > 0) acc3[(i)] += now_ - last3; last3 = now_; } while (0)
#else
#define TSA_STAMP(i) do { } while (0)
#define TSA_ACC(i) do { } while (0)
#endif

typedef short tsa_v4s __attribute__((ext_vector_type(4)));
typedef short tsa_v8s __attribute__((ext_vector_type(8)));

// transposed fragment of a row-major [16 keys][64 d] tile (rows of 128 B, 16-byte chunks XOR-swizzled by row & 7): lane (fr, fg)
// receives X[key = fg*4 + j][d = db*16 + fr], j = 0..3, in k-slots 0..3; slots 4..7 are zero (T <= 16: one key block)
template <typename T>
__device__ __forceinline__ typename TT<T>::v8 tsa_frag_tr(const char* lds, int db, int fr, int fg) {
    const int u = db * 4 + (fr & 3);
    const int r0 = fg * 4 + (fr >> 2);
    const int a0 = r0 * 128 + (((u >> 1) ^ (r0 & 7)) * 16) + (u & 1) * 8;
    const tsa_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tsa_v4s __attribute__((address_space(3)))*)(lds + a0));
    const tsa_v8s r = {lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
    return __builtin_bit_cast(typename TT<T>::v8, r);
}

}  // namespace

#include "band.h"

namespace {

template <typename T, int KB>
__global__ __launch_bounds__(512) void tsa_fwd_kernel(TsaParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    constexpr int C = KB * 64, C8 = C / 8;
    const int Tn = p.T, HW = p.HW;
    char* IMG = smem;                                  // [KB][TSA_RP rows][128 B], 16-byte chunk index XOR (row & 7)
    char* BST = smem + KB * (TSA_RP * 128);            // two weight stages; per-wave V tiles in phase 3
    const int bands = HW / p.P;
    const int b = blockIdx.x / bands, p0 = (blockIdx.x - b * bands) * p.P;
    const int R = p.P * Tn;                            // real rows of the band; local row lr = pi * T + t
    const int row0 = b * Tn * HW + p0;                 // global row of (pixel p0, frame 0); frame t of pixel pi: row0 + t*HW + pi
    const int rcpT = (65536 + Tn - 1) / Tn;            // lr / T == (lr * rcpT) >> 16 for lr < 4096, T <= 16
    auto grow = [&](int lr) __attribute__((always_inline)) { const int pi = (lr * rcpT) >> 16; return row0 + (lr - pi * Tn) * HW + pi; };
    TSA_STAMP(0);
#ifdef TSA_STAMPS
    unsigned long long wait2[3] = {0, 0, 0}, wait4[3] = {0, 0, 0};
#define TSA_WAIT2 , wait2
#define TSA_WAIT4 , wait4
#else
#define TSA_WAIT2
#define TSA_WAIT4
#endif

    // ---- phase 1a: band rows -> LDS image ------------------------------------------------------------------------------------------
    band_load<KB>(IMG, p.x, p.x_bytes, R, grow, tid);
    // ---- phase 1b: LayerNorm in place; n goes to HBM from the image during phase 2, under the MFMAs -----------------------------------
    band_layernorm<T, KB>(IMG, p.gamma, p.beta, p.eps, p.stats, R, grow, tid);
    TSA_STAMP(2);
    // rows this lane owns in the accumulator layout: block (wave / 4) * 5 + i, row block * 16 + fr
    int growr[TSA_MBW];
#pragma unroll
    for (int i = 0; i < TSA_MBW; ++i) {
        const int lr = ((wave >> 2) * TSA_MBW + i) * 16 + fr;
        growr[i] = lr < R ? grow(lr) : -1;
    }

    // ---- phase 2: qkv = n W_qkv^T ------------------------------------------------------------------------------------------------
    T* QKV = reinterpret_cast<T*>(p.qkv);
    const int N3 = 3 * C;
    {
        int rowoff[TSA_MBW];
#pragma unroll
        for (int i = 0; i < TSA_MBW; ++i) rowoff[i] = growr[i] >= 0 ? growr[i] * N3 * 2 : TSA_OOB;
#ifdef TSA_SKIP_QKV_STORE
#pragma unroll
        for (int i = 0; i < TSA_MBW; ++i) if (p.T > 0) rowoff[i] = TSA_OOB;
#endif
        __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(p.qkv, 0, p.x_bytes * 3, 0x00020000);
        tsa_band_gemm<T, KB>(IMG, BST, p.wqkv, p.wqkv_bytes, N3, tid, [&](int) __attribute__((always_inline)) {},
                             [&](int, int) __attribute__((always_inline)) {},
                             [&](int pass, f32x4 (&acc)[TSA_NG][2][TSA_MBW]) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g) {
                const int n = pass * TSA_PW + (g * 4 + (wave & 3)) * 32 + fg * 8;
#pragma unroll
                for (int i = 0; i < TSA_MBW; ++i) {
                    const float f[8] = {acc[g][0][i][0], acc[g][0][i][1], acc[g][0][i][2], acc[g][0][i][3],
                                        acc[g][1][i][0], acc[g][1][i][1], acc[g][1][i][2], acc[g][1][i][3]};
                    tsa_store16(tsa_pack8<T>(f), rsQ, n < N3 ? rowoff[i] + n * 2 : TSA_OOB, 0);
                }
            }
        }, p.n1, p.x_bytes, R, grow TSA_WAIT2);
    }
    TSA_STAMP(3);
    // every wave's q/k/v stores are complete (vmcnt(0) before the last barrier) and visible to the other waves of this CU

    // ---- phase 3: attention over the frames of each (pixel, head); two problems per wave in flight, the next two being fetched ---------
    {
        char* VsA = BST + wave * 4096;
        char* VsB = VsA + 2048;
        const int nprob = p.P * p.heads;
        const int tq = min(fr, Tn - 1);
        struct Loaded { v8 q0, q1, k0, k1; uint4 va, vb; };
        auto load_prob = [&](int pr) __attribute__((always_inline)) {
            Loaded L;
            pr = min(pr, nprob - 1);
            const int pi = pr / p.heads, hd = pr - pi * p.heads;
            const T* base = QKV + (size_t)(row0 + pi) * N3 + hd * 64;
            const T* qr = base + (size_t)tq * HW * N3;
            L.q0 = *reinterpret_cast<const v8*>(qr + fg * 8);
            L.q1 = *reinterpret_cast<const v8*>(qr + 32 + fg * 8);
            L.k0 = *reinterpret_cast<const v8*>(qr + C + fg * 8);
            L.k1 = *reinterpret_cast<const v8*>(qr + C + 32 + fg * 8);
            const int r = lane >> 3, lc = (lane & 7) ^ (r & 7);
            L.va = *reinterpret_cast<const uint4*>(base + (size_t)min(r, Tn - 1) * HW * N3 + 2 * C + lc * 8);
            L.vb = *reinterpret_cast<const uint4*>(base + (size_t)min(r + 8, Tn - 1) * HW * N3 + 2 * C + lc * 8);   // (r + 8) & 7 == r & 7
            return L;
        };
        auto stage_v = [&](const Loaded& L, char* Vs) __attribute__((always_inline)) {
            *reinterpret_cast<uint4*>(Vs + (lane >> 3) * 128 + (lane & 7) * 16) = L.va;
            *reinterpret_cast<uint4*>(Vs + ((lane >> 3) + 8) * 128 + (lane & 7) * 16) = L.vb;
        };
        auto scores = [&](const Loaded& L) __attribute__((always_inline)) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            s = TT<T>::mfma(L.k0, L.q0, s);
            s = TT<T>::mfma(L.k1, L.q1, s);                          // s[e] = <q_fr, k_(4 fg + e)>
            float mx = -1e30f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (fg * 4 + e >= Tn) s[e] = -1e30f;                 // padded frames carry no weight
                mx = fmaxf(mx, s[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] = __builtin_amdgcn_exp2f((s[e] - mx) * p.sl2); sum += s[e]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            v8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf[e] = from_f<T>(s[e] * inv); pf[4 + e] = from_f<T>(0.f); }
            return pf;
        };
        auto output = [&](int pr, const v8& pf, const char* Vs) __attribute__((always_inline)) {
            const int pi = pr / p.heads, hd = pr - pi * p.heads;
            const int lr = pi * Tn + fr;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const v8 vf = tsa_frag_tr<T>(Vs, db, fr, fg);
                const f32x4 o4 = TT<T>::mfma(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f});     // o4[e] = O[frame fr][d = db*16 + 4 fg + e]
                if (fr < Tn) {
                    Vec4<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(o4[e]);
                    const int c = db * 2 + (fg >> 1);
                    *reinterpret_cast<Vec4<T>*>(IMG + hd * (TSA_RP * 128) + lr * 128 + ((c ^ (lr & 7)) * 16) + (fg & 1) * 8) = o;
                }
            }
        };
#ifdef TSA_STAMPS
        unsigned long long acc3[5] = {0, 0, 0, 0, 0}, last3 = 0;
#endif
        Loaded ca = load_prob(wave), cb = load_prob(wave + TSA_WAVES);
        for (int pr = wave; pr < nprob; pr += 2 * TSA_WAVES) {
            TSA_ACC(0);
            const Loaded na = load_prob(pr + 2 * TSA_WAVES), nb = load_prob(pr + 3 * TSA_WAVES);      // the next two problems travel while these are computed
            const bool has_b = pr + TSA_WAVES < nprob;
            TSA_ACC(1);
            stage_v(ca, VsA);
            if (has_b) stage_v(cb, VsB);
            TSA_ACC(2);
            const v8 pa = scores(ca);
            const v8 pb = scores(cb);
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's V tiles are in LDS
            __builtin_amdgcn_wave_barrier();
            TSA_ACC(3);
            output(pr, pa, VsA);
            if (has_b) output(pr + TSA_WAVES, pb, VsB);
            __builtin_amdgcn_wave_barrier();
            TSA_ACC(4);
            ca = na;
            cb = nb;
        }
#ifdef TSA_STAMPS
        const unsigned long long t3 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef TSA_STAMPS
        if (tid == 0) {
            for (int i = 1; i < 5; ++i) p.stamps[blockIdx.x * 16 + 8 + i] = acc3[i];
            p.stamps[blockIdx.x * 16 + 13] = __builtin_readcyclecounter() - t3;
        }
#endif
    }
    TSA_STAMP(4);

    // ---- phase 4: h1 = o W_o^T + b_o + cvec + h ----------------------------------------------------------------------------------
    {
        int rowoff[TSA_MBW];
#pragma unroll
        for (int i = 0; i < TSA_MBW; ++i) rowoff[i] = growr[i] >= 0 ? growr[i] * C * 2 : TSA_OOB;
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(p.h1, 0, p.x_bytes, 0x00020000);
        tsa_u4 r8[TSA_NG][TSA_MBW];
        tsa_band_gemm<T, KB>(IMG, BST, p.wo, p.wo_bytes, C, tid,
                             [&](int pass) __attribute__((always_inline)) {
            // the residual rows of this pass travel while its last K-step runs (buffer loads: rows of the padding and columns beyond
            // C read zeros)
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g) {
                const int n = pass * TSA_PW + (g * 4 + (wave & 3)) * 32 + fg * 8;
#pragma unroll
                for (int i = 0; i < TSA_MBW; ++i)
                    r8[g][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, n < C ? rowoff[i] + n * 2 : TSA_OOB, 0, 0);
            }
        },
                             [&](int, int) __attribute__((always_inline)) {},
                             [&](int pass, f32x4 (&acc)[TSA_NG][2][TSA_MBW]) __attribute__((always_inline)) {
            int nb[TSA_NG];
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g) nb[g] = pass * TSA_PW + (g * 4 + (wave & 3)) * 32 + fg * 8;
            float4 bb[TSA_NG][2];
#pragma unroll
            for (int g = 0; g < TSA_NG; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e) bb[g][e] = *reinterpret_cast<const float4*>(p.bo + min(nb[g], C - 8) + 4 * e);
#pragma unroll
            for (int i = 0; i < TSA_MBW; ++i) {
                const int gr = max(growr[i], 0);
                const float* rv = p.cvec ? p.cvec + (size_t)(p.rv_mod ? gr % p.rv_mod : gr / p.rv_rpg) * p.rv_ld : nullptr;
                float4 rv4[TSA_NG][2];
#pragma unroll
                for (int g = 0; g < TSA_NG; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        rv4[g][e] = rv ? *reinterpret_cast<const float4*>(rv + min(nb[g], C - 8) + 4 * e) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < TSA_NG; ++g) {
                    const Vec8<T> rr = __builtin_bit_cast(Vec8<T>, r8[g][i]);
                    const float add[8] = {bb[g][0].x + rv4[g][0].x, bb[g][0].y + rv4[g][0].y, bb[g][0].z + rv4[g][0].z, bb[g][0].w + rv4[g][0].w,
                                          bb[g][1].x + rv4[g][1].x, bb[g][1].y + rv4[g][1].y, bb[g][1].z + rv4[g][1].z, bb[g][1].w + rv4[g][1].w};
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = acc[g][e >> 2][i][e & 3] + add[e] + to_f<T>(rr.v[e]);
                    tsa_store16(tsa_pack8<T>(o), rsH, nb[g] < C ? rowoff[i] + nb[g] * 2 : TSA_OOB, 0);
                }
            }
        }, p.o, p.x_bytes, R, grow TSA_WAIT4);
    }
    TSA_STAMP(5);
#ifdef TSA_STAMPS
    if (tid == 0) {
        p.stamps[blockIdx.x * 16 + 6] = wait2[0]; p.stamps[blockIdx.x * 16 + 7] = wait4[0];
        p.stamps[blockIdx.x * 16 + 14] = (wait2[1] << 32) | wait2[2]; p.stamps[blockIdx.x * 16 + 15] = (wait4[1] << 32) | wait4[2];
    }
#endif
#endif
}

}  // namespace

extern "C" int svdx_tsa_pixels_per_band(int T, int HW) {
    if (T <= 0 || T > 16 || HW <= 0) return 0;
    int best = 0;
    for (int P = 1; P * T <= TSA_RP && P <= HW; ++P)
        if (HW % P == 0) best = P;
    return best;
}

extern "C" int svdx_tsa_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* wqkv, const void* wo,
                            const float* bo, const float* cvec, int rv_ld, int rv_rows_per_group, int rv_mod, void* n1, float* stats,
                            void* qkv, void* o, void* h1, int B, int T, int HW, int C, int heads, float scale, int dtype, void* stream) {
    SVDX_CHECK_ARG(x && gamma && beta && wqkv && wo && bo && stats && qkv && o && h1, "svdx_tsa_fwd: null argument");
    SVDX_CHECK_ARG(B > 0 && T > 0 && T <= 16 && HW > 0 && C % 64 == 0 && C <= TSA_MAXC && heads * 64 == C,
                   "svdx_tsa_fwd: needs T <= 16, C = 64 * heads <= %d (got T=%d C=%d heads=%d)", TSA_MAXC, T, C, heads);
    SVDX_CHECK_ARG(!cvec || rv_mod > 0 || rv_rows_per_group > 0, "svdx_tsa_fwd: cvec needs a grouping");
    SVDX_CHECK_ARG(!cvec || ((((uintptr_t)cvec) & 15) == 0 && rv_ld % 4 == 0), "svdx_tsa_fwd: cvec rows must be 16-byte aligned");
    SVDX_CHECK_ARG((((uintptr_t)bo) & 15) == 0, "svdx_tsa_fwd: bo must be 16-byte aligned");
    SVDX_CHECK_ARG((((uintptr_t)x | (uintptr_t)wqkv | (uintptr_t)wo | (uintptr_t)qkv | (uintptr_t)o | (uintptr_t)h1 | (uintptr_t)n1) & 15) == 0,
                   "svdx_tsa_fwd: operands must be 16-byte aligned");
    const long M = (long)B * T * HW;
    SVDX_CHECK_ARG(M * 3 * C * 2 < (1L << 31), "svdx_tsa_fwd: activation too large for 32-bit buffer offsets");
    TsaParams p;
    p.x = x; p.gamma = gamma; p.beta = beta; p.eps = eps; p.wqkv = wqkv; p.wo = wo; p.bo = bo;
    p.cvec = cvec; p.rv_ld = rv_ld; p.rv_rpg = rv_rows_per_group; p.rv_mod = rv_mod;
    p.n1 = n1; p.stats = stats; p.qkv = qkv; p.o = o; p.h1 = h1;
    p.B = B; p.T = T; p.HW = HW; p.C = C; p.heads = heads; p.P = svdx_tsa_pixels_per_band(T, HW); p.sl2 = scale * TSA_LOG2E;
    p.x_bytes = (int)(M * C * 2); p.wqkv_bytes = 3 * C * C * 2; p.wo_bytes = C * C * 2;
    SVDX_CHECK_ARG(p.P > 0, "svdx_tsa_fwd: no band size for T=%d HW=%d", T, HW);
    const int lds = (C / 64) * (TSA_RP * 128) + TSA_NSTG * TSA_BST;
    const int blocks = B * (HW / p.P);
#define TSA_LAUNCH(KBV)                                                                                                              \
    case KBV: {                                                                                                                       \
        static bool attr_set = false;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsa_fwd_kernel<T, KBV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      KBV * (TSA_RP * 128) + TSA_NSTG * TSA_BST);                                                     \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((tsa_fwd_kernel<T, KBV>), dim3(blocks), dim3(64 * TSA_WAVES), lds, (hipStream_t)stream, p);                           \
    } break;
    DISPATCH_DTYPE(dtype, {
        switch (C / 64) { TSA_LAUNCH(1) TSA_LAUNCH(2) TSA_LAUNCH(3) TSA_LAUNCH(4) TSA_LAUNCH(5) }
    });
#undef TSA_LAUNCH
    SVDX_LAUNCH_CHECK("svdx_tsa_fwd");
    return 0;
}
