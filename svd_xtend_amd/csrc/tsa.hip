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
//            layer-normalised IN PLACE (16 lanes per row, fp32 statistics); n and (mean, rstd) also go to HBM -- the weight-gradient
//            GEMM dW_qkv = dqkv^T n and the LayerNorm backward need them
//   phase 2  qkv = n W_qkv^T on v_mfma_f32_16x16x32: the image is the resident A operand, W_qkv streams through a double-buffered
//            LDS stage (lean buffer_load ... lds pieces, as gemm_v4); 4 waves x 48 columns per pass, 9 x 3 accumulator blocks per wave;
//            q, k, v go to HBM (saved for the backward) and are read back by phase 3 from L2
//   phase 3  per (pixel, head): S^T = K Q^T (2 MFMAs, T padded to 16 and masked), softmax over the 4 lanes that share a query,
//            O^T = V^T P^T (4 MFMAs, V^T fragments by ds_read_b64_tr_b16 from a per-wave 2 KiB tile); O overwrites the image (the
//            out-projection's A operand) and goes to HBM (dW_o = dh1^T o needs it)
//   phase 4  h1 = o W_o^T + b_o + cvec + h with the same streaming GEMM loop; the residual is re-read from L2
// HBM traffic per row: read h once; write n, q, k, v, o, h1 once (all but h1 are needed by the backward).
#include "common.h"

namespace {

constexpr float TSA_LOG2E = 1.4426950408889634f;
constexpr int TSA_MB = 9;                      // 16-row blocks per band (up to 144 rows)
constexpr int TSA_RP = TSA_MB * 16;
constexpr int TSA_NB = 3;                      // 16-column blocks per wave per pass
constexpr int TSA_WN = 16 * TSA_NB;            // 48 columns per wave
constexpr int TSA_PW = 4 * TSA_WN;             // 192 columns per pass
constexpr int TSA_BST = TSA_PW * 128;          // bytes of one B stage (K extent 64)
constexpr int TSA_NPC = TSA_BST / 1024 / 4;    // DMA pieces per wave per stage (6)
constexpr int TSA_MAXC = 320;

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
#else
#define TSA_STAMP(i) do { } while (0)
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

// acc[j][i] = IMG[rows i*16.., K] * Bmat[N, K]^T for the wave's 48 columns of every 192-column pass; `epi(pass, acc)` runs when a
// pass has seen all of K (the loop body is gemm_v4's: fragments of one 32-deep half step in registers, then its 27 MFMAs).  The first
// stage of the next pass is already travelling when the epilogue runs.  Ends with every wave past a barrier and its stores complete.
template <typename T, typename Epi>
__device__ __forceinline__ void tsa_band_gemm(const char* IMG, char* BST, const void* Bmat, int b_bytes, int N, int Kd, int tid, Epi&& epi,
                                              unsigned long long* wait_cycles = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename TT<T>::v8 v8;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int KS = Kd / 64, npass = (N + TSA_PW - 1) / TSA_PW;
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Bmat), 0, b_bytes, 0x00020000);
    int vob[TSA_NPC];
    auto set_pass = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TSA_NPC; ++i) {
            const int id = (i * 4 + wave) * 64 + lane;           // 16-byte unit of the stage: row = id / 8, physical chunk = id % 8
            const int r = id >> 3, pc = id & 7, lc = pc ^ (r & 7);
            const int n = pass * TSA_PW + r;
            vob[i] = n < N ? (n * Kd + lc * 8) * 2 : (int)0x80000000;   // beyond N: out of range of the buffer -> zeros
        }
    };
    auto issue = [&](int stage, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TSA_NPC; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(BST + stage * TSA_BST + (i * 4 + wave) * 1024), 16,
                                                     vob[i], ks * 128, 0, 0);
    };
    set_pass(0);
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        f32x4 acc[TSA_NB][TSA_MB];
#pragma unroll
        for (int j = 0; j < TSA_NB; ++j)
#pragma unroll
            for (int i = 0; i < TSA_MB; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool live = pass * TSA_PW + wave * TSA_WN < N;     // a wave whose 48 columns lie beyond N idles through the pass
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                issue(cur ^ 1, ks + 1);
            } else if (pass + 1 < npass) {
                set_pass(pass + 1);
                issue(cur ^ 1, 0);
            }
            if (live) {
                const char* As = IMG + ks * (TSA_RP * 128);
                const char* Bs = BST + cur * TSA_BST;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int chunk = ((kk * 4 + fg) ^ (fr & 7)) * 16;
                    v8 af[TSA_MB], bf[TSA_NB];
#pragma unroll
                    for (int i = 0; i < TSA_MB; ++i) af[i] = *reinterpret_cast<const v8*>(As + (i * 16 + fr) * 128 + chunk);
#pragma unroll
                    for (int j = 0; j < TSA_NB; ++j) bf[j] = *reinterpret_cast<const v8*>(Bs + (wave * TSA_WN + j * 16 + fr) * 128 + chunk);
#pragma unroll
                    for (int i = 0; i < TSA_MB; ++i)
#pragma unroll
                        for (int j = 0; j < TSA_NB; ++j) acc[j][i] = TT<T>::mfma(bf[j], af[i], acc[j][i]);
                }
            }
#ifdef TSA_STAMPS
            const unsigned long long w0 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef TSA_STAMPS
            if (wait_cycles) *wait_cycles += __builtin_readcyclecounter() - w0;
#endif
            cur ^= 1;
        }
        if (live) epi(pass, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

template <typename T>
__global__ __launch_bounds__(256, 2) void tsa_fwd_kernel(TsaParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename TT<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int C = p.C, KB = C / 64, C8 = C / 8, Tn = p.T, HW = p.HW;
    char* IMG = smem;                                  // [KB][TSA_RP rows][128 B], 16-byte chunk index XOR (row & 7)
    char* BST = smem + KB * (TSA_RP * 128);            // two weight stages; per-wave V tiles in phase 3
    const int bands = HW / p.P;
    const int b = blockIdx.x / bands, p0 = (blockIdx.x - b * bands) * p.P;
    const int R = p.P * Tn;                            // real rows of the band; local row lr = pi * T + t
    const int row0 = b * Tn * HW + p0;                 // global row of (pixel p0, frame 0); frame t of pixel pi: row0 + t*HW + pi
    auto grow = [&](int lr) __attribute__((always_inline)) { const int pi = lr / Tn; return row0 + (lr - pi * Tn) * HW + pi; };
    const T* X = reinterpret_cast<const T*>(p.x);
    TSA_STAMP(0);
#ifdef TSA_STAMPS
    unsigned long long wait2 = 0, wait4 = 0;
#define TSA_WAIT2 , &wait2
#define TSA_WAIT4 , &wait4
#else
#define TSA_WAIT2
#define TSA_WAIT4
#endif

    // ---- phase 1a: band rows -> LDS image (one DMA piece = 8 rows x 128 B of one 64-channel block) --------------------------------
    {
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
        const int npieces = KB * (TSA_RP / 8);
        for (int pq = wave; pq < npieces; pq += 4) {
            const int kblk = pq / (TSA_RP / 8), rg = pq - kblk * (TSA_RP / 8);
            const int row = rg * 8 + (lane >> 3), lc = (lane & 7) ^ (lane >> 3);
            const int voff = row < R ? (grow(row) * C + kblk * 64 + lc * 8) * 2 : (int)0x80000000;     // padding rows read zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(IMG + pq * 1024), 16, voff, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    TSA_STAMP(1);
    // ---- phase 1b: LayerNorm in place; 16 lanes per row, 4 rows per wave per step, 3 independent steps in flight -----------------------
    {
        const int l16 = lane & 15;
        constexpr int NCH = (TSA_MAXC / 8 + 15) / 16;       // 16-byte chunks per lane (3 at C = 320)
        constexpr int G = 3;                                // row groups in flight
        float gm[NCH][8], bt[NCH][8];
        bool cv[NCH];
        int cl[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = l16 + 16 * j;
            cv[j] = c < C8;
            cl[j] = min(c, C8 - 1);                         // invalid chunks read a valid address and are masked to zero
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gm[j][e] = p.gamma[cl[j] * 8 + e];
                bt[j][e] = p.beta[cl[j] * 8 + e];
            }
        }
        T* N1 = reinterpret_cast<T*>(p.n1);
        const float invC = 1.f / (float)C;
        for (int it0 = 0; it0 < TSA_RP / 16; it0 += G) {
            float v[G][NCH][8], mean[G], rstd[G];
            int row[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                row[g] = wave * (TSA_RP / 4) + (it0 + g) * 4 + (lane >> 4);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    load8<T>(reinterpret_cast<const T*>(IMG + (cl[j] >> 3) * (TSA_RP * 128) + row[g] * 128 + (((cl[j] & 7) ^ (row[g] & 7)) * 16)), v[g][j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[g][j][e] = cv[j] ? v[g][j][e] : 0.f;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += v[g][j][e];
                mean[g] = row16_sum(s) * invC;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = cv[j] ? v[g][j][e] - mean[g] : 0.f; ss += d * d; }
                rstd[g] = rsqrtf(row16_sum(ss) * invC + p.eps);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool real = row[g] < R;
                const int gr = real ? grow(row[g]) : 0;
                if (real && l16 == 0) *reinterpret_cast<float2*>(p.stats + (size_t)gr * 2) = float2{mean[g], rstd[g]};
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = real ? (v[g][j][e] - mean[g]) * rstd[g] * gm[j][e] + bt[j][e] : 0.f;
                    if (cv[j]) {
                        store8<T>(reinterpret_cast<T*>(IMG + (cl[j] >> 3) * (TSA_RP * 128) + row[g] * 128 + (((cl[j] & 7) ^ (row[g] & 7)) * 16)), o);
                        if (real && N1) store8<T>(N1 + (size_t)gr * C + cl[j] * 8, o);
                    }
                }
            }
        }
        __syncthreads();
    }
    TSA_STAMP(2);
    // rows this lane owns in the accumulator layout: block i, row i*16 + fr
    int growr[TSA_MB];
#pragma unroll
    for (int i = 0; i < TSA_MB; ++i) growr[i] = (i * 16 + fr) < R ? grow(i * 16 + fr) : -1;

    // ---- phase 2: qkv = n W_qkv^T ------------------------------------------------------------------------------------------------
    T* QKV = reinterpret_cast<T*>(p.qkv);
    const int N3 = 3 * C;
    tsa_band_gemm<T>(IMG, BST, p.wqkv, p.wqkv_bytes, N3, C, tid, [&](int pass, f32x4 (&acc)[TSA_NB][TSA_MB]) __attribute__((always_inline)) {
        const int nb = pass * TSA_PW + wave * TSA_WN + fg * 4;
#ifdef TSA_SKIP_QKV_STORE
        if (p.T > 0) return;
#endif
#pragma unroll
        for (int i = 0; i < TSA_MB; ++i) {
            if (growr[i] < 0) continue;
            T* dst = QKV + (size_t)growr[i] * N3 + nb;
#pragma unroll
            for (int j = 0; j < TSA_NB; ++j) {
                if (nb + j * 16 < N3) {
                    Vec4<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(acc[j][i][e]);
                    *reinterpret_cast<Vec4<T>*>(dst + j * 16) = o;
                }
            }
        }
    } TSA_WAIT2);
    TSA_STAMP(3);
    // every wave's q/k/v stores are complete (vmcnt(0) before the last barrier) and visible to the other waves of this CU

    // ---- phase 3: attention over the frames of each (pixel, head); two problems per wave in flight, the next two being fetched ---------
    {
        T* Og = reinterpret_cast<T*>(p.o);
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
                    *reinterpret_cast<Vec4<T>*>(Og + (size_t)(row0 + pi + fr * HW) * C + hd * 64 + db * 16 + fg * 4) = o;
                }
            }
        };
        Loaded ca = load_prob(wave), cb = load_prob(wave + 4);
        for (int pr = wave; pr < nprob; pr += 8) {
            const Loaded na = load_prob(pr + 8), nb = load_prob(pr + 12);      // the next two problems travel while these are computed
            const bool has_b = pr + 4 < nprob;
            stage_v(ca, VsA);
            if (has_b) stage_v(cb, VsB);
            const v8 pa = scores(ca);
            const v8 pb = scores(cb);
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's V tiles are in LDS
            __builtin_amdgcn_wave_barrier();
            output(pr, pa, VsA);
            if (has_b) output(pr + 4, pb, VsB);
            __builtin_amdgcn_wave_barrier();
            ca = na;
            cb = nb;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    TSA_STAMP(4);

    // ---- phase 4: h1 = o W_o^T + b_o + cvec + h ----------------------------------------------------------------------------------
    T* H1 = reinterpret_cast<T*>(p.h1);
    tsa_band_gemm<T>(IMG, BST, p.wo, p.wo_bytes, C, C, tid, [&](int pass, f32x4 (&acc)[TSA_NB][TSA_MB]) __attribute__((always_inline)) {
        const int nb = pass * TSA_PW + wave * TSA_WN + fg * 4;
        // residual rows first: the compiler cannot know that h1 does not alias h, and a load issued between two stores costs a memory
        // round trip (27 of them in a chain: 61 k cycles in the first version of this epilogue)
        Vec4<T> r4[TSA_NB][TSA_MB];
#pragma unroll
        for (int j = 0; j < TSA_NB; ++j)
#pragma unroll
            for (int i = 0; i < TSA_MB; ++i) {
                const bool ok = growr[i] >= 0 && nb + j * 16 < C;
                r4[j][i] = *reinterpret_cast<const Vec4<T>*>(X + (size_t)(ok ? growr[i] : row0) * C + (ok ? nb + j * 16 : 0));
            }
        float bb[TSA_NB][4];
#pragma unroll
        for (int j = 0; j < TSA_NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[j][e] = (nb + j * 16 < C) ? p.bo[nb + j * 16 + e] : 0.f;
#pragma unroll
        for (int i = 0; i < TSA_MB; ++i) {
            const int gr = growr[i];
            if (gr < 0) continue;
            const float* rv = p.cvec ? p.cvec + (size_t)(p.rv_mod ? gr % p.rv_mod : gr / p.rv_rpg) * p.rv_ld : nullptr;
#pragma unroll
            for (int j = 0; j < TSA_NB; ++j) {
                const int n = nb + j * 16;
                if (n < C) {
                    Vec4<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(acc[j][i][e] + bb[j][e] + (rv ? rv[n + e] : 0.f) + to_f<T>(r4[j][i].v[e]));
                    *reinterpret_cast<Vec4<T>*>(H1 + (size_t)gr * C + n) = o;
                }
            }
        }
    } TSA_WAIT4);
    TSA_STAMP(5);
#ifdef TSA_STAMPS
    if (tid == 0) { p.stamps[blockIdx.x * 16 + 6] = wait2; p.stamps[blockIdx.x * 16 + 7] = wait4; }
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
    const int lds = (C / 64) * (TSA_RP * 128) + 2 * TSA_BST;
    const int blocks = B * (HW / p.P);
    DISPATCH_DTYPE(dtype, {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsa_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (TSA_MAXC / 64) * (TSA_RP * 128) + 2 * TSA_BST);
            attr_set = true;
        }
        hipLaunchKernelGGL((tsa_fwd_kernel<T>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, p);
    });
    SVDX_LAUNCH_CHECK("svdx_tsa_fwd");
    return 0;
}
