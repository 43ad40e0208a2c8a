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
// `pre(pass)` runs at the head of a pass's last K-step (the out-projection starts its residual loads there), `epi(pass, acc)` after it.
// Ends with every wave past a barrier and all its stores complete.
template <typename T, int KS, typename Pre, typename Epi, typename Grow>
__device__ __forceinline__ void tsa_band_gemm(const char* IMG, char* BST, const void* Bmat, int b_bytes, int N, int tid, Pre&& pre, Epi&& epi,
                                              void* side, int side_bytes, int R, Grow&& grow, unsigned long long* wait_cycles = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename TT<T>::v8 v8;
    constexpr int Kd = KS * 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int h = wave >> 2, cw = wave & 3;
    const int npass = (N + TSA_PW - 1) / TSA_PW, total = npass * KS;
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Bmat), 0, b_bytes, 0x00020000);
    int vob[TSA_NPC];
#pragma unroll
    for (int i = 0; i < TSA_NPC; ++i) {
        const int id = (i * TSA_WAVES + wave) * 64 + lane;       // 16-byte unit of the stage: LDS row = id / 8, physical chunk = id % 8
        const int r = id >> 3, pc = id & 7, lc = pc ^ (r & 7);
        const int n = (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);    // the weight row this LDS row holds
        vob[i] = (n * Kd + lc * 8) * 2;                          // rows beyond N lie beyond b_bytes: the descriptor returns zeros
    }
    int i_ks = 0, i_stage = 0, i_shift = 0;                      // the DMA's position: K-step, stage, byte shift of its pass
    auto issue_piece = [&](int i) __attribute__((always_inline)) {       // literal i
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(BST + i_stage * TSA_BST + (i * TSA_WAVES + wave) * 1024), 16,
                                                 vob[i] + i_shift, i_ks * 128, 0, 0);
    };
    auto issue_done = [&]() __attribute__((always_inline)) {
        i_stage ^= 1;
        if (++i_ks == KS) { i_ks = 0; i_shift += TSA_PW * Kd * 2; }
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
        for (int g = 0; g < TSA_NG; ++g) live[g] = pass * TSA_PW + (g * 4 + cw) * 32 < N;
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
#pragma unroll
            for (int i = 0; i < TSA_MBW; ++i)
                if (live[0] && rows_ok(i)) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[0][b][i] = TT<T>::mfma(b1[0][b], a1[i], acc[0][b][i]);
                    if (live[1]) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[1][b][i] = TT<T>::mfma(b1[1][b], a1[i], acc[1][b][i]);
                    }
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

    // ---- phase 1a: band rows -> LDS image (one DMA piece = 8 rows x 128 B of one 64-channel block) --------------------------------
    {
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
        const int npieces = KB * (TSA_RP / 8);
        for (int pq = wave; pq < npieces; pq += TSA_WAVES) {
            const int kblk = pq / (TSA_RP / 8), rg = pq - kblk * (TSA_RP / 8);
            const int row = rg * 8 + (lane >> 3), lc = (lane & 7) ^ (lane >> 3);
            const int voff = row < R ? (grow(row) * C + kblk * 64 + lc * 8) * 2 : (int)0x80000000;     // padding rows read zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(IMG + pq * 1024), 16, voff, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    TSA_STAMP(1);
    // ---- phase 1b: LayerNorm in place; 16 lanes per row, 4 rows per wave per step, 3 independent steps in flight; row quad
    //      (step * 8 + wave) of the band's 36.  The arithmetic is on float pairs (v_pk_add/mul/fma_f32): with two waves per SIMD this
    //      phase is VALU-bound.  n goes to HBM from the image during phase 2, under the MFMAs ----------------------------------------------
    {
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
    }
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
