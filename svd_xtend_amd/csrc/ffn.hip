// ffn.hip -- LayerNorm + the GEGLU projection of a transformer feed-forward as ONE kernel for gfx950:
//   n = LayerNorm(x);  [a | g] = n W1^T + b1;  pre = [a | g] (saved for the backward);  h = a * gelu(g)
// (diffusers FeedForward(activation_fn="geglu").net[0] behind norm3 / norm_in of BasicTransformerBlock and
// TemporalBasicTransformerBlock, instantiated at /root/reference/src/unet_spatio_temporal_condition.py:170-192).  At the 64x40 level of
// the benched shape this is a 35840 x 2560 x 320 projection: five K-steps per output tile, so the tiled GEMM (svdx_gemm with
// SVDX_EPI_GEGLU_FWD, 141 us) spends its time on prologues, epilogues and on staging the same 128 x 320 activation tile once per column
// tile.  Here a workgroup keeps a BAND of 144 consecutive rows resident in LDS (band.h: the swizzled image, LayerNorm in place) and
// streams all of W1 past it: the activation is staged once, LayerNorm costs no launch and no HBM round trip of n.
//
// Column pairing: a pass of the streaming GEMM covers 128 value columns and the 128 gate columns of the same indices (the weight rows
// of groups 4-7 of a stage are taken F rows further on), so the lane that holds 8 consecutive value columns holds their gates too and
// the epilogue is register-local: round, three 16-byte stores (pre value half, pre gate half, h).
// HBM traffic per row: read x once; write pre (2F), h (F), optionally n (C, for the weight gradient of a trainable W1) and (mean, rstd).
#include "common.h"
#include "band.h"

namespace {

struct LnGegluParams {
    const void* x; const float* gamma; const float* beta; float eps;
    const void* w1; const float* b1;
    void* n; float* stats; void* pre; void* hh;
    int M, F;
    int x_bytes, w_bytes, pre_bytes, hh_bytes;
};

template <typename T, int KB>
__global__ __launch_bounds__(512) void ln_geglu_kernel(LnGegluParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    constexpr int C = KB * 64;
    char* IMG = smem;                                  // [KB][TSA_RP rows][128 B], 16-byte chunk index XOR (row & 7)
    char* BST = smem + KB * (TSA_RP * 128);            // two weight stages
    const int row0 = blockIdx.x * TSA_RP;
    const int R = min(TSA_RP, p.M - row0);
    auto grow = [&](int lr) __attribute__((always_inline)) { return row0 + lr; };
    band_load<KB>(IMG, p.x, p.x_bytes, R, grow, tid);
    band_layernorm<T, KB>(IMG, p.gamma, p.beta, p.eps, p.stats, R, grow, tid);

    const int F = p.F;
    int rowoff_pre[TSA_MBW], rowoff_h[TSA_MBW];        // byte offsets of this lane's accumulator rows (block (wave / 4) * 5 + i, row + fr)
#pragma unroll
    for (int i = 0; i < TSA_MBW; ++i) {
        const int lr = ((wave >> 2) * TSA_MBW + i) * 16 + fr;
        rowoff_pre[i] = lr < R ? (row0 + lr) * (2 * F) * 2 : TSA_OOB;
        rowoff_h[i] = lr < R ? (row0 + lr) * F * 2 : TSA_OOB;
    }
    __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(p.pre, 0, p.pre_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(p.hh, 0, p.hh_bytes, 0x00020000);
    // The GEGLU of a finished pass (40 gelu evaluations per lane: ~1 k VALU instructions, as much issue time as a K-step's MFMAs) is
    // not computed behind the pass: the rounded pre-activations wait in `pend` and one row block per K-step of the NEXT pass is turned
    // into h and stored between that step's MFMA halves -- VALU and store issue under the matrix work instead of after it.
    tsa_u4 pend_a[TSA_MBW], pend_g[TSA_MBW];
    auto finish = [&](int i, int dpass) __attribute__((always_inline)) {                   // literal i
        const int n = dpass * 128 + (wave & 3) * 32 + fg * 8;    // the lane's 8 value columns; their gates are columns F + n ..
        const Vec8<T> a8 = __builtin_bit_cast(Vec8<T>, pend_a[i]), g8 = __builtin_bit_cast(Vec8<T>, pend_g[i]);
        Vec8<T> h8;
#pragma unroll
        for (int e = 0; e < 8; ++e) h8.v[e] = from_f<T>(to_f<T>(a8.v[e]) * gelu_erf(to_f<T>(g8.v[e])));    // from the ROUNDED values, like svdx_geglu_fwd
        const int op = dpass >= 0 ? rowoff_pre[i] : TSA_OOB, oh = dpass >= 0 ? rowoff_h[i] : TSA_OOB;
        tsa_store16(pend_a[i], rsP, op + n * 2, 0);
        tsa_store16(pend_g[i], rsP, op + (F + n) * 2, 0);
        tsa_store16(__builtin_bit_cast(tsa_u4, h8), rsH, oh + n * 2, 0);
    };
    constexpr int KS = KB;
    tsa_band_gemm<T, KB, 16, true>(IMG, BST, p.w1, p.w_bytes, 2 * F, tid, [&](int) __attribute__((always_inline)) {},
                         [&](int pass, int ks) __attribute__((always_inline)) {
        // unconditional (no branch: the work has to sit in the basic block of the MFMAs it is interleaved with); during pass 0 there
        // is nothing pending and the stores are sent out of range
#pragma unroll
        for (int i = 0; i < TSA_MBW; ++i)
            if (i % KS == ks) finish(i, pass - 1);
    },
                         [&](int pass, f32x4 (&acc)[TSA_NG][2][TSA_MBW]) __attribute__((always_inline)) {
        const int n = pass * 128 + (wave & 3) * 32 + fg * 8;
        float4 ba[2], bg[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ba[e] = *reinterpret_cast<const float4*>(p.b1 + n + 4 * e);
            bg[e] = *reinterpret_cast<const float4*>(p.b1 + F + n + 4 * e);
        }
        const float bav[8] = {ba[0].x, ba[0].y, ba[0].z, ba[0].w, ba[1].x, ba[1].y, ba[1].z, ba[1].w};
        const float bgv[8] = {bg[0].x, bg[0].y, bg[0].z, bg[0].w, bg[1].x, bg[1].y, bg[1].z, bg[1].w};
#pragma unroll
        for (int i = 0; i < TSA_MBW; ++i) {
            float a[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = acc[0][e >> 2][i][e & 3] + bav[e];
                g[e] = acc[1][e >> 2][i][e & 3] + bgv[e];
            }
            pend_a[i] = tsa_pack8<T>(a);
            pend_g[i] = tsa_pack8<T>(g);
        }
    }, p.n, p.x_bytes, R, grow, nullptr, F, 128);
    const int last = F / 128 - 1;
#pragma unroll
    for (int i = 0; i < TSA_MBW; ++i) finish(i, last);
#endif
}

}  // namespace

extern "C" int svdx_ln_geglu_rows_per_band(void) { return TSA_RP; }

extern "C" int svdx_ln_geglu_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1, void* n,
                                 float* stats, void* pre, void* hh, int M, int C, int F, int dtype, void* stream) {
    SVDX_CHECK_ARG(x && gamma && beta && w1 && b1 && stats && pre && hh, "svdx_ln_geglu_fwd: null argument");
    SVDX_CHECK_ARG(M > 0 && C > 0 && C % 64 == 0 && C <= TSA_MAXC && F > 0 && F % 128 == 0,
                   "svdx_ln_geglu_fwd: needs C a multiple of 64 up to %d and F a multiple of 128 (got C=%d F=%d)", TSA_MAXC, C, F);
    SVDX_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)pre | (uintptr_t)hh | (uintptr_t)n | (uintptr_t)b1) & 15) == 0,
                   "svdx_ln_geglu_fwd: operands must be 16-byte aligned");
    SVDX_CHECK_ARG((long)M * 2 * F * 2 < (1L << 31), "svdx_ln_geglu_fwd: pre-activation too large for 32-bit buffer offsets");
    LnGegluParams p;
    p.x = x; p.gamma = gamma; p.beta = beta; p.eps = eps; p.w1 = w1; p.b1 = b1; p.n = n; p.stats = stats; p.pre = pre; p.hh = hh;
    p.M = M; p.F = F;
    p.x_bytes = (int)((long)M * C * 2); p.w_bytes = 2 * F * C * 2; p.pre_bytes = (int)((long)M * 2 * F * 2); p.hh_bytes = (int)((long)M * F * 2);
    const int lds = (C / 64) * (TSA_RP * 128) + TSA_NSTG * TSA_BST;
    const int blocks = (M + TSA_RP - 1) / TSA_RP;
#define FFN_LAUNCH(KBV)                                                                                                               \
    case KBV: {                                                                                                                       \
        static bool attr_set = false;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_geglu_kernel<T, KBV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      KBV * (TSA_RP * 128) + TSA_NSTG * TSA_BST);                                                     \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((ln_geglu_kernel<T, KBV>), dim3(blocks), dim3(64 * TSA_WAVES), lds, (hipStream_t)stream, p);               \
    } break;
    DISPATCH_DTYPE(dtype, {
        switch (C / 64) { FFN_LAUNCH(1) FFN_LAUNCH(2) FFN_LAUNCH(3) FFN_LAUNCH(4) FFN_LAUNCH(5) }
    });
#undef FFN_LAUNCH
    SVDX_LAUNCH_CHECK("svdx_ln_geglu_fwd");
    return 0;
}
