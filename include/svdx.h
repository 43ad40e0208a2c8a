/* svdx.h -- C-ABI of libsvdx.so: the MI355X (gfx950) kernels behind the SVD UNet training step.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its hot path is the Python call
 * `unet(inp_noisy_latents, timesteps, encoder_hidden_states, added_time_ids)` at
 * /root/reference/train_svd.py:1021, `accelerator.backward(loss)` at :1044 and `optimizer.step()` at
 * :1047, all of which dispatch into ATen/cuDNN/cuBLAS/flash-SDPA kernels through diffusers modules
 * (/root/reference/src/unet_spatio_temporal_condition.py:7-13).  This header is what a ctypes binding
 * of those kernels binds instead; each entry cites the reference/diffusers operation it replaces.
 *
 * Conventions
 *  - Activations are row-major [rows, C] with rows = (b, t, y, x) flattened ("(B*T, HW, C)" layout),
 *    stored in `dtype` (SVDX_F16 / SVDX_BF16); statistics, biases, master weights, grads are float.
 *  - All memory is owned by the caller (PyTorch's caching allocator).  The library never allocates or
 *    frees device memory, never synchronises, never retains a pointer past the call; every launch goes
 *    to the explicit `stream` (a hipStream_t passed as void*), so all calls are hipGraph-capturable.
 *  - Return 0 on success, negative on error; the message is retrievable with svdx_last_error()
 *    (thread-local: forward runs on the Python main thread, backward may run on another).
 */
#ifndef SVDX_H
#define SVDX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVDX_F16 0
#define SVDX_BF16 1

#define SVDX_OUT_ACT 0      /* store in activation dtype            */
#define SVDX_OUT_F32 1      /* store float                          */
#define SVDX_OUT_F32_ATOMIC 2 /* atomicAdd float (grad accumulation, split-K) */
#define SVDX_OUT_F32_ADD 4    /* float read-modify-write without atomics (each element has one owner block) */
#define SVDX_OUT_F32_SLAB 3   /* split-K: split z stores its partial sums (float) to C + z*M*ldc; no bias/res here */

#define SVDX_EPI_NONE 0
#define SVDX_EPI_GEGLU_FWD 1  /* B = GEGLU proj [2F,K]: C = pre [M,2F] and aux_out = h [M,F] = pre[:, :F] * gelu(pre[:, F:])       */
#define SVDX_EPI_GEGLU_BWD 2  /* GEMM result = dh [M,F] (not stored); aux_in = pre [M,2F]; C = dpre [M,2F] (GEGLU backward)    */

#define SVDX_GATHER_PLAIN 0
#define SVDX_GATHER_CONV3X3 1     /* 3x3, pad 1, stride 1|2, optional nearest x2 upsampled source */
#define SVDX_GATHER_CONV3X3_DGRAD2 2 /* data-gradient of the stride-2 3x3 conv (transposed conv) */
#define SVDX_GATHER_TEMPORAL3 3   /* Conv3d kernel (3,1,1) pad (1,0,0) over the frame axis */
#define SVDX_GATHER_CONV3X3_PAD0 4 /* 3x3 stride 2 over F.pad(x, (0,1,0,1)): the VAE encoder's Downsample2D(padding=0) -- zeros only
                                    * below / right of the image (diffusers vae.Encoder, reached from train_svd.py:283-291) */

/* Implicit-GEMM A-operand addressing: row m of the GEMM is an output pixel, K = taps*cin,
 * k = tap*cin + ci.  Replaces cuDNN/MIOpen conv2d/conv3d reached from diffusers ResnetBlock2D /
 * TemporalResnetBlock / Downsample2D / Upsample2D. */
typedef struct svdx_gather {
    int mode;
    int n_img;      /* images (B*T) for modes 1,2; batch B for mode 3 */
    int hi, wi;     /* logical source height/width (after the optional x2 upsample)      */
    int ho, wo;     /* output height/width; GEMM rows M = n_img*ho*wo (mode 3: B*T*hw)     */
    int cin;        /* channels per tap                                                    */
    int stride;     /* mode 1: 1 or 2                                                      */
    int ups;        /* mode 1: source stored at (hi/2, wi/2), read through nearest x2      */
    int t, hw;      /* mode 3: frames per clip, rows per frame                             */
    int lda;        /* row stride of the source tensor in elements                         */
} svdx_gather;

/* ABI revision of this header: bumped whenever an entry changes its argument list or meaning (100 = rounds 1-3; 400 = round 4).
 * svdx_version() returns the value the library was built with; the ctypes binding refuses a library whose number differs. */
#define SVDX_VERSION 600
int         svdx_version(void);
int         svdx_last_error(char* buf, size_t n);
/* 1 when the binary was built for gfx950 and a device is usable */
int         svdx_device_ok(void);

/* ---- GEMM family: nn.Linear / conv2d / conv3d fwd and data-grad, weight-grad (NT form) ------------
 * acc[m,n] = sum_k Aeff[m,k] * B[n,k];  v = alpha*acc + bias[n] + rowvec[g(m)*rv_ld + n] + res[m*ldres+n]
 * g(m) = rv_mod ? m % rv_mod : m / rv_rows_per_group.   B is [N,K] row-major (ldb).
 * out_mode ACT: C[m*ldc+n] = (dtype)v ; F32: float store ; F32_ATOMIC: atomicAdd (split_k >= 1) ; F32_SLAB: slice z of the K-tiles (ceil(K/64/split_k)
 * tiles each) to its own float slab -- a split_k whose last slices would own no K-tile is an argument error (their slabs would stay unwritten).
 * epilogue (variant >= 2): SVDX_EPI_GEGLU_FWD / _BWD fuse diffusers' GEGLU (attention.py) into the projection GEMMs, aux_dim = F.
 * variant = output tile of the launch (rows x columns, LDS stages of the K-loop, waves):  0 / 1 the plain 128x128 kernel with 64-bit addressing
 * (operands beyond the 2 GiB buffer reach);  4 heuristic among 6 / 7 / 8 = 160x160 / 128x160 / 128x128, two stages, four waves, two workgroups per CU;
 * ring-staged, one workgroup per CU:  16 / 17 / 18 = 256x160 / 256x128 / 256x256 (eight waves; 3, 3, 2 stages),  20 / 21 = 128x160 / 128x128
 * (four waves, 4 stages),  23 / 22 = 192x160 / 192x128 (eight waves, 3 stages),  25 / 24 = 96x160 / 96x128 (four waves, 4 stages);
 * 26 = 192x128, eight waves, TWO stages (80 KB of LDS: two workgroups per CU -- the tile under the GEGLU epilogues).
 * 28 / 27 = 128x160 / 128x128, eight waves, two stages (72 / 64 KB: two workgroups per CU; tuner candidates).
 * A 160-wide variant takes its 128-wide sibling when N % 160 != 0 or under the GEGLU-forward epilogue; 18 needs N % 256 == 0 (else 17).
 * The host side picks per problem (svd_xtend_amd/ops.py: choose_cfg). */
int svdx_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
              const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
              const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
              int out_mode, float alpha, int split_k, int variant, int epilogue, const void* aux_in, void* aux_out, int aux_dim,
              int dtype, void* stream);

/* svdx_gemm with activation output, PLUS the GroupNorm statistics of the tensor it writes: gn_stats (the opaque buffer of svdx_gn_stats,
 * zeroed by the caller) receives sum / sum of squares of the ROUNDED results per (sample, group), sample = m / gn_rows, group =
 * n / gn_cg (M = whole samples, N = whole groups; G = N / gn_cg, n_s = M / gn_rows) -- the svdx_gn_stats pass over C that the consuming
 * ResnetBlock2D / TemporalResnetBlock / TransformerSpatioTemporalModel norm would need is gone.  Same fixed-point
 * integer sums as svdx_gn_stats: run-to-run identical.  Needs a variant-4 tile whose width divides N (the statistics are taken in the
 * coalesced store loop), ldc % 8 == 0, a tile that touches <= 8 samples and <= 36 groups; no split-K, no fused epilogue. */
int svdx_gemm_gn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                 const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                 const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                 float alpha, int variant, float* gn_stats, int gn_rows, int gn_cg, int dtype, void* stream);

/* Same GEMM with a second, plain operand pair reduced into the same accumulators before the epilogue:
 *   C = alpha * (gather(A) B^T + A2 B2^T) (+ bias + rowvec + res),  A2 [M, K2] row pitch lda2, B2 [N, K2] row pitch ldb2.
 * One launch computes a LoRA-adapted projection y = x W^T + (s x A^T) B^T (train_svd_lora.py:659-674, peft's Linear.forward)
 * instead of a second GEMM that re-reads and re-writes y.  Variant-4 kernels only (variant >= 2), no split-K, no fused epilogue;
 * K2 a multiple of 64, both pitches multiples of 8 elements, 16-byte aligned bases.
 * a2_seg_n > 0 (fused q/k/v adapters): output columns [j*a2_seg_n, (j+1)*a2_seg_n) pair with A2 columns [j*K2, (j+1)*K2), so A2 is
 * [M, (N / a2_seg_n) * K2] and B2 stays [N, K2]; a2_seg_n must divide N and be a multiple of the kernel's tile width (160 when
 * N % 160 == 0 and variant != 8, else 128). */
int svdx_gemm_dual(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                   const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                   const void* res, int ldres, const svdx_gather* gather, const void* zero_page,
                   int out_mode, float alpha, int variant, const void* A2, const void* B2, int K2, int lda2, int ldb2,
                   int a2_seg_n, int dtype, void* stream);

/* Weight-gradient GEMM in TN form: C[n*ldc + k] (+)= sum_r A[r*lda + n] * B[r*ldb + k]  (A = dY [R,N], B = X [R,K], float C).
 * Replaces the dW part of autograd's Linear backward (train_svd.py:1044) without materialising transposes.
 * out_mode: F32 (store), F32_ADD (+=), F32_SLAB (split z -> C + z*N*ldc), F32_ATOMIC.
 * a_colsum (may be NULL): the column sums of A -- the Linear bias gradient, computed on the MFMA pipe from the dY tiles the GEMM stages
 * anyway.  Unsplit (split_k == 1): a_colsum[n] += sum_r A[r*lda + n] (one owner per column, no atomics).  SVDX_OUT_F32_SLAB:
 * a_colsum is float[split_k][N] and slice z STORES its partial into row z; svdx_gemm_finalize(colsum_slabs = a_colsum, ...) adds
 * the rows in order -- a fixed summation order, so the bias gradient is run-to-run identical.  With split_k > 1 a_colsum needs the
 * slab mode.  stages selects the kernel: 0 / 2 = four waves, 128 x 128 output tiles, two LDS stages (two workgroups per CU, drained
 * every K-step); 3 / 4 = the same tile with 2 / 3 row tiles in flight across the barrier (one workgroup per CU); 18 = eight waves,
 * 256 x 256 output tiles, one workgroup per CU (outputs of >= 1024 x 512 that 180-256 such tiles x row slices cover).
 */
/* stages | SVDX_TN_FLAT: the flat global_load_lds staging that operands of 2 GiB and more get (no buffer descriptor reaches them), requested
 * for any operand -- how the tests exercise that path without allocating 2 GiB. */
#define SVDX_TN_FLAT 64
/* found_inf (may be NULL; SVDX_OUT_F32 / SVDX_OUT_F32_ADD only): *found_inf = 1.0f when a value this launch leaves in C is not finite --
 * GradScaler's inf check (accelerate fp16; train_svd.py:1047) done where the gradient is written; pass &opt_state[3]. */
int svdx_gemm_tn(const void* A, const void* B, float* C, int R, int N, int K, int lda, int ldb, int ldc,
                 float* a_colsum, const void* zero_page, int out_mode, int split_k, int stages, float* found_inf, int dtype, void* stream);

/* Epilogue of a split-K GEMM run with SVDX_OUT_F32_SLAB: v = sum_z acc[z*slab_stride + m*N + n] + bias + rowvec + res (same operand
 * meaning as svdx_gemm); c_is_f32_accumulate: 0 -> C[m*ldc+n] = (dtype)v, 1 -> ((float*)C)[m*ldc+n] += v, 2 -> ((float*)C)[m*ldc+n] = v
 * (a weight gradient that is written exactly once per step needs neither a zeroed destination nor the read of it).
 * colsum_slabs (may be NULL): float[nsplit][colsum_n] partial column sums left by svdx_gemm_tn; colsum_out[n] += their sum. */
int svdx_gemm_finalize(const float* acc, int nsplit, int64_t slab_stride, void* C, int c_is_f32_accumulate, int M, int N, int ldc,
                       const float* bias, const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod,
                       const void* res, int ldres, const float* colsum_slabs, float* colsum_out, int colsum_n, int dtype, void* stream);

/* svdx_gemm_finalize with activation output, PLUS the GroupNorm statistics of the tensor it writes (see svdx_gemm_gn; the split-K
 * convolutions of the 16x10 / 8x5 levels end here).  N * gn_rows >= 1024 (a block's 1024 consecutive elements touch <= 2 samples),
 * N / gn_cg <= 64 groups. */
int svdx_gemm_finalize_gn(const float* acc, int nsplit, int64_t slab_stride, void* C, int M, int N, int ldc, const float* bias,
                          const float* rowvec, int rv_ld, int rv_rows_per_group, int rv_mod, const void* res, int ldres,
                          float* gn_stats, int gn_rows, int gn_cg, int dtype, void* stream);

/* Skinny linears (M <= 64): time/added-id embedding MLPs, time_emb_proj, time_pos_embed, the KV-length-1
 * cross-attention.  X, Y float; W in dtype.
 * trans=0: Y[m,n] (+)= sum_k act(X[m,k]) W[n*ldw+k] + bias[n];  trans=1: Y[m,k] (+)= sum_n X[m,n] W[n*ldw+k]. */
int svdx_small_linear(const float* X, const void* W, const float* bias, float* Y, int M, int N, int K,
                      int ldw, int trans, int silu_in, int accumulate, int dtype, void* stream);
/* dW[n*K+k] += scale * sum_m dY[m,n] X[m,k]   (all float; weight-grad of a skinny linear) */
int svdx_outer_acc(const float* dY, const float* X, float* dW, int M, int N, int K, float scale, void* stream);
/* The same skinny kernels over a table of jobs in ONE launch (the KV-length-1 cross-attention of the 32 transformer blocks is a chain of
 * 64 + 16 skinny linears and 48 outer products per step, each ~5 us of launch latency: diffusers Attention.to_v / to_out[0] behind
 * src/unet_spatio_temporal_condition.py:170-192, 219-234).  `jobs` is a HOST array: it is copied into the kernel arguments
 * (SVDX_BATCH_MAX_JOBS per launch; longer tables take several launches), so a captured hipGraph owns its copy.  Every job runs
 * the device function of the single-job entry (same arithmetic in the same order); all jobs of a call share M, trans and dtype.  Jobs of one call must not
 * depend on each other. */
#define SVDX_BATCH_MAX_JOBS 48
typedef struct svdx_lin_job {
    const float* X;    /* [M, K] (trans = 0) or [M, N] (trans = 1), float */
    const void* W;     /* [N, ldw] in dtype */
    const float* bias; /* [N] or NULL (trans = 0 only) */
    float* Y;          /* [M, N] (trans = 0) or [M, K] (trans = 1) */
    int N, K, ldw;
    int flags;         /* bit 0: SiLU on the input (trans = 0 only); bit 1: accumulate into Y */
} svdx_lin_job;
int svdx_small_linear_batch(const svdx_lin_job* jobs, int n_jobs, int M, int trans, int dtype, void* stream);
typedef struct svdx_outer_job {
    const float* dY;   /* [M, N] */
    const float* X;    /* [M, K]; NULL with K = 1: a column of ones (bias gradient) */
    float* dW;         /* [N, K], accumulated into WITHOUT atomics: the jobs of one call run concurrently, so their dW must be distinct */
    int N, K;
    float scale;
    int reserved;
} svdx_outer_job;
int svdx_outer_acc_batch(const svdx_outer_job* jobs, int n_jobs, int M, void* stream);
/* The float forms of svdx_gemm_finalize (c_is_f32_accumulate 1 / 2) over a table of jobs: dst[i] (+)= sum_z acc[z*slab_stride + i] for
 * i < count, slice 0 first (the order svdx_gemm_finalize adds them in: same bits), and colsum_out[n] += sum_z colsum_slabs[z*colsum_n + n].
 * Replaces the weight-grad half of autograd's Linear backward epilogues (train_svd.py:1044): the reducing launches of the row-sliced
 * svdx_gemm_tn weight gradients of a sweep, none of which is read before the optimizer, run as one launch per 48.  The destinations of
 * one call must be distinct. */
typedef struct svdx_gradfin_job {
    const float* acc;           /* [nsplit][slab_stride] float slabs */
    float* dst;                 /* [count] floats, 16-byte aligned */
    const float* colsum_slabs;  /* [nsplit][colsum_n] or NULL */
    float* colsum_out;          /* [colsum_n], accumulated into */
    int64_t slab_stride, count; /* floats; multiples of 4 */
    int nsplit, colsum_n;
    int store;                  /* 1: dst = sum (a gradient written once per step), 0: dst += sum */
    int reserved;
    float* found_inf;           /* NULL, or &opt_state[3]: set to 1.0f when a value left in dst is not finite (see svdx_gemm_tn) */
} svdx_gradfin_job;
int svdx_grad_finalize_batch(const svdx_gradfin_job* jobs, int n_jobs, void* stream);
/* out[i, :] = [cos(t_i f_j), sin(t_i f_j)], f_j = exp(-ln(1e4) j / (dim/2))  (diffusers Timesteps,
 * flip_sin_to_cos=True, shift 0; src/unet_spatio_temporal_condition.py:138,143) */
int svdx_timestep_embed(const float* t, float* out, int n, int dim, void* stream);

/* ---- GroupNorm(32) (+SiLU) over n_s samples of `rows` rows x C channels (2-D: sample = frame;
 *      3-D: sample = clip, rows = T*HW).  stats / bstats are OPAQUE buffers of SVDX_GN_STAT_FLOATS * SVDX_GN_REPLICAS * n_s * G
 *      floats (8-byte aligned): int64[SVDX_GN_REPLICAS][n_s][G][2] partial sums in 64-bit fixed point (sum, sum of squares; backward:
 *      sum dz*gamma, sum dz*gamma*xhat).  Blocks spread their atomics over the replicas, readers add them; integer addition is
 *      associative, so the statistics are run-to-run identical.  The kernels zero the buffer first unless `prezeroed` (the host
 *      zeroes one arena per pass instead of ~200 tiny memsets). ------------------------------------------------------------ */
#define SVDX_GN_REPLICAS 8
#define SVDX_GN_STAT_FLOATS 4   /* floats of storage per (replica, sample, group): two int64 */
int svdx_gn_stats(const void* x, float* stats, int n_s, int rows, int C, int G, int prezeroed, int dtype, void* stream);
int svdx_gn_apply(const void* x, const float* stats, const float* gamma, const float* beta, void* y,
                  int n_s, int rows, int C, int G, float eps, int silu, int dtype, void* stream);
/* bstats[n_s,G,2] = (sum dz*gamma, sum dz*gamma*xhat), dz = dy * silu'(z) when silu */
int svdx_gn_bwd_stats(const void* dy, const void* x, const float* stats, const float* gamma, const float* beta,
                      float* bstats, int n_s, int rows, int C, int G, float eps, int silu, int prezeroed, int dtype,
                      void* stream);
int svdx_gn_bwd_apply(const void* dy, const void* x, const float* stats, const float* bstats,
                      const float* gamma, const float* beta, const void* add, void* dx,
                      int n_s, int rows, int C, int G, float eps, int silu, int dtype, void* stream);

/* ---- LayerNorm over C; stats[rows,2] = (mean, rstd) ------------------------------------------------- */
int svdx_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                int rows, int C, float eps, int dtype, void* stream);
/* dx = LN'(dy) (+ add) (+ add2_scale * add2);  dgamma/dbeta (float, accumulated into, may both be NULL).  scratch: NULL (block sums go in by
 * float atomics) or SVDX_LN_PARTIAL_ROWS*2*C floats of workspace (per-block partial rows + a reducing pass: ~4x faster). */
#define SVDX_LN_PARTIAL_ROWS 2048
/* defer_reduce (needs scratch): the reducing pass is NOT launched -- scratch then holds svdx_ln_bwd_blocks(rows, C) partial rows of 2*C floats
 * (dgamma | dbeta) that svdx_ln_param_reduce_batch adds into dgamma / dbeta later, many LayerNorms per launch (a backward sweep of
 * train_svd.py's trainable set has 48 of them, train_svd.py:761-766). */
int svdx_ln_bwd(const void* dy, const void* x, const float* stats, const float* gamma, const void* add, const void* add2,
                float add2_scale, void* dx, float* dgamma, float* dbeta, float* scratch, int rows, int C, int defer_reduce, int dtype,
                void* stream);
int svdx_ln_bwd_blocks(int rows, int C);
typedef struct svdx_lnred_job {
    const float* partial; /* [nblk][2*C] left by svdx_ln_bwd(defer_reduce = 1) */
    float* dgamma;        /* [C], accumulated into */
    float* dbeta;         /* [C], accumulated into */
    int nblk, C;
} svdx_lnred_job;
/* jobs: HOST array (copied into the kernel arguments, SVDX_BATCH_MAX_JOBS per launch) */
int svdx_ln_param_reduce_batch(const svdx_lnred_job* jobs, int n_jobs, void* stream);

/* ---- spatial self-attention (head_dim 64), flash form; replaces F.scaled_dot_product_attention in
 *      diffusers AttnProcessor2_0.  q,k,v element (n,s,h,d) at (n*S+s)*ld + h*64 + d; o at pitch ld_o.
 *      Operands whose reduction index must be contiguous (V^T, K^T, Q^T, dO^T) are read from the row-major tiles with
 *      gfx950's transposing LDS read: no transposed copies exist. ---- */
int svdx_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int nb, int heads, int S,
                  int ld, int ld_o, float scale, int dtype, void* stream);
/* D[n,h,s] = sum_d o*do */
int svdx_attn_bwd_prep(const void* o, const void* d_o, float* D, int nb, int heads, int S, int ld_o, int dtype, void* stream);
int svdx_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* d_o,
                      const float* lse, const float* D, void* dk, void* dv, int nb, int heads, int S,
                      int ld, int ld_o, int ld_d, float scale, int dtype, void* stream);
int svdx_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o,
                     const float* lse, const float* D, void* dq, int nb, int heads, int S,
                     int ld, int ld_o, int ld_d, float scale, int dtype, void* stream);

/* ---- temporal self-attention across frames: element (b,t,p,h,d) at
 *      ((b*T+t)*HW+p)*ld + h*64 + d -- addressed in place, no (B*T,HW,C)<->(B*HW,T,C) transpose. ------ */
int svdx_tattn_fwd(const void* q, const void* k, const void* v, void* o, int B, int T, int HW, int heads,
                   int ld, int ld_o, float scale, int dtype, void* stream);
int svdx_tattn_bwd(const void* q, const void* k, const void* v, const void* d_o, void* dq, void* dk, void* dv,
                   int B, int T, int HW, int heads, int ld, int ld_o, int ld_d, float scale, int dtype, void* stream);

/* ---- elementwise / data movement -------------------------------------------------------------------- */
int svdx_geglu_fwd(const void* pre, void* out, int M, int F, int dtype, void* stream);       /* out = pre[:, :F] * gelu(pre[:, F:]) */
int svdx_geglu_bwd(const void* dout, const void* pre, void* dpre, int M, int F, int dtype, void* stream);
int svdx_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* alpha = sigmoid(*mix_factor) (diffusers AlphaBlender, image_only_indicator == 0) */
int svdx_blend(const void* a, const void* b, const float* mix_factor, void* out, int64_t n, int dtype, void* stream);
/* da may be NULL (only db = (1-alpha)*dy is produced) */
int svdx_blend_bwd(const void* dy, const float* mix_factor, void* da, void* db, int64_t n, int dtype, void* stream);
int svdx_add_rowvec(const void* x, const float* vec, void* out, int rows, int C, int rv_ld,
                    int rows_per_group, int mod, int dtype, void* stream);
/* out[g, c] (+)= sum over rows of group g (float; zeroed first unless accumulate).  scratch (may be NULL): float[nslab][n_groups][C]
 * with nslab = ceil(rows of the largest group / SVDX_COLSUM_SLAB) -- row slabs then leave partial sums there and a second launch adds
 * them in slab order (run-to-run identical); without it the slabs add with float atomics. */
#define SVDX_COLSUM_SLAB 512
int svdx_colsum(const void* x, float* out, int rows, int C, int ldx, int n_groups, int rows_per_group, int mod,
                int accumulate, float* scratch, int dtype, void* stream);
/* out[c*ld_out + r] = in[r*ld_in + c]; columns r in [rows, ld_out) zero-filled */
int svdx_transpose(const void* in, int ld_in, void* out, int ld_out, int rows, int cols, int dtype, void* stream);
int svdx_concat2(const void* a, int Ca, const void* b, int Cb, void* out, int rows, int dtype, void* stream);
int svdx_split2(const void* in, void* a, int Ca, void* b, int Cb, int rows, int dtype, void* stream);
/* bwd of nearest x2: out[n,y,x,:] = sum of the 2x2 block of in[n,2y..,2x..,:] */
int svdx_sum2x2(const void* in, void* out, int n_img, int h, int w, int C, int dtype, void* stream);
int svdx_cast_from_f32(const float* in, void* out, int64_t n, int dtype, void* stream);
/* out[c*R + r] = (dtype) in[r*Ccols + c] */
int svdx_cast_transpose_from_f32(const float* in, void* out, int R, int Ccols, int dtype, void* stream);
/* NCHW float <-> rows: out[(n*H*W + y*W + x)*ld + c] (channels >= C zero-filled up to ld on the way in) */
int svdx_nchw_to_rows(const float* in, void* out, int n_img, int C, int H, int W, int ld, float mul, int dtype, void* stream);
int svdx_rows_to_nchw(const void* in, float* out, int n_img, int C, int H, int W, int ld, int dtype, void* stream);
int svdx_zero(void* p, size_t bytes, void* stream);

/* ---- fused temporal self-attention forward (csrc/tsa.hip): norm1 -> attn1 -> residual of diffusers' TemporalBasicTransformerBlock
 * (the block built at /root/reference/src/unet_spatio_temporal_condition.py:170-192) in ONE launch,
 *   n   = LayerNorm(x; gamma, beta, eps)                                   [M, C]   (also stored, with stats[M,2] = mean, rstd)
 *   qkv = n wqkv^T                                                        [M, 3C]  (stored: the backward needs q, k, v)
 *   o   = softmax over the T frames of each (clip, pixel, head) of q k^T * scale, times v      [M, C] (stored)
 *   h1  = o wo^T + bo + cvec[group(row)] + x                               [M, C]
 * rows are (b, t, pixel) ordered, M = B*T*HW; wqkv [3C, C] and wo [C, C] row-major in the activation dtype; cvec (float, may be NULL)
 * is indexed like svdx_gemm's rowvec.  Needs T <= 16, C = 64*heads <= 320.  A workgroup owns svdx_tsa_pixels_per_band(T, HW) pixels
 * (the largest divisor of HW with at most 144 rows) of one clip. */
int svdx_tsa_pixels_per_band(int T, int HW);
int svdx_tsa_fwd(const void* x, const float* gamma, const float* beta, float eps, const void* wqkv, const void* wo, const float* bo,
                 const float* cvec, int rv_ld, int rv_rows_per_group, int rv_mod, void* n1, float* stats, void* qkv, void* o, void* h1,
                 int B, int T, int HW, int C, int heads, float scale, int dtype, void* stream);

/* ---- frozen conditioners either side of the step (csrc/encoders.hip) --------------------------------------
 * svdx_patch_rows: im2col of a few-channel NCHW float image into GEMM rows,
 *   out[(n*ho + y)*wo + x][(c*kh + dy)*kw + dx] = mul * in[n][c][y*stride + dy - pad][x*stride + dx - pad]   (0 outside, 0 for k >= C*kh*kw up to ldk)
 *   -- the k order of a flattened torch conv weight [Cout, C, kh, kw].  Replaces the first convolution of diffusers' vae.Encoder
 *   (`conv_in`, reached from train_svd.py:286) and CLIPVisionEmbeddings.patch_embedding (train_svd.py:872).
 * svdx_softmax_rows: out[r][c] = softmax_c(scale * in[r][c]) for c < cols, 0 for cols <= c < cols_out (fp32 inside): the attention of
 *   a head dimension other than 64 is GEMM -> this -> GEMM (VAE mid-block attention, CLIP self-attention).
 * svdx_act_rows: out = gelu_erf(in) (act 0) or in * sigmoid(1.702 in) (act 1, quick_gelu). */
int svdx_patch_rows(const float* in, void* out, int n_img, int C, int H, int W, int kh, int kw, int stride, int pad, int ho, int wo,
                    int ldk, float mul, int dtype, void* stream);
int svdx_softmax_rows(const void* in, void* out, int rows, int cols, int cols_out, int64_t ld_in, int64_t ld_out, float scale,
                      int dtype, void* stream);
int svdx_act_rows(const void* in, void* out, int64_t n, int act, int dtype, void* stream);
/* The anti-aliased resize in front of the CLIP tower (`_resize_with_antialiasing`, train_svd.py:140-248), float NCHW planes:
 * svdx_blur_axis: one pass of the separable Gaussian blur (`_filter2d` with F.pad(mode="reflect")) along x (axis 0) or y (axis 1);
 *   `taps` (device floats, odd count) is the normalised window of `_gaussian`.
 * svdx_bicubic_affine: F.interpolate(mode="bicubic", align_corners=True) to (ho, wo), then out = scale[c] * v + shift[c] -- the
 *   un-normalisation (x + 1) / 2 and CLIP's mean / std normalisation of train_svd.py:861-871 in one affine map per channel. */
int svdx_blur_axis(const float* in, float* out, int planes, int H, int W, const float* taps, int ntaps, int axis, void* stream);
int svdx_bicubic_affine(const float* in, float* out, int n_img, int C, int H, int W, int ho, int wo, const float* scale,
                        const float* shift, void* stream);
/* Self-attention over a short sequence with any head dimension (CLIP ViT-H/14: 257 tokens, 16 heads of 80; transformers
 * CLIPAttention reached from train_svd.py:875).  qkv rows [n_img*S, ld]; head h: q at column h*dp, k at (heads + h)*dp, v at
 * (2*heads + h)*dp, d real channels out of the dp the packed projection gives every head; out rows [n_img*S, ld_o], head h at h*dp
 * (channels d..dp written as zeros).  Any S; dp <= 128, dp % 8 == 0; matrix-pipe kernel (key tiles of 64 through LDS). */
int svdx_attn_small_fwd(const void* qkv, void* out, int n_img, int S, int heads, int d, int dp, int64_t ld, int64_t ld_o, float scale,
                        int dtype, void* stream);
/* base[off .. off+cnt) = 0 for each (off, cnt) pair of `spans` (int pairs; off and cnt multiples of 4): ONE launch for the scattered
 * gradient slots that are accumulated with atomics and so must start from zero (biases, LayerNorm, skinny cross-attention weights). */
int svdx_zero_spans(float* base, const int* spans, int n_spans, void* stream);
/* Measurement aid: *slot = the device's constant-rate wall clock (ticks of svdx_wall_clock_khz kHz), written by a one-lane kernel in stream
 * order; capturable.  bench.py brackets the GEMM-family launches of a captured step with it. */
int svdx_stamp(uint64_t* slot, void* stream);
int svdx_wall_clock_khz(void);

/* ---- Launch plans: one pass of the path replayed from C (SURVEY.md 8b asks for svdx_unet_forward / svdx_unet_backward / svdx_workspace_bytes_*:
 * what a non-Python host would bind in place of `unet(...)` at /root/reference/train_svd.py:1021 and `accelerator.backward(loss)` at :1044).
 * The step is ~1,500 launches issued by the host-side operators, which own every buffer; a plan is that launch list -- kernel, grid,
 * block, LDS bytes, argument bytes (the device pointers among them) -- recorded on the calling thread while the pass runs or is being
 * captured, and re-issued by svdx_plan_replay with no Python, torch or hipGraph involved.  Every svdx_* entry called between begin and end
 * on that thread is recorded in call order (and still launched / captured as usual).  The buffers the recorded pointers refer to must
 * stay alive and in place for as long as the plan is replayed (the captured step's memory pool does that); replays are stream-ordered,
 * nothing synchronises.  svdx_plan_bytes = host bytes the plan holds. */
int     svdx_plan_begin(void);
int     svdx_plan_end(void** plan);
int64_t svdx_plan_launches(const void* plan);
int64_t svdx_plan_bytes(const void* plan);
int     svdx_plan_replay(const void* plan, void* stream);
int     svdx_plan_free(void* plan);

/* ---- The data-parallel gradient sum without a collective library (svdx_allreduce_grads; replaces the all-reduce that
 * DistributedDataParallel runs inside accelerator.backward, /root/reference/train_svd.py:815 + :1044, when RCCL's choice of algorithm
 * is a ring: xGMI is point-to-point, the direct exchange uses all seven links of a GPU at once).
 *   peers  HOST array of `world` device pointers: the same float buffer [n] of every rank of the node (index = rank, own buffer
 *          included), each mapped into this process by the caller (hipIpcOpenMemHandle / peer access enabled)
 *   phase  -1: system-scope release only (publishes what earlier launches of this device wrote)
 *           0: reduce-scatter -- slice `rank` of my buffer = sum over q of peer q's slice `rank`, added in rank order (one owner per
 *              element: identical bits on every rank and from run to run)
 *           1: all-gather -- every other slice of my buffer = its owner's reduced slice
 *          slice q = floats [q * per, min(n, (q + 1) * per)), per = ceil(n / world / 4) * 4.
 * The CALLER orders the ranks: all ranks past phase -1 (their gradients complete) before anyone's phase 0, all past phase 0 before
 * phase 1, all past phase 1 before anyone writes its buffer again (svd_xtend_amd/train.py DirectAllReduce: a stream-ordered one-element
 * RCCL all-reduce, or a host barrier).  n a multiple of 4, buffers 16-byte aligned, world <= SVDX_MAX_PEERS. */
#define SVDX_MAX_PEERS 16
int svdx_allreduce_grads(float* const* peers, int world, int rank, int64_t n, int phase, void* stream);

/* ---- EDM loss (train_svd.py:1025-1036) fused with its gradient.  pred rows [B*T*HW, ld]; noisy/target
 *      float NCHW-per-frame [B,T,4,H,W]; sigma[B].  loss (float, accumulated; zero it first) and
 *      dpred rows = loss_scale * dLoss/dpred, loss_scale read from opt_state[1].  scratch: SVDX_EDM_LOSS_SCRATCH floats
 *      (per-workgroup partial sums, added in a fixed order by a second launch: the loss is run-to-run identical). --- */
#define SVDX_EDM_LOSS_SCRATCH 1024
int svdx_edm_loss(const void* pred, int ld, const float* noisy, const float* target, const float* sigma,
                  float* loss, void* dpred, int B, int T, int C, int HW, const float* opt_state, float* scratch,
                  int dtype, void* stream);

/* ---- optimizer: AdamW (train_svd.py:767-773) + GradScaler semantics (accelerate fp16) + the learning-rate schedule
 *      (diffusers get_scheduler, train_svd.py:807-813, stepped at :1048), all on device.
 *      opt_state float[SVDX_OPT_STATE_FLOATS]: 0 step (optimizer steps that were not skipped), 1 loss_scale, 2 growth_tracker,
 *      3 found_inf, 4 inv_scale, 5 bc1, 6 bc2, 7 skip, 8 lr multiplier of the current step (written by svdx_optim_prep),
 *      9 schedule kind (SVDX_SCHED_*), 10 warmup steps, 11 total steps, 12 cycles, 13 power, 14 lr_end / lr_init (polynomial),
 *      15 scheduler steps per optimizer step (accelerate: num_processes; 0 is read as 1).
 *      Step k (1-based, skipped steps not counted) runs at lr * lambda((k - 1) * opt_state[15]). */
#define SVDX_OPT_STATE_FLOATS 16
#define SVDX_SCHED_CONSTANT 0
#define SVDX_SCHED_CONSTANT_WITH_WARMUP 1
#define SVDX_SCHED_LINEAR 2
#define SVDX_SCHED_COSINE 3
#define SVDX_SCHED_COSINE_WITH_RESTARTS 4
#define SVDX_SCHED_POLYNOMIAL 5
/* diffusers' piecewise_constant (step rules "m0:s0,m1:s1,...,m_last"; train_svd.py:807-812 passes --lr_scheduler straight through):
 * opt_state[10] = n rules (<= SVDX_SCHED_MAX_RULES) and, BEHIND the 16 floats, opt_state[16 + 2i] = boundary s_i, opt_state[17 + 2i] =
 * multiplier m_i for i < n, opt_state[16 + 2n] = m_last: lambda(step) = m_i of the first s_i > step, m_last beyond the last boundary.
 * Only this kind reads beyond SVDX_OPT_STATE_FLOATS (the buffer is then SVDX_OPT_STATE_FLOATS + 2 * SVDX_SCHED_MAX_RULES + 1 floats). */
#define SVDX_SCHED_PIECEWISE_CONSTANT 6
#define SVDX_SCHED_MAX_RULES 8
int svdx_check_finite(const float* g, int64_t n, float* opt_state, void* stream);
/* The same over `n_spans` (offset, count) pairs of g (ints; multiples of 4): with svdx_gemm_tn / svdx_grad_finalize_batch raising found_inf
 * for the gradients they store, only the accumulated slots of the flat buffer (the span table of svdx_zero_spans) are left to test. */
int svdx_check_finite_spans(const float* g, const int* spans, int n_spans, float* opt_state, void* stream);
int svdx_optim_prep(float* opt_state, float beta1, float beta2, float growth, float backoff, int growth_interval,
                    int dynamic, void* stream);
/* param_mode of the two AdamW entries.  SVDX_PARAMS_F32: parameters and moments are fp32 (the default of this library: the 16-bit copies the
 * kernels read derive from fp32 masters).  SVDX_PARAMS_BF16_REFERENCE: the reference's LoRA recipe under --mixed_precision bf16
 * (/root/reference/train_svd_lora.py:666-674: the UNet is cast to bf16 before add_adapter, so adapters, gradients and torch.optim.AdamW's
 * state are bf16 tensors): torch.optim.AdamW's op sequence on bf16 tensors, each op in float and rounded to bf16, on values kept in the same
 * float buffers (the caller rounds the initial parameters to bf16 once).  torch forms 1 - lr * wd, 1 - beta1, 1 - beta2 in double and hands its
 * kernels their float roundings; so does this mode -- which is why lr, the betas, eps, wd and grad_mul are doubles here. */
#define SVDX_PARAMS_F32 0
#define SVDX_PARAMS_BF16_REFERENCE 1
/* (the hyper-parameters travel as the doubles the host holds: see param_mode) */
int svdx_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
               double eps, double wd, double grad_mul, const float* opt_state, void* p_act, int param_mode, int dtype, void* stream);

/* Same update over a table of tiles (6 ints each: element offset, row pitch, rows <= 64, cols <= 64 -- multiples of 4 --,
 * transposed-twin offset or -1, transposed-twin row pitch): every tile updates p/m/v, writes the row-major 16-bit twin
 * p_act[off + r*ld + c] and, when its transposed-twin offset is >= 0, pt_act[wt_off + c*ldwt + r] (the [K,N] operand of the
 * data-grad GEMM of a trainable nn.Linear), so no separate transposition pass is needed after the optimizer step.
 * All offsets must be multiples of 4 elements; tiles must not overlap. */
int svdx_adamw_tiled(float* p, const float* g, float* m, float* v, const int* tiles, int n_tiles, double lr, double beta1,
                     double beta2, double eps, double wd, double grad_mul, const float* opt_state, void* p_act, void* pt_act,
                     int param_mode, int dtype, void* stream);

/* ---- EMA of the trainable weights (diffusers EMAModel.step, train_svd.py:1053-1054): shadow -= one_minus_decay * (shadow - p)
 *      over n floats (both buffers 16-byte aligned).  The decay itself follows EMAModel.get_decay on the host. */
int svdx_ema_lerp(float* shadow, const float* p, int64_t n, float one_minus_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVDX_H */
