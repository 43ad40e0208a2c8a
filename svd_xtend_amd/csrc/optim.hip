// optim.hip -- EDM loss (+gradient), gradient finiteness check, loss-scale state machine and fused AdamW (gfx950).
//
// Replaces, on device and without host synchronisation: the loss arithmetic of train_svd.py:1025-1036,
// torch.optim.AdamW (train_svd.py:767-773) and accelerate's GradScaler (scale / unscale / inf-skip / growth).
// opt_state float[8]: 0 step, 1 loss_scale, 2 growth_tracker, 3 found_inf, 4 inv_scale, 5 bc1, 6 bc2, 7 skip.
#include "common.h"

namespace {

// The loss is a single number: every workgroup leaves its partial sum in `partial[block]` and a second, tiny launch adds them in
// block order -- a fixed reduction tree, run-to-run identical (per-block float atomics were not; one 1024-thread workgroup over the
// whole tensor was, but took 240 us).
template <typename T>
__global__ __launch_bounds__(256) void edm_loss_kernel(const T* __restrict__ pred, int ld, const float* __restrict__ noisy,
                                                       const float* __restrict__ target, const float* __restrict__ sigma,
                                                       float* __restrict__ partial, T* __restrict__ dpred, int ld_d, int B, int T_, int C,
                                                       int HW, const float* __restrict__ opt_state) {
    __shared__ float red[4];
    const long n = (long)B * T_ * C * HW;
    const float norm = 1.f / (float)n;
    const float lscale = opt_state[1];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i indexes the NCHW-per-frame float tensors: ((b*T + t)*C + c)*HW + p
        const int p = (int)(i % HW);
        const long t1 = i / HW;
        const int c = (int)(t1 % C);
        const long bt = t1 / C;
        const int b = (int)(bt / T_);
        const float s = sigma[b];
        const float s2 = s * s;
        const float c_out = -s / sqrtf(s2 + 1.f);
        const float c_skip = 1.f / (s2 + 1.f);
        const float w = (1.f + s2) / s2;
        const size_t ro = ((size_t)bt * HW + p);
        const float pr = to_f<T>(pred[ro * ld + c]);
        const float diff = c_out * pr + c_skip * noisy[i] - target[i];
        acc += w * diff * diff;
        dpred[ro * ld_d + c] = from_f<T>(lscale * 2.f * w * diff * c_out * norm);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * norm;
}

__global__ __launch_bounds__(64) void edm_loss_reduce_kernel(const float* __restrict__ partial, int n, float* loss) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *loss += acc;
}

__global__ void check_finite_kernel(const float* __restrict__ g, long n, float* opt_state) {
    bool bad = false;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !isfinite(v[e]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) bad |= !isfinite(g[n4 * 4 + threadIdx.x]);
    if (__any(bad)) {
        if ((threadIdx.x & 63) == 0) opt_state[3] = 1.f;
    }
}

// the same test over a table of (offset, count) spans: the slots of the flat gradient buffer that are ACCUMULATED (biases, LayerNorm, skinny
// cross-attention weights, alignment gaps) -- the gradients that one GEMM stores per step are tested by the kernel that stores them
__global__ __launch_bounds__(256) void check_finite_spans_kernel(const float* __restrict__ g, const int* __restrict__ spans, float* opt_state) {
    const int off = spans[2 * blockIdx.x], cnt = spans[2 * blockIdx.x + 1];
    bool bad = false;
    for (int i = threadIdx.x * 4; i < cnt; i += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + off + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !isfinite(v[e]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) opt_state[3] = 1.f;
}

// ---- gradient sum over the ranks of one node, straight over xGMI (svdx_allreduce_grads) --------------------------------------------
// xGMI is point-to-point: every GPU has a link to each of the other seven, so the bandwidth-optimal exchange is the DIRECT one -- rank r
// pulls slice r of every peer's buffer (reduce-scatter), then pulls the reduced slices of the other ranks (all-gather): 2 x (7/8) x S
// bytes per GPU spread over seven links at once, where a ring moves the same bytes over ONE link at a time (2.6 ms
// against 18 ms for the 1.59 GB gradient buffer of 8 ranks).  Slice q = floats [q * per, min(n, (q + 1) * per)), per = ceil(n / world / 4) * 4.
// Every element of the sum is formed by ONE thread of ONE rank, adding the ranks in rank order: all ranks end with identical bits, and
// the same bits from run to run.  Visibility between devices: a workgroup starts with a system-scope acquire (drops cached lines of
// peer memory) and ends with a system-scope release (writes its own dirty lines back); the caller's barrier between the phases orders
// the ranks.  phase -1 is the release alone: it publishes whatever earlier kernels of this device left in its caches.
struct PeerPtrs { float* p[SVDX_MAX_PEERS]; };

template <int W>
__global__ __launch_bounds__(256) void peer_reduce_scatter_kernel(PeerPtrs peers, int world, int rank, long n, long per) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const long lo = (long)rank * per, hi = lo + per < n ? lo + per : n;
    float* mine = peers.p[rank];
    for (long i = lo + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < hi; i += (long)gridDim.x * blockDim.x * 4) {
        f32x4 v[W > 0 ? W : SVDX_MAX_PEERS];
        if (W > 0) {
#pragma unroll
            for (int q = 0; q < W; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(peers.p[q] + i));     // all in flight
        } else {
            for (int q = 0; q < world; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(peers.p[q] + i));
        }
        f32x4 acc = v[0];
        if (W > 0) {
#pragma unroll
            for (int q = 1; q < W; ++q) acc += v[q];
        } else {
            for (int q = 1; q < world; ++q) acc += v[q];
        }
        *reinterpret_cast<f32x4*>(mine + i) = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

__global__ __launch_bounds__(256) void peer_all_gather_kernel(PeerPtrs peers, int world, int rank, long n, long per) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    float* mine = peers.p[rank];
    // blockIdx.y walks the OTHER ranks starting with the next one, so that at any moment the seven links carry one slice each
    const int q = (rank + 1 + (int)blockIdx.y) % world;
    const long lo = (long)q * per, hi = lo + per < n ? lo + per : n;
    const float* src = peers.p[q];
    for (long i = lo + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < hi; i += (long)gridDim.x * blockDim.x * 4)
        *reinterpret_cast<f32x4*>(mine + i) = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + i));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

__global__ __launch_bounds__(64) void peer_publish_kernel() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }

// Learning-rate multiplier lambda(n) of diffusers.optimization.get_scheduler (train_svd.py:807-813), n = scheduler steps taken so
// far.  Evaluated on the device from the optimizer's own step counter so that a step replayed from a hipGraph follows the
// schedule without host involvement.  st[9] kind, st[10] warmup, st[11] total, st[12] cycles, st[13] power, st[14] lr_end / lr_init.
__device__ float lr_lambda(const float* st, float n) {
    const int kind = (int)st[9];
    const float warm = st[10], total = st[11], cycles = st[12], power = st[13], end_ratio = st[14];
    if (kind == SVDX_SCHED_CONSTANT) return 1.f;
    if (kind == SVDX_SCHED_PIECEWISE_CONSTANT) {                       // get_piecewise_constant_schedule: the first boundary beyond n decides
        const int nr = min((int)st[10], SVDX_SCHED_MAX_RULES);
        for (int i = 0; i < nr; ++i)
            if (n < st[SVDX_OPT_STATE_FLOATS + 2 * i]) return st[SVDX_OPT_STATE_FLOATS + 2 * i + 1];
        return st[SVDX_OPT_STATE_FLOATS + 2 * nr];
    }
    if (kind == SVDX_SCHED_POLYNOMIAL) {
        if (n < warm) return n / fmaxf(1.f, warm);
        if (n > total) return end_ratio;
        const float remaining = 1.f - (n - warm) / (total - warm);
        return (1.f - end_ratio) * powf(remaining, power) + end_ratio;
    }
    if (n < warm) return n / fmaxf(1.f, warm);
    if (kind == SVDX_SCHED_CONSTANT_WITH_WARMUP) return 1.f;
    if (kind == SVDX_SCHED_LINEAR) return fmaxf(0.f, (total - n) / fmaxf(1.f, total - warm));
    const float progress = (n - warm) / fmaxf(1.f, total - warm);
    const float pi = 3.14159265358979323846f;
    if (kind == SVDX_SCHED_COSINE) return fmaxf(0.f, 0.5f * (1.f + cosf(pi * cycles * 2.f * progress)));
    if (progress >= 1.f) return 0.f;                                   // SVDX_SCHED_COSINE_WITH_RESTARTS
    return fmaxf(0.f, 0.5f * (1.f + cosf(pi * fmodf(cycles * progress, 1.f))));
}

__global__ void optim_prep_kernel(float* st, float beta1, float beta2, float growth, float backoff, int interval, int dynamic) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool found = st[3] > 0.f;
    float step = st[0], scale = st[1], tracker = st[2];
    const float inv = 1.f / scale;
    // accelerate steps the scheduler st[15] (= num_processes) times after every optimizer step that was not skipped, so the
    // k-th successful step runs at lambda((k - 1) * st[15])
    st[8] = lr_lambda(st, step * fmaxf(1.f, st[15]));
    if (dynamic) {
        if (found) { scale *= backoff; tracker = 0.f; }
        else {
            tracker += 1.f;
            if (tracker >= (float)interval) { scale *= growth; tracker = 0.f; }
        }
    }
    if (!found) step += 1.f;
    st[0] = step; st[1] = scale; st[2] = tracker; st[3] = 0.f; st[4] = inv;
    st[5] = 1.f - powf(beta1, step);
    st[6] = 1.f - powf(beta2, step);
    st[7] = found ? 1.f : 0.f;
}

// EMA of the trainable weights (diffusers EMAModel.step, train_svd.py:1053-1054): shadow -= (1 - decay) * (shadow - p)
__global__ __launch_bounds__(256) void ema_lerp_kernel(float* __restrict__ shadow, const float* __restrict__ p, long n, float omd) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(shadow) + i);
        const f32x4 w = *reinterpret_cast<const f32x4*>(p + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] -= omd * (s[e] - w[e]);
        __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(shadow) + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        shadow[i] -= omd * (shadow[i] - p[i]);
    }
}

// One AdamW element.  REF = false: fp32 parameters and moments (this library's default: the 16-bit copies the kernels read are derived from
// fp32 masters).  REF = true: the reference's LoRA recipe under bf16 (/root/reference/train_svd_lora.py:666-674: the UNet is cast to bf16
// BEFORE add_adapter, so the adapters, their gradients and torch.optim.AdamW's moments are all bf16 tensors): the op sequence of
// torch.optim.AdamW on bf16 tensors -- mul_, lerp_, mul_ + addcmul_, sqrt, div, add, addcdiv_ -- each computed in float and rounded to
// bf16, on values held in the same float buffers (every stored value is bf16-representable).  Pinned to torch.optim.AdamW itself on
// bf16 CPU tensors through the emulation (tests/test_host_logic.py).
__device__ __forceinline__ float rb16(float x) { return (float)(bf16)x; }
struct AdamScalars {           // per-launch scalars of one AdamW step (device side)
    float gmul, decay, beta1, beta2, omb1, omb2, eps, step_size, bc2_sqrt, inv_bc2_sqrt;
};
// lr / wd / betas arrive as the doubles the host holds (torch forms 1 - lr * wd, 1 - beta1, 1 - beta2 in double and hands the kernels their
// float roundings: 1.f - 0.999f is off by 1.3e-5 relative from (float)(1 - 0.999), enough to move 1 % of a bf16 moment by one step)
__device__ __forceinline__ AdamScalars adam_scalars(const float* st, double lr, double beta1, double beta2, double eps, double wd, double grad_mul,
                                                    bool ref) {
    AdamScalars a;
    const float lr_f = (float)lr * st[8];                            // schedule multiplier of this step (optim_prep)
    a.gmul = st[4] * (float)grad_mul;
    a.step_size = lr_f / st[5];
    a.inv_bc2_sqrt = rsqrtf(st[6]);
    a.bc2_sqrt = sqrtf(st[6]);
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
    if (ref) {
        a.decay = (float)(1.0 - lr * (double)st[8] * wd);
        a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    } else {
        a.decay = 1.f - lr_f * (float)wd;
        a.omb1 = 1.f - a.beta1; a.omb2 = 1.f - a.beta2;
    }
    return a;
}
template <bool REF>
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, const AdamScalars& a) {
    if (!REF) {
        const float gg = g * a.gmul;
        p *= a.decay;
        m = a.beta1 * m + a.omb1 * gg;
        v = a.beta2 * v + a.omb2 * gg * gg;
        const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
        p -= a.step_size * m / denom;
    } else {
        const float gg = rb16(g * a.gmul);
        p = rb16(p * a.decay);
        m = rb16(fmaf(a.omb1, gg - m, m));                       // lerp_(grad, 1 - beta1): weight < 0.5, fused as torch's kernel
        v = rb16(v * a.beta2);
        v = rb16(v + a.omb2 * gg * gg);                          // addcmul_(grad, grad, value = 1 - beta2)
        float d = rb16(sqrtf(v));
        d = rb16(d / a.bc2_sqrt);
        d = rb16(d + a.eps);
        p = rb16(p + (-a.step_size) * (m / d));                  // addcdiv_(exp_avg, denom, value = -step_size)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, double lr, double beta1, double beta2, double eps,
                                                    double wd, double grad_mul, const float* __restrict__ st, T* __restrict__ p_act, int ref) {
    if (st[7] > 0.f) return;     // inf/nan in the gradients: skip the step (GradScaler semantics)
    const AdamScalars a = adam_scalars(st, lr, beta1, beta2, eps, wd, grad_mul, ref != 0);
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i * 4);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + i * 4);
        f32x4 vv = *reinterpret_cast<const f32x4*>(v + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];            // (vector elements do not bind to references)
            if (ref) adamw_elem<true>(pe, gv[e], me, ve, a);
            else adamw_elem<false>(pe, gv[e], me, ve, a);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        *reinterpret_cast<f32x4*>(p + i * 4) = pv;
        *reinterpret_cast<f32x4*>(m + i * 4) = mv;
        *reinterpret_cast<f32x4*>(v + i * 4) = vv;
        if (p_act) {
            Vec4<T> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[e] = from_f<T>(pv[e]);
            *reinterpret_cast<Vec4<T>*>(p_act + i * 4) = o;
        }
    }
}

// AdamW over a table of <= 64 x 64 tiles of the flat buffers: tile {off, ld, rows, cols, wt_off, ldwt} covers elements
// off + r*ld + c.  Besides the float update and the row-major 16-bit twin (p_act), a tile with wt_off >= 0 also writes the
// TRANSPOSED 16-bit twin pt_act[wt_off + c*ldwt + r] through LDS -- the [K, N] operand of the data-grad GEMMs -- so the
// per-step re-transposition of all trainable weights (a separate read + write of every weight) disappears.
template <typename T>
__global__ __launch_bounds__(256) void adamw_tiled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const int* __restrict__ tiles, double lr, double beta1,
                                                          double beta2, double eps, double wd, double grad_mul,
                                                          const float* __restrict__ st, T* __restrict__ p_act, T* __restrict__ pt_act, int ref) {
    if (st[7] > 0.f) return;     // inf/nan in the gradients: skip the step (GradScaler semantics)
    __shared__ T tile[64][68];
    const int* tl = tiles + (size_t)blockIdx.x * 6;
    const int off = tl[0], ld = tl[1], rows = tl[2], cols = tl[3], wt_off = tl[4], ldwt = tl[5];
    const AdamScalars a = adam_scalars(st, lr, beta1, beta2, eps, wd, grad_mul, ref != 0);
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
    // one pass over 12.6 GB that nothing re-reads before the next step: streaming loads and stores throughout.  All sixteen operand
    // loads of the thread (four rows x p, g, m, v) are requested before the first is used (round 6: the row-at-a-time loop kept
    // four in flight and drained them before every row's arithmetic -- two IEEE divisions and a square root per element).
    f32x4 pv[4], gv[4], mv[4], vv[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        live[i] = r < rows && c4 < cols;
        // threads beyond the tile read its first element group (in range, never stored): no branch between the loads
        const long idx = live[i] ? (long)off + (long)r * ld + c4 : (long)off;
        pv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + idx));
        gv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + idx));
        mv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + idx));
        vv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + idx));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        if (live[i]) {
            const long idx = (long)off + (long)r * ld + c4;
            Vec4<T> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[i][e], me = mv[i][e], ve = vv[i][e];
                if (ref) adamw_elem<true>(pe, gv[i][e], me, ve, a);
                else adamw_elem<false>(pe, gv[i][e], me, ve, a);
                pv[i][e] = pe; mv[i][e] = me; vv[i][e] = ve;
                o.v[e] = from_f<T>(pe);
            }
            __builtin_nontemporal_store(pv[i], reinterpret_cast<f32x4*>(p + idx));
            __builtin_nontemporal_store(mv[i], reinterpret_cast<f32x4*>(m + idx));
            __builtin_nontemporal_store(vv[i], reinterpret_cast<f32x4*>(v + idx));
            if (p_act) *reinterpret_cast<Vec4<T>*>(p_act + idx) = o;
            if (wt_off >= 0) *reinterpret_cast<Vec4<T>*>(&tile[r][c4]) = o;
        }
    }
    if (wt_off < 0) return;      // block-uniform
    __syncthreads();
    // transposed store: thread -> wt row (= tile column) cc, 4 consecutive wt columns (= tile rows) r4..r4+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = r0 + 16 * i, r4 = c4;
        if (cc < cols && r4 < rows) {
            Vec4<T> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[e] = tile[r4 + e][cc];
            *reinterpret_cast<Vec4<T>*>(pt_act + (long)wt_off + (long)cc * ldwt + r4) = o;
        }
    }
}

}  // namespace

extern "C" int svdx_edm_loss(const void* pred, int ld, const float* noisy, const float* target, const float* sigma,
                             float* loss, void* dpred, int B, int T_, int C, int HW, const float* opt_state, float* scratch, int dtype,
                             void* stream) {
    SVDX_CHECK_ARG(pred && noisy && target && sigma && loss && dpred && opt_state && scratch, "svdx_edm_loss: null argument");
    const long n = (long)B * T_ * C * HW;
    const int ld_d = ((C + 63) / 64) * 64;
    const int blocks = (int)std::min<long>((n + 255) / 256, SVDX_EDM_LOSS_SCRATCH);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((edm_loss_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)pred,
                                             ld, noisy, target, sigma, scratch, (T*)dpred, ld_d, B, T_, C, HW, opt_state));
    SVDX_LAUNCH_CHECK("svdx_edm_loss");
    hipLaunchKernelGGL(edm_loss_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, blocks, loss);
    SVDX_LAUNCH_CHECK("svdx_edm_loss(reduce)");
    return 0;
}

extern "C" int svdx_check_finite(const float* g, int64_t n, float* opt_state, void* stream) {
    SVDX_CHECK_ARG(g && opt_state && n > 0 && ((uintptr_t)g & 15) == 0, "svdx_check_finite: bad args");
    const int blocks = (int)std::min<long>((n / 4 + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(check_finite_kernel, dim3(std::max(1, blocks)), dim3(256), 0, (hipStream_t)stream, g, (long)n, opt_state);
    SVDX_LAUNCH_CHECK("svdx_check_finite");
    return 0;
}

extern "C" int svdx_allreduce_grads(float* const* peers, int world, int rank, int64_t n, int phase, void* stream) {
    SVDX_CHECK_ARG(peers && world >= 1 && world <= SVDX_MAX_PEERS && rank >= 0 && rank < world && n > 0 && n % 4 == 0 &&
                       phase >= -1 && phase <= 1, "svdx_allreduce_grads: bad args (1..%d ranks, n a multiple of 4, phase -1 / 0 / 1)", SVDX_MAX_PEERS);
    PeerPtrs pp;
    for (int q = 0; q < SVDX_MAX_PEERS; ++q) pp.p[q] = q < world ? peers[q] : nullptr;
    for (int q = 0; q < world; ++q)
        SVDX_CHECK_ARG(pp.p[q] && ((uintptr_t)pp.p[q] & 15) == 0, "svdx_allreduce_grads: peer %d buffer null or not 16-byte aligned", q);
    const long per = ((n + world - 1) / world + 3) / 4 * 4;
    hipStream_t st = (hipStream_t)stream;
    if (phase == -1) {
        hipLaunchKernelGGL(peer_publish_kernel, dim3(1024), dim3(64), 0, st);       // every XCD's L2 sees a wave that writes it back
    } else if (phase == 0) {
        const int blocks = (int)std::max<long>(1, std::min<long>((per / 4 + 255) / 256, 1024));
        if (world == 8) hipLaunchKernelGGL(peer_reduce_scatter_kernel<8>, dim3(blocks), dim3(256), 0, st, pp, world, rank, (long)n, per);
        else if (world == 4) hipLaunchKernelGGL(peer_reduce_scatter_kernel<4>, dim3(blocks), dim3(256), 0, st, pp, world, rank, (long)n, per);
        else if (world == 2) hipLaunchKernelGGL(peer_reduce_scatter_kernel<2>, dim3(blocks), dim3(256), 0, st, pp, world, rank, (long)n, per);
        else hipLaunchKernelGGL(peer_reduce_scatter_kernel<0>, dim3(blocks), dim3(256), 0, st, pp, world, rank, (long)n, per);
    } else if (world > 1) {
        const int blocks = (int)std::max<long>(1, std::min<long>((per / 4 + 255) / 256, 256));
        hipLaunchKernelGGL(peer_all_gather_kernel, dim3(blocks, world - 1), dim3(256), 0, st, pp, world, rank, (long)n, per);
    }
    SVDX_LAUNCH_CHECK("svdx_allreduce_grads");
    return 0;
}

extern "C" int svdx_check_finite_spans(const float* g, const int* spans, int n_spans, float* opt_state, void* stream) {
    if (n_spans <= 0) return 0;
    SVDX_CHECK_ARG(g && spans && opt_state && ((uintptr_t)g & 15) == 0, "svdx_check_finite_spans: bad args");
    hipLaunchKernelGGL(check_finite_spans_kernel, dim3(n_spans), dim3(256), 0, (hipStream_t)stream, g, spans, opt_state);
    SVDX_LAUNCH_CHECK("svdx_check_finite_spans");
    return 0;
}

extern "C" int svdx_optim_prep(float* opt_state, float beta1, float beta2, float growth, float backoff, int growth_interval,
                               int dynamic, void* stream) {
    SVDX_CHECK_ARG(opt_state, "svdx_optim_prep: null state");
    hipLaunchKernelGGL(optim_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, opt_state, beta1, beta2, growth, backoff,
                       growth_interval, dynamic);
    SVDX_LAUNCH_CHECK("svdx_optim_prep");
    return 0;
}

extern "C" int svdx_ema_lerp(float* shadow, const float* p, int64_t n, float one_minus_decay, void* stream) {
    SVDX_CHECK_ARG(shadow && p && n > 0 && (((uintptr_t)shadow | (uintptr_t)p) & 15) == 0, "svdx_ema_lerp: bad args (16-byte aligned buffers)");
    const int blocks = (int)std::min<long>((n / 4 + 255) / 256 + 1, 256 * 8);
    hipLaunchKernelGGL(ema_lerp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, shadow, p, (long)n, one_minus_decay);
    SVDX_LAUNCH_CHECK("svdx_ema_lerp");
    return 0;
}

extern "C" int svdx_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                          double wd, double grad_mul, const float* opt_state, void* p_act, int param_mode, int dtype, void* stream) {
    SVDX_CHECK_ARG(p && g && m && v && opt_state && n > 0 && n % 4 == 0, "svdx_adamw: bad args (n must be a multiple of 4)");
    SVDX_CHECK_ARG(param_mode == SVDX_PARAMS_F32 || param_mode == SVDX_PARAMS_BF16_REFERENCE, "svdx_adamw: param_mode %d", param_mode);
    const int blocks = (int)std::min<long>((n / 4 + 255) / 256, 256 * 8);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((adamw_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                                             (long)n, lr, beta1, beta2, eps, wd, grad_mul, opt_state, (T*)p_act, param_mode));
    SVDX_LAUNCH_CHECK("svdx_adamw");
    return 0;
}

extern "C" int svdx_adamw_tiled(float* p, const float* g, float* m, float* v, const int* tiles, int n_tiles, double lr, double beta1,
                                double beta2, double eps, double wd, double grad_mul, const float* opt_state, void* p_act, void* pt_act,
                                int param_mode, int dtype, void* stream) {
    SVDX_CHECK_ARG(p && g && m && v && tiles && opt_state && n_tiles > 0, "svdx_adamw_tiled: bad args");
    SVDX_CHECK_ARG(param_mode == SVDX_PARAMS_F32 || param_mode == SVDX_PARAMS_BF16_REFERENCE, "svdx_adamw_tiled: param_mode %d", param_mode);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((adamw_tiled_kernel<T>), dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, p, g, m, v, tiles,
                                             lr, beta1, beta2, eps, wd, grad_mul, opt_state, (T*)p_act, (T*)pt_act, param_mode));
    SVDX_LAUNCH_CHECK("svdx_adamw_tiled");
    return 0;
}
