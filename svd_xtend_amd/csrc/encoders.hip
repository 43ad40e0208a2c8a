// encoders.hip -- the few kernels the frozen conditioners either side of the UNet step need beyond the UNet's own set
// (SURVEY.md 8f ranks 1-2): the VAE encoder of `tensor_to_vae_latent` (/root/reference/train_svd.py:283-291) and the CLIP image
// tower of `encode_image` (:857-876).  Their convolutions, linears, GroupNorm / LayerNorm run on the kernels of gemm.hip / norm.hip;
// here are
//   * patch_rows     -- im2col of a FEW-channel image (cin = 3) into GEMM rows: the VAE's conv_in (3x3, pad 1) and CLIP's patch
//                       embedding (14x14, stride 14).  K = cin*kh*kw is zero-padded to the GEMM's K granule; 27 -> 64 costs 2.4x the
//                       MFMA work of the real 27 instead of the 21x of padding the CHANNELS to 64.
//   * softmax_rows   -- row softmax (fp32 inside) between the two plain GEMMs of an attention whose head dimension is not 64 and
//                       too wide for one fused kernel: the VAE mid-block's single head of 512 over 2560 tokens.
//   * attn_small_fwd -- fused attention on the matrix pipe for heads of up to 128 channels: CLIP ViT-H's heads of 80 over 257 tokens.
//   * gelu_rows      -- bias-free exact-erf GELU (the GEMM epilogue already added the bias), CLIP's MLP activation.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void patch_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int n_img, int C, int H, int W,
                                                         int kh, int kw, int stride, int pad, int ho, int wo, int ldk, float mul) {
    const int kk = C * kh * kw;
    const long n = (long)n_img * ho * wo * (ldk / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % (ldk / 8));
        const long m = i / (ldk / 8);
        const int x = (int)(m % wo);
        const long t = m / wo;
        const int y = (int)(t % ho);
        const long im = t / ho;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k8 * 8 + e;
            float val = 0.f;
            if (k < kk) {
                const int dx = k % kw, r = k / kw;
                const int dy = r % kh, c = r / kh;
                const int ys = y * stride + dy - pad, xs = x * stride + dx - pad;
                if (ys >= 0 && ys < H && xs >= 0 && xs < W) val = in[((im * C + c) * H + ys) * (long)W + xs] * mul;
            }
            v[e] = val;
        }
        store8<T>(out + m * ldk + k8 * 8, v);
    }
}

// one 256-thread block per row; the row (<= a few thousand scores) is re-read from L2 rather than held in registers
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int cols, int cols_out,
                                                           long ld_in, long ld_out, float scale_l2) {
    __shared__ float red[8];
    const T* x = in + (long)blockIdx.x * ld_in;
    T* y = out + (long)blockIdx.x * ld_out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c8 = cols / 8;
    float mx = -1e30f;
    for (int i = tid; i < c8; i += 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[e]);
    }
    for (int c = c8 * 8 + tid; c < cols; c += 256) mx = fmaxf(mx, to_f<T>(x[c]));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_l2;
    float sum = 0.f;
    for (int i = tid; i < c8; i += 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f(fmaf(v[e], scale_l2, -mx));
    }
    for (int c = c8 * 8 + tid; c < cols; c += 256) sum += __builtin_amdgcn_exp2f(fmaf(to_f<T>(x[c]), scale_l2, -mx));
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid; i < cols_out / 8; i += 256) {
        float v[8], o[8];
        if (i < c8) load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = i * 8 + e;
            const float xv = i < c8 ? v[e] : (c < cols ? to_f<T>(x[c]) : 0.f);
            o[e] = c < cols ? __builtin_amdgcn_exp2f(fmaf(xv, scale_l2, -mx)) * inv : 0.f;     // columns beyond `cols`: zero padding
        }
        store8<T>(y + i * 8, o);
    }
}

// act 0: exact-erf GELU, 1: x * sigmoid(1.702 x) (CLIP's quick_gelu)
template <typename T>
__global__ void act_rows_kernel(const T* __restrict__ in, T* __restrict__ out, long n8, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float v[8], o[8];
        load8<T>(in + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = act == 0 ? gelu_erf(v[e]) : v[e] * sigmoidf_(1.702f * v[e]);
        store8<T>(out + i * 8, o);
    }
}


// ---- anti-aliased resize of encode_image (train_svd.py:140-248): separable Gaussian blur with reflect padding, then bicubic ----
// axis 0: along x (W), axis 1: along y (H); planes = n * C images of H x W floats; taps odd (the reference makes them odd)
__global__ void blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int H, int W, const float* __restrict__ taps,
                                 int nt, int axis) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % H);
        const long pl = t / H;
        const float* src = in + pl * (long)H * W;
        const int half = (nt - 1) / 2;
        float acc = 0.f;
        for (int k = 0; k < nt; ++k) {
            int xs = x, ys = y;
            if (axis == 0) { xs = x + k - half; if (xs < 0) xs = -xs; if (xs >= W) xs = 2 * (W - 1) - xs; }
            else { ys = y + k - half; if (ys < 0) ys = -ys; if (ys >= H) ys = 2 * (H - 1) - ys; }
            acc += taps[k] * src[(long)ys * W + xs];
        }
        out[i] = acc;
    }
}

__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {      // torch's upsample_bicubic2d, A = -0.75
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// out[n][c][yo][xo] = scale[c] * bicubic(in[n][c])(yo, xo; align_corners = True) + shift[c]
__global__ void bicubic_affine_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int C, int H, int W, int ho, int wo,
                                      float ry, float rx, const float* __restrict__ scale, const float* __restrict__ shift) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % wo);
        const long t = i / wo;
        const int yo = (int)(t % ho);
        const long pl = t / ho;
        const int c = (int)(pl % C);
        const float* src = in + pl * (long)H * W;
        const float fy = ry * yo, fx = rx * xo;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_weights(fy - iy, wy);
        cubic_weights(fx - ix, wx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ys = min(max(iy - 1 + a, 0), H - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) row += wx[b] * src[(long)ys * W + min(max(ix - 1 + b, 0), W - 1)];
            acc += wy[a] * row;
        }
        out[i] = acc * scale[c] + shift[c];
    }
}

// ---- self-attention of a short sequence whose head dimension is not 64 (CLIP ViT-H: 257 tokens, 16 heads of 80) ------------------
// qkv rows [n*S, ld]: head h of q at column h*dp, of k at (heads + h)*dp, of v at (2 heads + h)*dp; d <= dp real channels (the rest of
// a head's dp columns is padding and is written as zeros).  Block = (64 queries, head, image), one wave per 16 queries; key tiles of 64
// rows go through LDS (rows padded by 16 B: conflict-free for both read shapes), channels beyond d staged as zeros up to the next
// multiple of 32.  The arithmetic is csrc/attention.hip's: S^T = K Q^T on the matrix pipe, online softmax per query column, P^T packed
// straight from the accumulators as the B operand of O^T += V^T P^T, V^T fragments by the transposing LDS read.
// Round 5: replaces the VALU form (one query per wave at a time, 244 us per launch, a quarter of the frozen conditioners' time).
constexpr int AS_MAXD = 128, AS_KT = 64, AS_MAXPITCH = (AS_MAXD / 8) * 16 + 16;
template <typename T>
__device__ __forceinline__ typename TT<T>::v8 as_frag_tr(const char* lds, int pitch, int db, int pb0, int pb1, int fr, int fg) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const int u = db * 4 + (fr & 3);                 // 8-byte unit of the row: channels u*4 .. u*4+3
    const int r0 = pb0 * 16 + fg * 4 + (fr >> 2), r1 = pb1 * 16 + fg * 4 + (fr >> 2);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + r0 * pitch + u * 8));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + r1 * pitch + u * 8));
    const v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename TT<T>::v8, r);
}
template <typename T>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int S, int heads, int d, int dp,
                                                             long ld, long ld_o, float sl2) {
    typedef typename TT<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char smem[2 * AS_KT * AS_MAXPITCH];
    const int nk = (d + 31) / 32, nd = (d + 15) / 16;      // 32-channel steps of the scores, 16-channel blocks of the output
    const int dc = nk * 4, c8 = (d + 7) / 8;               // staged / real 16-byte chunks per row
    const int pitch = dc * 16 + 16;
    char* Ks = smem;
    char* Vs = smem + AS_KT * pitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, n = blockIdx.z, q0 = blockIdx.x * 64 + wave * 16;
    const T* base = qkv + (size_t)n * S * ld;
    const uint4 zero4 = {0u, 0u, 0u, 0u};

    v8 qf[AS_MAXD / 32];
#pragma unroll
    for (int ks = 0; ks < AS_MAXD / 32; ++ks) {
        const int c = ks * 4 + fg;
        const uint4 v = (ks < nk && c < c8) ? *reinterpret_cast<const uint4*>(base + (size_t)min(q0 + fr, S - 1) * ld + h * dp + c * 8) : zero4;
        qf[ks] = __builtin_bit_cast(v8, v);
    }
    f32x4 oacc[AS_MAXD / 16];
#pragma unroll
    for (int i = 0; i < AS_MAXD / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow = -1e30f, lrow = 0.f;

    const int ntiles = (S + AS_KT - 1) / AS_KT;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        for (int i = tid; i < AS_KT * dc; i += 256) {
            const int r = i / dc, c = i - r * dc;
            const T* row = base + (size_t)min(t * AS_KT + r, S - 1) * ld + h * dp + c * 8;
            const bool real = c < c8;
            *reinterpret_cast<uint4*>(Ks + r * pitch + c * 16) = real ? *reinterpret_cast<const uint4*>(row + (size_t)heads * dp) : zero4;
            *reinterpret_cast<uint4*>(Vs + r * pitch + c * 16) = real ? *reinterpret_cast<const uint4*>(row + (size_t)2 * heads * dp) : zero4;
        }
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < AS_MAXD / 32; ++ks)
            if (ks < nk) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const v8 kf = *reinterpret_cast<const v8*>(Ks + (kb * 16 + fr) * pitch + (ks * 4 + fg) * 16);
                    s[kb] = TT<T>::mfma(kf, qf[ks], s[kb]);            // S^T[key kb*16 + fg*4 + r][query fr]
                }
            }
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (t * AS_KT + kb * 16 + fg * 4 + r >= S) s[kb][r] = -1e30f;
                mx = fmaxf(mx, s[kb][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(mrow, fmaxf(mx * sl2, -1e30f));
        const float alpha = __builtin_amdgcn_exp2f(mrow - mn);
        mrow = mn;
        lrow *= alpha;
#pragma unroll
        for (int db = 0; db < AS_MAXD / 16; ++db) oacc[db] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(s[kb][r] * sl2 - mn);
                s[kb][r] = pr;
                lrow += pr;
            }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            v8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = from_f<T>(s[2 * k2][e]);
                pf[4 + e] = from_f<T>(s[2 * k2 + 1][e]);
            }
#pragma unroll
            for (int db = 0; db < AS_MAXD / 16; ++db)
                if (db < nd) oacc[db] = TT<T>::mfma(as_frag_tr<T>(Vs, pitch, db, 2 * k2, 2 * k2 + 1, fr, fg), pf, oacc[db]);   // O^T[channel][query fr]
        }
    }
    lrow += __shfl_xor(lrow, 16, 64);
    lrow += __shfl_xor(lrow, 32, 64);
    const int q = q0 + fr;
    if (q < S) {
        const float inv = 1.f / lrow;
        T* op = out + ((size_t)n * S + q) * ld_o + h * dp;
#pragma unroll
        for (int db = 0; db < AS_MAXD / 16; ++db) {
            const int c0 = db * 16 + fg * 4;
            if (c0 < dp) {                                               // dp % 8 == 0: the four channels are all inside the head's columns
                Vec4<T> o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.v[r] = from_f<T>(c0 + r < d ? oacc[db][r] * inv : 0.f);     // padding channels: zeros
                *reinterpret_cast<Vec4<T>*>(op + c0) = o;
            }
        }
    }
}

}  // namespace

extern "C" int svdx_patch_rows(const float* in, void* out, int n_img, int C, int H, int W, int kh, int kw, int stride, int pad,
                               int ho, int wo, int ldk, float mul, int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && n_img > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "svdx_patch_rows: bad args");
    SVDX_CHECK_ARG(ldk % 8 == 0 && ldk >= C * kh * kw && (((uintptr_t)out) & 15) == 0, "svdx_patch_rows: ldk must be a multiple of 8 and >= C*kh*kw");
    SVDX_CHECK_ARG(ho == (H + 2 * pad - kh) / stride + 1 && wo == (W + 2 * pad - kw) / stride + 1, "svdx_patch_rows: output size mismatch");
    const long n = (long)n_img * ho * wo * (ldk / 8);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((patch_rows_kernel<T>), dim3((int)std::min<long>((n + 255) / 256, 65536)), dim3(256), 0,
                                             (hipStream_t)stream, in, (T*)out, n_img, C, H, W, kh, kw, stride, pad, ho, wo, ldk, mul));
    SVDX_LAUNCH_CHECK("svdx_patch_rows");
    return 0;
}

extern "C" int svdx_softmax_rows(const void* in, void* out, int rows, int cols, int cols_out, int64_t ld_in, int64_t ld_out, float scale,
                                 int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && rows > 0 && cols > 0 && cols_out >= cols, "svdx_softmax_rows: bad args");
    SVDX_CHECK_ARG(ld_in % 8 == 0 && ld_out % 8 == 0 && cols_out % 8 == 0 && ld_out >= cols_out && ld_in >= cols && (((uintptr_t)in | (uintptr_t)out) & 15) == 0,
                   "svdx_softmax_rows: rows must be 16-byte aligned, cols_out a multiple of 8");
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out,
                                             cols, cols_out, (long)ld_in, (long)ld_out, scale * 1.4426950408889634f));
    SVDX_LAUNCH_CHECK("svdx_softmax_rows");
    return 0;
}

extern "C" int svdx_act_rows(const void* in, void* out, int64_t n, int act, int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && n > 0 && n % 8 == 0 && (act == 0 || act == 1) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0, "svdx_act_rows: bad args");
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_rows_kernel<T>), dim3((int)std::min<long>((n / 8 + 255) / 256, 4096)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)in, (T*)out, (long)(n / 8), act));
    SVDX_LAUNCH_CHECK("svdx_act_rows");
    return 0;
}

extern "C" int svdx_blur_axis(const float* in, float* out, int planes, int H, int W, const float* taps, int ntaps, int axis, void* stream) {
    SVDX_CHECK_ARG(in && out && taps && planes > 0 && H > 0 && W > 0 && ntaps > 0 && ntaps % 2 == 1 && (axis == 0 || axis == 1),
                   "svdx_blur_axis: bad args (odd tap count, axis 0 = x / 1 = y)");
    SVDX_CHECK_ARG((ntaps - 1) / 2 < (axis == 0 ? W : H), "svdx_blur_axis: reflect padding needs (taps - 1) / 2 < extent");
    const long total = (long)planes * H * W;
    hipLaunchKernelGGL(blur_axis_kernel, dim3((int)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, in, out,
                       total, H, W, taps, ntaps, axis);
    SVDX_LAUNCH_CHECK("svdx_blur_axis");
    return 0;
}

extern "C" int svdx_bicubic_affine(const float* in, float* out, int n_img, int C, int H, int W, int ho, int wo, const float* scale,
                                   const float* shift, void* stream) {
    SVDX_CHECK_ARG(in && out && scale && shift && n_img > 0 && C > 0 && H > 0 && W > 0 && ho > 0 && wo > 0, "svdx_bicubic_affine: bad args");
    const float ry = ho > 1 ? (float)(H - 1) / (float)(ho - 1) : 0.f, rx = wo > 1 ? (float)(W - 1) / (float)(wo - 1) : 0.f;
    const long total = (long)n_img * C * ho * wo;
    hipLaunchKernelGGL(bicubic_affine_kernel, dim3((int)std::min<long>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, in, out,
                       total, C, H, W, ho, wo, ry, rx, scale, shift);
    SVDX_LAUNCH_CHECK("svdx_bicubic_affine");
    return 0;
}

extern "C" int svdx_attn_small_fwd(const void* qkv, void* out, int n_img, int S, int heads, int d, int dp, int64_t ld, int64_t ld_o, float scale,
                                   int dtype, void* stream) {
    SVDX_CHECK_ARG(qkv && out && n_img > 0 && S > 0 && heads > 0 && d > 0 && d <= dp && dp <= AS_MAXD && dp % 8 == 0,
                   "svdx_attn_small_fwd: needs d <= dp <= %d, dp %% 8 == 0 (got S=%d d=%d dp=%d)", AS_MAXD, S, d, dp);
    SVDX_CHECK_ARG(ld % 8 == 0 && ld >= 3L * heads * dp && ld_o % 4 == 0 && ld_o >= (long)heads * dp && (((uintptr_t)qkv) & 15) == 0 &&
                   (((uintptr_t)out) & 7) == 0, "svdx_attn_small_fwd: alignment");
    SVDX_CHECK_ARG(heads <= 65535 && n_img <= 65535, "svdx_attn_small_fwd: grid");
    dim3 grid((S + 63) / 64, heads, n_img);
    DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((attn_small_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)qkv, (T*)out, S, heads, d, dp,
                           (long)ld, (long)ld_o, scale * 1.4426950408889634f);
    });
    SVDX_LAUNCH_CHECK("svdx_attn_small_fwd");
    return 0;
}
