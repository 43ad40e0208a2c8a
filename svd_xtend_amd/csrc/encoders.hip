// encoders.hip -- the few kernels the frozen conditioners either side of the UNet step need beyond the UNet's own set
// (SURVEY.md 8f ranks 1-2): the VAE encoder of `tensor_to_vae_latent` (/root/reference/train_svd.py:283-291) and the CLIP image
// tower of `encode_image` (:857-876).  Their convolutions, linears, GroupNorm / LayerNorm run on the kernels of gemm.hip / norm.hip;
// here are
//   * patch_rows     -- im2col of a FEW-channel image (cin = 3) into GEMM rows: the VAE's conv_in (3x3, pad 1) and CLIP's patch
//                       embedding (14x14, stride 14).  K = cin*kh*kw is zero-padded to the GEMM's K granule; 27 -> 64 costs 2.4x the
//                       MFMA work of the real 27 instead of the 21x of padding the CHANNELS to 64.
//   * softmax_rows   -- row softmax (fp32 inside) between the two plain GEMMs of an attention whose head dimension is not 64:
//                       the VAE mid-block's single head of 512 over 2560 tokens, CLIP ViT-H's heads of 80 over 257 tokens.
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
