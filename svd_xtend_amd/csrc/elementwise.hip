// elementwise.hip -- HBM-bound glue kernels of the SVD UNet step for gfx950: GEGLU, AlphaBlender, broadcast
// row-vector add / grouped column sums, transposes (weight-grad operands, head-transposed attention operands),
// channel concat/split, nearest-x2 backward, dtype casts, NCHW<->rows.  All accesses are 16 B per lane.
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
static inline int ew_blocks(long n_items) { return (int)std::min<long>((n_items + EW_THREADS - 1) / EW_THREADS, 256 * 16); }


template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ pre, T* __restrict__ out, int M, int F) {
    const long n8 = (long)M * (F / 8);
    const int f8n = F / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / f8n;
        const int f = (int)(i - m * f8n) * 8;
        float a[8], g[8], o[8];
        load8<T>(pre + m * 2 * F + f, a);
        load8<T>(pre + m * 2 * F + F + f, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_erf(g[e]);
        store8<T>(out + m * F + f, o);
    }
}

template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ pre, T* __restrict__ dpre, int M, int F) {
    const long n8 = (long)M * (F / 8);
    const int f8n = F / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / f8n;
        const int f = (int)(i - m * f8n) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        load8<T>(pre + m * 2 * F + f, a);
        load8<T>(pre + m * 2 * F + F + f, g);
        load8<T>(dout + m * F + f, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            da[e] = d[e] * gelu_erf(g[e]);
            dg[e] = d[e] * a[e] * gelu_erf_grad(g[e]);
        }
        store8<T>(dpre + m * 2 * F + f, da);
        store8<T>(dpre + m * 2 * F + F + f, dg);
    }
}

// OP 0: out = a + b ; 1: out = al*a + (1-al)*b ; 2 (bwd): o1 = al*a, o2 = (1-al)*a
template <typename T, int OP>
__global__ void binary_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ mix,
                              T* __restrict__ o1, T* __restrict__ o2, long n) {
    float al = 0.f;
    if (OP != 0) al = sigmoidf_(mix[0]);
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float x[8], y[8], r[8], s[8];
        load8<T>(a + i * 8, x);
        if (OP != 2) load8<T>(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (OP == 0) r[e] = x[e] + y[e];
            else if (OP == 1) r[e] = al * x[e] + (1.f - al) * y[e];
            else { r[e] = al * x[e]; s[e] = (1.f - al) * x[e]; }
        }
        if (OP != 2 || o1) store8<T>(o1 + i * 8, r);
        if (OP == 2) store8<T>(o2 + i * 8, s);
    }
}

template <typename T>
__global__ void add_rowvec_kernel(const T* __restrict__ x, const float* __restrict__ vec, T* __restrict__ out, int rows,
                                  int C, int rv_ld, int rpg, int mod) {
    const int c8n = C / 8;
    const long n8 = (long)rows * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / c8n);
        const int c = (int)(i - (long)m * c8n) * 8;
        const int g = mod ? m % mod : m / rpg;
        float v[8];
        load8<T>(x + (size_t)m * C + c, v);
        const float* rv = vec + (size_t)g * rv_ld + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
        store8<T>(out + (size_t)m * C + c, v);
    }
}

constexpr int CS_SLAB = 512;
// grid (groups, row slabs of 512, column blocks of 256 channels).  A block = 32 column chunks (16 B) x 8 row lanes; every
// thread streams 64 rows with 4 loads in flight, the 8 row lanes are reduced through LDS and one atomicAdd per column per
// block goes out (float atomics are the scarce resource: ~15/ns chip-wide).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* out, int rows, int C, int ldx, int rpg, int mod,
                                                     float* partial) {
    __shared__ float red[8][32 * 8 + 8];
    const int g = blockIdx.x, slab = blockIdx.y;
    const int cnt = mod ? (rows - g + mod - 1) / mod : min(rpg, rows - g * rpg);
    const int i0 = slab * CS_SLAB, i1 = min(cnt, i0 + CS_SLAB);
    const int jj = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int j = blockIdx.z * 32 + jj;
    const bool active = j * 8 < C;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const size_t rstride = (size_t)(mod ? mod : 1) * ldx;
        const T* base = x + (size_t)(mod ? g : g * rpg) * ldx + j * 8;
        int i = i0 + rs;
        for (; i + 24 < i1; i += 32) {
            float v0[8], v1[8], v2[8], v3[8];
            load8<T>(base + (size_t)i * rstride, v0);
            load8<T>(base + (size_t)(i + 8) * rstride, v1);
            load8<T>(base + (size_t)(i + 16) * rstride, v2);
            load8<T>(base + (size_t)(i + 24) * rstride, v3);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; i < i1; i += 8) {
            float v[8];
            load8<T>(base + (size_t)i * rstride, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rs][jj * 8 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;          // 256 columns of this block
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][c];
    const int col = blockIdx.z * 256 + c;
    if (partial) {                  // deterministic form: slab `slab` of group g leaves its sums in partial[slab][g][C] (zeros when empty)
        if (col < C) partial[((size_t)slab * gridDim.x + g) * C + col] = i0 < i1 ? s : 0.f;
    } else if (col < C && i0 < i1) {
        atomicAdd(out + (size_t)g * C + col, s);
    }
}

// out[i] (+)= sum over the row slabs, in slab order
__global__ void colsum_reduce_kernel(const float* __restrict__ partial, float* out, int nslab, long n, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float t = 0.f;
        for (int sl = 0; sl < nslab; ++sl) t += partial[(size_t)sl * n + i];
        out[i] = accumulate ? out[i] + t : t;
    }
}

// batched tiled transpose: out[b][c*ld_out + r] = in[b][r*ld_in + c], r in [0, ld_out) zero-filled beyond rows.
// batch b -> (b1 = b / nb2, b2 = b % nb2); in offset = b1*in_s1 + b2*in_s2; out offset = b*out_s.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int rows, int cols,
                                                        int ld_in, int ld_out, int nb2, long in_s1, long in_s2,
                                                        long out_s) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const TI* ip = in + (b / nb2) * in_s1 + (b % nb2) * in_s2;
    TO* op = out + (long)b * out_s;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < rows && c < cols) ? (float)ip[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (c < cols && r < ld_out) op[(size_t)c * ld_out + r] = (TO)tile[tx][cc];
    }
}

template <typename T>
__global__ void concat2_kernel(const T* __restrict__ a, int Ca, const T* __restrict__ b, int Cb, T* __restrict__ out, int rows) {
    const int C = Ca + Cb, c8n = C / 8;
    const long n8 = (long)rows * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / c8n;
        const int c = (int)(i - m * c8n) * 8;
        const uint4 v = c < Ca ? *reinterpret_cast<const uint4*>(a + m * Ca + c)
                               : *reinterpret_cast<const uint4*>(b + m * Cb + (c - Ca));
        *reinterpret_cast<uint4*>(out + m * C + c) = v;
    }
}

template <typename T>
__global__ void split2_kernel(const T* __restrict__ in, T* __restrict__ a, int Ca, T* __restrict__ b, int Cb, int rows) {
    const int C = Ca + Cb, c8n = C / 8;
    const long n8 = (long)rows * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long m = i / c8n;
        const int c = (int)(i - m * c8n) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(in + m * C + c);
        if (c < Ca) *reinterpret_cast<uint4*>(a + m * Ca + c) = v;
        else *reinterpret_cast<uint4*>(b + m * Cb + (c - Ca)) = v;
    }
}

template <typename T>
__global__ void sum2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int n_img, int h, int w, int C) {
    const int c8n = C / 8;
    const long n8 = (long)n_img * h * w * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / c8n;
        const int c = (int)(i - pix * c8n) * 8;
        const int x = (int)(pix % w);
        const long t = pix / w;
        const int y = (int)(t % h);
        const long n = t / h;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                load8<T>(in + ((n * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx) * C + c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        store8<T>(out + pix * C + c, acc);
    }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        store8<T>(out + i * 8, v);
    }
    // tail
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n8 * 8)) out[n8 * 8 + threadIdx.x] = from_f<T>(in[n8 * 8 + threadIdx.x]);
}

template <typename T>
__global__ void nchw_to_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int n_img, int C, int HW, int ld, float mul) {
    const long n = (long)n_img * HW * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const long pix = i / ld;
        const int p = (int)(pix % HW);
        const long im = pix / HW;
        out[i] = c < C ? from_f<T>(in[(im * C + c) * HW + p] * mul) : from_f<T>(0.f);
    }
}

template <typename T>
__global__ void rows_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int n_img, int C, int HW, int ld) {
    const long n = (long)n_img * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C);
        const long im = t / C;
        out[i] = to_f<T>(in[(im * HW + p) * ld + c]);
    }
}

}  // namespace

#define EW_ALIGN_CHECK(name, cond) SVDX_CHECK_ARG(cond, name ": sizes must be multiples of 8 / pointers 16-byte aligned")
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int svdx_geglu_fwd(const void* pre, void* out, int M, int F, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_geglu_fwd", F % 8 == 0 && al16(pre) && al16(out));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((geglu_fwd_kernel<T>), dim3(ew_blocks((long)M * F / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)pre, (T*)out, M, F));
    SVDX_LAUNCH_CHECK("svdx_geglu_fwd");
    return 0;
}

extern "C" int svdx_geglu_bwd(const void* dout, const void* pre, void* dpre, int M, int F, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_geglu_bwd", F % 8 == 0 && al16(pre) && al16(dout) && al16(dpre));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((geglu_bwd_kernel<T>), dim3(ew_blocks((long)M * F / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)dout, (const T*)pre, (T*)dpre, M, F));
    SVDX_LAUNCH_CHECK("svdx_geglu_bwd");
    return 0;
}

extern "C" int svdx_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_add", n % 8 == 0 && al16(a) && al16(b) && al16(out));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((binary_kernel<T, 0>), dim3(ew_blocks(n / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)a, (const T*)b, (const float*)nullptr, (T*)out,
                                             (T*)nullptr, (long)n));
    SVDX_LAUNCH_CHECK("svdx_add");
    return 0;
}

extern "C" int svdx_blend(const void* a, const void* b, const float* mix_factor, void* out, int64_t n, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_blend", n % 8 == 0 && al16(a) && al16(b) && al16(out) && mix_factor);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((binary_kernel<T, 1>), dim3(ew_blocks(n / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)a, (const T*)b, mix_factor, (T*)out, (T*)nullptr,
                                             (long)n));
    SVDX_LAUNCH_CHECK("svdx_blend");
    return 0;
}

extern "C" int svdx_blend_bwd(const void* dy, const float* mix_factor, void* da, void* db, int64_t n, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_blend_bwd", n % 8 == 0 && al16(dy) && al16(da) && db && al16(db) && mix_factor);   // da may be NULL
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((binary_kernel<T, 2>), dim3(ew_blocks(n / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)dy, (const T*)nullptr, mix_factor, (T*)da, (T*)db,
                                             (long)n));
    SVDX_LAUNCH_CHECK("svdx_blend_bwd");
    return 0;
}

extern "C" int svdx_add_rowvec(const void* x, const float* vec, void* out, int rows, int C, int rv_ld, int rows_per_group,
                               int mod, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_add_rowvec", C % 8 == 0 && al16(x) && al16(out) && (mod > 0 || rows_per_group > 0));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_rowvec_kernel<T>), dim3(ew_blocks((long)rows * C / 8)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, (const T*)x, vec, (T*)out, rows, C, rv_ld, rows_per_group, mod));
    SVDX_LAUNCH_CHECK("svdx_add_rowvec");
    return 0;
}

extern "C" int svdx_colsum(const void* x, float* out, int rows, int C, int ldx, int n_groups, int rows_per_group, int mod,
                           int accumulate, float* scratch, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_colsum", C % 8 == 0 && ldx % 8 == 0 && al16(x) && (mod > 0 || rows_per_group > 0));
    hipStream_t st = (hipStream_t)stream;
    const int maxcnt = mod ? cdiv(rows, mod) : std::min(rows_per_group, rows);
    const int nslab = cdiv(maxcnt, CS_SLAB);
    if (!scratch && !accumulate) { if (int rc = svdx_zero(out, sizeof(float) * n_groups * C, stream)) return rc; }     // a kernel: see svdx_zero
    dim3 grid(n_groups, nslab, cdiv(C, 256));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, st, (const T*)x, out, rows, C, ldx,
                                             rows_per_group, mod, scratch));
    SVDX_LAUNCH_CHECK("svdx_colsum");
    if (scratch) {
        const long n = (long)n_groups * C;
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, scratch, out, nslab, n, accumulate);
        SVDX_LAUNCH_CHECK("svdx_colsum(reduce)");
    }
    return 0;
}

extern "C" int svdx_transpose(const void* in, int ld_in, void* out, int ld_out, int rows, int cols, int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_out >= rows, "svdx_transpose: bad args");
    dim3 grid(cdiv(ld_out, 64), cdiv(cols, 64), 1);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T, T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                             (T*)out, rows, cols, ld_in, ld_out, 1, 0L, 0L, 0L));
    SVDX_LAUNCH_CHECK("svdx_transpose");
    return 0;
}

extern "C" int svdx_cast_transpose_from_f32(const float* in, void* out, int R, int Ccols, int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && R > 0 && Ccols > 0, "svdx_cast_transpose_from_f32: bad args");
    dim3 grid(cdiv(R, 64), cdiv(Ccols, 64), 1);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<float, T>), grid, dim3(256), 0, (hipStream_t)stream, in, (T*)out,
                                             R, Ccols, Ccols, R, 1, 0L, 0L, 0L));
    SVDX_LAUNCH_CHECK("svdx_cast_transpose_from_f32");
    return 0;
}

extern "C" int svdx_concat2(const void* a, int Ca, const void* b, int Cb, void* out, int rows, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_concat2", Ca % 8 == 0 && Cb % 8 == 0 && al16(a) && al16(b) && al16(out));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((concat2_kernel<T>), dim3(ew_blocks((long)rows * (Ca + Cb) / 8)), dim3(EW_THREADS),
                                             0, (hipStream_t)stream, (const T*)a, Ca, (const T*)b, Cb, (T*)out, rows));
    SVDX_LAUNCH_CHECK("svdx_concat2");
    return 0;
}

extern "C" int svdx_split2(const void* in, void* a, int Ca, void* b, int Cb, int rows, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_split2", Ca % 8 == 0 && Cb % 8 == 0 && al16(a) && al16(b) && al16(in));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((split2_kernel<T>), dim3(ew_blocks((long)rows * (Ca + Cb) / 8)), dim3(EW_THREADS),
                                             0, (hipStream_t)stream, (const T*)in, (T*)a, Ca, (T*)b, Cb, rows));
    SVDX_LAUNCH_CHECK("svdx_split2");
    return 0;
}

extern "C" int svdx_sum2x2(const void* in, void* out, int n_img, int h, int w, int C, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_sum2x2", C % 8 == 0 && al16(in) && al16(out));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sum2x2_kernel<T>), dim3(ew_blocks((long)n_img * h * w * C / 8)), dim3(EW_THREADS),
                                             0, (hipStream_t)stream, (const T*)in, (T*)out, n_img, h, w, C));
    SVDX_LAUNCH_CHECK("svdx_sum2x2");
    return 0;
}

extern "C" int svdx_cast_from_f32(const float* in, void* out, int64_t n, int dtype, void* stream) {
    EW_ALIGN_CHECK("svdx_cast_from_f32", al16(in) && al16(out) && n > 0);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cast_kernel<T>), dim3(std::max(1, ew_blocks(n / 8))), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, in, (T*)out, (long)n));
    SVDX_LAUNCH_CHECK("svdx_cast_from_f32");
    return 0;
}

extern "C" int svdx_nchw_to_rows(const float* in, void* out, int n_img, int C, int H, int W, int ld, float mul, int dtype,
                                 void* stream) {
    SVDX_CHECK_ARG(in && out && ld >= C, "svdx_nchw_to_rows: bad args");
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nchw_to_rows_kernel<T>), dim3(ew_blocks((long)n_img * H * W * ld)),
                                             dim3(EW_THREADS), 0, (hipStream_t)stream, in, (T*)out, n_img, C, H * W, ld, mul));
    SVDX_LAUNCH_CHECK("svdx_nchw_to_rows");
    return 0;
}

extern "C" int svdx_rows_to_nchw(const void* in, float* out, int n_img, int C, int H, int W, int ld, int dtype, void* stream) {
    SVDX_CHECK_ARG(in && out && ld >= C, "svdx_rows_to_nchw: bad args");
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rows_to_nchw_kernel<T>), dim3(ew_blocks((long)n_img * H * W * C)),
                                             dim3(EW_THREADS), 0, (hipStream_t)stream, (const T*)in, out, n_img, C, H * W, ld));
    SVDX_LAUNCH_CHECK("svdx_rows_to_nchw");
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void zero_spans_kernel(float* base, const int* __restrict__ spans) {
    const int off = spans[blockIdx.x * 2], cnt = spans[blockIdx.x * 2 + 1];
    f32x4* p = reinterpret_cast<f32x4*>(base + off);
    for (int i = threadIdx.x; i < cnt / 4; i += 256) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
}  // namespace

extern "C" int svdx_zero_spans(float* base, const int* spans, int n_spans, void* stream) {
    if (n_spans <= 0) return 0;
    SVDX_CHECK_ARG(base && spans && ((uintptr_t)base & 15) == 0, "svdx_zero_spans: bad args");
    hipLaunchKernelGGL(zero_spans_kernel, dim3(n_spans), dim3(256), 0, (hipStream_t)stream, base, spans);
    SVDX_LAUNCH_CHECK("svdx_zero_spans");
    return 0;
}

namespace {
// 16 bytes per lane, grid-stride; the unaligned head / tail bytes (never more than 15 each) by single lanes
__global__ __launch_bounds__(256) void zero_kernel(char* p, size_t bytes) {
    const size_t head = min(bytes, (size_t)((16 - ((uintptr_t)p & 15)) & 15));
    const size_t n16 = (bytes - head) / 16;
    f32x4* q = reinterpret_cast<f32x4*>(p + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) p[threadIdx.x] = 0;
        const size_t tail0 = head + n16 * 16;
        if (tail0 + threadIdx.x < bytes && threadIdx.x < 16) p[tail0 + threadIdx.x] = 0;
    }
}
}  // namespace

namespace {
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace

// Device-side time stamp on the launch stream: one single-lane kernel writes the constant-rate wall clock (svdx_wall_clock_khz) to
// *slot.  Two stamps around a launch, captured with it in a hipGraph, give that launch's duration under replay conditions -- bench.py's
// `roofline` uses them (HIP events around eager launches add ~5 us of marker handling per launch; stamps cost one ~1.5 us kernel
// boundary, calibrated from back-to-back stamps and subtracted).
extern "C" int svdx_stamp(uint64_t* slot, void* stream) {
    SVDX_CHECK_ARG(slot && ((uintptr_t)slot & 7) == 0, "svdx_stamp: slot must be 8-byte aligned");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slot));
    SVDX_LAUNCH_CHECK("svdx_stamp");
    return 0;
}
extern "C" int svdx_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

// A kernel, not hipMemsetAsync.  Round 4 (MI355X, ROCm 7.2, torch 2.10; profiles/r4_graph_replay_hazard.txt): once the process has issued
// ANY host <-> device copy or certain eager torch launches between two replays of a captured step, the hipMemsetAsync calls captured
// in it (memset NODES of the hipGraph: the statistics arenas cleared at the head of every sweep) no longer take effect in order with the
// kernel nodes around them -- every later replay computed with wrong GroupNorm statistics (loss 0.905 where the undisturbed replay and
// the eager step give 0.988; sometimes NaN), silently and for good.  A training loop copies a new batch in before every replay, so this
// hit any real use of GraphedStep; the fixed-batch bench and tests never saw it.  With a kernel in place of the memset the replays are
// bit-identical whatever runs between them (the memset form left the library in round 6; tests/test_e2e_gpu.py keeps the regression test).
extern "C" int svdx_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    const int blocks = (int)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (char*)p, bytes);
    SVDX_LAUNCH_CHECK("svdx_zero");
    return 0;
}
