// sim_rt.h -- wave64 functional simulator for the libsvdx kernel sources (TEST INFRASTRUCTURE; nothing under svd_xtend_amd/ uses it).
//
// tests/sim/build_sim.py compiles the UNMODIFIED .hip sources of svd_xtend_amd/csrc for the host with this header standing in for
// <hip/hip_runtime.h>: every GPU thread becomes a fiber, a workgroup's fibers are scheduled wave by wave, and every cross-lane
// instruction (MFMA, DPP, transposing LDS read, shuffles, readfirstlane, votes) is a rendezvous of the wave's lanes that computes the
// documented lane mapping.  Memory-side asynchrony is modelled at its LATEST legal point: an LDS-DMA (`buffer_load ... lds`,
// `global_load_lds`) lands only when the issuing wave executes the `s_waitcnt vmcnt(N)` that retires it (or __syncthreads(), whose fence
// is vmcnt(0)); raw `s_barrier` retires nothing.  Waves run with maximal skew (one wave as far as it can go before the next is looked
// at, in ascending or descending order: SVDX_SIM_ORDER), so a missing wait or barrier shows up as wrong data, not as a lucky pass.
// Not modelled: timing, bank conflicts, the wide-store data hazard (tests/test_store_hazard.py scans the ISA for that one).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __HIP_DEVICE_COMPILE__ 1
#define SVDX_SIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

// ---- launch geometry -------------------------------------------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct SimLane { dim3 tidx; int lane, wave, tid; };
struct SimBlockCtx { dim3 bidx, bdim, gdim; char* dyn_lds; };
extern thread_local SimLane* sim_lane;
extern thread_local SimBlockCtx* sim_blk;
#define threadIdx (sim_lane->tidx)
#define blockIdx (sim_blk->bidx)
#define blockDim (sim_blk->bdim)
#define gridDim (sim_blk->gdim)
constexpr int warpSize = 64;

void sim_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) sim_launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })
inline char* sim_dyn_lds() { return sim_blk->dyn_lds; }

// ---- the little of the HIP host API the library touches ----------------------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "simulator"; }
template <typename F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
constexpr int hipDeviceAttributeWallClockRate = 1;
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 100000; return hipSuccess; }      // the simulator's wall clock: 100 MHz, a counter
inline unsigned long long wall_clock64() { static unsigned long long t = 0; return t += 100; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = 256; return hipSuccess; }

// ---- vector types ------------------------------------------------------------------------------------------------------------------
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
typedef _Float16 sim_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sim_h4 __attribute__((ext_vector_type(4)));
typedef __bf16 sim_b8 __attribute__((ext_vector_type(8)));
typedef short sim_s4 __attribute__((ext_vector_type(4)));
typedef float sim_f4 __attribute__((ext_vector_type(4)));
typedef unsigned sim_u4 __attribute__((ext_vector_type(4)));

// ---- scalar helpers HIP puts in the global namespace ----------------------------------------------------------------------------------
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
inline long min(long a, int b) { return a < b ? a : b; }
inline long min(int a, long b) { return a < b ? a : b; }
inline long max(long a, int b) { return a > b ? a : b; }
inline long max(int a, long b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline long long __float2ll_rn(float x) { return llrintf(x); }
inline float sim_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float sim_amdgcn_exp2f(float x) { return exp2f(x); }
inline float sim_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline float sim_amdgcn_sqrtf(float x) { return sqrtf(x); }
inline float sim_amdgcn_logf(float x) { return log2f(x); }

float atomicAdd(float* p, float v);
int atomicAdd(int* p, int v);
unsigned atomicAdd(unsigned* p, unsigned v);
unsigned long long atomicAdd(unsigned long long* p, unsigned long long v);
double atomicAdd(double* p, double v);

// ---- wave / workgroup rendezvous --------------------------------------------------------------------------------------------------------
enum SimOp { SIM_OP_MFMA32_F16 = 1, SIM_OP_MFMA32_BF16, SIM_OP_MFMA16_F16, SIM_OP_MFMA16_BF16, SIM_OP_DPP, SIM_OP_SHFL_XOR, SIM_OP_READFIRST,
             SIM_OP_TR16, SIM_OP_ANY, SIM_OP_ALL, SIM_OP_BARRIER, SIM_OP_BPERMUTE };
void sim_wave_op(int op, const void* in, void* out, uint64_t imm);
void sim_block_barrier(bool fence_vm);          // fence_vm: __syncthreads() (its fence retires every LDS-DMA of the wave first)
void sim_dma(char* lds_dst, const void* src, int size);   // src == nullptr: out-of-range buffer read, zeros
void sim_waitcnt_vm(int n);
inline void sim_waitcnt_lgkm(int) {}

inline void __syncthreads() { sim_block_barrier(true); }
inline void sim_amdgcn_s_barrier() { sim_block_barrier(false); }
inline void sim_amdgcn_wave_barrier() { sim_wave_op(SIM_OP_BARRIER, nullptr, nullptr, 0); }   // lanes are fibers here: lockstep has to be made
inline void sim_amdgcn_sched_barrier(int) {}
inline void sim_amdgcn_sched_group_barrier(int, int, int) {}
inline void sim_amdgcn_s_setprio(int) {}
inline void sim_amdgcn_s_sleep(int) {}
inline void sim_amdgcn_fence(int, const char*) {}              // one address space, sequentially consistent: nothing to order
inline void sim_amdgcn_s_waitcnt(int imm) {                      // gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4
    const int vm = (imm & 0xf) | ((imm >> 14) & 3) << 4;
    if (vm != 63) sim_waitcnt_vm(vm);
}

struct SimMfmaIn { float a[8], b[8], c[4]; };
template <typename V8> inline sim_f4 sim_mfma32(int op, V8 a, V8 b, sim_f4 c) {
    SimMfmaIn in;
    for (int i = 0; i < 8; ++i) { in.a[i] = (float)a[i]; in.b[i] = (float)b[i]; }
    for (int i = 0; i < 4; ++i) in.c[i] = c[i];
    float d[4];
    sim_wave_op(op, &in, d, 0);
    return sim_f4{d[0], d[1], d[2], d[3]};
}
inline sim_f4 sim_amdgcn_mfma_f32_16x16x32_f16(sim_h8 a, sim_h8 b, sim_f4 c, int, int, int) { return sim_mfma32(SIM_OP_MFMA32_F16, a, b, c); }
inline sim_f4 sim_amdgcn_mfma_f32_16x16x32_bf16(sim_b8 a, sim_b8 b, sim_f4 c, int, int, int) { return sim_mfma32(SIM_OP_MFMA32_BF16, a, b, c); }
inline sim_f4 sim_amdgcn_mfma_f32_16x16x16f16(sim_h4 a, sim_h4 b, sim_f4 c, int, int, int) {
    SimMfmaIn in;
    for (int i = 0; i < 4; ++i) { in.a[i] = (float)a[i]; in.b[i] = (float)b[i]; in.c[i] = c[i]; }
    float d[4];
    sim_wave_op(SIM_OP_MFMA16_F16, &in, d, 0);
    return sim_f4{d[0], d[1], d[2], d[3]};
}
inline float sim_bf16_bits_to_f(short s) { uint32_t u = (uint32_t)(uint16_t)s << 16; float f; memcpy(&f, &u, 4); return f; }
inline sim_f4 sim_amdgcn_mfma_f32_16x16x16bf16_1k(sim_s4 a, sim_s4 b, sim_f4 c, int, int, int) {
    SimMfmaIn in;
    for (int i = 0; i < 4; ++i) { in.a[i] = sim_bf16_bits_to_f(a[i]); in.b[i] = sim_bf16_bits_to_f(b[i]); in.c[i] = c[i]; }
    float d[4];
    sim_wave_op(SIM_OP_MFMA16_BF16, &in, d, 0);
    return sim_f4{d[0], d[1], d[2], d[3]};
}
inline int sim_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    int in[2] = {old, src}, out = 0;
    sim_wave_op(SIM_OP_DPP, in, &out, (uint64_t)(unsigned)ctrl | (uint64_t)(row_mask & 0xf) << 16 | (uint64_t)(bank_mask & 0xf) << 20 | (uint64_t)bound_ctrl << 24);
    return out;
}
inline int sim_amdgcn_readfirstlane(int v) { int out = v; sim_wave_op(SIM_OP_READFIRST, &v, &out, 0); return out; }
inline int sim_amdgcn_ds_bpermute(int byte_addr, int v) { int in[2] = {byte_addr, v}, out = 0; sim_wave_op(SIM_OP_BPERMUTE, in, &out, 0); return out; }
template <typename P> inline sim_s4 sim_amdgcn_ds_read_tr16_b64_v4i16(P lds_ptr) {
    const void* p = (const void*)lds_ptr;
    short out[4];
    sim_wave_op(SIM_OP_TR16, &p, out, 0);
    return sim_s4{out[0], out[1], out[2], out[3]};
}
template <typename V> inline V __shfl_xor(V v, int mask, int width = 64) {
    static_assert(sizeof(V) <= 8, "shuffle of at most 8 bytes");
    uint64_t in = 0, out = 0;
    memcpy(&in, &v, sizeof(V));
    sim_wave_op(SIM_OP_SHFL_XOR, &in, &out, (uint64_t)(unsigned)mask | (uint64_t)(unsigned)width << 32);
    V r;
    memcpy(&r, &out, sizeof(V));
    return r;
}
inline int __any(int pred) { int out = 0; sim_wave_op(SIM_OP_ANY, &pred, &out, 0); return out; }
inline int __all(int pred) { int out = 0; sim_wave_op(SIM_OP_ALL, &pred, &out, 0); return out; }

// ---- buffer resources and LDS-DMA ---------------------------------------------------------------------------------------------------------
struct SimRsrc { char* base; uint32_t num_records; };
typedef SimRsrc __amdgpu_buffer_rsrc_t;
inline SimRsrc sim_amdgcn_make_buffer_rsrc(void* p, short, int num_records, int) { return SimRsrc{(char*)p, (uint32_t)num_records}; }
// raw buffer addressing: the range check sees voffset + the instruction offset, the scalar offset only moves the address.  The sources use
// voffset = 0x80000000 for padding rows; anything where the two readings of the rule could disagree is reported.
const char* sim_buffer_addr(const SimRsrc& rs, int voffset, int soffset, int imm, int size);
template <typename P> inline void sim_amdgcn_raw_ptr_buffer_load_lds(SimRsrc rs, P lds, int size, int voffset, int soffset, int imm, int) {
    sim_dma((char*)lds + sim_lane->lane * size + imm, sim_buffer_addr(rs, voffset, soffset, imm, size), size);
}
// csrc/gemm.hip hidden_dma: buffer_load_dword[x4] ... lds issued from an asm statement, with a hand-built raw descriptor {base lo, base hi16, num_records, flags}
typedef int sim_desc4 __attribute__((ext_vector_type(4)));
template <int SIZE> inline void sim_hidden_dma(sim_desc4 d, char* lds, unsigned voff) {
    char* base = (char*)(((uint64_t)(uint32_t)d[0]) | ((uint64_t)((uint32_t)d[1] & 0xffffu) << 32));
    sim_amdgcn_raw_ptr_buffer_load_lds(SimRsrc{base, (uint32_t)d[2]}, lds, SIZE, (int)voff, 0, 0, 0);
}
template <typename G, typename P> inline void sim_amdgcn_global_load_lds(G gptr, P lds, int size, int imm, int) {
    sim_dma((char*)lds + sim_lane->lane * size + imm, (const char*)gptr + imm, size);
}
inline sim_u4 sim_amdgcn_raw_buffer_load_b128(SimRsrc rs, int voffset, int soffset, int) {
    sim_u4 r = {0u, 0u, 0u, 0u};
    const char* a = sim_buffer_addr(rs, voffset, soffset, 0, 16);
    if (a) memcpy(&r, a, 16);
    return r;
}
void sim_buffer_store(const void* v, int size, const SimRsrc& rs, int voffset, int soffset);
