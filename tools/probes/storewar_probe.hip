// Does a gfx950 vector store read its data VGPRs at issue?  Each wave issues buffer_store_dwordx4 v[20:23] and overwrites v20 right
// behind it (explicit physical registers, so the compiler cannot rename); the host counts stores whose first dword arrived as the
// marker, for 0..16 wait states between the two and for `s_waitcnt expcnt(0)`.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/storewar_probe.hip -o tools/probes/storewar_probe && tools/probes/storewar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STORE_THEN(GAP)                                                                                                                   \
    asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, 2\n v_mov_b32 v23, 3\n s_nop 4\n"                                   \
                 "buffer_store_dwordx4 v[20:23], %2, %3, 0 offen\n" GAP "v_mov_b32 v20, 0xdeadbeef\n" ::"v"(it), "v"(tid), "v"(off), "s"(rs) \
                 : "v20", "v21", "v22", "v23", "memory")

// MODE: number of independent SALU instructions (s_nop 0 = one wait state each) between the store and the overwrite; 100 = s_waitcnt expcnt(0)
template <int MODE>
__global__ __launch_bounds__(512) void k(void* out, int bytes, int iters) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        const int off = (it * n + tid) * 16;
        if (MODE == 0) STORE_THEN("");
        else if (MODE == 1) STORE_THEN("s_nop 0\n");
        else if (MODE == 2) STORE_THEN("s_nop 0\n s_nop 0\n");
        else if (MODE == 3) STORE_THEN("s_nop 0\n s_nop 0\n s_nop 0\n");
        else if (MODE == 4) STORE_THEN("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
        else if (MODE == 6) STORE_THEN("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
        else if (MODE == 8) STORE_THEN("s_nop 7\n");
        else if (MODE == 16) STORE_THEN("s_nop 7\n s_nop 7\n");
        else STORE_THEN("s_waitcnt expcnt(0)\n");
    }
}

// the same experiment with an 8-byte store (LLVM assumes no hazard at all for <= 64 bits of data): MODE = wait states
template <int MODE>
__global__ __launch_bounds__(512) void k2(void* out, int bytes, int iters) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        const int off = (it * n + tid) * 16;
        if (MODE == 0)
            asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n s_nop 4\n buffer_store_dwordx2 v[20:21], %2, %3, 0 offen\n v_mov_b32 v20, 0xdeadbeef\n"
                         ::"v"(it), "v"(tid), "v"(off), "s"(rs) : "v20", "v21", "memory");
        else
            asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n s_nop 4\n buffer_store_dwordx2 v[20:21], %2, %3, 0 offen\n s_nop 0\n v_mov_b32 v20, 0xdeadbeef\n"
                         ::"v"(it), "v"(tid), "v"(off), "s"(rs) : "v20", "v21", "memory");
    }
}

template <int MODE>
long run2(unsigned* out, long bytes, int blocks, int iters, std::vector<unsigned>& h) {
    long bad_total = 0;
    const long n = (long)blocks * 512;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, bytes);
        hipLaunchKernelGGL(k2<MODE>, dim3(blocks), dim3(512), 0, 0, out, (int)bytes, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost);
        for (long i = 0; i < n * iters; ++i)
            if (h[i * 4] != (unsigned)(i / n)) ++bad_total;
    }
    return bad_total;
}

template <int MODE>
long run(unsigned* out, long bytes, int blocks, int iters, std::vector<unsigned>& h) {
    long bad_total = 0;
    const long n = (long)blocks * 512;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, bytes);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, (int)bytes, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost);
        for (long i = 0; i < n * iters; ++i)
            if (h[i * 4] != (unsigned)(i / n)) ++bad_total;
    }
    return bad_total;
}

int main() {
    const int blocks = 256, iters = 64;
    const long n = (long)blocks * 512, bytes = n * iters * 16;
    unsigned* out;
    hipMalloc(&out, bytes);
    std::vector<unsigned> h(bytes / 4);
    const long total = 3 * n * iters;
    printf("stores whose first data dword arrived as the value written to the register AFTER the store was issued, of %ld:\n", total);
    printf("  0 wait states: %ld\n", run<0>(out, bytes, blocks, iters, h));
    printf("  1 wait state : %ld\n", run<1>(out, bytes, blocks, iters, h));
    printf("  2 wait states: %ld\n", run<2>(out, bytes, blocks, iters, h));
    printf("  3 wait states: %ld\n", run<3>(out, bytes, blocks, iters, h));
    printf("  4 wait states: %ld\n", run<4>(out, bytes, blocks, iters, h));
    printf("  6 wait states: %ld\n", run<6>(out, bytes, blocks, iters, h));
    printf("  s_nop 7 (8)  : %ld\n", run<8>(out, bytes, blocks, iters, h));
    printf("  2 x s_nop 7  : %ld\n", run<16>(out, bytes, blocks, iters, h));
    printf("  s_waitcnt expcnt(0): %ld\n", run<100>(out, bytes, blocks, iters, h));
    printf("buffer_store_dwordx2 (8 bytes), same experiment:\n");
    printf("  0 wait states: %ld\n", run2<0>(out, bytes, blocks, iters, h));
    printf("  1 wait state : %ld\n", run2<1>(out, bytes, blocks, iters, h));
    return 0;
}
