// Store-shape microbenchmark: bytes per clock a CU moves to HBM/L2 for the store patterns an MFMA epilogue can produce.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/store_probe.hip -o tools/probes/store_probe && tools/probes/store_probe
// Each wave writes ITER instructions; pattern p decides how the 64 lanes of one instruction are laid over rows of `ld` bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Pat { const char* name; int bytes_per_lane; int lanes_per_row; };
// rows_per_instr = 64 / lanes_per_row; row segment = lanes_per_row * bytes_per_lane contiguous bytes

template <int BPL>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ld, int lanes_per_row, int iters, long block_stride, unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane / lanes_per_row, c = lane % lanes_per_row;
    const int rows = 64 / lanes_per_row, seg = lanes_per_row * BPL;
    char* base = out + (long)blockIdx.x * block_stride;
    const unsigned long long t0 = __builtin_readcyclecounter();
    // wave w owns column strip [w * seg, (w+1) * seg) of every row; instruction it covers rows it*rows .. +rows-1
    for (int it = 0; it < iters; ++it) {
        char* p = base + (long)(it * rows + r) * ld + wave * seg + c * BPL;
        if constexpr (BPL == 16) *reinterpret_cast<uint4*>(p) = uint4{(unsigned)it, 1u, 2u, 3u};
        else if constexpr (BPL == 8) *reinterpret_cast<uint2*>(p) = uint2{(unsigned)it, 1u};
        else *reinterpret_cast<unsigned*>(p) = (unsigned)it;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main() {
    const int ld = 1920;                      // bytes per row (960 fp16 columns, the q/k/v buffer of the fused temporal attention)
    const int iters = 256;
    char* out; unsigned long long* cyc;
    const long block_stride = 64L * iters * ld;            // upper bound on rows one block touches
    hipMalloc(&out, 256 * block_stride + (1 << 20)); hipMalloc(&cyc, 256 * 8);
    const Pat pats[] = {{"8B/lane, 4 lanes/row (16 rows x 32 B) [MFMA epilogue]", 8, 4},  {"16B/lane, 4 lanes/row (16 rows x 64 B)", 16, 4},
                        {"16B/lane, 8 lanes/row (8 rows x 128 B)", 16, 8},               {"8B/lane, 16 lanes/row (4 rows x 128 B)", 8, 16},
                        {"16B/lane, 6 lanes/row (10 rows x 96 B)", 16, 6},               {"4B/lane, 16 lanes/row (4 rows x 64 B)", 4, 16},
                        {"16B/lane, 16 lanes/row (4 rows x 256 B)", 16, 16},             {"8B/lane, 8 lanes/row (8 rows x 64 B)", 8, 8}};
    for (int blocks : {1, 256}) {
        for (const Pat& p : pats) {
            for (int rep = 0; rep < 2; ++rep) {
                if (p.bytes_per_lane == 16) hipLaunchKernelGGL(store_kernel<16>, dim3(blocks), dim3(512), 0, 0, out, ld, p.lanes_per_row, iters, block_stride, cyc);
                else if (p.bytes_per_lane == 8) hipLaunchKernelGGL(store_kernel<8>, dim3(blocks), dim3(512), 0, 0, out, ld, p.lanes_per_row, iters, block_stride, cyc);
                else hipLaunchKernelGGL(store_kernel<4>, dim3(blocks), dim3(512), 0, 0, out, ld, p.lanes_per_row, iters, block_stride, cyc);
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
            const int lanes_used = (64 / p.lanes_per_row) * p.lanes_per_row;
            const double bytes = 8.0 * iters * lanes_used * p.bytes_per_lane;
            printf("blocks=%3d  %-58s %8.0f cycles  %6.1f B/clk/CU  %5.1f clk/instr\n", blocks, p.name, mean, bytes / mean, mean / (8.0 * iters));
        }
    }
    return 0;
}
