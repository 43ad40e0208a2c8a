// Probe of ds_read_b64_tr_b16 semantics on gfx950: which lane's address feeds which output element.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_halfs, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int a = addr_halfs[threadIdx.x];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
    int h_addr[64]; short h_out[256];
    int *d_addr; short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 2; ++mode) {
        for (int l = 0; l < 64; ++l) h_addr[l] = mode == 0 ? l * 4 : ((l * 37) % 64) * 100 + 4 * (l % 3);   // multiples of 4 halfs
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15;
            for (int j = 0; j < 4; ++j) {
                int expect = h_addr[g * 16 + 4 * j + (i >> 2)] + (i & 3);   // hypothesis
                if (h_out[l * 4 + j] != (short)expect) ++bad;
            }
        }
        printf("mode %d hypothesis mismatches: %d\n", mode, bad);
        if (bad) for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
    }
    return 0;
}
