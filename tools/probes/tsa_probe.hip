// Phase-by-phase shader-clock stamps of the fused temporal self-attention kernel (svd_xtend_amd/csrc/tsa.hip compiled into this
// probe with -DTSA_STAMPS).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSA_STAMPS [-DTSA_SKIP_QKV_STORE] tools/probes/tsa_probe.hip svd_xtend_amd/csrc/common.cpp -o /tmp/tsa_probe && /tmp/tsa_probe
#include "../../svd_xtend_amd/csrc/tsa.hip"
#include <vector>
#include <algorithm>

int main() {
    const int B = 1, T = 14, HW = 2560, heads = 5, C = 320;
    const long M = (long)B * T * HW;
    f16 *x, *wqkv, *wo, *n1, *qkv, *o, *h1; float *gamma, *beta, *bo, *stats; unsigned long long* stamps;
    hipMalloc(&x, M * C * 2); hipMalloc(&n1, M * C * 2); hipMalloc(&qkv, M * 3 * C * 2); hipMalloc(&o, M * C * 2); hipMalloc(&h1, M * C * 2);
    hipMalloc(&wqkv, 3 * C * C * 2); hipMalloc(&wo, C * C * 2); hipMalloc(&gamma, C * 4); hipMalloc(&beta, C * 4); hipMalloc(&bo, C * 4);
    hipMalloc(&stats, M * 8);
    std::vector<f16> hx(M * C), hw(3 * C * C), hwo(C * C);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = (f16)(2.f * rnd());
    for (auto& v : hw) v = (f16)(0.2f * rnd());
    for (auto& v : hwo) v = (f16)(0.2f * rnd());
    std::vector<float> ones(C, 1.f), zeros(C, 0.f);
    hipMemcpy(x, hx.data(), M * C * 2, hipMemcpyHostToDevice); hipMemcpy(wqkv, hw.data(), 3 * C * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(wo, hwo.data(), C * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(beta, zeros.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(bo, zeros.data(), C * 4, hipMemcpyHostToDevice);
    TsaParams p;
    p.x = x; p.gamma = gamma; p.beta = beta; p.eps = 1e-5f; p.wqkv = wqkv; p.wo = wo; p.bo = bo; p.cvec = nullptr; p.rv_ld = 0; p.rv_rpg = 1; p.rv_mod = 0;
    p.n1 = n1; p.stats = stats; p.qkv = qkv; p.o = o; p.h1 = h1; p.B = B; p.T = T; p.HW = HW; p.C = C; p.heads = heads;
    p.P = svdx_tsa_pixels_per_band(T, HW); p.sl2 = 0.125f * 1.4426950408889634f;
    p.x_bytes = (int)(M * C * 2); p.wqkv_bytes = 3 * C * C * 2; p.wo_bytes = C * C * 2;
    const int blocks = B * (HW / p.P);
    hipMalloc(&stamps, blocks * 16 * 8);
    p.stamps = stamps;
    const int lds = (C / 64) * (TSA_RP * 128) + TSA_NSTG * TSA_BST;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tsa_fwd_kernel<f16, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nb : {blocks, 1}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((tsa_fwd_kernel<f16, 5>), dim3(nb), dim3(512), lds, 0, p);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((tsa_fwd_kernel<f16, 5>), dim3(nb), dim3(512), lds, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nb * 16);
        hipMemcpy(h.data(), stamps, nb * 16 * 8, hipMemcpyDeviceToHost);
        double ph[7] = {0}, a3[6] = {0}, ex[4] = {0};
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < 5; ++i) ph[i] += (double)(h[b * 16 + i + 1] - h[b * 16 + i]);
            ph[5] += (double)h[b * 16 + 6]; ph[6] += (double)h[b * 16 + 7];
            for (int i = 1; i < 6; ++i) a3[i] += (double)h[b * 16 + 8 + i];
            ex[0] += (double)(h[b * 16 + 14] >> 32); ex[1] += (double)(h[b * 16 + 14] & 0xffffffffull);
            ex[2] += (double)(h[b * 16 + 15] >> 32); ex[3] += (double)(h[b * 16 + 15] & 0xffffffffull);
        }
        printf("blocks=%d  %.1f us/launch | mean cycles/block: dma %.0f  ln %.0f  qkv-gemm %.0f (waits %.0f)  attention %.0f  out-proj %.0f (waits %.0f)  total %.0f\n",
               nb, ms * 100.f, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[5] / nb, ph[3] / nb, ph[4] / nb, ph[6] / nb,
               (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / nb);
        printf("   attention, wave 0: issue-next-loads %.0f  stage-V %.0f  scores+softmax %.0f  outputs %.0f  final drain+barrier %.0f\n",
               a3[1] / nb, a3[2] / nb, a3[3] / nb, a3[4] / nb, a3[5] / nb);
        printf("   qkv-gemm: epilogues %.0f  final drain %.0f | out-proj: epilogues %.0f  final drain %.0f\n", ex[0] / nb, ex[1] / nb, ex[2] / nb, ex[3] / nb);
    }
    return 0;
}
