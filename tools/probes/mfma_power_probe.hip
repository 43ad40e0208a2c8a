// Which MFMA shape does the same FLOPs for less power?  (round 6: the step runs power-managed -- tools/clock_probe.py -- so energy per FLOP is rate.)
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_power_probe.hip -o tools/probes/mfma_power_probe -lpthread && tools/probes/mfma_power_probe
// Register-only loops (no LDS, no memory): every wave runs independent accumulator chains of ONE instruction --
//   v_mfma_f32_16x16x32_f16 (16 KFLOP, 1024 operand elements, 4 accumulator registers)   or
//   v_mfma_f32_32x32x16_f16 (32 KFLOP, 1024 operand elements, 16 accumulator registers) --
// for ~2 s per case on all 256 CUs, two waves per SIMD, while a host thread reads the device's pp_dpm_sclk / power1_average every 50 ms.
// Prints TFLOP/s, the median shader clock and socket power, and FLOP per joule.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int CH>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    } else {
        f32x16 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    }
    if (s == 12345.678f) out[0] = s;
}

static std::string find_hwmon(const std::string& dev) {
    std::string base = dev + "/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) return "";
    std::string r;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5) == 0) {
            for (const char* f : {"power1_average", "power1_input"}) {
                std::string p = base + "/" + e->d_name + "/" + f;
                std::ifstream t(p);
                if (t.good()) { r = p; break; }
            }
        }
        if (!r.empty()) break;
    }
    closedir(d);
    return r;
}

struct Sampler {
    std::string sclk, power;
    std::vector<double> clk, w;
    std::atomic<bool> stop{false};
    std::thread th;
    void start() {
        clk.clear(); w.clear(); stop = false;
        th = std::thread([this] {
            while (!stop) {
                std::ifstream f(sclk);
                std::string line;
                while (std::getline(f, line))
                    if (line.find('*') != std::string::npos) { size_t c = line.find(':'); clk.push_back(atof(line.c_str() + c + 1)); }
                if (!power.empty()) { std::ifstream p(power); double v = 0; p >> v; w.push_back(v / 1e6); }
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    void finish() { stop = true; th.join(); }
    static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
};

template <int SHAPE, int CH>
void run_case(const char* name, Sampler& s, float* out, double flop_per_mfma) {
    const int iters = 40000 / CH, waves = 8;
    auto launch = [&] { hipLaunchKernelGGL((mfma_loop<SHAPE, CH>), dim3(256), dim3(64 * waves), 0, 0, out, iters); };
    launch();
    hipDeviceSynchronize();
    s.start();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    while (el < 2.0) {
        for (int i = 0; i < 10; ++i) launch();
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    s.finish();
    const double flops = (double)n * 256 * waves * iters * CH * flop_per_mfma;
    const double tf = flops / el / 1e12, W = Sampler::med(s.w);
    printf("%-34s %8.1f TFLOP/s  sclk %6.0f MHz  %7.1f W  %6.2f GFLOP/J  (%zu samples)\n", name, tf, Sampler::med(s.clk), W, W > 0 ? tf * 1e3 / W : 0.0, s.clk.size());
}

int main() {
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    std::string id = bus;
    for (auto& ch : id) ch = (char)tolower(ch);
    const std::string dev = "/sys/bus/pci/devices/" + id;
    Sampler s;
    s.sclk = dev + "/pp_dpm_sclk";
    s.power = find_hwmon(dev);
    printf("# device %s  power node %s\n", id.c_str(), s.power.empty() ? "(none)" : s.power.c_str());
    float* out;
    hipMalloc(&out, 4);
    run_case<16, 8>("16x16x32 f16, 8 chains / wave", s, out, 2.0 * 16 * 16 * 32);
    run_case<32, 4>("32x32x16 f16, 4 chains / wave", s, out, 2.0 * 32 * 32 * 16);
    run_case<16, 8>("16x16x32 f16, 8 chains / wave", s, out, 2.0 * 16 * 16 * 32);
    run_case<32, 4>("32x32x16 f16, 4 chains / wave", s, out, 2.0 * 32 * 32 * 16);
    return 0;
}
