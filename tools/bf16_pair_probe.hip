// bf16_pair_probe.hip - where the bf16 first per-pair layer (gemm_bf16_pp_pair_kernel, BASELINE configs[2]) spends a tile:
// per-workgroup stamps (shader cycles + 10-ns wall ticks) at entry / after the prologue / after the main loop / after the
// epilogue, next to the plain bf16 ping-pong GEMM (A already bf16 in HBM) and the split-f16 pair kernel of the fp32 path on
// the same shape in the same process.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16_pair_probe.hip -o tools/bf16_pair_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>
#include <random>
#include "../dl-channel-estimation-mamimo_amd/csrc/gemm_hs.hip.h"
using namespace csi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static std::mt19937 rng(3);
static std::vector<float> rnd(size_t n, float scale) {
    std::vector<float> h(n);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = scale * nd(rng);
    return h;
}
template <typename T>
static T* dput(const std::vector<T>& h) {
    T* d; CK(hipMalloc(&d, (h.size() + 256) * sizeof(T))); CK(hipMemset(d, 0, (h.size() + 256) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <typename F>
static void report(const char* name, F&& launch, unsigned long long* st, unsigned nblocks, double flops, double ideal_cycles) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 7; ++i) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    CK(hipMemset(st, 0, (size_t)nblocks * 12 * 8));
    launch(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nblocks * 12);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc[3] = {0, 0, 0}, wall[3] = {0, 0, 0};
    size_t n = 0;
    for (unsigned blk = 0; blk < nblocks; ++blk) {
        const unsigned long long* p = &h[(size_t)blk * 12];
        if (!p[0] || !p[6]) continue;
        for (int i = 0; i < 3; ++i) { cyc[i] += (double)(p[2 * (i + 1)] - p[2 * i]); wall[i] += (double)(p[2 * (i + 1) + 1] - p[2 * i + 1]); }
        ++n;
    }
    printf("%-34s %.3f ms  %5.0f TFLOP/s | per tile: prologue %6.0f cyc %5.2f us, main loop %7.0f cyc %6.2f us (ideal %.0f cyc), epilogue %6.0f cyc %5.2f us | clock %.2f GHz\n",
           name, ts[3], flops / ts[3] / 1e9, cyc[0] / n, wall[0] / n / 100, cyc[1] / n, wall[1] / n / 100, ideal_cycles, cyc[2] / n, wall[2] / n / 100,
           (cyc[0] + cyc[1] + cyc[2]) / (wall[0] + wall[1] + wall[2]) / 10.0);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nt = 64, K = 1024, N = 1024, M = 262144, M1 = M / nt;
    auto hL0 = rnd((size_t)M1 * K, 1.f), hT = rnd((size_t)nt * K, 1.f), hW = rnd((size_t)N * K, 0.03f), hb = rnd(N, 0.1f);
    float *L0 = dput(hL0), *T = dput(hT), *bias = dput(hb);
    std::vector<float> one(N, 1.f), zero(N, 0.f);
    float *sc = dput(one), *sh = dput(zero);
    std::vector<uint16_t> hWb(hW.size());
    for (size_t i = 0; i < hW.size(); ++i) hWb[i] = f2bf(hW[i]);
    bf16_t* Wb = reinterpret_cast<bf16_t*>(dput(hWb));
    bf16_t* C; CK(hipMalloc(&C, (size_t)(M + 256) * N * 2));
    bf16_t* A; CK(hipMalloc(&A, (size_t)(M + 256) * K * 2)); CK(hipMemset(A, 0x3c, (size_t)(M + 256) * K * 2));
    const int tiles_m = M / 256, tiles_n = N / 256;
    const unsigned nblocks = pp_grid(tiles_m, tiles_n);
    unsigned long long* st; CK(hipMalloc(&st, (size_t)nblocks * 12 * 8));
    const double fl = 2.0 * M * N * K;

    GemmBf16Args g{};
    g.Bt = Wb; g.ldb = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.k_per_split = K; g.tiles_n = tiles_n;
    g.bias = bias; g.scale = sc; g.shift = sh; g.stamps = st;
    PairSrc ps{L0, T, K, nt};
    auto kpair = gemm_bf16_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true>;
    const size_t lds = (size_t)PPP_RING_FLOATS * 4;
    CK(hipFuncSetAttribute((const void*)kpair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // 32 sub-tiles of 32 k x 16 MFMAs per wave x 2 waves per SIMD x 32 cycles
    report("bf16 pair layer (A generated)", [&] { hipLaunchKernelGGL(kpair, dim3(nblocks), dim3(PP_THREADS), lds, 0, g, ps); }, st, nblocks, fl, 32.0 * 16 * 2 * 32);

    // split-f16 pair kernel of the fp32 path, same rows / widths (3 MFMAs per product)
    std::vector<uint16_t> hz((size_t)N * 2 * K, 0);
    {
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const float x = hW[(size_t)n * K + k] * 8192.f;
                const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
                memcpy(&hz[(size_t)n * 2 * K + (k >> 4) * 32 + (k & 15)], &hi, 2);
                memcpy(&hz[(size_t)n * 2 * K + (k >> 4) * 32 + 16 + (k & 15)], &lo, 2);
            }
    }
    uint16_t* Wh = dput(hz);
    uint16_t* Ch; CK(hipMalloc(&Ch, (size_t)(M + 256) * 2 * N * 2 + 4096));
    GemmHsArgs gh{};
    gh.Bt = Wh; gh.ldb = 2 * K; gh.C = Ch; gh.ldc = 2 * N; gh.M = M; gh.N = N; gh.K = K; gh.k_per_split = K; gh.tiles_n = tiles_n;
    gh.acc_scale = std::ldexp(1.f, -(4 + 13)); gh.out_scale = 16.f; gh.bias = bias; gh.scale = sc; gh.shift = sh; gh.stamps = st; gh.xcd_cols = 1; gh.c_blk = 1;
    std::vector<float> hTs(hT);
    for (auto& v : hTs) v *= 16.f;
    float* Ts = dput(hTs);
    PairSrc psh{L0, Ts, K, nt};
    auto khs = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 0, false, 3>;
    const size_t lds5 = (size_t)5 * PP_SUBF * 4;
    CK(hipFuncSetAttribute((const void*)khs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5));
    report("split-f16 pair layer (3 MFMA/product)", [&] { hipLaunchKernelGGL(khs, dim3(nblocks), dim3(PP_THREADS), lds5, 0, gh, psh, 16.f, PairRegArgs{}); }, st, nblocks, fl,
           64.0 * 24 * 2 * 32);
    printf("(the bf16 kernel executes 1 MFMA per product: at the same tile time it would show 3x the TFLOP/s of the split-f16 one)\n");
    return 0;
}
