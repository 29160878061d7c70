// band_probe.hip - correctness (vs fp64 on the host) and timing of the fused pair-layer + regressor band kernel
// (csrc/gemm_hs_band.hip.h) against the two separate kernels it replaces, in one process.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/band_probe.hip -o tools/band_probe.bin
// Run:   band_probe.bin            correctness (M = 1024 + 24 rows, ragged last band), run-to-run identity
//        band_probe.bin time       timing at M = 262144 (alternating with pair + regressor), ablations
//        band_probe.bin stamps     per-workgroup cycle / wall stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>
#include <random>
#include "gemm_hs_band4.hip.h"
using namespace csi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static std::mt19937 rng(7);
static std::vector<float> rnd(size_t n, float scale, bool normal = true) {
    std::vector<float> h(n);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    for (auto& v : h) v = scale * (normal ? nd(rng) : ud(rng));
    return h;
}
template <typename T>
static T* dput(const std::vector<T>& h, size_t pad = 256) {
    T* d; CK(hipMalloc(&d, (h.size() + pad) * sizeof(T))); CK(hipMemset(d, 0, (h.size() + pad) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
static uint16_t* to_hs(const float* d_src, int ld, int rows, int cols, int* ldh, float scale) {
    *ldh = 2 * ((cols + 15) / 16 * 16);
    uint16_t* d; CK(hipMalloc(&d, (size_t)rows * *ldh * 2 + 4096)); CK(hipMemset(d, 0, (size_t)rows * *ldh * 2 + 4096));
    hipLaunchKernelGGL(f32_to_hs_kernel, dim3(2048), dim3(256), 0, 0, d_src, ld, rows, cols, d, *ldh, scale, 0);
    CK(hipDeviceSynchronize());
    return d;
}
static double rel_rows(const std::vector<double>& ref, const std::vector<float>& got, int M, int N, int* worst_row = nullptr) {
    double worst = 0;
    for (int m = 0; m < M; ++m) {
        double e = 0, r = 0;
        for (int n = 0; n < N; ++n) { const double d = got[(size_t)m * N + n] - ref[(size_t)m * N + n]; e += d * d; r += ref[(size_t)m * N + n] * ref[(size_t)m * N + n]; }
        const double v = std::sqrt(e / std::max(r, 1e-300));
        if (v > worst) { worst = v; if (worst_row) *worst_row = m; }
    }
    return worst;
}
template <typename F>
static double time_ms(F&& launch, int iters = 7) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const std::string mode = argc > 1 ? argv[1] : "check";
    const bool timing = mode == "time" || mode == "stamps" || mode == "loop";
    const int nt = argc > 2 ? atoi(argv[2]) : 32;
    const int K = 1024, N = 1024, NO = 234;
    const int M = getenv("BAND_M") ? atoi(getenv("BAND_M")) : (timing ? 262144 : 1024 + 24);      // BAND_M: rows of the timing runs (round 5: launch-tail experiment)
    const int M1 = (M + nt - 1) / nt;
    auto hW = rnd((size_t)N * K, 0.054f, false);
    auto hW2 = rnd((size_t)NO * N, 0.07f, false);
    auto hb = rnd(N, 0.1f), hsc = rnd(N, 0.3f), hsh = rnd(N, 0.1f), hb2 = rnd(NO, 0.1f);
    for (auto& v : hsc) v = 1.f + v;
    auto hL0 = rnd((size_t)M1 * K, 1.f), hT = rnd((size_t)nt * K, 1.f), hs0 = rnd(K, 0.3f);
    for (auto& v : hs0) v = 1.f + v;
    float *b = dput(hb), *sc = dput(hsc), *sh = dput(hsh);
    float *L0 = dput(hL0);
    const int sa = 4;
    auto wshift = [](const std::vector<float>& w) { float m = 0; for (float v : w) m = std::max(m, std::fabs(v)); int e; std::frexp(m, &e); return 13 - e; };
    // operands as the library prepares them: bn0's scale folded into W1, the pilot table pre-scaled, bn1's scale folded
    // into W2 and its shift into the regressor bias
    std::vector<float> hWf(hW.size()), hTs(hT.size());
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) hWf[(size_t)n * K + k] = (float)((double)hW[(size_t)n * K + k] * hs0[k]);
    for (size_t i = 0; i < hT.size(); ++i) hTs[i] = std::ldexp(hT[i], sa);
    const int swf = wshift(hWf);
    int ldbf;
    float* Wf = dput(hWf);
    uint16_t* Wfh = to_hs(Wf, K, N, K, &ldbf, std::ldexp(1.f, swf));
    float* Ts = dput(hTs);
    std::vector<float> hW2f(hW2.size()), hb2f(NO);
    for (int n = 0; n < NO; ++n) {
        double acc = hb2[n];
        for (int k = 0; k < N; ++k) { hW2f[(size_t)n * N + k] = (float)((double)hW2[(size_t)n * N + k] * hsc[k]); acc += (double)hsh[k] * hW2[(size_t)n * N + k]; }
        hb2f[n] = (float)acc;
    }
    const int sw2f = wshift(hW2f);
    int ldb2f;
    float* W2f = dput(hW2f);
    uint16_t* W2fh = to_hs(W2f, N, NO, N, &ldb2f, std::ldexp(1.f, sw2f));          // plain order: the separate regressor kernel
    uint16_t* W2p; CK(hipMalloc(&W2p, (size_t)256 * ldb2f * 2 + 4096)); CK(hipMemset(W2p, 0, (size_t)256 * ldb2f * 2 + 4096));
    hipLaunchKernelGGL(f32_to_hs_band_w2_kernel, dim3(256), dim3(256), 0, 0, W2f, N, NO, N, W2p, ldb2f, std::ldexp(1.f, sw2f));
    CK(hipDeviceSynchronize());
    float* b2f = dput(hb2f);
    printf("shifts: activations 2^%d, W1 2^%d, W2 2^%d; nt %d, M %d\n", sa, swf, sw2f, nt, M);

    float* O; CK(hipMalloc(&O, (size_t)M * NO * 4 + 4096));
    float* O2; CK(hipMalloc(&O2, (size_t)M * NO * 4 + 4096));
    uint16_t* Ch; CK(hipMalloc(&Ch, (size_t)(M + 256) * 2 * N * 2 + 4096)); CK(hipMemset(Ch, 0, (size_t)(M + 256) * 2 * N * 2 + 4096));
    unsigned* peak; CK(hipMalloc(&peak, 64)); CK(hipMemset(peak, 0, 64));

    BandArgs ba{};
    ba.L0 = L0; ba.Ts = Ts; ba.ldl = K; ba.nt = nt; ba.in_scale = std::ldexp(1.f, sa);
    ba.W1 = Wfh; ba.ldb1 = ldbf; ba.bias1 = b; ba.M = M; ba.K1 = K; ba.N1 = N;
    ba.acc_scale1 = std::ldexp(1.f, -(sa + swf)); ba.out_scale = std::ldexp(1.f, sa);
    ba.W2p = W2p; ba.ldb2 = ldb2f; ba.bias2 = b2f; ba.n2 = NO; ba.acc_scale2 = std::ldexp(1.f, -(sa + sw2f));
    ba.out = O; ba.ldo = NO; ba.peak = peak;
    const int nbands = (M + BAND_ROWS - 1) / BAND_ROWS;
    const size_t lds_band = (size_t)BAND_NSLOT * BAND_SLOT_BYTES + (size_t)(N + 256) * 4;
    auto kband = gemm_hs_band_kernel<0>;
    CK(hipFuncSetAttribute((const void*)kband, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_band));
    auto launch_band = [&] { hipLaunchKernelGGL(kband, dim3(nbands), dim3(BAND_THREADS), lds_band, 0, ba); };

    // the two kernels it replaces (library defaults: VM 3 pair kernel, blocked h2, generic regressor)
    const int tiles_m = (M + 255) / 256;
    GemmHsArgs gp{};
    gp.Bt = Wfh; gp.ldb = ldbf; gp.M = M; gp.N = N; gp.K = K; gp.k_per_split = K; gp.tiles_n = N / 256;
    gp.acc_scale = std::ldexp(1.f, -(sa + swf)); gp.bias = b; gp.scale = sc; gp.shift = sh; gp.C = Ch; gp.ldc = 2 * N;
    gp.out_scale = std::ldexp(1.f, sa); gp.xcd_cols = 1; gp.c_blk = 1;
    PairSrc ps{L0, Ts, K, nt};
    const size_t lds_pair = (size_t)5 * PP_SUBF * 4;
    auto kpair = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 0, false, 3>;
    CK(hipFuncSetAttribute((const void*)kpair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
    dim3 grid(pp_grid(tiles_m, gp.tiles_n));
    // separate path applies bn1 (scale, shift) in the pair epilogue and the plain W2 / bias
    std::vector<float> hW2s(hW2);
    const int sw2 = wshift(hW2s);
    int ldb2;
    float* W2 = dput(hW2s);
    uint16_t* W2h = to_hs(W2, N, NO, N, &ldb2, std::ldexp(1.f, sw2));
    float* b2 = dput(hb2);
    GemmHsArgs gr{};
    gr.A = Ch; gr.lda = 2 * N; gr.a_blk = 1; gr.Bt = W2h; gr.ldb = ldb2; gr.M = M; gr.N = NO; gr.K = N; gr.k_per_split = N; gr.tiles_n = 1;
    gr.C = O2; gr.ldc = NO; gr.acc_scale = std::ldexp(1.f, -(sa + sw2)); gr.bias = b2;
    dim3 gridr(pp_grid(tiles_m, 1));
    auto launch_sep = [&] {
        hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{});
        hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gr);
    };
    (void)W2fh;

    // ---- the assembly form (8 waves, two per SIMD): code object built by csrc/band_kernel_gen.py
    const char* hsaco = getenv("BAND8_HSACO");
    hipModule_t mod8 = nullptr;
    auto get8 = [&](const char* name) {
        hipFunction_t f = nullptr;
        if (mod8 && hipModuleGetFunction(&f, mod8, name) != hipSuccess) f = nullptr;
        return f;
    };
    if (hsaco && hipModuleLoad(&mod8, hsaco) != hipSuccess) { printf("cannot load %s\n", hsaco); mod8 = nullptr; }
    hipFunction_t k8 = get8(getenv("BAND8_NAME") ? getenv("BAND8_NAME") : "csi_band8");
    // the staged kernels (every variant without "nostage" in its name) stream the pilot table in slab order
    float* Tsw; CK(hipMalloc(&Tsw, ((size_t)(K / 16 + 1) * nt * 16 + 256) * sizeof(float)));
    hipLaunchKernelGGL(band_tsw_kernel<16>, dim3(256), dim3(256), 0, 0, Ts, K, nt, K, Tsw);
    CK(hipDeviceSynchronize());
    Band8Args a8_plain = band8_args(ba);
    BandArgs bas = ba; bas.Ts = Tsw;
    const Band8Args a8_staged = band8_args(bas);
    const char* k8name = getenv("BAND8_NAME") ? getenv("BAND8_NAME") : "csi_band8";
    Band8Args a8 = strstr(k8name, "nostage") ? a8_plain : a8_staged;
    auto launch8f = [&](hipFunction_t f, const Band8Args& args) {
        Band8Args tmp = args;
        size_t sz = sizeof(tmp);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &tmp, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        CK(hipModuleLaunchKernel(f, nbands, 1, 1, BAND8_THREADS, 1, 1, 0, 0, nullptr, extra));
    };
    auto launch_band8 = [&] { launch8f(k8, a8); };
    if (k8) printf("band8 code object %s loaded; shapes served: %d\n", hsaco, (int)band8_serves(ba));
    if (const char* only = getenv("BAND8_ONLY")) {      // bring-up: one launch of one variant, nothing else
        hipFunction_t f = get8(only);
        if (!f) { printf("no kernel %s\n", only); return 2; }
        printf("launching %s once (M = %d, %d bands)...\n", only, M, nbands);
        launch8f(f, a8);
        const hipError_t e = hipDeviceSynchronize();
        printf("   -> %s\n", hipGetErrorString(e));
        std::vector<float> o8((size_t)4 * NO);
        CK(hipMemcpy(o8.data(), O, o8.size() * 4, hipMemcpyDeviceToHost));
        printf("   out[0][0..3] = %g %g %g %g\n", o8[0], o8[1], o8[2], o8[3]);
        unsigned pw[16]; CK(hipMemcpy(pw, peak, 64, hipMemcpyDeviceToHost));
        printf("   host: out %p peak %p ldo %d M %d n2 %d\n   guard words:", (void*)O, (void*)peak, NO, M, NO);
        for (int i = 0; i < 16; ++i) printf(" %08x", pw[i]);
        printf("\n");
        return 0;
    }

    if (!timing) {
        std::vector<double> reff((size_t)M * NO);
        {
            std::vector<double> h1(K), h2(N);
            for (int m = 0; m < M; ++m) {
                const int pr = m / nt, t = m % nt;
                for (int k = 0; k < K; ++k) h1[k] = std::max((double)hL0[(size_t)pr * K + k] + hT[(size_t)t * K + k], 0.0) * hs0[k];
                for (int n = 0; n < N; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += h1[k] * hW[(size_t)n * K + k];
                    h2[n] = std::max(s + hb[n], 0.0) * hsc[n] + hsh[n];
                }
                for (int n = 0; n < NO; ++n) {
                    double s = 0;
                    for (int k = 0; k < N; ++k) s += h2[k] * hW2[(size_t)n * N + k];
                    reff[(size_t)m * NO + n] = s + hb2[n];
                }
            }
        }
        std::vector<float> go((size_t)M * NO), go2(go.size());
        CK(hipMemset(O, 0xff, (size_t)M * NO * 4));
        launch_band();
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        CK(hipMemcpy(go.data(), O, go.size() * 4, hipMemcpyDeviceToHost));
        int wr = -1;
        const double e1 = rel_rows(reff, go, M, NO, &wr);
        printf("band kernel (pair layer + regressor fused)  worst row rel err %.3g (row %d)\n", e1, wr);
        if (e1 > 1e-5) {
            for (int n = 0; n < 12; ++n) printf("   row %d col %d: got %.6g ref %.6g\n", wr, n, go[(size_t)wr * NO + n], reff[(size_t)wr * NO + n]);
            for (int m : {0, 1, 31, 32, 127, 128, M - 1}) {
                double e = 0, r = 0;
                for (int n = 0; n < NO; ++n) { const double d = go[(size_t)m * NO + n] - reff[(size_t)m * NO + n]; e += d * d; r += reff[(size_t)m * NO + n] * reff[(size_t)m * NO + n]; }
                printf("   row %d rel err %.3g\n", m, std::sqrt(e / r));
            }
        }
        CK(hipMemset(O, 0, (size_t)M * NO * 4));
        launch_band();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(go2.data(), O, go2.size() * 4, hipMemcpyDeviceToHost));
        printf("   second run bit-identical: %s\n", memcmp(go.data(), go2.data(), go.size() * 4) == 0 ? "yes" : "NO");
        unsigned pk[2]; CK(hipMemcpy(pk, peak, 8, hipMemcpyDeviceToHost));
        printf("   range guard words: %08x %08x\n", pk[0], pk[1]);
        launch_sep();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(go2.data(), O2, go2.size() * 4, hipMemcpyDeviceToHost));
        printf("separate pair + regressor kernels           worst row rel err %.3g\n", rel_rows(reff, go2, M, NO));
        double e8 = 0;
        if (k8) {
            CK(hipMemset(O, 0xff, (size_t)M * NO * 4));
            CK(hipMemset(peak, 0, 64));
            launch_band8();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(go.data(), O, go.size() * 4, hipMemcpyDeviceToHost));
            e8 = rel_rows(reff, go, M, NO, &wr);
            printf("band8 (assembly, 8 waves)                   worst row rel err %.3g (row %d)\n", e8, wr);
            if (!(e8 <= 1e-5)) {
                for (int n = 0; n < 8; ++n) printf("   row %d col %d: got %.6g ref %.6g\n", wr, n, go[(size_t)wr * NO + n], reff[(size_t)wr * NO + n]);
                for (int m : {0, 1, 31, 32, 33, 64, 127, 128, 129, 1023, 1024, M - 1}) {
                    double e = 0, r = 0;
                    for (int n = 0; n < NO; ++n) { const double d = go[(size_t)m * NO + n] - reff[(size_t)m * NO + n]; e += d * d; r += reff[(size_t)m * NO + n] * reff[(size_t)m * NO + n]; }
                    printf("   row %d rel err %.3g\n", m, std::sqrt(e / r));
                }
                for (int c0 = 0; c0 < NO; c0 += 16) {
                    double e = 0, r = 0;
                    for (int m = 0; m < 128; ++m) for (int n = c0; n < std::min(NO, c0 + 16); ++n) { const double d = go[(size_t)m * NO + n] - reff[(size_t)m * NO + n]; e += d * d; r += reff[(size_t)m * NO + n] * reff[(size_t)m * NO + n]; }
                    printf("   band 0, columns %d..%d rel err %.3g\n", c0, c0 + 15, std::sqrt(e / r));
                }
            }
            CK(hipMemset(O, 0, (size_t)M * NO * 4));
            launch_band8();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(go2.data(), O, go2.size() * 4, hipMemcpyDeviceToHost));
            printf("   second run bit-identical: %s\n", memcmp(go.data(), go2.data(), go.size() * 4) == 0 ? "yes" : "NO");
            CK(hipMemcpy(pk, peak, 8, hipMemcpyDeviceToHost));
            printf("   range guard words: %08x %08x\n", pk[0], pk[1]);
        }
        return e1 > 1e-5 || !(e8 <= 1e-5);
    }
    const double fl = 2.0 * M * N * K + 2.0 * M * N * NO;
    if (mode == "loop") {            // band_probe.bin loop <nt> <kernel name> <seconds>: one variant back to back (tools/power_probe.py samples power / clock meanwhile)
        const char* nm = argc > 3 ? argv[3] : "csi_band8";
        const double secs = argc > 4 ? atof(argv[4]) : 5.0;
        hipFunction_t f = get8(nm);
        if (!f) { printf("no kernel %s (BAND8_HSACO?)\n", nm); return 2; }
        const Band8Args& av = strstr(nm, "nostage") ? a8_plain : a8_staged;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        launch8f(f, av); CK(hipDeviceSynchronize());
        printf("LOOP START\n"); fflush(stdout);
        CK(hipEventRecord(e0));
        int n = 0; float ms = 0;
        do {
            for (int i = 0; i < 20; ++i) launch8f(f, av);
            n += 20;
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        } while (ms < secs * 1e3);
        printf("loop %s: %d launches, %.3f ms each, %.0f TF f16 executed (incl. 256-column regressor tile)\n", nm, n, ms / n,
               3 * (2.0 * M * N * K + 2.0 * M * N * 256) / (ms / n) / 1e9);
        return 0;
    }
    if (mode == "stamps") {
        unsigned long long* st; CK(hipMalloc(&st, (size_t)nbands * 12 * 8));
        BandArgs a = ba; a.stamps = st;
        auto l = [&] { hipLaunchKernelGGL(kband, dim3(nbands), dim3(BAND_THREADS), lds_band, 0, a); };
        l(); CK(hipDeviceSynchronize());
        CK(hipMemset(st, 0, (size_t)nbands * 12 * 8));
        l(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)nbands * 12);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc[3] = {0, 0, 0}, wall[3] = {0, 0, 0};
        unsigned long long t_min = ~0ull, t_max = 0; size_t n = 0;
        for (int bnd = 0; bnd < nbands; ++bnd) {
            const unsigned long long* p = &h[(size_t)bnd * 12];
            if (!p[0] || !p[6]) continue;
            for (int i = 0; i < 3; ++i) { cyc[i] += (double)(p[2 * (i + 1)] - p[2 * i]); wall[i] += (double)(p[2 * (i + 1) + 1] - p[2 * i + 1]); }
            t_min = std::min(t_min, p[1]); t_max = std::max(t_max, p[7]); ++n;
        }
        printf("band kernel %zu workgroups: head %.0f cyc (%.2f us)  band %.0f cyc (%.2f us)  output %.0f cyc (%.2f us)  | clock %.2f GHz | launch span %.1f us\n",
               n, cyc[0] / n, wall[0] / n / 100, cyc[1] / n, wall[1] / n / 100, cyc[2] / n, wall[2] / n / 100,
               (cyc[0] + cyc[1] + cyc[2]) / (wall[0] + wall[1] + wall[2]) / 10.0, (t_max - t_min) / 100.0);
        printf("(ideal band: 4 x (64 + 16) sub-steps x 24 MFMA x 32 cycles = 245760 cycles of the SIMD's matrix pipe)\n");
        if (k8) {
            Band8Args a = a8; a.stamps = st;
            auto l8 = [&] { launch8f(k8, a); };
            l8(); CK(hipDeviceSynchronize());
            CK(hipMemset(st, 0, (size_t)nbands * 12 * 8));
            l8(); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            double c2[3] = {0, 0, 0}, w2[3] = {0, 0, 0};
            t_min = ~0ull; t_max = 0; n = 0;
            for (int bnd = 0; bnd < nbands; ++bnd) {
                const unsigned long long* p = &h[(size_t)bnd * 12];
                if (!p[0] || !p[6]) continue;
                for (int i = 0; i < 3; ++i) { c2[i] += (double)(p[2 * (i + 1)] - p[2 * i]); w2[i] += (double)(p[2 * (i + 1) + 1] - p[2 * i + 1]); }
                t_min = std::min(t_min, p[1]); t_max = std::max(t_max, p[7]); ++n;
            }
            printf("band8 kernel %zu workgroups: head %.0f cyc (%.2f us)  band %.0f cyc (%.2f us)  output %.0f cyc (%.2f us)  | clock %.2f GHz | launch span %.1f us\n",
                   n, c2[0] / n, w2[0] / n / 100, c2[1] / n, w2[1] / n / 100, c2[2] / n, w2[2] / n / 100,
                   (c2[0] + c2[1] + c2[2]) / (w2[0] + w2[1] + w2[2]) / 10.0, (t_max - t_min) / 100.0);
        }
        return 0;
    }
    auto kb1 = gemm_hs_band_kernel<1>; auto kb2 = gemm_hs_band_kernel<2>; auto kb3 = gemm_hs_band_kernel<3>;
    auto kb4 = gemm_hs_band_kernel<7>; auto kb8 = gemm_hs_band_kernel<8>; auto kb16 = gemm_hs_band_kernel<23>; auto kb31 = gemm_hs_band_kernel<31>;
    for (auto k : {kb1, kb2, kb3, kb4, kb8, kb16, kb31}) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_band));
    for (int rep = 0; rep < 3; ++rep) {
        double ms = time_ms(launch_band);
        printf("band kernel (fused)          %.3f ms  %.0f TF fp32-equivalent (%.0f TF f16 executed incl. 256-column regressor tile)\n", ms, fl / ms / 1e9,
               3 * (2.0 * M * N * K + 2.0 * M * N * 256) / ms / 1e9);
        if (k8) {
            const double m8 = time_ms(launch_band8);
            printf("band8 (assembly, fused)      %.3f ms  %.0f TF fp32-equivalent (%.0f TF f16 executed incl. 256-column regressor tile)\n", m8, fl / m8 / 1e9,
                   3 * (2.0 * M * N * K + 2.0 * M * N * 256) / m8 / 1e9);
        }
        double ms2 = time_ms(launch_sep);
        printf("pair + regressor (separate)  %.3f ms  %.0f TF fp32-equivalent\n", ms2, fl / ms2 / 1e9);
        double p1 = time_ms([&] { hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
        double p2 = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gr); });
        printf("   pair alone %.3f ms, regressor alone %.3f ms\n", p1, p2);
        if (rep == 0) {
            auto t = [&](auto k, const char* what) {
                const double m = time_ms([&] { hipLaunchKernelGGL(k, dim3(nbands), dim3(BAND_THREADS), lds_band, 0, ba); });
                printf("   band, %-52s %.3f ms\n", what, m);
            };
            t(kb1, "no A-side requests (invalid)");
            t(kb2, "no conversion (invalid)");
            t(kb3, "neither (invalid)");
            t(kb4, "... and no LDS-DMA (invalid)");
            t(kb8, "full, no barriers (invalid)");
            t(kb16, "no A side, no DMA, no fragment reads (invalid)");
            t(kb31, "MFMA only (invalid)");
            for (const char* nm : {"csi_band8_noconv", "csi_band8_noreq", "csi_band8_noaside", "csi_band8_noaside_nodma", "csi_band8_noaside_noread", "csi_band8_skeleton", "csi_band8_skeleton_rnd", "csi_band8_skeleton_rnd_nobarrier", "csi_band8_noaside_rnd", "csi_band8_p2first_noaside", "csi_band8_p2first", "csi_band8_prio1", "csi_band8_prio0", "csi_band8", "csi_band8_nodma", "csi_band8_noread", "csi_band8_nobarrier", "csi_band8_stagger", "csi_band8_ownpieces", "csi_band8_nointerleave", "csi_band8_nostage", "csi_band8_nostage_noreq", "csi_band8"}) {
                hipFunction_t f = get8(nm);
                if (!f) continue;
                const Band8Args& av = strstr(nm, "nostage") ? a8_plain : a8_staged;
                const double m = time_ms([&] { launch8f(f, av); });
                printf("   %-58s %.3f ms\n", nm, m);
            }
        }
    }
    return 0;
}
