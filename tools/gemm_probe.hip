// gemm_probe.hip - within-process A/B of gemm_f32_kernel variants at the bench shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip -o /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../dl-channel-estimation-mamimo_amd/csrc/gemm_f32.hip.h"
#include "../dl-channel-estimation-mamimo_amd/csrc/gemm_bf16.hip.h"
using namespace csi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float* dalloc(size_t n, bool rnd, float scale = 1.f) {
    float* d; CK(hipMalloc(&d, (n + 64) * 4)); CK(hipMemset(d, 0, (n + 64) * 4));
    if (rnd) { std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() & 0xffff) / 32768.f - 1.f);
        CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); }
    return d;
}

template <int EPI>
double run(const char* name, GemmArgs g, int splits, int iters, std::vector<float>* out = nullptr) {
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    const int tmm = (g.M + G_BM - 1) / G_BM;
    g.tiles_m = ((g.tiles_n >= 8) && tile_map_pays(tmm, g.tiles_n, splits)) ? tmm : 0;
    dim3 grid(g.tiles_m ? tile_map_grid(tmm, g.tiles_n) : tmm * g.tiles_n, 1, splits);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_f32_kernel<EPI>), grid, dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm_f32_kernel<EPI>), grid, dim3(256), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("%-34s med %.3f ms  %.1f TF   (best %.1f TF)\n", name, med, fl / med / 1e9, fl / ts[0] / 1e9);
    if (out) { out->resize(256); CK(hipMemcpy(out->data(), g.C, 256 * 4, hipMemcpyDeviceToHost)); }
    return med;
}

template <int EPI>
double run256(const char* name, GemmArgs g, int splits, int iters) {
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    const int tmm = (g.M + G2_BM - 1) / G2_BM;
    g.tiles_m = tile_map_pays(tmm, g.tiles_n, splits) ? tmm : 0;
    dim3 grid(g.tiles_m ? tile_map_grid(tmm, g.tiles_n) : tmm * g.tiles_n, 1, splits);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256_f32_kernel<EPI>), grid, dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm256_f32_kernel<EPI>), grid, dim3(256), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("%-34s med %.3f ms  %.1f TF   (best %.1f TF)\n", name, med, fl / med / 1e9, fl / ts[0] / 1e9);
    return med;
}

template <int EPI, int TPW>
double run_pair(const char* name, GemmArgs g, int iters, std::vector<float>* out = nullptr) {
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    dim3 grid(((g.M + G_BM - 1) / G_BM) * g.tiles_n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI, TPW>), grid, dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI, TPW>), grid, dim3(256), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("%-34s med %.3f ms  %.1f TF   (best %.1f TF)\n", name, med, fl / med / 1e9, fl / ts[0] / 1e9);
    if (out) { out->resize(256); CK(hipMemcpy(out->data(), g.C, 256 * 4, hipMemcpyDeviceToHost)); }
    return med;
}

template <int EPI, int DBG = 0>
double run_pair256(const char* name, GemmArgs g, int iters, std::vector<float>* out = nullptr) {
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    dim3 grid(((g.M + P2_BM - 1) / P2_BM) * g.tiles_n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pair_gemm256_f32_kernel<EPI, 1, 1, DBG>), grid, dim3(256), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((pair_gemm256_f32_kernel<EPI, 1, 1, DBG>), grid, dim3(256), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("%-34s med %.3f ms  %.1f TF   (best %.1f TF)\n", name, med, fl / med / 1e9, fl / ts[0] / 1e9);
    if (out) { out->resize(256); CK(hipMemcpy(out->data(), g.C, 256 * 4, hipMemcpyDeviceToHost)); }
    return med;
}

template <int EPI, bool OB, int WM, int WN, int MI, int NJ, int NS>
double run_bf16(const char* name, GemmBf16Args g, int splits, int iters) {
    constexpr int BM = WM * MI * 32, BN = WN * NJ * 32;
    g.tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(((g.M + BM - 1) / BM) * g.tiles_n, 1, splits);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OB, WM, WN, MI, NJ, NS>), grid, dim3(64 * WM * WN), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OB, WM, WN, MI, NJ, NS>), grid, dim3(64 * WM * WN), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("bf16 %-22s %dx%d waves %dx%d tiles/wave NS=%d  med %.3f ms  %.0f TF (best %.0f)\n", name, WM, WN, MI, NJ, NS, med, fl / med / 1e9, fl / ts[0] / 1e9);
    return med;
}

static bf16_t* dalloc_bf16(size_t n, float scale) {
    bf16_t* d; CK(hipMalloc(&d, (n + 128) * 2)); CK(hipMemset(d, 0, (n + 128) * 2));
    std::vector<bf16_t> h(n);
    for (size_t i = 0; i < n; ++i) { float f = scale * ((rand() & 0xffff) / 32768.f - 1.f); uint32_t u; memcpy(&u, &f, 4); h[i] = (bf16_t)(u >> 16); }
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}


// checksum of a bf16 output sample (first 64 rows x 1024 cols) for cross-kernel comparison
static double bf16_checksum(const void* C, int ldc) {
    std::vector<bf16_t> h((size_t)64 * ldc);
    CK(hipMemcpy(h.data(), C, h.size() * 2, hipMemcpyDeviceToHost));
    double s = 0; for (size_t i = 0; i < h.size(); ++i) { uint32_t u = (uint32_t)h[i] << 16; float f; memcpy(&f, &u, 4); s += f * (double)((i % 7) + 1); }
    return s;
}

template <int EPI, bool OB, int NSUB, int DBG = 0>
double run_bf16_pp(const char* name, GemmBf16Args g, int splits, int iters) {
    g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    dim3 grid(pp_grid(tiles_m, g.tiles_n), 1, splits);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OB, NSUB, DBG>), grid, dim3(PP_THREADS), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OB, NSUB, DBG>), grid, dim3(PP_THREADS), 0, 0, g);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    double med = ts[ts.size() / 2], fl = 2.0 * g.M * g.N * g.K;
    printf("bf16 %-22s ping-pong 256x256 NSUB=%d DBG=%d  med %.3f ms  %.0f TF (best %.0f)  checksum %.6g\n", name, NSUB, DBG, med, fl / med / 1e9, fl / ts[0] / 1e9, OB ? bf16_checksum(g.C, g.ldc) : 0.0);
    return med;
}


int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "bf16") {
        const int M = 262144, H = 1024;
        bf16_t* A = dalloc_bf16((size_t)M * H, 1.f);
        bf16_t* W = dalloc_bf16((size_t)H * H, 0.05f);
        float* bias = dalloc(H, true); float* sc = dalloc(H, true); float* sh = dalloc(H, true);
        bf16_t* C; CK(hipMalloc(&C, (size_t)M * H * 2));
        GemmBf16Args g{}; g.A = A; g.Bt = W; g.C = C; g.M = M; g.N = H; g.K = H; g.lda = H; g.ldb = H; g.ldc = H; g.k_per_split = H;
        g.bias = bias; g.scale = sc; g.shift = sh;
        for (int rep = 0; rep < 2; ++rep) {
            run_bf16<EPI_BIAS_RELU_AFFINE, true, 2, 2, 2, 2, 2>("hidden 1024x1024", g, 1, 7);
            printf("   checksum lock-step %.6g\n", bf16_checksum(g.C, g.ldc));
            CK(hipMemset(C, 0, (size_t)M * H * 2));
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 4>("hidden 1024x1024", g, 1, 7);
            CK(hipMemset(C, 0, (size_t)M * H * 2));
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 1>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 2>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 4>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 12>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 14>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 10>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 9>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 8>("hidden 1024x1024", g, 1, 7);
            run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5, 11>("hidden 1024x1024", g, 1, 7);
            { GemmBf16Args g4 = g; g4.K = 4096; g4.k_per_split = 4096; g4.lda = 1024; g4.ldb = 1024; g4.M = 65536;   // K-loop 4x longer over the same memory (rows wrap inside the buffers)
              run_bf16_pp<EPI_BIAS_RELU_AFFINE, true, 5>("K=4096 M=65536 (aliased)", g4, 1, 7); }
            run_bf16<EPI_BIAS_RELU_AFFINE, true, 2, 4, 4, 2, 2>("hidden 1024x1024", g, 1, 7);
            run_bf16<EPI_BIAS_RELU_AFFINE, true, 2, 2, 4, 4, 2>("hidden 1024x1024", g, 1, 7);
        }
        return 0;
    }
    const int nt = 32, M1 = 8192, M2 = M1 * nt, H = 1024, KL = 10240, NO = 234;
    float* ltf = dalloc((size_t)M1 * KL, true);
    float* W0 = dalloc((size_t)H * 10272, true, 0.02f);
    float* L0 = dalloc((size_t)M1 * H * 2, true);
    float* T = dalloc((size_t)nt * H, true);
    float* s0 = dalloc(H, true); float* t0 = dalloc(H, true);
    float* W1 = dalloc((size_t)H * H, true, 0.05f);
    float* W2 = dalloc((size_t)NO * H, true, 0.05f);
    float* bias = dalloc(H, true); float* sc = dalloc(H, true); float* sh = dalloc(H, true);
    float* h2 = dalloc((size_t)M2 * H, false);
    float* out = dalloc((size_t)M2 * NO, false);
    GemmArgs p{}; p.A = L0; p.lda = H; p.T = T; p.s0 = s0; p.t0 = t0; p.nt = nt; p.M = M2; p.K = H; p.N = H;
    p.Bt = W1; p.ldb = H; p.bias = bias; p.scale = sc; p.shift = sh; p.C = h2; p.ldc = H; p.k_per_split = H;
    GemmArgs l{}; l.A = ltf; l.lda = KL; l.Bt = W0; l.ldb = 10272; l.C = L0; l.ldc = H; l.M = M1; l.N = H; l.K = KL; l.k_per_split = 5120;
    GemmArgs r{}; r.A = h2; r.lda = H; r.Bt = W2; r.ldb = H; r.M = M2; r.N = NO; r.K = H; r.bias = bias; r.C = out; r.ldc = NO; r.k_per_split = H;
    std::vector<float> o1, o2, o3;
    for (int rep = 0; rep < 3; ++rep) {
        run_pair<EPI_BIAS_RELU_AFFINE, 1>("pair_dense tpw1", p, 7, &o1);
        run_pair256<EPI_BIAS_RELU_AFFINE>("pair_dense 256x128", p, 7, &o2);
        run_pair256<EPI_BIAS_RELU_AFFINE, 1>("pair_dense 256x128 no DMA", p, 7);
        run_pair256<EPI_BIAS_RELU_AFFINE, 2>("pair_dense 256x128 no h1 gen", p, 7);
        run_pair256<EPI_BIAS_RELU_AFFINE, 3>("pair_dense 256x128 no DMA/h1", p, 7);
        run_pair256<EPI_BIAS_RELU_AFFINE, 4>("pair_dense 256x128 no setprio", p, 7);
        run_pair256<EPI_BIAS_RELU_AFFINE, 8>("pair_dense 256x128 no stores", p, 7);
        run_pair256<EPI_BIAS_RELU_AFFINE, 11>("pair_dense 256x128 MFMA skeleton", p, 7);
        { double d = 0; for (int i = 0; i < 256; ++i) d = std::max(d, (double)std::fabs(o1[i] - o2[i])); printf("   max |diff| 128 vs 256 tile: %g (value %g)\n", d, o1[7]); }
        run<EPI_RAW>("layer0 (split 2)", l, 2, 7);
        run256<EPI_RAW>("layer0 (split 2) 256x128", l, 2, 7);
        run<EPI_BIAS>("regressor", r, 1, 7);
        run256<EPI_BIAS>("regressor 256x128", r, 1, 7);
    }
    return 0;
}
