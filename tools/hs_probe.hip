// hs_probe.hip - correctness (vs fp64 on the host) and timing of the split-f16 GEMM kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hs_probe.hip -o /tmp/hs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>
#include <random>
#include "../dl-channel-estimation-mamimo_amd/csrc/gemm_hs.hip.h"
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
    hipLaunchKernelGGL(f32_to_hs_kernel, dim3(2048), dim3(256), 0, 0, d_src, ld, rows, cols, d, *ldh, scale);
    CK(hipDeviceSynchronize());
    return d;
}
// power experiment: zero the low `bits` mantissa bits of every lo half (hi halves untouched) of an hs matrix
__global__ void mask_lo_kernel(uint16_t* m, size_t n_groups, unsigned mask) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * blockDim.x)
        for (int i = 0; i < 16; ++i) m[g * 32 + 16 + i] &= (uint16_t)mask;
}
static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }

// norm-relative error per row, worst row
static double rel_rows(const std::vector<double>& ref, const std::vector<float>& got, int M, int N) {
    double worst = 0;
    for (int m = 0; m < M; ++m) {
        double e = 0, r = 0;
        for (int n = 0; n < N; ++n) { const double d = got[(size_t)m * N + n] - ref[(size_t)m * N + n]; e += d * d; r += ref[(size_t)m * N + n] * ref[(size_t)m * N + n]; }
        worst = std::max(worst, std::sqrt(e / std::max(r, 1e-300)));
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

// hs_lo_pair (v_fma_mix) against hs_split2 (cvt / sub / cvt) on every kind of value: must agree bit for bit
__global__ void split_check_kernel(const float* x, size_t n, unsigned* mismatches, unsigned* first) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = x[2 * i], b = x[2 * i + 1];
        uint32_t h0, l0;
        hs_split2(a, b, h0, l0);
        const uint32_t h1 = hs_hi_pair(a, b), l1 = hs_lo_pair(a, b, h1);
        if (h0 != h1 || l0 != l1) {
            if (atomicAdd(mismatches, 1u) == 0) { first[0] = __builtin_bit_cast(unsigned, a); first[1] = __builtin_bit_cast(unsigned, b); first[2] = l0; first[3] = l1; }
        }
    }
}
static int split_check() {
    const size_t n = 1 << 24;
    std::vector<float> h(n);
    std::mt19937 r(11);
    for (size_t i = 0; i < n; ++i) {
        // random bit patterns with a biased exponent range that covers f16 normals, denormals, overflow and zero
        const unsigned mant = r() & 0x7fffffu, sign = (r() & 1u) << 31;
        const unsigned ex = 127 - 40 + (r() % 64);
        unsigned bits = sign | (ex << 23) | mant;
        if (i % 97 == 0) bits = sign;                       // +-0
        if (i % 101 == 0) bits = sign | (ex << 23);         // powers of two (ties)
        memcpy(&h[i], &bits, 4);
    }
    float* d = dput(h);
    unsigned* cnt; CK(hipMalloc(&cnt, 32)); CK(hipMemset(cnt, 0, 32));
    hipLaunchKernelGGL(split_check_kernel, dim3(1024), dim3(256), 0, 0, d, n, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    unsigned res[5];
    CK(hipMemcpy(res, cnt, 20, hipMemcpyDeviceToHost));
    printf("split check: %u mismatches of %zu pairs between v_fma_mix lo halves and cvt/sub/cvt", res[0], n / 2);
    if (res[0]) printf(" (first: a=%08x b=%08x lo %08x vs %08x)", res[1], res[2], res[3], res[4]);
    printf("\n");
    return res[0] != 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "split") return split_check();
    const bool timing = argc > 1 && (std::string(argv[1]) == "time" || std::string(argv[1]) == "loop");
    const float in_mag = argc > 2 ? atof(argv[2]) : 1.f;
    const bool relu_zeros = argc > 3 && std::string(argv[3]) == "relu";     // A operands with ~50 % exact zeros (power experiment)
    const bool stamps = argc > 4 && std::string(argv[4]) == "stamps";       // per-workgroup phase timing
    const bool loop = argc > 1 && std::string(argv[1]) == "loop";           // loop <mag> <relu|-> <mfma|hs2hs|pair|regressor|cast> <seconds>: for tools/power_probe.py
    const int nt = 32, K = 1024, N = 1024, NO = 234;
    const int M = timing ? 262144 : 1000 * 1 + 24;        // ragged last tile in the check
    const int M1 = (M + nt - 1) / nt;
    // ---------------- operands
    auto hA = rnd((size_t)M * K, in_mag);
    auto hW = rnd((size_t)N * K, 0.054f, false);
    auto hW2 = rnd((size_t)NO * K, 0.07f, false);
    auto hb = rnd(N, 0.1f), hsc = rnd(N, 0.3f), hsh = rnd(N, 0.1f);
    for (auto& v : hsc) v = 1.f + v;
    if (relu_zeros) for (auto& v : hA) v = std::max(v, 0.f);
    auto hL0 = rnd((size_t)M1 * K, in_mag), hT = rnd((size_t)nt * K, in_mag), hs0 = rnd(K, 0.3f), ht0 = rnd(K, 0.1f);
    for (auto& v : hs0) v = 1.f + v;
    for (auto& v : ht0) v = 0.f;          // the pair kernel applies no BN shift (the library folds it into the next layer's bias)
    float *A = dput(hA), *W = dput(hW), *W2 = dput(hW2), *b = dput(hb), *sc = dput(hsc), *sh = dput(hsh);
    float *L0 = dput(hL0), *T = dput(hT);
    const int sa = 4;
    auto wshift = [](const std::vector<float>& w) { float m = 0; for (float v : w) m = std::max(m, std::fabs(v)); int e; std::frexp(m, &e); return 13 - e; };
    const int sw = wshift(hW), sw2 = wshift(hW2);
    printf("shifts: activations 2^%d, W 2^%d, W2 2^%d\n", sa, sw, sw2);
    int lda, ldb, ldb2;
    uint16_t* Ah = to_hs(A, K, M, K, &lda, std::ldexp(1.f, sa));
    uint16_t* Wh = to_hs(W, K, N, K, &ldb, std::ldexp(1.f, sw));
    uint16_t* W2h = to_hs(W2, K, NO, K, &ldb2, std::ldexp(1.f, sw2));
    // pair kernel operands as the library prepares them: bn0's scale folded into the rows of the weights, the pilot
    // table pre-scaled by the (power-of-two) activation scale
    std::vector<float> hWf(hW.size()), hTs(hT.size());
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) hWf[(size_t)n * K + k] = (float)((double)hW[(size_t)n * K + k] * hs0[k]);
    for (size_t i = 0; i < hT.size(); ++i) hTs[i] = std::ldexp(hT[i], sa);
    const int swf = wshift(hWf);
    int ldbf;
    float* Wf = dput(hWf);
    uint16_t* Wfh = to_hs(Wf, K, N, K, &ldbf, std::ldexp(1.f, swf));
    float* Ts = dput(hTs);

    float* C; CK(hipMalloc(&C, (size_t)M * N * 4 + 4096));
    uint16_t* Ch; CK(hipMalloc(&Ch, (size_t)M * 2 * N * 2 + 4096)); CK(hipMemset(Ch, 0, (size_t)M * 2 * N * 2 + 4096));
    float* O; CK(hipMalloc(&O, (size_t)M * NO * 4 + 4096));

    GemmHsArgs g{};
    g.A = Ah; g.lda = lda; g.Bt = Wh; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.k_per_split = K; g.tiles_n = (N + 255) / 256;
    g.acc_scale = std::ldexp(1.f, -(sa + sw)); g.bias = b; g.scale = sc; g.shift = sh; g.C = C; g.ldc = N;
    const int tiles_m = (M + 255) / 256;
    dim3 grid(pp_grid(tiles_m, g.tiles_n));
    GemmHsArgs gh = g; gh.C = Ch; gh.ldc = 2 * N; gh.out_scale = std::ldexp(1.f, sa);
    GemmHsArgs gr = g; gr.A = Ch; gr.lda = 2 * N; gr.Bt = W2h; gr.ldb = ldb2; gr.N = NO; gr.tiles_n = 1; gr.C = O; gr.ldc = NO;
    gr.acc_scale = std::ldexp(1.f, -(sa + sw2));
    dim3 gridr(pp_grid(tiles_m, 1));
    GemmHsArgs gp = gh;                                   // pair kernel: A generated from L0 / T
    gp.Bt = Wfh; gp.ldb = ldbf; gp.acc_scale = std::ldexp(1.f, -(sa + swf));
    PairSrc ps{L0, Ts, K, nt};
    // the library's defaults: pair layer with hand-counted memory operations and merged segments (VM 3, five ring slots),
    // layer 0 with the extra look-ahead (VM 2); kpair0 / kcast0 = the round-1/2 schedule (builtin LDS-DMA, drain per sub-tile)
    const size_t lds_pair = (size_t)5 * PP_SUBF * 4;
    auto kpair = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 0, false, 3>;
    auto kpair0 = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false>;
    CK(hipFuncSetAttribute((const void*)kpair0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PPP_RING_FLOATS * 4)));
    auto kcast = gemm_hs_pp_pair_kernel<EPI_RAW, false, true, 0, false, 2>;
    auto kcast0 = gemm_hs_pp_pair_kernel<EPI_RAW, false, true>;
    CK(hipFuncSetAttribute((const void*)kcast0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PPP_RING_FLOATS * 4)));
    CK(hipFuncSetAttribute((const void*)kpair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
    CK(hipFuncSetAttribute((const void*)kcast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PPP_RING_FLOATS * 4)));
    GemmHsArgs gc = g; gc.C = C; gc.ldc = N;              // CAST: A = fp32 rows, raw output
    PairSrc pc{A, nullptr, K, 1};

    // ---- fused pair + regressor (hs_fused_regressor): W2 rows scaled by the pair layer's BN scale, its shift in the bias
    std::vector<float> hW2f(hW2.size()), hb2f(NO);
    for (int n = 0; n < NO; ++n) {
        double acc = hb[n];
        for (int k = 0; k < K; ++k) { hW2f[(size_t)n * K + k] = (float)((double)hW2[(size_t)n * K + k] * hsc[k]); acc += (double)hsh[k] * hW2[(size_t)n * K + k]; }
        hb2f[n] = (float)acc;
    }
    const int sw2f = wshift(hW2f);
    int ldb2f;
    float* W2f = dput(hW2f);
    uint16_t* W2fh = to_hs(W2f, K, NO, K, &ldb2f, std::ldexp(1.f, sw2f));
    float* b2f = dput(hb2f);
    const int tiles_nf = N / 256;
    float* fslabs; CK(hipMalloc(&fslabs, (size_t)(tiles_nf - 1) * M * NO * 4 + 4096));
    unsigned* fflags; CK(hipMalloc(&fflags, (size_t)(tiles_m + 64) * 4)); CK(hipMemset(fflags, 0, (size_t)(tiles_m + 64) * 4));
    PairRegArgs rg{};
    rg.B2 = W2fh; rg.ldb2 = ldb2f; rg.n2 = NO; rg.bias2 = b2f; rg.out = O; rg.slabs = fslabs; rg.flags = fflags; rg.err = fflags + tiles_m + 8;
    rg.acc_scale2 = std::ldexp(1.f, -(sa + sw2f));
    GemmHsArgs gf = gp;                     // stage 1 = the pair kernel's operands; its BN scale / shift are NOT applied (folded above)
    gf.out_scale = std::ldexp(1.f, sa);
    auto kfuse = gemm_hs_pp_pair_kernel<EPI_RAW, false, false, 0, true>;
    CK(hipFuncSetAttribute((const void*)kfuse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PR_LDS_FLOATS * 4)));
    auto launch_fused = [&] {
        hipMemsetAsync(fflags, 0, (size_t)tiles_m * 4, 0);
        hipLaunchKernelGGL(kfuse, grid, dim3(PP_THREADS), PR_LDS_FLOATS * 4, 0, gf, ps, std::ldexp(1.f, sa), rg);
    };

    if (!timing) {
        // ---- fp64 references
        std::vector<double> ref((size_t)M * N), refh((size_t)M * N), refraw((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
                refraw[(size_t)m * N + n] = s;
                ref[(size_t)m * N + n] = s + hb[n];
                refh[(size_t)m * N + n] = std::max(s + hb[n], 0.0) * hsc[n] + hsh[n];
            }
        std::vector<float> got((size_t)M * N);
        hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), grid, dim3(PP_THREADS), 0, 0, g);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost));
        printf("generic  fp32 out (bias)            worst row rel err %.3g\n", rel_rows(ref, got, M, N));
        // fp32 sgemm-like reference error for scale: plain float accumulation
        {
            std::vector<float> f32((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(hA[(size_t)m * K + k], hW[(size_t)n * K + k], s); f32[(size_t)m * N + n] = s + hb[n]; }
            printf("   (host fp32 sequential fma         worst row rel err %.3g)\n", rel_rows(ref, f32, M, N));
        }
        // hs output
        hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> hh((size_t)M * 2 * N);
        CK(hipMemcpy(hh.data(), Ch, hh.size() * 2, hipMemcpyDeviceToHost));
        auto decode = [&](std::vector<float>& out) {
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    const size_t o = (size_t)m * 2 * N + (n >> 4) * 32 + (n & 15);
                    out[(size_t)m * N + n] = std::ldexp(h2f(hh[o]) + h2f(hh[o + 16]), -sa);
                }
        };
        decode(got);
        printf("generic  hs out (bias, relu, bn)    worst row rel err %.3g\n", rel_rows(refh, got, M, N));
        // regressor on the hs activations
        std::vector<double> refo((size_t)M * NO);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < NO; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += refh[(size_t)m * N + k] * hW2[(size_t)n * K + k];
                refo[(size_t)m * NO + n] = s + hb[n];
            }
        hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gr);
        CK(hipDeviceSynchronize());
        std::vector<float> go((size_t)M * NO);
        CK(hipMemcpy(go.data(), O, go.size() * 4, hipMemcpyDeviceToHost));
        printf("two layers, N=234 fp32 out          worst row rel err %.3g\n", rel_rows(refo, go, M, NO));
        // pair kernel
        std::vector<double> refp((size_t)M * N);
        {
            std::vector<double> h1(K);
            for (int m = 0; m < M; ++m) {
                const int pr = m / nt, t = m % nt;
                for (int k = 0; k < K; ++k) h1[k] = std::max((double)hL0[(size_t)pr * K + k] + hT[(size_t)t * K + k], 0.0) * hs0[k] + ht0[k];
                for (int n = 0; n < N; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += h1[k] * hW[(size_t)n * K + k];
                    refp[(size_t)m * N + n] = std::max(s + hb[n], 0.0) * hsc[n] + hsh[n];
                }
            }
        }
        CK(hipMemset(Ch, 0, (size_t)M * 2 * N * 2));
        hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{});
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hh.data(), Ch, hh.size() * 2, hipMemcpyDeviceToHost));
        decode(got);
        printf("pair (A generated), hs out          worst row rel err %.3g\n", rel_rows(refp, got, M, N));
        // fused pair + regressor: out = (relu(h1 W1 + b) sc + sh) W2 + b2, with sc / sh folded into W2 / b2
        {
            std::vector<double> reff((size_t)M * NO);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < NO; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += refp[(size_t)m * N + k] * hW2[(size_t)n * K + k];
                    reff[(size_t)m * NO + n] = s + hb[n];
                }
            CK(hipMemset(O, 0, (size_t)M * NO * 4));
            launch_fused();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(go.data(), O, go.size() * 4, hipMemcpyDeviceToHost));
            unsigned e = 0; CK(hipMemcpy(&e, rg.err, 4, hipMemcpyDeviceToHost));
            printf("pair + fused regressor, fp32 out    worst row rel err %.3g   (timeout flag %u)\n", rel_rows(reff, go, M, NO), e);
            std::vector<float> go2(go.size());
            launch_fused();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(go2.data(), O, go2.size() * 4, hipMemcpyDeviceToHost));
            printf("   second run bit-identical: %s\n", memcmp(go.data(), go2.data(), go.size() * 4) == 0 ? "yes" : "NO");
        }
        // CAST kernel, two K splits
        GemmHsArgs g2 = gc; g2.k_per_split = 512;
        float* slabs; CK(hipMalloc(&slabs, (size_t)2 * M * N * 4 + 4096));
        g2.C = slabs;
        hipLaunchKernelGGL(kcast, dim3(grid.x, 1, 2), dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, g2, pc, std::ldexp(1.f, sa), PairRegArgs{});
        CK(hipDeviceSynchronize());
        std::vector<float> s2((size_t)2 * M * N);
        CK(hipMemcpy(s2.data(), slabs, s2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * N; ++i) got[i] = s2[i] + s2[i + (size_t)M * N];
        printf("cast (A from fp32 rows), 2 slabs    worst row rel err %.3g\n", rel_rows(refraw, got, M, N));
        return 0;
    }
    const double fl = 2.0 * M * N * K;
    if (argc > 4 && std::string(argv[4]) == "masklo") {
        // does the matrix pipe draw less with fewer significant bits in the lo halves?  MFMA-only form and full kernel,
        // lo halves of A and W truncated to 10 (as stored), 8, 6, 4, 0 mantissa bits
        // fresh copies so that every variant starts from the full-precision halves
        std::vector<uint16_t> a0((size_t)M * lda), w0((size_t)N * ldb);
        CK(hipMemcpy(a0.data(), Ah, a0.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(w0.data(), Wh, w0.size() * 2, hipMemcpyDeviceToHost));
        auto run = [&](const char* what, int keep_a, int keep_w) {
            CK(hipMemcpy(Ah, a0.data(), a0.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(Wh, w0.data(), w0.size() * 2, hipMemcpyHostToDevice));
            auto mk = [](int keep) { return keep == 0 ? 0u : (0xffffu << (10 - keep)) & 0xffffu; };
            hipLaunchKernelGGL(mask_lo_kernel, dim3(2048), dim3(256), 0, 0, Ah, (size_t)M * lda / 32, mk(keep_a));
            hipLaunchKernelGGL(mask_lo_kernel, dim3(256), dim3(256), 0, 0, Wh, (size_t)N * ldb / 32, mk(keep_w));
            CK(hipDeviceSynchronize());
            double m1 = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 57>), grid, dim3(PP_THREADS), 0, 0, gh); });
            double m2 = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh); });
            printf("%-34s lo mantissa bits A %2d W %2d: MFMA only %.3f ms   full hs->hs %.3f ms\n", what, keep_a, keep_w, m1, m2);
        };
        for (int rep = 0; rep < 2; ++rep) {
            run("as stored", 10, 10);
            run("weights only", 10, 8);
            run("weights only", 10, 6);
            run("both", 8, 8);
            run("both", 6, 6);
            run("as stored (again)", 10, 10);
        }
        return 0;
    }
    if (loop) {
        const std::string mode = argc > 4 ? argv[4] : "hs2hs";
        const double secs = argc > 5 ? atof(argv[5]) : 4.0;
        auto one = [&] {
            if (mode == "mfma") hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 57>), grid, dim3(PP_THREADS), 0, 0, gh);
            else if (mode == "pair") hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{});
            else if (mode == "regressor") hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gr);
            else if (mode == "cast") hipLaunchKernelGGL(kcast, grid, dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, gc, pc, 16.f, PairRegArgs{});      // layer 0's kernel (A = fp32 rows) at K = 1024
            else hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh);
        };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        one(); CK(hipDeviceSynchronize());
        printf("LOOP START\n"); fflush(stdout);
        CK(hipEventRecord(e0));
        int n = 0;
        float ms = 0;
        do {
            for (int i = 0; i < 50; ++i) one();
            n += 50;
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        } while (ms < secs * 1e3);
        printf("loop %s: %d launches, %.3f ms each\n", mode.c_str(), n, ms / n);
        return 0;
    }
    if (stamps) {
        // where a workgroup's time goes: entry -> prologue done -> main loop done -> epilogue done (wave 0), in shader
        // cycles and in wall time, averaged over all workgroups of a launch; plus the launch's span
        unsigned long long* st; CK(hipMalloc(&st, (size_t)grid.x * 12 * 8)); 
        auto report = [&](const char* name, auto&& launch, unsigned nblocks) {
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemset(st, 0, (size_t)grid.x * 12 * 8));
            launch(); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)nblocks * 12);
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            double cyc[3] = {0, 0, 0}, wall[3] = {0, 0, 0};
            unsigned long long t_min = ~0ull, t_max = 0; size_t n = 0;
            for (unsigned b = 0; b < nblocks; ++b) {
                const unsigned long long* p = &h[(size_t)b * 12];
                if (!p[0] || !p[6]) continue;               // tile outside the matrix
                for (int i = 0; i < 3; ++i) { cyc[i] += (double)(p[2 * (i + 1)] - p[2 * i]); wall[i] += (double)(p[2 * (i + 1) + 1] - p[2 * i + 1]); }
                t_min = std::min(t_min, p[1]); t_max = std::max(t_max, p[7]); ++n;
            }
            printf("%-18s %zu workgroups: prologue %.0f cyc (%.2f us)  main loop %.0f cyc (%.2f us)  epilogue %.0f cyc (%.2f us)  | clock %.2f GHz | launch span %.1f us\n",
                   name, n, cyc[0] / n, wall[0] / n / 100, cyc[1] / n, wall[1] / n / 100, cyc[2] / n, wall[2] / n / 100,
                   (cyc[0] + cyc[1] + cyc[2]) / (wall[0] + wall[1] + wall[2]) / 10.0, (t_max - t_min) / 100.0);
        };
        GemmHsArgs a = gh; a.stamps = st;
        report("generic hs->hs", [&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, a); }, grid.x);
        a = g; a.stamps = st;
        report("generic hs->fp32", [&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), grid, dim3(PP_THREADS), 0, 0, a); }, grid.x);
        a = gp; a.stamps = st;
        report("pair (fused A)", [&] { hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, a, ps, std::ldexp(1.f, sa), PairRegArgs{}); }, grid.x);
        report("pair, old schedule", [&] { hipLaunchKernelGGL(kpair0, grid, dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, a, ps, std::ldexp(1.f, sa), PairRegArgs{}); }, grid.x);
        a = gc; a.stamps = st;
        report("cast (A fp32)", [&] { hipLaunchKernelGGL(kcast, grid, dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, a, pc, 16.f, PairRegArgs{}); }, grid.x);
        report("cast, old schedule", [&] { hipLaunchKernelGGL(kcast0, grid, dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, a, pc, 16.f, PairRegArgs{}); }, grid.x);
        {   // fused kernel: main loop -> [both halves of the regressor stage] -> [flag wait] -> [output] per column-tile class
            GemmHsArgs af = gf; af.stamps = st;
            auto lf = [&] { hipMemsetAsync(fflags, 0, (size_t)tiles_m * 4, 0); hipLaunchKernelGGL(kfuse, grid, dim3(PP_THREADS), PR_LDS_FLOATS * 4, 0, af, ps, std::ldexp(1.f, sa), rg); };
            lf(); CK(hipDeviceSynchronize());
            CK(hipMemset(st, 0, (size_t)grid.x * 12 * 8));
            lf(); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)grid.x * 12);
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            for (int cls = 0; cls < 2; ++cls) {              // 0: column tiles 0..2 (slab writers), 1: the last one (reducer)
                double w[5] = {0, 0, 0, 0, 0}; size_t n = 0;
                for (unsigned b = 0; b < grid.x; ++b) {
                    const unsigned long long* p = &h[(size_t)b * 12];
                    const int tn = (b >> 3) % tiles_nf;
                    if (!p[0] || !p[6] || (tn == tiles_nf - 1) != (cls == 1)) continue;
                    // slots: 0 entry, 1 prologue, 2 main loop, 4 halves done, 5 wait over, 3 end
                    const int order[6] = {0, 1, 2, 4, 5, 3};
                    for (int i = 0; i < 5; ++i) w[i] += (double)(p[2 * order[i + 1] + 1] - p[2 * order[i] + 1]);
                    ++n;
                }
                printf("fused %-14s %zu workgroups: prologue %.2f us  main loop %.2f us  regressor stage %.2f us  flag wait %.2f us  output %.2f us\n",
                       cls ? "reducer tiles" : "slab tiles", n, w[0] / n / 100, w[1] / n / 100, w[2] / n / 100, w[3] / n / 100, w[4] / n / 100);
            }
        }
        a = gr; a.stamps = st;
        report("regressor N=234", [&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, a); }, gridr.x);
        printf("(ideal main loop: 64 sub-tiles x 48 MFMA x 32 cycles = 98304 cycles of the SIMD's matrix pipe)\n");
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("generic hs->hs   %.3f ms  %.0f TF fp32-equivalent (%.0f TF f16 executed)\n", ms, fl / ms / 1e9, 3 * fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 8>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   no stores     %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 64>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   epilogue arithmetic + LDS staging, no global stores %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 9>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   no stores/DMA %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 25>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   ... and no barriers %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 57>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   ... and no fragment reads (MFMA only) %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 41>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   MFMA + barriers, no reads/DMA/stores %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true, 5, 16>), grid, dim3(PP_THREADS), 0, 0, gh); });
        printf("   full kernel, no barriers (invalid results) %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), grid, dim3(PP_THREADS), 0, 0, g); });
        printf("generic hs->fp32 %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
        printf("pair (fused A)   %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        {
            GemmHsArgs gx = gp; gx.xcd_cols = 1;
            ms = time_ms([&] { hipLaunchKernelGGL(kpair, grid, dim3(PP_THREADS), lds_pair, 0, gx, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
            printf("pair, column tile per XCD  %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
            gx = gh; gx.xcd_cols = 1;
            ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gx); });
            printf("generic hs->hs, column tile per XCD  %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        }
        {
            auto k1 = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 1>;
            auto k2 = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 2>;
            auto k3 = gemm_hs_pp_pair_kernel<EPI_BIAS_RELU_AFFINE, true, false, 3>;
            CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
            CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
            CK(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
            ms = time_ms([&] { hipLaunchKernelGGL(k1, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
            printf("   pair, no L0/T requests (invalid)      %.3f ms\n", ms);
            ms = time_ms([&] { hipLaunchKernelGGL(k2, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
            printf("   pair, no conversion (invalid)         %.3f ms\n", ms);
            ms = time_ms([&] { hipLaunchKernelGGL(k3, grid, dim3(PP_THREADS), lds_pair, 0, gp, ps, std::ldexp(1.f, sa), PairRegArgs{}); });
            printf("   pair, neither (invalid)               %.3f ms\n", ms);
        }
        ms = time_ms([&] { hipLaunchKernelGGL(kcast, grid, dim3(PP_THREADS), PPP_RING_FLOATS * 4, 0, gc, pc, 16.f, PairRegArgs{}); });
        printf("cast (A fp32)    %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gr); });
        printf("regressor N=234  %.3f ms  %.0f TF  (%.0f GB/s of hs input)\n", ms, 2.0 * M * NO * K / ms / 1e9, 4.0 * M * K / ms / 1e6);
        {   // the same regressor reading its A operand in the blocked activation layout
            uint16_t* Chb; CK(hipMalloc(&Chb, (size_t)(M + 16) * 2 * N * 2 + 4096)); CK(hipMemset(Chb, 0, (size_t)(M + 16) * 2 * N * 2 + 4096));
            hipLaunchKernelGGL(f32_to_hs_kernel, dim3(2048), dim3(256), 0, 0, A, K, M, K, Chb, 2 * N, std::ldexp(1.f, sa), 1);
            GemmHsArgs gb = gr; gb.A = Chb; gb.a_blk = 1;
            ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS, false>), gridr, dim3(PP_THREADS), 0, 0, gb); });
            printf("regressor N=234, blocked A  %.3f ms  %.0f TF  (%.0f GB/s of hs input)\n", ms, 2.0 * M * NO * K / ms / 1e9, 4.0 * M * K / ms / 1e6);
            GemmHsArgs gh2 = gh; gh2.c_blk = 1; gh2.C = Chb;
            ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh2); });
            printf("generic hs->hs, blocked C   %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
            GemmHsArgs gh3 = gh2; gh3.A = Chb; gh3.a_blk = 1; gh3.C = Ch;  gh3.c_blk = 0;
            ms = time_ms([&] { hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI_BIAS_RELU_AFFINE, true>), grid, dim3(PP_THREADS), 0, 0, gh3); });
            printf("generic hs->hs, blocked A   %.3f ms  %.0f TF\n", ms, fl / ms / 1e9);
        }
        ms = time_ms(launch_fused);
        printf("pair + fused regressor %.3f ms  %.0f TF (both products)   [pair + regressor kernels above: their sum]\n", ms, (fl + 2.0 * M * NO * K) / ms / 1e9);
    }
    return 0;
}
