// csi_dnn_hs.hpp - host side of the split-f16 engine of fp32 contexts (gemm_hs.hip.h): which GEMMs
// of a packet chunk run on it and how they are launched.  The engine serves the throughput regime
// (256x256 tiles, one workgroup per CU); small calls stay on the native fp32 MFMA kernels, and since
// the layer-0 product L0 is plain fp32 either way the two engines mix freely per layer.
#pragma once
#include "csi_context.hpp"

namespace {

constexpr int HS_L0_MAX_SPLITS = 8;

// shapes the kernels can serve: every hidden width a multiple of 16 (hs groups), split weights present
bool hs_static_ok(const csi_ctx* c, const Model& m) {
    const csi_config& cf = c->cfg;
    if (c->f32_engine == 0 || cf.dtype != CSI_DTYPE_F32 || cf.nt <= 0 || (cf.len_ltf % HS_G) != 0) return false;
    if (!m.hs_repr_ok) return false;          // split copies of this model's weights are not fp32-grade (csi_load_weights): fp32 MFMA kernels
    for (int i = 0; i < cf.n_hidden; ++i)
        if (cf.hidden[i] % HS_G) return false;
    for (size_t i = 0; i < m.layers.size(); ++i)
        if (!m.layers[i].Wh || (i >= 1 && !m.layers[i].bias_hs)) return false;
    return true;
}

long hs_tiles(int M, int N) { return (long)((M + PP_BM - 1) / PP_BM) * ((N + PP_BN - 1) / PP_BN); }

// per-pair layers of a chunk of M2 rows on the split engine?  ("force_tile" = 128 / 256 keeps addressing the
// fp32 MFMA kernels of that tile height for small shapes - tests - and 128 excludes this engine)
bool hs_tail_wanted(const csi_ctx* c, int M2, int n1) {
    if (c->force_pair_tile == 128) return false;
    return c->f32_engine == 1 || hs_tiles(M2, n1) >= c->hs_min_blocks;
}

// Split-K of the layer-0 product on the split engine: the count (<= 8, >= 512 k-columns each) whose
// last round of 256 workgroups is fullest; 0 = leave layer 0 to the native kernels.
int hs_layer0_splits(const csi_ctx* c, int M1, int h1, int K, int* k_per_split) {
    if (M1 <= 8 || c->force_pair_tile == 128) return 0;
    const long tiles = (long)((M1 + PP_BM - 1) / PP_BM + 7) / 8 * 8 * ((h1 + PP_BN - 1) / PP_BN);       // as launched (pp_grid)
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= HS_L0_MAX_SPLITS; ++s) {
        if (s > 1 && K / s < 512) break;
        const long blocks = tiles * s;
        const double eff = (double)blocks / (double)((blocks + 255) / 256 * 256) - 0.01 * (s - 1);
        if (eff > best_eff) { best_eff = eff; best = s; }
    }
    const int kps = ((K + best - 1) / best + 63) / 64 * 64;
    const int real = (K + kps - 1) / kps;
    if (kps / HS_G < 3) return 0;
    if (c->f32_engine != 1 && hs_tiles(M1, h1) * real < std::max(c->hs_min_blocks, 128)) return 0;
    *k_per_split = kps;
    return real;
}

// Range guard: has any split-engine GEMM since the last check converted an operand near / beyond the
// f16 limit (*hit = the magnitude seen), or a row whose scaled magnitudes all sat in the f16 denormal
// range (*low)?  The caller has synchronised the stream.
int hs_range_check(csi_ctx* c, float* hit, bool* low) {
    *hit = 0.f;
    *low = false;
    if (!c->hs_peak || c->hs_launches == c->hs_checked) return CSI_OK;
    c->hs_checked = c->hs_launches;
    unsigned bits[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpy(bits, c->hs_peak, sizeof(bits), hipMemcpyDeviceToHost));
    if (bits[0] || bits[1] || bits[2]) {
        std::memcpy(hit, &bits[0], 4);
        *low = bits[1] != 0;
        HIP_TRY(c, hipMemset(c->hs_peak, 0, sizeof(bits)));
    }
    if (bits[2])
        return fail(c, CSI_ERR_HIP, "fused regressor: a workgroup waited 2 s for the partial sums of its row tile - results of the calls since the "
                    "last check are not valid (set hs_fuse_regressor to 0 and report this)");
    return CSI_OK;
}

template <typename Kern>
int hs_dynamic_lds(csi_ctx* c, Kern kern, size_t bytes, size_t* have) {
    if (*have < bytes) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        *have = bytes;
    }
    return CSI_OK;
}

// Layer 0 of a mid-size call on the weight-streaming kernel (l0_hs_stream.hip.h): the number of k ranges, 0 = the call is not its
// (at most 8 preambles: layer0_skinny_kernel / the one-packet path; more than 256: the 256 x 256 kernels have enough row tiles)
int l0_stream_splits(const csi_ctx* c, const Model& m, int M1, int h1, int K, int* k_per_split) {
    if (c->hs_in_shift != HS_SHIFT_AUTO) return 0;        // a fixed input scale ("hs_in_shift") addresses the kernels that take one scale per launch
    if (!c->l0_stream || M1 <= 8 || M1 > c->l0_stream_max_rows || (K % L0S_KC) != 0 || K < 2 * L0S_KC || c->force_pair_tile || !m.layers[0].Wh) return 0;
    const int groups = (h1 + L0S_COLS - 1) / L0S_COLS;
    const int blocks = (M1 + 255) / 256;                              // row blocks (gridDim.z)
    int ks = c->l0_stream_ks > 0 ? c->l0_stream_ks : std::max(1, (M1 <= 128 ? 256 : 128) / (groups * blocks));
    ks = std::max(1, std::min(ks, K / (2 * L0S_KC)));                 // at least two chunks per range
    const int kps = ((K + ks - 1) / ks + L0S_KC - 1) / L0S_KC * L0S_KC;
    *k_per_split = kps;
    return (K + kps - 1) / kps;
}

int l0_stream_launch(csi_ctx* c, Model& m, const float* x, int ldx, int M1, int h1, int K, int kps, int splits, float* slabs) {
    const Layer& l0 = m.layers[0];
    L0StreamArgs a{};
    a.x = x; a.Wh = l0.Wh; a.slabs = slabs;
    a.M = M1; a.N = h1; a.K = K; a.lda = ldx; a.ldwh = l0.ldwh; a.kps = kps; a.wshift = l0.wshift;
    const int blocks = (M1 + 255) / 256;
    const int rt = ((M1 + 31) / 32 + blocks - 1) / blocks;           // row tiles per workgroup: the row blocks are (nearly) equal
    const int rt_inst = rt <= 4 ? rt : (rt <= 6 ? 6 : 8);
    const dim3 grid((unsigned)((h1 + L0S_COLS - 1) / L0S_COLS), (unsigned)splits, (unsigned)((M1 + 32 * rt_inst - 1) / (32 * rt_inst)));
    ++c->l0_stream_launches;
    if (M1 > c->l0_stream_prepass_rows) {
        // row maxima over the whole K by their own small kernel (beyond 64 preambles cheaper than the first pass of every workgroup)
        if (!m.l0_rowmax && hipMalloc((void**)&m.l0_rowmax, 4096 * sizeof(float)) != hipSuccess)
            return fail(c, CSI_ERR_NOMEM, "device allocation of the row maxima failed");
        hipLaunchKernelGGL(l0_row_max_kernel, dim3((unsigned)M1), dim3(256), 0, c->stream, x, ldx, K, m.l0_rowmax);
        HIP_TRY(c, hipGetLastError());
        a.row_max = m.l0_rowmax;
    }
    ProfScope ps(c, K_LAYER0_LTF, 2.0 * (double)M1 * h1 * K, 4.0 * ((double)M1 * K + (double)h1 * K + (double)M1 * h1 * splits));
    auto go = [&](auto kern, int rtt, size_t* attr) {
        const size_t lds = l0s_lds_bytes(rtt);
        int rc = hs_dynamic_lds(c, kern, lds, attr);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, c->stream, a);
        HIP_TRY(c, hipGetLastError());
        return (int)CSI_OK;
    };
    if (rt <= 1) return go(l0_hs_stream_kernel<1>, 1, &c->hs_lds_attr[14]);
    if (rt <= 2) return go(l0_hs_stream_kernel<2>, 2, &c->hs_lds_attr[15]);
    if (rt <= 3) return go(l0_hs_stream_kernel<3>, 3, &c->hs_lds_attr[16]);
    if (rt <= 4) return go(l0_hs_stream_kernel<4>, 4, &c->hs_lds_attr[17]);
    if (rt <= 6) return go(l0_hs_stream_kernel<6>, 6, &c->hs_lds_attr[18]);
    return go(l0_hs_stream_kernel<8>, 8, &c->hs_lds_attr[19]);
}

// layer 0: slabs[z][M1][h1] = (X[M1][K] * W0[0:K, :]) over k range z, X converted inside the kernel
int hs_launch_layer0(csi_ctx* c, const Model& m, const float* x, int ldx, int M1, int h1, int K, int kps, int splits, float* slabs) {
    const Layer& l0 = m.layers[0];
    GemmHsArgs g{};
    g.Bt = l0.Wh; g.ldb = l0.ldwh;
    g.C = slabs; g.ldc = h1;
    g.M = M1; g.N = h1; g.K = K;
    g.k_per_split = kps;
    g.tiles_n = (h1 + PP_BN - 1) / PP_BN;
    const bool auto_scale = c->hs_in_shift == HS_SHIFT_AUTO;
    const int in_shift = auto_scale ? 0 : c->hs_in_shift;
    g.acc_scale = std::ldexp(1.f, -(in_shift + l0.wshift));
    g.peak = c->hs_peak;
    g.wshift = l0.wshift;
    ++c->hs_launches;
    if (auto_scale) {
        // magnitude estimate of these rows (1-KiB blocks, at most ~4 MiB of them), stream-ordered in front of the GEMM
        unsigned* dyn = c->hs_peak + 8 + (&m == &c->model[1] ? 1 : 0);      // one word per component model (small calls run them on two streams)
        HIP_TRY(c, hipMemsetAsync(dyn, 0, sizeof(unsigned), c->stream));
        const size_t n4 = (size_t)M1 * ldx / 4;
        const size_t nblk = (n4 + 63) / 64;                                   // 1-KiB blocks
        const size_t step = std::max<size_t>(1, nblk / 4096);                 // at most ~4 MiB (a million samples) are read
        const unsigned blocks = (unsigned)std::min<size_t>((nblk / step + 3) / 4, 1024);
        hipLaunchKernelGGL(hs_absmax_sample_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, c->stream, x, n4, step, dyn);
        HIP_TRY(c, hipGetLastError());
        g.dyn_max = dyn;
    }
    const double flops = 2.0 * (double)M1 * h1 * K;
    const double bytes = 4.0 * ((double)M1 * K + (double)h1 * K + (double)M1 * h1 * splits);
    ProfScope ps(c, K_LAYER0_LTF, flops, bytes);
    const size_t lds = (size_t)PPP_RING_FLOATS * sizeof(float);
    PairSrc src{x, nullptr, ldx, 1};
    const int tiles_m = (M1 + PP_BM - 1) / PP_BM;
    auto go = [&](auto kern, size_t* attr) {
        int rc = hs_dynamic_lds(c, kern, lds, attr);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n), 1, (unsigned)splits), dim3(PP_THREADS), lds, c->stream, g, src,
                           std::ldexp(1.f, in_shift), PairRegArgs{});
        HIP_TRY(c, hipGetLastError());
        return (int)CSI_OK;
    };
    // "hs_vm": hand-counted vector-memory operations (gemm_hs.hip.h) - 2 = with the extra sub-tile of look-ahead for the fp32 rows
    if (c->hs_vm_cast == 3) {
        const size_t lds5 = (size_t)5 * PP_SUBF * sizeof(float);
        auto kern = gemm_hs_pp_pair_kernel<EPI_RAW, false, true, 0, false, 3>;
        int rc = hs_dynamic_lds(c, kern, lds5, &c->hs_lds_attr[11]);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n), 1, (unsigned)splits), dim3(PP_THREADS), lds5, c->stream, g, src,
                           std::ldexp(1.f, in_shift), PairRegArgs{});
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    if (c->hs_vm_cast == 2) return go(gemm_hs_pp_pair_kernel<EPI_RAW, false, true, 0, false, 2>, &c->hs_lds_attr[5]);
    if (c->hs_vm_cast == 1) return go(gemm_hs_pp_pair_kernel<EPI_RAW, false, true, 0, false, 1>, &c->hs_lds_attr[6]);
    return go(gemm_hs_pp_pair_kernel<EPI_RAW, false, true>, &c->hs_lds_attr[0]);
}

// first per-pair layer: A generated from (L0, T, bn0); hs output (a hidden layer follows) or fp32
// output with bias only (the regressor follows layer 0 directly)
// shift of the OUTPUT activations of hidden layer li
// (pre: the consumer reads this layer's bare relu output - its BatchNormalization scale sits in the consumer's weights)
int hs_act_shift_of(const csi_ctx* c, const Model& m, int li, bool pre = false) {
    return c->hs_act_shift == HS_SHIFT_AUTO ? (pre ? m.layers[li].ashift_pre : m.layers[li].ashift) : c->hs_act_shift;
}

template <int EPI, bool OUT_HS>
int hs_launch_pair(csi_ctx* c, int kid, GemmHsArgs g, const PairSrc& src, int in_shift) {
    g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
    g.peak = c->hs_peak;
    g.xcd_cols = 1;            // a column tile per XCD: its 1 MiB of weights stays in that L2 (the A side is 32 KB per row tile); -2.5 %
    ++c->hs_launches;
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 4.0 * ((double)g.M / src.nt * g.K + (double)src.nt * g.K + (double)g.N * g.K + (double)g.M * g.N);
    ProfScope ps(c, kid, flops, bytes);
    const size_t lds = (size_t)PPP_RING_FLOATS * sizeof(float);
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    auto go = [&](auto kern, size_t* attr) {
        int rc = hs_dynamic_lds(c, kern, lds, attr);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n)), dim3(PP_THREADS), lds, c->stream, g, src, std::ldexp(1.f, in_shift), PairRegArgs{});
        HIP_TRY(c, hipGetLastError());
        return (int)CSI_OK;
    };
    if (c->hs_vm_pair == 3) {
        const size_t lds5 = (size_t)5 * PP_SUBF * sizeof(float);
        auto kern = gemm_hs_pp_pair_kernel<EPI, OUT_HS, false, 0, false, 3>;
        int rc = hs_dynamic_lds(c, kern, lds5, &c->hs_lds_attr[OUT_HS ? 12 : 13]);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n)), dim3(PP_THREADS), lds5, c->stream, g, src, std::ldexp(1.f, in_shift), PairRegArgs{});
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    if (c->hs_vm_pair == 2) return go(gemm_hs_pp_pair_kernel<EPI, OUT_HS, false, 0, false, 2>, &c->hs_lds_attr[OUT_HS ? 7 : 8]);
    if (c->hs_vm_pair == 1) return go(gemm_hs_pp_pair_kernel<EPI, OUT_HS, false, 0, false, 1>, &c->hs_lds_attr[OUT_HS ? 9 : 10]);
    return go(gemm_hs_pp_pair_kernel<EPI, OUT_HS, false>, &c->hs_lds_attr[OUT_HS ? 1 : 2]);
}

template <int EPI, bool OUT_HS>
int hs_launch_gemm(csi_ctx* c, int kid, GemmHsArgs g) {
    g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
    g.peak = c->hs_peak;
    ++c->hs_launches;
    g.k_per_split = (g.K + HS_G - 1) / HS_G * HS_G;
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
    ProfScope ps(c, kid, flops, bytes);
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    hipLaunchKernelGGL((gemm_hs_pp_kernel<EPI, OUT_HS>), dim3(pp_grid(tiles_m, g.tiles_n)), dim3(PP_THREADS), 0, c->stream, g);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// ---- the fused band kernel (gemm_hs_band.hip.h, band_kernel_gen.py): code object embedded at build time
#if __has_include("band8_hsaco.inc")
#include "band8_hsaco.inc"
#define CSI_HAVE_BAND8 1
#else
#warning "band8_hsaco.inc not found: this build has NO fused band kernel (the separate pair + regressor kernels serve every call; option band_available reads 0).  _lib.build_library() generates it: band_kernel_gen.py -> clang -x assembler -mcpu=gfx950 -> ld.lld"
#endif

int band8_function(csi_ctx* c, hipFunction_t* fn, bool bf16 = false, bool staged = false) {
    *fn = nullptr;
#ifdef CSI_HAVE_BAND8
    if (c->band_failed) return CSI_OK;
    if (!c->band_mod) {
        // timing experiments (tools/): CSI_BAND8_HSACO = a code object built by tools/build_band8.sh (every ablation variant of
        // band_kernel_gen.py), CSI_BAND8_NAME / CSI_BAND8_BF16_NAME = the variants to run in place of the two product kernels
        // Honoured only together with CSI_DEBUG_HOOKS=1: a production process never loads a code object named by its environment.
        const char* hooks = std::getenv("CSI_DEBUG_HOOKS");
        const char* ext = hooks && hooks[0] == '1' ? std::getenv("CSI_BAND8_HSACO") : nullptr;
        const char* n_hs = std::getenv("CSI_BAND8_NAME");
        const char* n_bf = std::getenv("CSI_BAND8_BF16_NAME");
        const hipError_t le = ext && *ext ? hipModuleLoad(&c->band_mod, ext) : hipModuleLoadData(&c->band_mod, band8_hsaco);
        if (le != hipSuccess || hipModuleGetFunction(&c->band_fn, c->band_mod, ext && n_hs ? n_hs : "csi_band8") != hipSuccess ||
            hipModuleGetFunction(&c->band_fn_bf16, c->band_mod, ext && n_bf ? n_bf : "csi_band8_bf16") != hipSuccess ||
            hipModuleGetFunction(&c->band_fn_bf16_ns, c->band_mod, "csi_band8_bf16_nostage") != hipSuccess ||
            hipModuleGetFunction(&c->band_fn_ns, c->band_mod, "csi_band8_nostage") != hipSuccess) {
            (void)hipGetLastError();
            c->band_failed = true;       // not fatal: the separate kernels serve the call
            c->band_fn = c->band_fn_bf16 = c->band_fn_bf16_ns = c->band_fn_ns = nullptr;
            return CSI_OK;
        }
        if (hipModuleGetFunction(&c->band_fn_cs, c->band_mod, "csi_band8_cs") != hipSuccess) {      // (an external code object may lack it)
            (void)hipGetLastError();
            c->band_fn_cs = nullptr;
        }
        if (hipModuleGetFunction(&c->band_fn_bf16_cs, c->band_mod, "csi_band8_bf16_cs") != hipSuccess) {
            (void)hipGetLastError();
            c->band_fn_bf16_cs = nullptr;
        }
        if (hipModuleGetFunction(&c->band_fn4, c->band_mod, "csi_band4") != hipSuccess) {
            (void)hipGetLastError();
            c->band_fn4 = nullptr;
        }
        c->band_hs_threads = (ext && n_hs && std::strncmp(n_hs, "csi_band4", 9) == 0) ? 256 : BAND8_THREADS;
        const auto ends_p = [](const char* n) { const size_t l = n ? std::strlen(n) : 0; return l > 2 && n[l - 2] == '_' && n[l - 1] == 'p'; };
        c->band_hs_persist = ext && ends_p(n_hs);          // hooked A/B runs of the persistent forms (csi_band4_p / csi_band4_bf16_p)
        c->band_bf16_persist = ext && ends_p(n_bf);
        if (hipModuleGetFunction(&c->band_fn4_p, c->band_mod, "csi_band4_p") != hipSuccess) { (void)hipGetLastError(); c->band_fn4_p = nullptr; }
        if (hipModuleGetFunction(&c->band_fn4_bf16_p, c->band_mod, "csi_band4_bf16_p") != hipSuccess) { (void)hipGetLastError(); c->band_fn4_bf16_p = nullptr; }
        if (hipModuleGetFunction(&c->band_fn4_cs, c->band_mod, "csi_band4_cs") != hipSuccess) { (void)hipGetLastError(); c->band_fn4_cs = nullptr; }
        if (hipModuleGetFunction(&c->band_fn4_bf16_cs, c->band_mod, "csi_band4_bf16_cs") != hipSuccess) { (void)hipGetLastError(); c->band_fn4_bf16_cs = nullptr; }
        if (hipModuleGetFunction(&c->band_fn4_bf16, c->band_mod, "csi_band4_bf16") != hipSuccess) {
            (void)hipGetLastError();
            c->band_fn4_bf16 = nullptr;
        }
        c->band_bf16_threads = (ext && n_bf && std::strncmp(n_bf, "csi_band4", 9) == 0) ? 256 : BAND8_THREADS;
    }
    *fn = bf16 ? (staged ? c->band_fn_bf16 : c->band_fn_bf16_ns) : (staged ? c->band_fn : c->band_fn_ns);
#endif
    return CSI_OK;
}

// Static part of "the column-split band kernel serves this model's per-pair layers" (band8_serves + band8_staged + band8_splits on
// the shapes alone): with it a call of a few hundred pair rows is faster on the split engine than on the fp32 MFMA kernels -
// 5 ... 20 packets of the shipped shape 136-142 us against 155-260 (profiles/r05_band_split_probe.txt)
bool band_split_static_ok(csi_ctx* c, const Model& m) {
    const csi_config& cf = c->cfg;
    if (cf.n_hidden != 2 || !c->hs_band || c->hs_band == 3 || c->hs_fuse_regressor || c->band_split == 0 || c->band_split == 1 ||
        c->force_pair_tile == 128)
        return false;
    if (m.layers.size() < 3 || !m.layers[2].Wh_p) return false;
    const int h1 = cf.hidden[0], n1 = cf.hidden[1];
    if (cf.nt < 16 || cf.nt > 128 || h1 < 128 || (h1 % 64) != 0 || (n1 % 512) != 0 || n1 > BAND8_MAX_N1 || cf.n_out < 1 || cf.n_out > 256) return false;
    hipFunction_t fn = nullptr;
    if (band8_function(c, &fn, false, true) != CSI_OK || !fn) return false;
    return c->band_fn_cs != nullptr;
}

// The register-blocked band kernels (band4_kernel_gen.py) stream PRE-TILED weights: built once per model at first use (band4_tile_kernel), then the
// argument record's W1 / W2p point at the tiled copies.  bf16: sub-tiles of 32 k, 8 regressor fragments per column step; split-f16: 16 k, 16 fragments.
int band4_prepare(csi_ctx* c, Model& m, BandArgs& ba, bool bf16) {
    if (!m.tiled_ok) {
        const int ncol = ba.N1 / 256, nsub = ba.K1 / (bf16 ? 32 : 16), nq = bf16 ? 8 : 16;
        const size_t b1 = (size_t)(ncol * nsub + 4) * BAND_SLOT_BYTES, b2 = (size_t)ncol * nq * BAND_SLOT_BYTES;
        if (b1 >= 0x7fffffffull) return fail(c, CSI_ERR_INVALID_ARG, "band weights of %zu bytes: beyond what the tiled copy addresses", b1);
        if ((!m.Wt1 && hipMalloc((void**)&m.Wt1, b1) != hipSuccess) || (!m.Wt2 && hipMalloc((void**)&m.Wt2, b2) != hipSuccess))
            return fail(c, CSI_ERR_NOMEM, "device allocation of the tiled band weights failed");
        hipLaunchKernelGGL(band4_tile_kernel, dim3(512), dim3(256), 0, c->stream, ba.W1, ba.ldb1, ncol, nsub, 0, 4, m.Wt1);
        hipLaunchKernelGGL(band4_tile_kernel, dim3(512), dim3(256), 0, c->stream, ba.W2p, ba.ldb2, ncol, nq, bf16 ? 1 : 2, 0, m.Wt2);
        HIP_TRY(c, hipGetLastError());
        m.tiled_ok = true;
    }
    ba.W1 = m.Wt1;
    ba.W2p = m.Wt2;
    return CSI_OK;
}

int band8_launch(csi_ctx* c, hipFunction_t fn, const BandArgs& ba, double flops, double bytes) {
    ++c->band_launches;
    ProfScope ps(c, K_PAIR_DENSE, flops, bytes);
    Band8ArgsCs a8{band8_args(ba), nullptr, 0};
    const unsigned bands = (unsigned)((ba.M + BAND_ROWS - 1) / BAND_ROWS);
    // the persistent forms (round 6: csi_band4_p / csi_band4_bf16_p): one workgroup per CU walks bands x, x + P, ...; P travels in the record's last quadword
    const bool persist = fn == c->band_fn4_p || fn == c->band_fn4_bf16_p || (fn == c->band_fn && c->band_hs_persist) || (fn == c->band_fn_bf16 && c->band_bf16_persist);
    const unsigned grid = persist ? std::min(bands, (unsigned)std::max(c->n_cu, 1)) : bands;
    a8.pad = grid;
    size_t sz = persist ? sizeof(a8) : sizeof(a8.a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a8, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    // the register-blocked form is a workgroup of 4 waves (band4_kernel_gen.py), the others of 8
    const unsigned threads = (fn == c->band_fn4_bf16 || fn == c->band_fn4 || fn == c->band_fn4_p || fn == c->band_fn4_bf16_p) ? 256u :
                             (fn == c->band_fn_bf16 ? (unsigned)c->band_bf16_threads : (fn == c->band_fn ? (unsigned)c->band_hs_threads : (unsigned)BAND8_THREADS));
    HIP_TRY(c, hipModuleLaunchKernel(fn, grid, 1, 1, threads, 1, 1, 0, c->stream, nullptr, extra));
    return CSI_OK;
}

// Column splits of a call of `bands` bands: a band is one workgroup's work for ~200 us, so a call with fewer bands than CUs leaves
// CUs idle for that long - 2 or 4 workgroups per band, each over N1 / splits hidden features, fill them ("band_split";
// profiles/r05_band_split_probe.txt: 24 packets 248 -> 149 us, 64 packets 299 -> 251 us)
int band8_splits(const csi_ctx* c, const BandArgs& ba, size_t part_capacity_floats, bool bf16 = false) {
    if (!(bf16 ? c->band_fn_bf16_cs : c->band_fn_cs) || c->band_split == 0 || c->band_split == 1 || ba.stamps) return 1;
    const long bands = (ba.M + BAND_ROWS - 1) / BAND_ROWS;
    int S = 1;
    if (c->band_split > 1) {
        S = c->band_split;
    } else {
        const long in_flight = bands * std::max(c->models_in_flight, 1);
        while (S < 4 && in_flight * (2 * S) <= 256) S *= 2;
        // a second round of workgroups that is at most a quarter full (129 ... 160 packets of the shipped shape): half-size workgroups
        // fill it better - 144 packets 502 -> 460 us, 160: 511 -> 476; from 192 packets on the split only costs (profiles/r05_band_split_probe.txt)
        if (S == 1 && in_flight > 256 && in_flight <= 320) S = 2;
    }
    while (S > 1 && (ba.N1 % (256 * S)) != 0) S >>= 1;
    const unsigned long long part = (unsigned long long)(S - 1) * (unsigned long long)ba.M * (unsigned long long)ba.ldo;
    if (S > 1 && (part > part_capacity_floats || part * 4ull >= 0xffffffffull || (unsigned long long)ba.N1 * ba.ldb1 * 2ull >= 0x7fffffffull)) return 1;
    return S;
}

// the column-split launch: partial outputs of splits 1 .. in `part`, added to split 0's output in split order
int band8_launch_split(csi_ctx* c, const BandArgs& ba, int S, float* part, double flops, double bytes, bool bf16 = false, bool blocked = false,
                       int kid = K_PAIR_DENSE) {
    ++c->band_launches;
    ++c->band_split_launches;
    BandArgs one = ba;
    one.N1 = ba.N1 / S;
    Band8ArgsCs a{};
    a.a = band8_args(one);
    a.part = part;
    size_t sz = sizeof(a);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    {
        ProfScope ps(c, kid, flops, bytes);
        // blocked: the register-blocked form (4 waves; ba.W1 / W2p are the tiled copies of the WHOLE layer: split y starts at its own column steps)
        const hipFunction_t fn = blocked ? (bf16 ? c->band_fn4_bf16_cs : c->band_fn4_cs) : (bf16 ? c->band_fn_bf16_cs : c->band_fn_cs);
        HIP_TRY(c, hipModuleLaunchKernel(fn, (unsigned)((ba.M + BAND_ROWS - 1) / BAND_ROWS), (unsigned)S, 1, blocked ? 256u : (unsigned)BAND8_THREADS, 1, 1, 0, c->stream, nullptr, extra));
    }
    const size_t n = (size_t)ba.M * ba.ldo;
    ProfScope ps(c, K_SPLITK_REDUCE, (double)(S - 1) * n, 4.0 * (S + 1) * (double)n);
    hipLaunchKernelGGL(band_split_sum_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, ba.out, part, n, n, S - 1);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// "band_tail_split" (round 6): one workgroup per CU computes a band for 75-190 us, so a call of `bands` bands runs in ceil(bands / CUs) rounds and a last
// round of a few bands costs a whole one (configs[2]: 10 000 bands = 39 rounds + 16 bands = 2.5 % of the kernel's time for 0.16 % of the work).  When that
// round holds at most half (a quarter) of the CUs' worth of bands it is launched separately in 2 (4) column splits - the rows of the full rounds through
// the unsplit kernel, the rest through the column-split one on shifted operand pointers.  Returns the split count of the tail (0: one launch) and its first row.
inline int band_tail_plan(const csi_ctx* c, const BandArgs& ba, size_t part_capacity_floats, bool bf16, long* row0) {
    // measured (tools/band_tail_ab.py, profiles/r06_band_probe.txt (G)): fp32 contexts -2.8 ... -3.2 % per call where it applies (2100 / 2600 / 3100 / 4150 packets of the
    // shipped shape); bf16 contexts 0 ... +1 % (bands of 75 us backfill the last round well enough; the 8-wave split kernels and the extra sum eat the rest): fp32 only
    if (bf16) return 0;
    if (!c->band_tail_split || c->models_in_flight > 1 || ba.stamps || c->band_split == 0 || c->band_split == 1) return 0;
    if (!(bf16 ? c->band_fn_bf16_cs : c->band_fn_cs) || ba.ldo != ba.n2) return 0;
    const long ncu = std::max(c->n_cu, 1), bands = (ba.M + BAND_ROWS - 1) / BAND_ROWS;
    const long full = bands / ncu * ncu, tail = bands - full;
    if (full < ncu || tail == 0) return 0;
    int S = 0;
    for (int s = 4; s >= 2; s >>= 1)
        if (tail * s <= ncu && (ba.N1 % (256 * s)) == 0) { S = s; break; }
    const long r0 = full * BAND_ROWS;
    if (!S || ba.nt < 1 || (r0 % ba.nt) != 0) return 0;
    const unsigned long long part = (unsigned long long)(S - 1) * (unsigned long long)(ba.M - r0) * (unsigned long long)ba.ldo;
    if (part > part_capacity_floats || (unsigned long long)ba.N1 * ba.ldb1 * 2ull >= 0x7fffffffull) return 0;
    *row0 = r0;
    return S;
}
// the two launches of a tail-split call: `full` = the arguments the unsplit kernel `fn` takes (tiled weights if it is a register-blocked form), `cs` = the
// arguments of the column-split launch (tiled weights iff cs_blocked)
inline int band_tail_launch(csi_ctx* c, hipFunction_t fn, const BandArgs& full, const BandArgs& cs, bool cs_blocked, int S, long r0, float* part, double flops,
                            double bytes, bool bf16) {
    const double f0 = (double)r0 / (double)full.M;
    BandArgs a0 = full;
    a0.M = (int)r0;
    int rc = band8_launch(c, fn, a0, flops * f0, bytes * f0);
    if (rc) return rc;
    BandArgs a1 = cs;
    a1.M = cs.M - (int)r0;
    a1.L0 = cs.L0 + (size_t)(r0 / cs.nt) * cs.ldl;
    a1.out = cs.out + (size_t)r0 * cs.ldo;
    ++c->band_tail_launches;
    return band8_launch_split(c, a1, S, part, flops * (1.0 - f0), bytes * (1.0 - f0), bf16, cs_blocked, K_PAIR_DENSE_TAIL);
}

// csi_profile_band_skeleton: the band kernel's MFMA + barrier skeleton (band_kernel_gen.py 'skeleton_rnd': no operand conversion, no
// L0 / T streams, no LDS-DMA, no fragment reads; its 48 operand registers filled once from the model's own split weights, relu-like
// zeros in the activation fragments) on `rows` pair rows of the loaded real model.  What it measures is the rate the matrix pipe of
// THIS part sustains on THIS data inside the power budget with everything else of the kernel removed - the practical ceiling the
// bench divides by next to the table peak (its output is garbage and goes to the workspace).
int band_skeleton_time(csi_ctx* c, int64_t rows, int iters, double* ms_per_launch, double* executed_flops) {
#ifdef CSI_HAVE_BAND8
    const csi_config& cf = c->cfg;
    Model& m = c->model[0];
    if (cf.dtype != CSI_DTYPE_F32 || cf.n_hidden != 2 || !m.loaded || !m.layers[2].Wh_p || rows <= 0 || rows > (1 << 30) || iters < 1)
        return fail(c, CSI_ERR_INVALID_ARG, "csi_profile_band_skeleton: needs an fp32 context with the two-hidden-layer model loaded, rows > 0");
    hipFunction_t fn = nullptr;
    int rc = band8_function(c, &fn, false, true);
    if (rc) return rc;
    hipFunction_t sk = nullptr;
    if (!fn || hipModuleGetFunction(&sk, c->band_mod, "csi_band8_skeleton_rnd") != hipSuccess || !sk) {
        (void)hipGetLastError();
        return fail(c, CSI_ERR_NOT_READY, "csi_profile_band_skeleton: the band kernel's code object holds no skeleton variant");
    }
    const Layer &l1 = m.layers[1], &lr = m.layers[2];
    const int h1 = cf.hidden[0], M2 = (int)rows;
    rc = ensure_bytes(c, &c->ws, &c->ws_bytes, (size_t)M2 * cf.n_out * sizeof(float) + ((size_t)M2 / std::max(cf.nt, 1) + 256) * h1 * sizeof(float));
    if (rc) return rc;
    BandArgs ba{};
    ba.out = reinterpret_cast<float*>(c->ws); ba.ldo = cf.n_out;
    ba.L0 = ba.out + (size_t)M2 * cf.n_out; ba.Ts = m.T_hs ? m.T_hs : m.T; ba.ldl = h1; ba.nt = cf.nt; ba.in_scale = 1.f;
    ba.W1 = l1.Wh; ba.ldb1 = l1.ldwh; ba.bias1 = l1.bias_hs; ba.M = M2; ba.K1 = h1; ba.N1 = l1.out;
    ba.acc_scale1 = 1.f; ba.out_scale = 1.f;
    ba.W2p = lr.Wh_p; ba.ldb2 = lr.ldwh; ba.bias2 = lr.bias_hs; ba.n2 = cf.n_out; ba.acc_scale2 = 1.f;
    ba.peak = nullptr;
    if (!band8_serves(ba)) return fail(c, CSI_ERR_INVALID_ARG, "csi_profile_band_skeleton: the band kernel does not serve this shape");
    Band8Args a8 = band8_args(ba);
    size_t sz = sizeof(a8);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a8, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    const unsigned grid = (unsigned)((M2 + BAND_ROWS - 1) / BAND_ROWS);
    struct Events {                                    // destroyed on every return path (an early HIP_TRY return leaked them)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
    } ev;
    HIP_TRY(c, hipEventCreate(&ev.a));
    HIP_TRY(c, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    for (int i = 0; i < 2; ++i) HIP_TRY(c, hipModuleLaunchKernel(sk, grid, 1, 1, BAND8_THREADS, 1, 1, 0, c->stream, nullptr, extra));
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) HIP_TRY(c, hipModuleLaunchKernel(sk, grid, 1, 1, BAND8_THREADS, 1, 1, 0, c->stream, nullptr, extra));
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_launch) *ms_per_launch = (double)ms / iters;
    // executed f16 flop: 3 products, the padded 256-column regressor tile included (what the pipe really does)
    if (executed_flops) *executed_flops = 3.0 * (2.0 * (double)grid * BAND_ROWS * l1.out * h1 + 2.0 * (double)grid * BAND_ROWS * 256.0 * l1.out);
    return CSI_OK;
#else
    return fail(c, CSI_ERR_NOT_READY, "csi_profile_band_skeleton: the library was built without the band kernel");
#endif
}

// the per-pair layers of one chunk: l0sum [M1][h1] fp32 -> out [M2][n_out] fp32; hbuf0 / hbuf1 are the
// ping-pong activation buffers of the fp32 path re-used as hs matrices (same 4 bytes per element)
int hs_tail(csi_ctx* c, Model& m, const float* l0sum, int M2, float* hbuf0, float* hbuf1, float* out) {
    const csi_config& cf = c->cfg;
    if (int rc_ls = ls_deferred_fire(c)) return rc_ls;       // csi_estimate_device with "ls_overlap_cus": the LS kernel starts here, beside the per-pair kernels
    const int nh = cf.n_hidden, h1 = cf.hidden[0];
    const Layer& l1 = m.layers[1];
    const int s0 = hs_act_shift_of(c, m, 0, true);
    // bn0 is not applied by the pair kernel: its scale multiplies the rows of layer 1's split weights, its shift
    // (like every later BatchNormalization shift) lives in layer 1's bias (Layer::bias_hs) - A = relu(2^s0 L0 + Ts)
    // with Ts = 2^s0 T, a copy of the pilot table rebuilt when the table or the shift changes
    if (m.T_hs_shift != s0) {
        const size_t bytes = ((size_t)cf.nt * h1 + G_SLACK_FLOATS) * sizeof(float);
        if (!m.T_hs) {
            if (hipMalloc((void**)&m.T_hs, bytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "pilot table allocation failed");
            HIP_TRY(c, hipMemsetAsync(m.T_hs, 0, bytes, c->stream));
        }
        ProfScope ps(c, K_PILOT_TABLE, 2.0 * cf.nt * cf.nt * h1, 4.0 * (cf.nt * cf.nt + 2.0 * cf.nt * h1));
        hipLaunchKernelGGL(pilot_table_kernel, dim3((h1 + 255) / 256, cf.nt), dim3(256), 0, c->stream, c->P, m.W0p, m.layers[0].bias, m.T_hs,
                           cf.nt, h1, std::ldexp(1.f, s0));
        HIP_TRY(c, hipGetLastError());
        m.T_hs_shift = s0;
        m.T_sw_ok = false;
    }
    PairSrc src{l0sum, m.T_hs, h1, cf.nt};
    GemmHsArgs p{};
    p.Bt = l1.Wh; p.ldb = l1.ldwh;
    p.M = M2; p.N = l1.out; p.K = h1;
    p.bias = l1.bias_hs; p.scale = l1.scale; p.shift = c->hs_zero;
    p.acc_scale = std::ldexp(1.f, -(s0 + l1.wshift));
    if (nh == 1) {
        p.C = out; p.ldc = cf.n_out;
        return hs_launch_pair<EPI_BIAS, false>(c, K_REGRESSOR, p, src, s0);
    }
    const Layer& lr = m.layers[nh];
    if (nh == 2 && c->hs_band && !c->hs_fuse_regressor && lr.Wh_p) {
        // first per-pair layer + regressor in one kernel, h2 in registers: the assembly band kernel (8 waves per 128-row band)
        const int s1 = hs_act_shift_of(c, m, 1, true);
        BandArgs ba{};
        ba.L0 = l0sum; ba.Ts = m.T_hs; ba.ldl = h1; ba.nt = cf.nt; ba.in_scale = std::ldexp(1.f, s0);
        ba.W1 = l1.Wh; ba.ldb1 = l1.ldwh; ba.bias1 = l1.bias_hs; ba.M = M2; ba.K1 = h1; ba.N1 = l1.out;
        ba.acc_scale1 = std::ldexp(1.f, -(s0 + l1.wshift)); ba.out_scale = std::ldexp(1.f, s1);
        ba.W2p = lr.Wh_p; ba.ldb2 = lr.ldwh; ba.bias2 = lr.bias_hs; ba.n2 = cf.n_out; ba.acc_scale2 = std::ldexp(1.f, -(s1 + lr.wshift_f));
        ba.out = out; ba.ldo = cf.n_out; ba.peak = c->hs_peak;
        hipFunction_t fn = nullptr;
        const bool staged = band8_staged(ba, false) && c->hs_band != 3;       // "hs_band" = 3: the form with per-lane global loads of L0 / T (A/B runs)
        if (band8_serves(ba) && h1 == l1.in && lr.in == l1.out) {
            int rc = band8_function(c, &fn, false, staged);
            if (rc) return rc;
        }
        if (fn && staged) {     // the kernel streams the (pre-scaled) pilot table slab by slab through LDS: its slab-ordered copy
            if (!m.T_sw_ok) {
                // (+ 2 KiB: the last 1-KiB DMA chunk of a slab may reach past it, and the kernel requests one slab past the column step)
                const size_t floats = (size_t)(h1 / 16 + 1) * cf.nt * 16;
                if (!m.T_sw && hipMalloc((void**)&m.T_sw, (floats + 512) * sizeof(float)) != hipSuccess)
                    return fail(c, CSI_ERR_NOMEM, "device allocation of the slab-ordered pilot table failed");
                hipLaunchKernelGGL(band_tsw_kernel<16>, dim3(256), dim3(256), 0, c->stream, m.T_hs, h1, cf.nt, h1, m.T_sw);
                HIP_TRY(c, hipGetLastError());
                m.T_sw_ok = true;
            }
            ba.Ts = m.T_sw;
        }
        if (fn) {
            ++c->hs_launches;
            const double flops = 2.0 * (double)M2 * l1.out * h1 + 2.0 * (double)M2 * cf.n_out * l1.out;
            const double bytes = 4.0 * ((double)M2 / cf.nt * h1 + (double)cf.nt * h1 + (double)l1.out * h1 + (double)cf.n_out * l1.out + (double)M2 * cf.n_out);
            // (the band path leaves the activation buffers unused: hbuf0 - M2 x 1024 floats here - holds the partial outputs)
            const int S = (fn == c->band_fn && staged && ba.ldo == ba.n2) ? band8_splits(c, ba, (size_t)M2 * l1.out) : 1;
            // round 6: the register-blocked forms (band4_kernel_gen.py "csi_band4" / "csi_band4_cs": 4 waves x 512 registers, every weight fragment against two
            // row groups) on the same operands, their weight streams pre-tiled once per model
            const bool hooked4 = c->band_hs_threads == 256;
            const bool blocked = fn == c->band_fn && staged && c->band4 && c->band_fn4 && !ba.stamps;
            if (S > 1) {
                // (measured, tools/regime_probe.py band4=1 / 0 alternating: 2 splits - 64 packets - 192 against 195 us, 128 packets unsplit 304 against 315; 4 splits -
                // 24 packets - 120 against 117 us: a quarter band on four waves has less to hide its waits behind)
                const bool b4 = blocked && c->band_fn4_cs && S == 2;
                if (b4) { const int rc = band4_prepare(c, m, ba, false); if (rc) return rc; }
                return band8_launch_split(c, ba, S, hbuf0, flops, bytes, false, b4);
            }
            const BandArgs ba_plain = ba;          // (band4_prepare puts the tiled weight copies into ba)
            if (blocked || (fn == c->band_fn && staged && hooked4 && !ba.stamps)) {
                const int rc = band4_prepare(c, m, ba, false);
                if (rc) return rc;
                if (blocked) fn = c->band_fn4;
            }
            long r0 = 0;
            const int St = (fn == c->band_fn4 || (fn == c->band_fn && staged)) ? band_tail_plan(c, ba, (size_t)M2 * l1.out, false, &r0) : 0;
            if (St) {
                const bool b4 = blocked && c->band_fn4_cs && St == 2;
                return band_tail_launch(c, fn, ba, b4 ? ba : ba_plain, b4, St, r0, hbuf0, flops, bytes, false);
            }
            return band8_launch(c, fn, ba, flops, bytes);
        }
    }
    if (nh == 2 && c->hs_fuse_regressor && lr.Wh_f && cf.n_out <= PP_BN && l1.out % PP_BN == 0) {
        // regressor inside the pair kernel: h2 never leaves the CU (hs_fused_regressor)
        const int tiles_m = (M2 + PP_BM - 1) / PP_BM, tiles_n = l1.out / PP_BN;
        const size_t slab_bytes = (size_t)(tiles_n - 1) * M2 * cf.n_out * sizeof(float);
        int rc = ensure_bytes(c, &c->fuse_ws, &c->fuse_ws_bytes, slab_bytes + ((size_t)tiles_m + 64) * sizeof(unsigned));
        if (rc) return rc;
        const int s1 = hs_act_shift_of(c, m, 1, true);
        PairRegArgs rg{};
        rg.B2 = lr.Wh_f; rg.ldb2 = lr.ldwh; rg.n2 = cf.n_out;
        rg.bias2 = lr.bias_hs;
        rg.out = out;
        rg.slabs = reinterpret_cast<float*>(c->fuse_ws);
        rg.flags = reinterpret_cast<unsigned*>(c->fuse_ws + slab_bytes);
        rg.err = c->hs_peak + 2;
        rg.acc_scale2 = std::ldexp(1.f, -(s1 + lr.wshift_f));
        p.out_scale = std::ldexp(1.f, s1);
        p.tiles_n = tiles_n;
        p.peak = c->hs_peak;
        p.xcd_cols = 0;                     // the kernel's flag wait relies on the plain blockIdx -> (row tile, column tile) order: never the XCD remap
        HIP_TRY(c, hipMemsetAsync(rg.flags, 0, (size_t)tiles_m * sizeof(unsigned), c->stream));
        ++c->hs_launches;
        const double flops = 2.0 * (double)M2 * l1.out * h1 + 2.0 * (double)M2 * cf.n_out * l1.out;
        const double bytes = 4.0 * ((double)M2 / cf.nt * h1 + (double)cf.nt * h1 + (double)l1.out * h1 + (double)cf.n_out * l1.out +
                                    (double)M2 * cf.n_out * (2.0 * (tiles_n - 1) + 1.0));
        ProfScope ps(c, K_PAIR_DENSE, flops, bytes);
        auto kern = gemm_hs_pp_pair_kernel<EPI_RAW, false, false, 0, true>;
        rc = hs_dynamic_lds(c, kern, (size_t)PR_LDS_FLOATS * sizeof(float), &c->hs_lds_attr[3]);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, tiles_n)), dim3(PP_THREADS), (size_t)PR_LDS_FLOATS * sizeof(float), c->stream, p, src,
                           std::ldexp(1.f, s0), rg);
        HIP_TRY(c, hipGetLastError());
        return CSI_OK;
    }
    uint16_t* hb[2] = {reinterpret_cast<uint16_t*>(hbuf0), reinterpret_cast<uint16_t*>(hbuf1)};
    p.C = hb[0]; p.ldc = 2 * l1.out;
    p.c_blk = c->hs_blocked;            // activation matrices between split-engine layers: blocked layout (hs_blk_offset)
    p.out_scale = std::ldexp(1.f, hs_act_shift_of(c, m, 1));
    int rc = hs_launch_pair<EPI_BIAS_RELU_AFFINE, true>(c, K_PAIR_DENSE, p, src, s0);
    if (rc) return rc;
    int cur = 0;
    for (int li = 2; li <= nh; ++li) {
        const Layer& l = m.layers[li];
        GemmHsArgs q{};
        q.A = hb[cur]; q.lda = 2 * l.in;
        q.a_blk = q.c_blk = c->hs_blocked;
        q.Bt = l.Wh; q.ldb = l.ldwh;
        q.M = M2; q.N = l.out; q.K = l.in;
        q.bias = l.bias_hs; q.scale = l.scale; q.shift = c->hs_zero;
        q.acc_scale = std::ldexp(1.f, -(hs_act_shift_of(c, m, li - 1) + l.wshift));
        if (li == nh) {
            q.C = out; q.ldc = cf.n_out;
            rc = hs_launch_gemm<EPI_BIAS, false>(c, K_REGRESSOR, q);
        } else {
            q.out_scale = std::ldexp(1.f, hs_act_shift_of(c, m, li));
            q.C = hb[cur ^ 1]; q.ldc = 2 * l.out;
            rc = hs_launch_gemm<EPI_BIAS_RELU_AFFINE, true>(c, K_DENSE_HIDDEN, q);
            cur ^= 1;
        }
        if (rc) return rc;
    }
    return CSI_OK;
}

}  // namespace
