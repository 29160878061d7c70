// csi_dnn_bf16.hpp - host orchestration of the bf16-operand DNN path (BASELINE config 3).
#pragma once
#include "csi_context.hpp"

namespace {

// ---------------------------------------------------------------- bf16 mode
template <int EPI, bool OUT_BF16>
int launch_gemm_bf16(csi_ctx* c, int kid, GemmBf16Args g, int splits) {
    if (g.M <= 0) return CSI_OK;
    if ((g.lda & 7) || (g.ldb % B_BK))
        return fail(c, CSI_ERR_INVALID_ARG, "bf16 gemm: lda must be a multiple of 8 and ldb of 64 (lda=%d ldb=%d)", g.lda, g.ldb);
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (OUT_BF16 ? 2.0 : 4.0) * (double)g.M * g.N * splits;
    ProfScope ps(c, kid, flops, bytes);
    // 256x256 ping-pong tiles (8 waves, 1 workgroup per CU) once they fill the 256 CUs, 128x128
    // tiles (4 waves, 2 per CU) below
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    const long big_tiles = (long)tiles_m * ((g.N + PP_BN - 1) / PP_BN) * splits;
    const bool wide_ok = !OUT_BF16 || ((g.N & 7) == 0 && (g.ldc & 7) == 0);
    const int force = c->force_pair_tile;      // "force_tile" option: 256 -> ping-pong kernel, 128 -> 128x128 lock-step kernel
    if (wide_ok && (force == 256 || (force == 0 && big_tiles >= 256))) {
        g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
        dim3 grid(pp_grid(tiles_m, g.tiles_n), 1, (unsigned)splits);
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OUT_BF16, 5>), grid, dim3(PP_THREADS), 0, c->stream, g);
    } else if (big_tiles >= 256 && force != 128) {
        g.tiles_n = (g.N + 255) / 256;
        dim3 grid((unsigned)(((g.M + 255) / 256) * g.tiles_n), 1, (unsigned)splits);
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OUT_BF16, 2, 4, 4, 2, 2>), grid, dim3(512), 0, c->stream, g);
    } else {
        g.tiles_n = (g.N + 127) / 128;
        dim3 grid((unsigned)(((g.M + 127) / 128) * g.tiles_n), 1, (unsigned)splits);
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OUT_BF16, 2, 2, 2, 2, 2>), grid, dim3(256), 0, c->stream, g);
    }
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// Split-K of the bf16 layer-0 product ([M1 x len_ltf] x [len_ltf x h1], M1 = packets x rx): with one
// 256x256 workgroup per CU the row tiles alone rarely fill whole rounds of 256 CUs (config 3:
// 79 x 4 tiles = 1.23 rounds); choose the split count whose last round is fullest.
constexpr int BF16_L0_MAX_SPLITS = 8;
int bf16_layer0_splits(int M1, int h1, int K) {
    const long tiles = (long)((M1 + PP_BM - 1) / PP_BM + 7) / 8 * 8 * ((h1 + PP_BN - 1) / PP_BN);      // as launched (pp_grid)
    if (tiles < 256) return 1;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= BF16_L0_MAX_SPLITS; ++s) {
        if (K / s < 1024) break;
        const long blocks = tiles * s;
        const double eff = (double)blocks / (double)((blocks + 255) / 256 * 256) - 0.01 * (s - 1);     // slab traffic
        if (eff > best_eff) { best_eff = eff; best = s; }
    }
    return best;
}

// Layer 0 of a call of at most "l0_stream_max_rows" rx preambles on the weight-streaming kernel (l0_hs_stream.hip.h, bf16 form): k ranges, 0 = no
int bf16_l0_stream_splits(const csi_ctx* c, int M1, int h1, int K, int* k_per_split) {
    if (!c->l0_stream || M1 < 1 || M1 > c->l0_stream_max_rows || (K % L0S_KC) != 0 || K < 2 * L0S_KC || c->force_pair_tile) return 0;
    const int groups = (h1 + L0S_COLS - 1) / L0S_COLS, blocks = (M1 + 255) / 256;
    int ks = c->l0_stream_ks > 0 ? c->l0_stream_ks : std::max(1, (M1 <= 128 ? 256 : 128) / (groups * blocks));
    ks = std::max(1, std::min(ks, K / (2 * L0S_KC)));
    const int kps = ((K + ks - 1) / ks + L0S_KC - 1) / L0S_KC * L0S_KC;
    *k_per_split = kps;
    return (K + kps - 1) / kps;
}

int bf16_l0_stream_launch(csi_ctx* c, const Model& m, const float* x, int ldx, int M1, int h1, int K, int kps, int splits, float* slabs) {
    L0StreamBf16Args a{};
    a.x = x; a.Wb = m.layers[0].Wb; a.slabs = slabs;
    a.M = M1; a.N = h1; a.K = K; a.lda = ldx; a.ldwb = m.layers[0].ldwb; a.kps = kps;
    const int blocks = (M1 + 255) / 256;
    const int rt = ((M1 + 31) / 32 + blocks - 1) / blocks;
    const int rt_inst = rt <= 4 ? rt : (rt <= 6 ? 6 : 8);
    const dim3 grid((unsigned)((h1 + L0S_COLS - 1) / L0S_COLS), (unsigned)splits, (unsigned)((M1 + 32 * rt_inst - 1) / (32 * rt_inst)));
    ++c->l0_stream_launches;
    ProfScope ps(c, K_LAYER0_LTF, 2.0 * (double)M1 * h1 * K, 4.0 * (double)M1 * K + 2.0 * (double)h1 * K + 4.0 * (double)M1 * h1 * splits);
    auto go = [&](auto kern, int rtt) {
        hipLaunchKernelGGL(kern, grid, dim3(256), l0b_lds_bytes(rtt), c->stream, a);       // (at most 41 KB: below the default dynamic-LDS limit)
        HIP_TRY(c, hipGetLastError());
        return (int)CSI_OK;
    };
    if (rt <= 1) return go(l0_bf16_stream_kernel<1>, 1);
    if (rt <= 2) return go(l0_bf16_stream_kernel<2>, 2);
    if (rt <= 3) return go(l0_bf16_stream_kernel<3>, 3);
    if (rt <= 4) return go(l0_bf16_stream_kernel<4>, 4);
    if (rt <= 6) return go(l0_bf16_stream_kernel<6>, 6);
    return go(l0_bf16_stream_kernel<8>, 8);
}

// first per-pair layer with h1 generated inside the GEMM (gemm_bf16_pp_pair_kernel); usable when the
// grid is large, the reduction length fits the LDS copy of the bn0 vectors and rows are 16-byte aligned
bool pair_fused_ok(const csi_ctx* c, int M, int N, int K, bool out_bf16, int ldc) {
    if (c->force_pair_tile == 128 || c->bf16_fused_h1 == 0) return false;
    const long tiles = (long)((M + PP_BM - 1) / PP_BM) * ((N + PP_BN - 1) / PP_BN);
    if (tiles < 256 && c->force_pair_tile != 256) return false;
    if (K > 4096 || (K % PP_BK) != 0) return false;
    return !out_bf16 || ((N & 7) == 0 && (ldc & 7) == 0);
}

template <int EPI, bool OUT_BF16>
int launch_pair_bf16(csi_ctx* c, int kid, GemmBf16Args g, const PairSrc& ps) {
    if (g.M <= 0) return CSI_OK;
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 4.0 * ((double)g.M / ps.nt * g.K + (double)ps.nt * g.K) + 2.0 * (double)g.N * g.K + (OUT_BF16 ? 2.0 : 4.0) * (double)g.M * g.N;
    ProfScope psc(c, kid, flops, bytes);
    g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    const size_t lds = (size_t)PPP_RING_FLOATS * sizeof(float);
    auto kern = gemm_bf16_pp_pair_kernel<EPI, OUT_BF16>;
    static thread_local size_t attr_set = 0;       // per instantiation
    if (attr_set < lds) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n)), dim3(PP_THREADS), lds, c->stream, g, ps);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// layer 0 straight from the fp32 preambles (the bf16 conversion happens while the A image is built)
int launch_layer0_cast_bf16(csi_ctx* c, GemmBf16Args g, const float* x, int ldx, int splits) {
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 4.0 * (double)g.M * g.K + 2.0 * (double)g.N * g.K + 4.0 * (double)g.M * g.N * splits;
    ProfScope psc(c, K_LAYER0_LTF, flops, bytes);
    g.tiles_n = (g.N + PP_BN - 1) / PP_BN;
    const int tiles_m = (g.M + PP_BM - 1) / PP_BM;
    const size_t lds = (size_t)PPP_RING_FLOATS * sizeof(float);
    auto kern = gemm_bf16_pp_pair_kernel<EPI_RAW, false, true>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    PairSrc src{x, nullptr, ldx, 1};
    hipLaunchKernelGGL(kern, dim3(pp_grid(tiles_m, g.tiles_n), 1, (unsigned)splits), dim3(PP_THREADS), lds, c->stream, g, src);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

int cast_bf16(csi_ctx* c, const float* src, bf16_t* dst, size_t n) {
    ProfScope ps(c, K_CAST_BF16, 0.0, 6.0 * n);
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 8192);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, src, dst, n8);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// hidden layers 1.. and the regressor on a bf16 activation matrix hin [M][l.in]; writes d_out fp32
// fused: the first layer of the tail generates its A operand from (L0, T) - hin is then unused
int bf16_tail(csi_ctx* c, Model& m, const bf16_t* hin, int M, bf16_t* hb0, bf16_t* hb1, float* d_out, int first_layer,
              const PairSrc* fused = nullptr) {
    const csi_config& cf = c->cfg;
    bf16_t* hb[2] = {hb0, hb1};
    const bf16_t* cur = hin;
    int w = 0;
    for (int li = first_layer; li <= cf.n_hidden; ++li) {
        const Layer& l = m.layers[li];
        GemmBf16Args q{};
        q.A = cur; q.lda = l.in;
        q.Bt = l.Wb; q.ldb = l.ldwb;
        q.M = M; q.N = l.out; q.K = l.in;
        q.bias = l.bias; q.scale = l.scale; q.shift = l.shift;
        q.k_per_split = l.ldwb;
        const bool fuse = fused && li == first_layer;
        int rc;
        if (li == cf.n_hidden) {
            q.C = d_out; q.ldc = cf.n_out;
            rc = fuse ? launch_pair_bf16<EPI_BIAS, false>(c, K_REGRESSOR, q, *fused) : launch_gemm_bf16<EPI_BIAS, false>(c, K_REGRESSOR, q, 1);
        } else {
            q.C = hb[w]; q.ldc = l.out;
            const int kid = li == 1 ? K_PAIR_DENSE : K_DENSE_HIDDEN;
            rc = fuse ? launch_pair_bf16<EPI_BIAS_RELU_AFFINE, true>(c, kid, q, *fused)
                      : launch_gemm_bf16<EPI_BIAS_RELU_AFFINE, true>(c, kid, q, 1);
            cur = hb[w];
            w ^= 1;
        }
        if (rc) return rc;
    }
    return CSI_OK;
}

int predict_plane_bf16(csi_ctx* c, Model& m, const float* d_ltf, int64_t npkt, float* d_out) {
    const csi_config& cf = c->cfg;
    const int nt = cf.nt, nr = cf.nr, h1 = cf.hidden[0], nh = cf.n_hidden;
    int maxh = 0;
    for (int i = 1; i < nh; ++i) maxh = std::max(maxh, cf.hidden[i]);
    // per packet: bf16 preamble copy, fp32 layer-0 product, bf16 h1, bf16 ping-pong hidden buffers
    const size_t per_pkt = (size_t)nr * cf.len_ltf * 2 + (size_t)nr * h1 * 4 * (BF16_L0_MAX_SPLITS + 1) + (size_t)nr * nt * h1 * 2 +
                           (size_t)nr * nt * maxh * 2 * (nh >= 3 ? 2 : (nh >= 2 ? 1 : 0));
    // default 8 GiB: a whole config-3 step (5000 packets) in one chunk, so that layer 0 sees M1 = 20000 rows
    // (two component models in flight on two streams hold a workspace each: the cap is shared between them - ADVICE round 5)
    const size_t budget = (cf.workspace_bytes > 0 ? (size_t)cf.workspace_bytes : ((size_t)8 << 30)) / (size_t)std::max(c->models_in_flight, 1);
    int64_t cap = std::max<int64_t>(1, (int64_t)(budget / per_pkt));
    cap = std::min(cap, (int64_t)0x7fffffff / ((int64_t)nr * nt * 2));
    const int64_t nchunks = (npkt + cap - 1) / cap;
    const int64_t chunk = (npkt + nchunks - 1) / nchunks;
    int rc = ensure_bytes(c, &c->ws, &c->ws_bytes, per_pkt * (size_t)chunk + 1024);
    if (rc) return rc;
    char* base = c->ws;
    bf16_t* xb = reinterpret_cast<bf16_t*>(base);             base += (size_t)chunk * nr * cf.len_ltf * 2;
    float* l0_ws = reinterpret_cast<float*>(base);            base += (size_t)chunk * nr * h1 * 4 * (BF16_L0_MAX_SPLITS + 1);
    bf16_t* h1b = reinterpret_cast<bf16_t*>(base);            base += (size_t)chunk * nr * nt * h1 * 2;
    bf16_t* hb0 = reinterpret_cast<bf16_t*>(base);            base += (size_t)chunk * nr * nt * maxh * 2;
    bf16_t* hb1 = reinterpret_cast<bf16_t*>(base);
    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        const int M1 = (int)(np * nr), M2 = (int)(np * nr * nt);
        GemmBf16Args g{};
        g.A = xb; g.lda = cf.len_ltf;
        g.Bt = m.layers[0].Wb; g.ldb = m.layers[0].ldwb;
        g.C = l0_ws; g.ldc = h1;
        g.M = M1; g.N = h1; g.K = cf.len_ltf;
        int S = bf16_layer0_splits(M1, h1, cf.len_ltf);
        g.k_per_split = ((cf.len_ltf + S - 1) / S + B_BK - 1) / B_BK * B_BK;
        // between the streaming kernel's range and the fused 256 x 256 kernel's (fewer than 256 of its tiles): the 128 x 128 kernel, its K cut
        // into as many ranges as fill the 256 CUs (384 packets at Nt = 64: 96 workgroups over the whole K took 320-360 us per model)
        if (S == 1) {
            const long tiles128 = (long)((M1 + 127) / 128) * ((h1 + 127) / 128);
            int s2 = (int)std::min<long>(BF16_L0_MAX_SPLITS, (256 + tiles128 - 1) / tiles128);
            while (s2 > 1 && cf.len_ltf / s2 < 1024) --s2;
            const long tiles256 = (long)((M1 + PP_BM - 1) / PP_BM) * ((h1 + PP_BN - 1) / PP_BN);
            if (s2 > 1 && tiles256 < 256) {
                S = s2;
                g.k_per_split = ((cf.len_ltf + S - 1) / S + B_BK - 1) / B_BK * B_BK;
            }
        }
        // round 6: between the streaming kernel's range and 256 tiles of the fused kernel (321 ... 4095 packets at Nt = 64, Nr = 4) layer 0 took a cast pass
        // plus the 128 x 128 kernel (1000 packets: 100-130 us + 450 us per model, 0.15 of the bf16 peak; tools/ls_overlap_trace.sh shows it).  The fused
        // 256 x 256 kernel with its K cut so that tiles x ranges fill the CUs does the same product without the cast pass ("bf16_l0_fused_split" = 0: before)
        int kps_hint = 0;
        const int stream_splits_hint = bf16_l0_stream_splits(c, M1, h1, cf.len_ltf, &kps_hint);       // (the streaming kernel takes the call: nothing to choose)
        if (!stream_splits_hint && c->bf16_l0_fused_split && c->bf16_fused_h1 != 0 && c->force_pair_tile == 0 && (cf.len_ltf & 3) == 0) {
            const long launched = (long)(((M1 + PP_BM - 1) / PP_BM + 7) / 8 * 8) * ((h1 + PP_BN - 1) / PP_BN);      // as pp_grid launches them
            if (launched < 256) {
                // K ranges: rounds of 256 workgroups x the k extent of one range, plus the slabs written and read back (a 256 x 256 x k workgroup at the
                // kernel's measured rate ~40 ns per k; a slab of M1 x h1 floats at ~4 TB/s both ways).  1500 packets: 3 ranges = 288 workgroups = two rounds
                // (3079 us per call), 5 ranges = 480 of 512
                int s3 = 1;
                double best = 1e30;
                for (int sx = 1; sx <= BF16_L0_MAX_SPLITS; ++sx) {
                    const int kps = ((cf.len_ltf + sx - 1) / sx + B_BK - 1) / B_BK * B_BK;
                    if (sx > 1 && (cf.len_ltf / sx < 1024 || kps / PP_BK < 3)) break;
                    const double cost = (double)((launched * sx + 255) / 256) * kps * 40e-9 + (sx > 1 ? (double)sx * M1 * h1 * 8.0 / 4e12 : 0.0);
                    if (cost < best) { best = cost; s3 = sx; }
                }
                if (launched * s3 >= 128) {
                    ++c->bf16_l0_fused_split_launches;
                    S = s3;
                    g.k_per_split = ((cf.len_ltf + S - 1) / S + B_BK - 1) / B_BK * B_BK;
                }
            }
        }
        float* l0 = l0_ws;
        int kps_stream = 0;
        const int stream_splits = bf16_l0_stream_splits(c, M1, h1, cf.len_ltf, &kps_stream);
        const long l0_tiles = (long)((M1 + PP_BM - 1) / PP_BM) * ((h1 + PP_BN - 1) / PP_BN) * S;
        const long l0_launched = (long)(((M1 + PP_BM - 1) / PP_BM + 7) / 8 * 8) * ((h1 + PP_BN - 1) / PP_BN) * S;
        const bool l0_fused = c->bf16_fused_h1 != 0 && c->force_pair_tile != 128 && (l0_tiles >= 256 || c->force_pair_tile == 256 || (c->bf16_l0_fused_split && l0_launched >= 128)) &&
                              g.k_per_split / PP_BK >= 3 && (cf.len_ltf & 3) == 0;
        if (stream_splits) {
            // small and mid-size calls: the weight-streaming kernel, its k-range slabs in the split-K scratch of the context
            rc = ensure_bytes(c, &c->skbuf, &c->skbuf_bytes, (size_t)(stream_splits + 1) * M1 * h1 * sizeof(float));
            if (rc) return rc;
            l0 = reinterpret_cast<float*>(c->skbuf);
            S = stream_splits;
            rc = bf16_l0_stream_launch(c, m, d_ltf + (size_t)p0 * nr * cf.len_ltf, cf.len_ltf, M1, h1, cf.len_ltf, kps_stream, stream_splits, l0);
        } else if (l0_fused) {
            // the conversion happens inside the GEMM: no separate pass over the preambles
            rc = launch_layer0_cast_bf16(c, g, d_ltf + (size_t)p0 * nr * cf.len_ltf, cf.len_ltf, S);
        } else {
            rc = cast_bf16(c, d_ltf + (size_t)p0 * nr * cf.len_ltf, xb, (size_t)M1 * cf.len_ltf);
            if (rc) return rc;
            rc = launch_gemm_bf16<EPI_RAW, false>(c, K_LAYER0_LTF, g, S);
        }
        if (rc) return rc;
        const Layer& l1 = m.layers[1];
        const bool regressor_first = cf.n_hidden == 1;
        const bool fused_ok = pair_fused_ok(c, M2, l1.out, h1, !regressor_first, l1.out);
        // small calls (fewer row tiles than the fused pair kernel wants): the band kernel in its column-split launch serves them too
        const bool band_small = !fused_ok && nh == 2 && c->hs_band && m.layers[nh].Wb_p && (c->band_split != 0 && c->band_split != 1) && c->force_pair_tile == 0 &&
                                c->bf16_fused_h1 != 0 && (M2 + BAND_ROWS - 1) / BAND_ROWS * 2 <= 256 && nt >= 32 && nt <= 64 && h1 >= 256 && (h1 % 128) == 0 &&
                                (l1.out % 256) == 0 && l1.out <= BAND8_MAX_N1 && cf.n_out <= 256 && l1.in == h1 && l1.ldwb == h1 && m.layers[nh].ldwb == l1.out;
        bool done = false;
        if (fused_ok || band_small) {
            // h1 is generated inside the first per-pair GEMM from the (slab-summed) layer-0 product
            float* l0sum = S > 1 ? l0 + (size_t)S * M1 * h1 : l0;
            auto sum_slabs = [&]() -> int {          // the k-range slabs of layer 0 -> l0sum, once a kernel that reads it is known to run
                if (S <= 1) return CSI_OK;
                ProfScope ps(c, K_SPLITK_REDUCE, (double)S * M1 * h1, 4.0 * (S + 1) * (double)M1 * h1);
                const size_t n4 = (size_t)M1 * h1 / 4;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, c->stream, l0, l0sum, n4, S);
                HIP_TRY(c, hipGetLastError());
                return CSI_OK;
            };
            PairSrc src{l0sum, m.T, h1, nt};
            const Layer& lr = m.layers[nh];
            hipFunction_t fn = nullptr;
            BandArgs ba{};
            if (nh == 2 && c->hs_band && lr.Wb_p) {
                // first per-pair layer + regressor as one kernel, h2 in registers: the bf16 form of the assembly band kernel.  With the
                // L0 / T values of the fragments streamed through LDS (32 <= nt <= 64) it replaces the two bf16 kernels by default: 3.52
                // against 3.37 + 0.99 ms per launch at configs[2]; the form with per-lane global loads of those values (any nt) is bound by
                // the vector-memory path - 4.9 ms - and runs only on request ("hs_band" = 2): profiles/r03_bench_bf16_band_ab.txt
                ba.L0 = l0sum; ba.Ts = m.T; ba.ldl = h1; ba.nt = nt; ba.in_scale = 1.f;
                ba.W1 = reinterpret_cast<const uint16_t*>(l1.Wb); ba.ldb1 = l1.ldwb; ba.bias1 = l1.bias; ba.M = M2; ba.K1 = h1; ba.N1 = l1.out;
                ba.acc_scale1 = 1.f; ba.out_scale = 1.f;
                ba.W2p = reinterpret_cast<const uint16_t*>(lr.Wb_p); ba.ldb2 = lr.ldwb; ba.bias2 = lr.bias; ba.n2 = cf.n_out; ba.acc_scale2 = 1.f;
                ba.out = d_out + (size_t)p0 * nr * nt * cf.n_out; ba.ldo = cf.n_out; ba.peak = nullptr;
                if (band8_serves(ba, true) && (c->hs_band >= 2 || band8_staged(ba)) && l1.in == h1 && lr.in == l1.out && l1.ldwb == h1 && lr.ldwb == l1.out) {
                    rc = band8_function(c, &fn, true, band8_staged(ba) && c->hs_band != 3);
                    if (rc) return rc;
                }
                if (fn && band8_staged(ba) && c->hs_band != 3) {       // the kernel streams the pilot table slab by slab through LDS: its slab-ordered copy
                    if (!m.T_sw_ok) {
                        const size_t floats = (size_t)(h1 / 32 + 1) * nt * 32;
                        if (!m.T_sw && hipMalloc((void**)&m.T_sw, (floats + 512) * sizeof(float)) != hipSuccess)
                            return fail(c, CSI_ERR_NOMEM, "device allocation of the slab-ordered pilot table failed");
                        hipLaunchKernelGGL(band_tsw_kernel<32>, dim3(256), dim3(256), 0, c->stream, m.T, h1, nt, h1, m.T_sw);
                        HIP_TRY(c, hipGetLastError());
                        m.T_sw_ok = true;
                    }
                    ba.Ts = m.T_sw;
                }
            }
            if (fn) {
                rc = sum_slabs();
                if (rc) return rc;
                const double flops = 2.0 * (double)M2 * l1.out * h1 + 2.0 * (double)M2 * cf.n_out * l1.out;
                const double bytes = 4.0 * ((double)M2 / nt * h1 + (double)nt * h1) + 2.0 * ((double)l1.out * h1 + (double)cf.n_out * l1.out) + 4.0 * (double)M2 * cf.n_out;
                // small calls: the column-split launch (csi_dnn_hs.hpp); its partial outputs live in the h1 / activation buffers this path leaves unused
                const bool staged_fn = fn == c->band_fn_bf16;
                const int Sb = (staged_fn && ba.ldo == ba.n2 && (ba.N1 * (long)ba.ldb1 * 2 < 0x7fffffffL)) ?
                                   band8_splits(c, ba, ((size_t)M2 * h1 + (size_t)M2 * maxh) / 2, true) : 1;
                // round 6: the register-blocked form (4 waves x 512 registers) serves the same staged shapes with the same operand buffers
                const bool hooked4 = c->band_bf16_threads == 256;          // (CSI_DEBUG_HOOKS: a csi_band4* variant named in place of csi_band8_bf16)
                const bool blocked = staged_fn && c->band4 && c->band_fn4_bf16 && !ba.stamps;
                if (Sb > 1) {
                    const bool b4 = blocked && c->band_fn4_bf16_cs && Sb == 2;      // (4 splits: the 8-wave form, as in fp32 contexts - csi_dnn_hs.hpp)
                    if (b4) rc = band4_prepare(c, m, ba, true);
                    if (!rc) rc = band8_launch_split(c, ba, Sb, reinterpret_cast<float*>(h1b), flops, bytes, true, b4);
                } else {
                    const BandArgs ba_plain = ba;          // (band4_prepare puts the tiled weight copies into ba)
                    if (blocked || (staged_fn && hooked4 && !ba.stamps)) {
                        rc = band4_prepare(c, m, ba, true);
                        if (blocked) fn = c->band_fn4_bf16;
                    }
                    long r0 = 0;
                    const int St = (!rc && staged_fn && (ba.N1 * (long)ba.ldb1 * 2 < 0x7fffffffL)) ?
                                       band_tail_plan(c, ba, ((size_t)M2 * h1 + (size_t)M2 * maxh) / 2, true, &r0) : 0;
                    if (St) {
                        const bool b4 = blocked && c->band_fn4_bf16_cs && St == 2;
                        rc = band_tail_launch(c, fn, ba, b4 ? ba : ba_plain, b4, St, r0, reinterpret_cast<float*>(h1b), flops, bytes, true);
                    } else if (!rc) rc = band8_launch(c, fn, ba, flops, bytes);
                }
                done = true;
            } else if (fused_ok) {
                rc = sum_slabs();
                if (rc) return rc;
                rc = bf16_tail(c, m, nullptr, M2, hb0, hb1, d_out + (size_t)p0 * nr * nt * cf.n_out, 1, &src);
                done = true;
            }
        }
        if (!done) {
            {
                ProfScope ps(c, K_PAIR_H1_BF16, 3.0 * M2 * h1, 2.0 * M2 * h1 + 4.0 * M1 * h1);
                hipLaunchKernelGGL(pair_h1_bf16_kernel, dim3((unsigned)std::min(M1, 65536)), dim3(256), 0, c->stream, l0, S, (size_t)M1 * h1, m.T,
                                   m.layers[0].scale, m.layers[0].shift, h1b, M1, nt, h1);
                HIP_TRY(c, hipGetLastError());
            }
            rc = bf16_tail(c, m, h1b, M2, hb0, hb1, d_out + (size_t)p0 * nr * nt * cf.n_out, 1);
        }
        if (rc) return rc;
    }
    return CSI_OK;
}

}  // namespace
