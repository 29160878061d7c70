// csi_dnn_f32.hpp - host orchestration of the fp32 DNN path: GEMM launchers (tile height, XCD tile
// order, split-K choices) and the packet-chunk loop of one component model.
#pragma once
#include "csi_context.hpp"

namespace {

// ---------------------------------------------------------------- GEMM launch helpers
// Small-batch latency path: a GEMM with few output tiles (the reference's literal one-packet call
// has 8) is split along K over ~256 workgroups and combined by splitk_epilogue_kernel.
int small_batch_splits(long tiles, int K) {
    if (tiles >= 96 || K < 256) return 1;
    long s = 384 / std::max<long>(tiles, 1);
    s = std::min<long>(s, K / 64);           // >= 4 ring k-tiles per workgroup
    return (int)std::max<long>(s, 1);
}

template <int EPI>
int launch_splitk_epilogue(csi_ctx* c, const GemmArgs& g, const float* slabs, int S) {
    ProfScope ps(c, K_SPLITK_REDUCE, (double)S * g.M * g.N, 4.0 * (S + 1) * (double)g.M * g.N);
    const size_t total = (size_t)g.M * g.N;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL((splitk_epilogue_kernel<EPI>), dim3(blocks), dim3(256), 0, c->stream, slabs, S, g.M, g.N, g.C, g.ldc,
                       g.bias, g.scale, g.shift);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

template <int EPI>
int launch_gemm(csi_ctx* c, int kid, GemmArgs g, int splits) {
    if (g.M <= 0) return CSI_OK;
    if ((g.K & 3) || (g.lda & 3) || (g.ldb % G_BK))
        return fail(c, CSI_ERR_INVALID_ARG, "gemm: K/lda must be multiples of 4 and ldb of 32 (K=%d lda=%d ldb=%d)",
                    g.K, g.lda, g.ldb);
    const int tiles_m = (g.M + G_BM - 1) / G_BM;
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    if (EPI != EPI_RAW && splits == 1) {
        const int S = c->force_pair_tile ? 1 : small_batch_splits((long)tiles_m * g.tiles_n, g.K);
        if (S > 1) {
            int rc = ensure_bytes(c, &c->skbuf, &c->skbuf_bytes, (size_t)S * g.M * g.N * sizeof(float));
            if (rc) return rc;
            GemmArgs r = g;
            r.C = reinterpret_cast<float*>(c->skbuf);
            r.ldc = g.N;
            r.k_per_split = ((g.K + G_BK - 1) / G_BK + S - 1) / S * G_BK;
            const int real = (g.K + r.k_per_split - 1) / r.k_per_split;
            {
                ProfScope ps(c, kid, 2.0 * (double)g.M * g.N * g.K, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * real));
                r.tiles_m = 0;           // few tiles: linear order
                hipLaunchKernelGGL((gemm_f32_kernel<EPI_RAW>), dim3((unsigned)(tiles_m * g.tiles_n), 1, (unsigned)real), dim3(G_THREADS), 0, c->stream, r);
                HIP_TRY(c, hipGetLastError());
            }
            return launch_splitk_epilogue<EPI>(c, g, r.C, real);
        }
    }
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double a_rows = (double)g.M;
    const double bytes = 4.0 * (a_rows * g.K + (double)g.N * g.K + (double)g.M * g.N * splits);
    ProfScope ps(c, kid, flops, bytes);
    // 256-row tiles (fewer LDS-DMA instructions per MFMA) once they fill the 512 resident slots
    const int tiles_m256 = (g.M + G2_BM - 1) / G2_BM;
    const bool big = c->force_pair_tile == 256 || (c->force_pair_tile != 128 && (long)tiles_m256 * g.tiles_n * splits >= 512);
    const int tm_used = big ? tiles_m256 : tiles_m;
    // measured: traffic -50..60 %, time neutral for 8 column tiles and for the 256-row kernel,
    // -10 % for the 128-row kernel with 2 column tiles (not used there)
    const bool xcd_order = c->xcd_order >= 0 ? c->xcd_order != 0
                                             : (g.tiles_n >= 8 || big) && tile_map_pays(tm_used, g.tiles_n, splits);
    g.tiles_m = xcd_order ? tm_used : 0;
    const dim3 grid(xcd_order ? tile_map_grid(tm_used, g.tiles_n) : (unsigned)(tm_used * g.tiles_n), 1, (unsigned)splits);
    if (big) hipLaunchKernelGGL((gemm256_f32_kernel<EPI>), grid, dim3(G_THREADS), 0, c->stream, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<EPI>), grid, dim3(G_THREADS), 0, c->stream, g);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// first per-pair layer (fragment-time h1 kernels, 4 <= nt <= 128)
template <int EPI>
int launch_pair(csi_ctx* c, int kid, GemmArgs g) {
    if (g.nt < 4 || g.nt > 128)
        return fail(c, CSI_ERR_INVALID_ARG, "the per-pair layer supports 4 <= nt <= 128 (got %d)", g.nt);
    if (g.M <= 0) return CSI_OK;
    if ((g.K & 3) || (g.lda & 3) || (g.ldb % G_BK))
        return fail(c, CSI_ERR_INVALID_ARG, "pair gemm: K/lda must be multiples of 4 and ldb of 32 (K=%d lda=%d ldb=%d)",
                    g.K, g.lda, g.ldb);
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    const int tiles_m = (g.M + G_BM - 1) / G_BM;
    const int tiles_m256 = (g.M + P2_BM - 1) / P2_BM;
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 4.0 * ((double)g.M / g.nt * g.K + (double)g.N * g.K + (double)g.M * g.N);

    // small batch: split K over the chip, combine + epilogue in a second (tiny) kernel
    const int S = c->force_pair_tile ? 1 : small_batch_splits((long)tiles_m * g.tiles_n, g.K);
    if (S > 1) {
        int rc = ensure_bytes(c, &c->skbuf, &c->skbuf_bytes, (size_t)S * g.M * g.N * sizeof(float));
        if (rc) return rc;
        GemmArgs r = g;
        r.C = reinterpret_cast<float*>(c->skbuf);
        r.ldc = g.N;
        r.k_per_split = ((g.K + G_BK - 1) / G_BK + S - 1) / S * G_BK;
        const int real = (g.K + r.k_per_split - 1) / r.k_per_split;
        {
            ProfScope ps(c, kid, flops, bytes);
            const dim3 grid((unsigned)(tiles_m * g.tiles_n), 1, (unsigned)real);
            if (g.nt <= 64) hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI_RAW, 1>), grid, dim3(G_THREADS), 0, c->stream, r);
            else hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI_RAW, 2>), grid, dim3(G_THREADS), 0, c->stream, r);
            HIP_TRY(c, hipGetLastError());
        }
        return launch_splitk_epilogue<EPI>(c, g, r.C, real);
    }

    ProfScope ps(c, kid, flops, bytes);
    // 256-row tiles halve the LDS-DMA instructions per MFMA; use them once they fill the 512
    // resident workgroup slots, 128-row tiles (more workgroups) below that.
    const bool big = c->force_pair_tile == 256 || (c->force_pair_tile != 128 && (long)tiles_m256 * g.tiles_n >= 512);
    if (big) {
        const dim3 grid((unsigned)(tiles_m256 * g.tiles_n));
        if (g.nt < 8) hipLaunchKernelGGL((pair_gemm256_f32_kernel<EPI, 1, 2>), grid, dim3(G_THREADS), 0, c->stream, g);
        else if (g.nt <= 64) hipLaunchKernelGGL((pair_gemm256_f32_kernel<EPI, 1, 1>), grid, dim3(G_THREADS), 0, c->stream, g);
        else hipLaunchKernelGGL((pair_gemm256_f32_kernel<EPI, 2, 1>), grid, dim3(G_THREADS), 0, c->stream, g);
    } else {
        const dim3 grid((unsigned)(tiles_m * g.tiles_n));
        g.k_per_split = (g.K + G_BK - 1) / G_BK * G_BK;
        if (g.nt <= 64) hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI, 1>), grid, dim3(G_THREADS), 0, c->stream, g);
        else hipLaunchKernelGGL((pair_gemm_f32_kernel<EPI, 2>), grid, dim3(G_THREADS), 0, c->stream, g);
    }
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// Split-K factor of layer 0.  The grid should fill whole rounds of the 512 resident workgroups
// (256 CUs x 2): pick the smallest factor whose last round is >= 90 % full, else the fullest.
int choose_splits(int M, int N, int K, int* k_per_split) {
    const long tiles = (long)((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    const int ktiles = (K + G_BK - 1) / G_BK;
    static const int cand[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 32, 40};
    int best = 1;
    double best_eff = -1.0;
    for (int s : cand) {
        if (s > 1 && ktiles / s < 8) break;           // keep >= 8 k-tiles (of 32) per block
        const int kps = (ktiles + s - 1) / s;
        const int real = (ktiles + kps - 1) / kps;
        const double rounds = (double)tiles * real / 512.0;
        const double eff = rounds / std::ceil(rounds);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
        if (eff >= 0.9) { best = s; break; }
    }
    const int kps = (ktiles + best - 1) / best * G_BK;
    *k_per_split = kps;
    return (K + kps - 1) / kps;
}

int build_pilot_table(csi_ctx* c, Model& m) {
    if (!m.loaded || !c->pilot_ok || c->cfg.nt == 0) return CSI_OK;
    const int nt = c->cfg.nt, h1 = c->cfg.hidden[0];
    if (!m.T) {
        const size_t bytes = ((size_t)nt * h1 + G_SLACK_FLOATS) * sizeof(float);
        if (hipMalloc((void**)&m.T, bytes) != hipSuccess)
            return fail(c, CSI_ERR_NOMEM, "pilot table allocation failed");
        HIP_TRY(c, hipMemsetAsync(m.T, 0, bytes, c->stream));
    }
    ProfScope ps(c, K_PILOT_TABLE, 2.0 * nt * nt * h1, 4.0 * (nt * nt + 2.0 * nt * h1));
    hipLaunchKernelGGL(pilot_table_kernel, dim3((h1 + 255) / 256, nt), dim3(256), 0, c->stream,
                       c->P, m.W0p, m.layers[0].bias, m.T, nt, h1, 1.0f);
    HIP_TRY(c, hipGetLastError());
    m.table_ok = true;
    m.T_hs_shift = HS_SHIFT_AUTO;       // the split engine's pre-scaled copy is rebuilt on its next use
    m.T_sw_ok = false;                  // ... and the slab-ordered copy of the bf16 band kernel
    return CSI_OK;
}

// ---------------------------------------------------------------- DNN, shared layer 0
// d_ltf: [npkt][nr][len_ltf] one component plane; d_out: [npkt][nr][nt][n_out]
int predict_plane(csi_ctx* c, Model& m, const float* d_ltf, int64_t npkt, float* d_out) {
    const csi_config& cf = c->cfg;
    const int nt = cf.nt, nr = cf.nr, h1 = cf.hidden[0], nh = cf.n_hidden;
    int maxh = 0;
    for (int i = 1; i < nh; ++i) maxh = std::max(maxh, cf.hidden[i]);
    // Packet chunks: as few as the workspace allows, all of (nearly) the same size so that every
    // chunk fills the machine equally well.  Per packet: layer-0 slabs (+ their sum when split-K
    // is on - the factor depends on the chunk size, hence the loop) and the ping-pong buffers of
    // the hidden activations.
    const size_t hid_pkt = (size_t)nr * nt * maxh * 4 * (nh >= 3 ? 2 : (nh >= 2 ? 1 : 0));
    // 4 GiB: a config-2 step is one chunk.  Two component models in flight on two streams hold a workspace each: the cap is shared between them
    const size_t budget = (cf.workspace_bytes > 0 ? (size_t)cf.workspace_bytes : ((size_t)4 << 30)) / (size_t)std::max(c->models_in_flight, 1);
    const int64_t max_rows = (int64_t)0x7fffffff / ((int64_t)nr * nt * 2);     // M2 must fit an int
    int64_t nchunks = 1, chunk = npkt;
    int splits_max = 1;
    const bool hs_ok = hs_static_ok(c, m);          // split-f16 engine available for this model (csi_dnn_hs.hpp)
    for (;;) {
        chunk = (npkt + nchunks - 1) / nchunks;
        int kps_tmp;
        splits_max = choose_splits((int)std::min<int64_t>(chunk * nr, 1 << 30), h1, cf.len_ltf, &kps_tmp);
        if (hs_ok) splits_max = std::max(splits_max, hs_layer0_splits(c, (int)std::min<int64_t>(chunk * nr, 1 << 30), h1, cf.len_ltf, &kps_tmp));
        if (hs_ok) splits_max = std::max(splits_max, l0_stream_splits(c, m, (int)std::min<int64_t>(chunk * nr, 1 << 30), h1, cf.len_ltf, &kps_tmp));
        const size_t need = ((size_t)nr * h1 * 4 * (splits_max > 1 ? splits_max + 1 : 1) + hid_pkt) * (size_t)chunk;
        if ((need <= budget && chunk <= max_rows) || chunk == 1) break;
        nchunks = std::max(nchunks + 1, (int64_t)((double)nchunks * (double)need / (double)budget));
    }
    {   // the last chunk can be shorter and may want a different split factor
        int kps_tmp;
        const int64_t tail = npkt - (nchunks - 1) * chunk;
        if (tail > 0 && tail != chunk) {
            splits_max = std::max(splits_max, choose_splits((int)(tail * nr), h1, cf.len_ltf, &kps_tmp));
            if (hs_ok) splits_max = std::max(splits_max, hs_layer0_splits(c, (int)(tail * nr), h1, cf.len_ltf, &kps_tmp));
            if (hs_ok) splits_max = std::max(splits_max, l0_stream_splits(c, m, (int)(tail * nr), h1, cf.len_ltf, &kps_tmp));
        }
    }
    const size_t slab_floats = (size_t)chunk * nr * h1;
    // (+ 16 rows per activation buffer: the blocked hs layout addresses whole 16-row blocks)
    const size_t hid_pad = (size_t)16 * maxh * 4;
    const size_t per_chunk = slab_floats * 4 * (splits_max > 1 ? splits_max + 1 : 1) + hid_pkt * (size_t)chunk + 2 * hid_pad;
    int rc = ensure_bytes(c, &c->ws, &c->ws_bytes, per_chunk);
    if (rc) return rc;

    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        const int M1 = (int)(np * nr);
        const int M2 = (int)(np * nr * nt);
        float* slabs = reinterpret_cast<float*>(c->ws);
        float* l0sum = slabs + slab_floats * splits_max;              // unused when splits_max == 1
        float* hbuf[2];
        hbuf[0] = slabs + slab_floats * (splits_max > 1 ? splits_max + 1 : 1);
        hbuf[1] = hbuf[0] + (size_t)chunk * nr * nt * maxh + hid_pad / 4;

        // layer 0, LTF part: L0[M1][h1] = ltf[M1][len_ltf] * W0[0:len_ltf, :]
        const float* l0 = nullptr;
        if (M1 <= 8 && m.W0rm && !c->force_pair_tile) {
            // a handful of preambles: stream W0 once (HBM-bound) instead of running a GEMM
            const int S = (cf.len_ltf + 4 * SK_KS - 1) / (4 * SK_KS);
            rc = ensure_bytes(c, &c->l0skinny, &c->l0skinny_bytes, (size_t)(S + 1) * 8 * h1 * sizeof(float));
            if (rc) return rc;
            float* sl = reinterpret_cast<float*>(c->l0skinny);
            float* sum = sl + (size_t)S * M1 * h1;
            {
                ProfScope ps(c, K_LAYER0_LTF, 2.0 * M1 * h1 * cf.len_ltf, 4.0 * ((double)cf.len_ltf * h1 + (double)M1 * cf.len_ltf + (double)S * M1 * h1));
                hipLaunchKernelGGL((layer0_skinny_kernel<8>), dim3((unsigned)S, (unsigned)((h1 + 255) / 256)), dim3(256), 0, c->stream,
                                   d_ltf + (size_t)p0 * nr * cf.len_ltf, cf.len_ltf, M1, m.W0rm, h1, cf.len_ltf, sl);
                HIP_TRY(c, hipGetLastError());
            }
            {
                const size_t n4 = (size_t)M1 * h1 / 4;
                ProfScope ps(c, K_SPLITK_REDUCE, (double)(S - 1) * M1 * h1, 4.0 * (S + 1) * M1 * h1);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(64), 0, c->stream, sl, sum, n4, S);
                HIP_TRY(c, hipGetLastError());
            }
            l0 = sum;
        }
        int kps = 0;
        // 9 ... 256 preambles: the weight-streaming kernel of the split engine (l0_hs_stream.hip.h), k ranges summed below
        const int stream_splits = (hs_ok && !l0) ? l0_stream_splits(c, m, M1, h1, cf.len_ltf, &kps) : 0;
        if (stream_splits) {
            rc = l0_stream_launch(c, m, d_ltf + (size_t)p0 * nr * cf.len_ltf, cf.len_ltf, M1, h1, cf.len_ltf, kps, stream_splits, slabs);
            if (rc) return rc;
            l0 = slabs;
        }
        const int hs_splits = (hs_ok && !l0) ? hs_layer0_splits(c, M1, h1, cf.len_ltf, &kps) : 0;
        if (hs_splits) {
            rc = hs_launch_layer0(c, m, d_ltf + (size_t)p0 * nr * cf.len_ltf, cf.len_ltf, M1, h1, cf.len_ltf, kps, hs_splits, slabs);
            if (rc) return rc;
            l0 = slabs;
        }
        const int splits = stream_splits ? stream_splits : (hs_splits ? hs_splits : (l0 ? 1 : choose_splits(M1, h1, cf.len_ltf, &kps)));
        GemmArgs g{};
        g.A = d_ltf + (size_t)p0 * nr * cf.len_ltf;
        g.lda = cf.len_ltf;
        g.Bt = m.layers[0].Wt;
        g.ldb = m.layers[0].ldw;
        g.C = slabs;
        g.ldc = h1;
        g.M = M1; g.N = h1; g.K = cf.len_ltf;
        g.k_per_split = kps;
        if (!l0) {
            rc = launch_gemm<EPI_RAW>(c, K_LAYER0_LTF, g, splits);
            if (rc) return rc;
            l0 = slabs;
        }
        if (splits > 1) {
            const size_t n4 = (size_t)M1 * h1 / 4;
            ProfScope ps(c, K_SPLITK_REDUCE, (double)(splits - 1) * M1 * h1, 4.0 * (splits + 1) * M1 * h1);
            const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 4096);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, c->stream, slabs, l0sum, n4, splits);
            HIP_TRY(c, hipGetLastError());
            l0 = l0sum;
        }

        // first per-pair layer: h1 generated in the prologue from L0 + T
        float* out_chunk = d_out + (size_t)p0 * nr * nt * cf.n_out;
        if (hs_ok && (hs_tail_wanted(c, M2, m.layers[1].out) || band_split_static_ok(c, m))) {
            rc = hs_tail(c, m, l0, M2, hbuf[0], hbuf[1], out_chunk);
            if (rc) return rc;
            continue;
        }
        GemmArgs p{};
        p.A = l0; p.lda = h1;
        // BatchNormalization shifts live in the next layer's bias (Layer::bias_hs, csi_load_weights): the
        // activations keep the exact zeros of the relu, on which the matrix pipe draws less power
        const bool fold = m.layers[1].bias_hs != nullptr;
        p.T = m.T; p.s0 = m.layers[0].scale; p.t0 = fold ? c->hs_zero : m.layers[0].shift; p.nt = nt;
        p.M = M2; p.K = h1;
        const Layer& l1 = m.layers[1];
        p.Bt = l1.Wt; p.ldb = l1.ldw; p.N = l1.out;
        p.bias = fold ? l1.bias_hs : l1.bias; p.scale = l1.scale; p.shift = fold ? c->hs_zero : l1.shift;
        p.k_per_split = ((h1 + G_BK - 1) / G_BK) * G_BK;
        if (nh == 1) {
            p.C = out_chunk; p.ldc = cf.n_out;
            rc = launch_pair<EPI_BIAS>(c, K_REGRESSOR, p);
            if (rc) return rc;
            continue;
        }
        p.C = hbuf[0]; p.ldc = l1.out;
        rc = launch_pair<EPI_BIAS_RELU_AFFINE>(c, K_PAIR_DENSE, p);
        if (rc) return rc;
        int cur = 0;
        for (int li = 2; li <= nh; ++li) {
            const Layer& l = m.layers[li];
            GemmArgs q{};
            q.A = hbuf[cur]; q.lda = l.in;
            q.Bt = l.Wt; q.ldb = l.ldw;
            q.M = M2; q.N = l.out; q.K = l.in;
            q.bias = fold ? l.bias_hs : l.bias; q.scale = l.scale; q.shift = fold ? c->hs_zero : l.shift;
            q.k_per_split = ((l.in + G_BK - 1) / G_BK) * G_BK;
            if (li == nh) {
                q.C = out_chunk; q.ldc = cf.n_out;
                rc = launch_gemm<EPI_BIAS>(c, K_REGRESSOR, q, 1);
            } else {
                q.C = hbuf[cur ^ 1]; q.ldc = l.out;
                rc = launch_gemm<EPI_BIAS_RELU_AFFINE>(c, K_DENSE_HIDDEN, q, 1);
                cur ^= 1;
            }
            if (rc) return rc;
        }
    }
    return CSI_OK;
}

}  // namespace
