// csi_dnn_small.hpp - host side of the one-packet regime (small_call.hip.h): both component models of a small call (at most
// "small_rows" = 1024 pair rows and 64 rx preambles) in 1 + n_hidden launches on the context's stream - layer 0 as one
// weight-streaming kernel (up to 8 preambles) or on fp32-MFMA tiles, every layer behind it as 16 x 16 / 32 x 32 MFMA tiles over
// the whole K - instead of six launches per model on two streams.  Same arithmetic class as the fp32 MFMA
// kernels of gemm_f32.hip.h (exact fp32 products, fp32 accumulation); the reference call this serves is the literal per-packet
// predict of massiveMIMO_CSI_prediction_DNN.py:339-346.
#pragma once
#include "csi_context.hpp"
#include "small_call.hip.h"

namespace {

// which calls take it: fp32 contexts with a pilot input, both models loaded, at most 64 preambles and "small_rows" pair rows;
// "small_fused" = 0 restores the general kernels (A/B runs, tests)
bool small_call_ok(csi_ctx* c, int64_t npkt) {
    const csi_config& cf = c->cfg;
    if (!c->small_fused || cf.dtype != CSI_DTYPE_F32 || cf.nt < 1 || c->force_pair_tile) return false;
    if (c->f32_engine == 1) return false;        // "f32_engine" = 1 asks for the split-f16 engine wherever the shapes allow
    // up to 8 preambles layer 0 is the weight-streaming kernel; up to "small_rows" pair rows (default 1024: 8 packets of the shipped
    // shape) it is the tile kernel with the EPI_H1 epilogue - beyond that the general kernels (from 24 packets the split-f16 engine) win
    if (npkt * cf.nr * cf.nt > c->small_rows || npkt * cf.nr > 64) return false;
    for (int d = 0; d < 2; ++d)
        if (!c->model[d].loaded || !c->model[d].table_ok || !c->model[d].layers[0].Wt) return false;
    // where the column-split band kernel serves the model the general path (weight-streaming layer 0 + that kernel) wins from 3 packets of
    // the shipped shape on (102 us against 121 here; 2 packets: 111 against 61 - profiles/r05_band_split_probe.txt): this path keeps the
    // calls of at most "small_rows_band" rows
    // (calls of at most 8 preambles - layer 0 on the weight-streaming gemv - stay here whatever their rows: Nt = 64, 2 packets = 512 rows 99 us against 128)
    if (npkt * cf.nr * cf.nt > c->small_rows_band && npkt * cf.nr > SC_MAX_ROWS0 && c->f32_engine != 0 && hs_static_ok(c, c->model[0]) && hs_static_ok(c, c->model[1]) &&
        band_split_static_ok(c, c->model[0]) && band_split_static_ok(c, c->model[1]))
        return false;
    return true;
}

// (defined in csi_mamimo.hip behind the LS plan) the LS kernel of this context is the Walsh-Hadamard one in its default shape: its LDS bytes
bool ls_default_fwht2(const csi_ctx* c, size_t* lds_bytes);
LsArgs ls_args(const csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, float* d_h_re, float* d_h_im);

// csi_estimate_device: may the LS estimate of this call ride in the layer-0 launch?  The weight-streaming layer 0 (at most 8 preambles), the
// Walsh-Hadamard LS kernel in its default one-thread-per-bin shape on the Sylvester order itself (Nt = 16 / 32 / 64)
bool small_ls_fusable(csi_ctx* c, int64_t npkt) {
    const csi_config& cf = c->cfg;
    const int nt = cf.nt;
    if (!c->small_ls_fused || !(nt == 16 || nt == 32 || nt == 64) || cf.dtype != CSI_DTYPE_F32) return false;
    size_t lds = 0;
    if (!ls_default_fwht2(c, &lds)) return false;
    // ONLY beside the weight-streaming layer 0 of the one-packet path (plain v_fma_f32, no matrix instructions).  The same arrangement beside the
    // mid-size path's l0_hs_stream_kernel (f16 MFMAs) was built and measured this round: LS results wrong and different run to run at 3 and 8
    // packets (profiles/r06_small_calls.txt) - the signature round 4 saw with LS waves beside another workgroup's MFMAs on one SIMD
    // (profiles/r04_ls_ringb_variants.txt); and it bought nothing there (97.3 us either way: the second stream's chain is the longer one)
    return small_call_ok(c, npkt) && npkt * cf.nr <= SC_MAX_ROWS0;
}

int predict_small(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_out_re, float* d_out_im) {
    const csi_config& cf = c->cfg;
    const int nt = cf.nt, h1 = cf.hidden[0], nh = cf.n_hidden;
    const int M1 = (int)(npkt * cf.nr), M2 = M1 * nt;
    int maxh = 0;
    for (int i = 1; i < nh; ++i) maxh = std::max(maxh, cf.hidden[i]);
    // scratch: the per-pair layer's input rows of both models, then two ping-pong activation buffers per model
    const size_t h1_floats = (size_t)M2 * h1, act_floats = (size_t)M2 * maxh;
    int rc = ensure_bytes(c, &c->small_ws, &c->small_ws_bytes, (2 * h1_floats + 4 * act_floats) * sizeof(float));
    if (rc) return rc;
    float* h1buf = reinterpret_cast<float*>(c->small_ws);
    float* act = h1buf + 2 * h1_floats;         // [buffer][model][M2][maxh]
    Model* md[2] = {&c->model[0], &c->model[1]};
    if (M1 > SC_MAX_ROWS0) {
        // 9 ... 64 preambles: layer 0 on the 16 x 16 tiles, the per-pair layer's input rows written by its epilogue (EPI_H1)
        SmallGemmArgs g{};
        g.A[0] = d_ltf_re; g.A[1] = d_ltf_im;
        for (int d = 0; d < 2; ++d) {
            const Model& m = *md[d];
            const bool fold = m.layers[1].bias_hs != nullptr;
            g.Bt[d] = m.layers[0].Wt; g.T[d] = m.T; g.s0[d] = m.layers[0].scale; g.t0[d] = fold ? c->hs_zero : m.layers[0].shift;
            g.C[d] = h1buf + d * h1_floats;
        }
        g.M = M1; g.N = h1; g.K = cf.len_ltf; g.lda = cf.len_ltf; g.ldb = md[0]->layers[0].ldw; g.ldc = h1; g.nt = nt;
        const int tiles_n = (h1 + 15) / 16, tiles_m = (M1 + 15) / 16;
        int rg = 4;
        while (rg > 1 && (long)tiles_n * ((tiles_m + rg - 1) / rg) * 2 < 200) rg >>= 1;
        const dim3 grid((unsigned)tiles_n, (unsigned)((tiles_m + rg - 1) / rg), 2);
        ProfScope ps(c, K_LAYER0_LTF, 2.0 * 2.0 * M1 * h1 * cf.len_ltf, 2.0 * 4.0 * ((double)cf.len_ltf * h1 + (double)M1 * cf.len_ltf + (double)(nt + M2) * h1));
        if (rg == 4) hipLaunchKernelGGL((small_tile_gemm_kernel<EPI_H1, 4, 4>), grid, dim3(1024), 0, c->stream, g);
        else if (rg == 2) hipLaunchKernelGGL((small_tile_gemm_kernel<EPI_H1, 2, 4>), grid, dim3(1024), 0, c->stream, g);
        else hipLaunchKernelGGL((small_tile_gemm_kernel<EPI_H1, 1, 4>), grid, dim3(1024), 0, c->stream, g);
        HIP_TRY(c, hipGetLastError());
    } else {
        SmallL0Args a{};
        a.x[0] = d_ltf_re; a.x[1] = d_ltf_im;
        for (int d = 0; d < 2; ++d) {
            const Model& m = *md[d];
            // BatchNormalization shifts live in the next layer's bias where csi_load_weights folded them (Layer::bias_hs)
            const bool fold = m.layers[1].bias_hs != nullptr;
            a.Wt[d] = m.layers[0].Wt; a.T[d] = m.T; a.s0[d] = m.layers[0].scale; a.t0[d] = fold ? c->hs_zero : m.layers[0].shift;
            a.h1out[d] = h1buf + d * h1_floats;
        }
        a.M = M1; a.K = cf.len_ltf; a.lda = cf.len_ltf; a.ldw = md[0]->layers[0].ldw; a.h1 = h1; a.nt = nt;
        ProfScope ps(c, K_LAYER0_LTF, 2.0 * 2.0 * M1 * h1 * cf.len_ltf, 2.0 * 4.0 * ((double)cf.len_ltf * h1 + (double)M1 * cf.len_ltf + (double)(nt + M2) * h1));
        // 4 columns per workgroup, 2 k steps of 1024 in flight: every shape tried (4 / 8 columns, 2 ... 5 steps) lands at 16.2-17.0 us
        // for the 84 MB of the shipped model = 5.2 TB/s - the memory system's rate, not the kernel's (profiles/r05_small_call_trace.txt)
        const dim3 grid((unsigned)((h1 + 3) / 4), 2);
        if (c->small_ls_h_re && c->small_ls_h_im) {
            // csi_estimate_device: the LS estimate of the call inside this launch (small_l0_ls_kernel), one boundary and 8.9 us less per call
            LsArgs la = ls_args(c, d_ltf_re, d_ltf_im, c->small_ls_h_re, c->small_ls_h_im);
            size_t ls_lds = 0;
            if (!ls_default_fwht2(c, &ls_lds)) return fail(c, CSI_ERR_NOT_READY, "one-packet call: the LS kernel changed under the call");
            int nblk = M1, ls_blocks = M1;
            const dim3 fgrid(grid.x + (unsigned)ls_blocks, 2);
            const void* fn = nullptr;
#define SC_LS(NTV) (M1 > 4 ? (const void*)small_l0_ls_kernel<NTV, 8> : (const void*)small_l0_ls_kernel<NTV, 4>)
            fn = nt == 16 ? SC_LS(16) : (nt == 32 ? SC_LS(32) : SC_LS(64));
#undef SC_LS
            static thread_local const void* attr_fn[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            bool seen = false;
            for (const void* f : attr_fn) seen = seen || f == fn;
            if (!seen) {
                HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ls_lds));
                for (auto& f : attr_fn) if (!f) { f = fn; break; }
            }
            void* kargs[] = {(void*)&a, (void*)&la, (void*)&nblk, (void*)&ls_blocks};
            HIP_TRY(c, hipLaunchKernel(fn, fgrid, dim3(256), kargs, ls_lds, c->stream));
            c->small_ls_h_re = c->small_ls_h_im = nullptr;      // consumed
            ++c->small_ls_launches;
        } else if (M1 > 4) hipLaunchKernelGGL((small_l0_gemv_kernel<8, 4, 2>), grid, dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((small_l0_gemv_kernel<4, 4, 2>), grid, dim3(256), 0, c->stream, a);
        HIP_TRY(c, hipGetLastError());
    }
    float* out[2] = {d_out_re, d_out_im};
    int cur = 0;
    for (int li = 1; li <= nh; ++li) {
        SmallGemmArgs g{};
        const bool last = li == nh;
        for (int d = 0; d < 2; ++d) {
            const Model& m = *md[d];
            const Layer& l = m.layers[li];
            const bool fold = m.layers[1].bias_hs != nullptr;
            g.A[d] = li == 1 ? h1buf + d * h1_floats : act + ((size_t)cur * 2 + d) * act_floats;
            g.Bt[d] = l.Wt;
            g.bias[d] = fold ? l.bias_hs : l.bias;
            g.scale[d] = l.scale;
            g.shift[d] = fold ? c->hs_zero : l.shift;
            g.C[d] = last ? out[d] : act + ((size_t)(cur ^ 1) * 2 + d) * act_floats;
        }
        const Layer& l = md[0]->layers[li];
        g.M = M2; g.N = l.out; g.K = l.in; g.ldb = l.ldw;
        g.lda = li == 1 ? h1 : l.in;
        g.ldc = last ? cf.n_out : l.out;
        const bool force16 = c->debug_small_tile16 != 0;      // A/B runs: CSI_DEBUG_HOOKS=1 CSI_SMALL_TILE16=1 (read once, at csi_create) keeps every layer on the 16 x 16 tiles
        ProfScope ps(c, last ? K_REGRESSOR : (li == 1 ? K_PAIR_DENSE : K_DENSE_HIDDEN), 2.0 * 2.0 * (double)M2 * g.N * g.K,
                     2.0 * 4.0 * ((double)g.N * g.K + (double)M2 * g.N + (double)M2 * g.K));
        // 32 x 32 tiles (half the operand traffic per flop) where they still make ~256 workgroups, 16 x 16 tiles where the layer is
        // too narrow for that (the regressor's 234 columns); row tiles per workgroup: as few as leave that many workgroups - the k
        // split inside the workgroup grows as RG shrinks
        const int tn32 = (g.N + 31) / 32, tm32 = (M2 + 31) / 32;
        const bool t32 = (long)tn32 * tm32 * 2 >= 200 && !force16;
        const int tsz = t32 ? 32 : 16;
        const int tiles_n = (g.N + tsz - 1) / tsz, tiles_m = (M2 + tsz - 1) / tsz;
        int rg = 4;
        while (rg > 1 && (long)tiles_n * ((tiles_m + rg - 1) / rg) * 2 < 200) rg >>= 1;
        const dim3 grid((unsigned)tiles_n, (unsigned)((tiles_m + rg - 1) / rg), 2);
        // groups of 32 k a wave owns -> groups whose loads it requests at once
        const int per_wave = (((g.K + 31) >> 5) + (16 / rg) - 1) / (16 / rg);
        const int ub = t32 ? (per_wave >= 2 ? 2 : 1) : (per_wave >= 4 ? 4 : (per_wave >= 2 ? 2 : 1));
#define SC_K(KERNEL, EPIV, RGV, UBV) hipLaunchKernelGGL((KERNEL<EPIV, RGV, UBV>), grid, dim3(1024), 0, c->stream, g)
#define SC_LAUNCH3(EPIV, RGV)                                                                        \
    do {                                                                                             \
        if (t32) { if (ub == 2) SC_K(small_tile32_gemm_kernel, EPIV, RGV, 2); else SC_K(small_tile32_gemm_kernel, EPIV, RGV, 1); }        \
        else if (ub == 4) SC_K(small_tile_gemm_kernel, EPIV, RGV, 4);                                \
        else if (ub == 2) SC_K(small_tile_gemm_kernel, EPIV, RGV, 2);                                \
        else SC_K(small_tile_gemm_kernel, EPIV, RGV, 1);                                             \
    } while (0)
#define SC_LAUNCH(EPIV)                                                                              \
    do {                                                                                             \
        if (rg == 4) SC_LAUNCH3(EPIV, 4);                                                            \
        else if (rg == 2) SC_LAUNCH3(EPIV, 2);                                                       \
        else SC_LAUNCH3(EPIV, 1);                                                                    \
    } while (0)
        if (last) SC_LAUNCH(EPI_BIAS);
        else SC_LAUNCH(EPI_BIAS_RELU_AFFINE);
#undef SC_LAUNCH
#undef SC_LAUNCH3
#undef SC_K
        HIP_TRY(c, hipGetLastError());
        cur ^= 1;
    }
    ++c->small_calls;
    return CSI_OK;
}

}  // namespace
