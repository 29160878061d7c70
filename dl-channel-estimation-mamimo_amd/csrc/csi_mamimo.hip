// csi_mamimo.hip - C-ABI (include/csi_mamimo.h) and host-side orchestration of the MI355X
// channel-estimation hot path.  gfx950 only; built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC csi_mamimo.hip -o libcsi_mamimo.so
//
// What runs where (reference call sites in include/csi_mamimo.h):
//   csi_predict*        layer 0 once per (packet, rx)  -> gemm_f32_kernel<EPI_RAW> (optional split-K)
//                       (+ splitk_reduce_kernel)          then per pair (packet, rx, tx):
//                       hidden 1.. and regressor       -> pair_gemm_f32_kernel / gemm_f32_kernel
//   csi_predict_samples literal un-shared network      -> gemm_f32_kernel
//   csi_ls_estimate*    FFT + despread                 -> ls_estimate_kernel
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/csi_mamimo.h"
#include "gemm_f32.hip.h"
#include "gemm_bf16.hip.h"
#include "ls_estimate.hip.h"
#include "lmmse.hip.h"

using namespace csi;

namespace {

enum KernelId {
    K_LAYER0_LTF = 0,    // layer 0, LTF part, once per (packet, rx)
    K_SPLITK_REDUCE,     // deterministic split-K combine of layer 0
    K_PAIR_DENSE,        // first per-pair layer, h1 generated in the prologue  (dominant)
    K_DENSE_HIDDEN,      // further hidden layers
    K_REGRESSOR,         // fc_regressor
    K_LS_ESTIMATE,       // FFT + despread
    K_NAIVE_DENSE0,      // un-shared layer 0 of csi_predict_samples
    K_SYNTH_WHITE,
    K_PILOT_TABLE,
    K_CAST_BF16,         // fp32 -> bf16 of the preambles (bf16 mode)
    K_PAIR_H1_BF16,      // materialise h1 in bf16 (bf16 mode)
    K_LMMSE,             // Levinson solve of the LMMSE smoother
    K_COUNT
};
const char* const kKernelNames[K_COUNT] = {
    "layer0_ltf_gemm", "splitk_reduce", "pair_dense_gemm", "dense_hidden_gemm", "regressor_gemm",
    "ls_estimate", "naive_dense0_gemm", "synth_white", "pilot_table", "cast_bf16", "pair_h1_bf16", "lmmse_levinson"};

thread_local std::string g_create_error;

struct Layer {
    float* Wt = nullptr;      // [out][ldw] (K-major, ldw = in rounded up to 32, zero padded)   fp32 mode
    int ldw = 0;
    bf16_t* Wb = nullptr;     // [out][ldwb] bf16 (K-major, ldwb = in rounded up to 64)          bf16 mode
    int ldwb = 0;
    float* bias = nullptr;    // [out]
    float* scale = nullptr;   // [out]  BN: gamma * rsqrt(var + eps)   (1 without BN)
    float* shift = nullptr;   // [out]  BN: beta - mean * scale        (0 without BN)
    int in = 0, out = 0;
};

struct Model {
    std::vector<Layer> layers;   // n_hidden dense layers + regressor (last)
    float* W0p = nullptr;        // [nt][H1] pilot rows of fc_dense0.kernel, row-major
    float* W0rm = nullptr;       // [lenLTF][H1] LTF rows of fc_dense0.kernel as stored (skinny layer-0 kernel)
    float* T = nullptr;          // [nt][H1] pilot table incl. bias
    bool loaded = false;
    bool table_ok = false;
};

struct GraphEntry {           // one captured csi_predict_device call
    const void* in_re; const void* in_im; void* out_re; void* out_im;
    int64_t npkt;
    int seen;                 // eager runs with this key so far (capture happens on the 2nd call)
    hipGraphExec_t exec;
};

struct ProfSpan {
    int id;
    hipEvent_t beg, end;
};

}  // namespace

struct csi_ctx {
    csi_config cfg;
    int d_in = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Model model[2];
    float* P = nullptr;          // device [nt][nt]
    bool pilot_ok = false;
    // LS constants
    float* tw = nullptr;         // [2][256]
    int* bin_pos = nullptr;      // [234]
    float* denom = nullptr;      // [234]
    // activation workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // layer-0 slabs + sum of the one-packet (skinny) path
    char* l0skinny = nullptr;
    size_t l0skinny_bytes = 0;
    // split-K slabs of the small-batch path
    char* skbuf = nullptr;
    size_t skbuf_bytes = 0;
    // staging for host-buffer entry points
    char* stage = nullptr;
    size_t stage_bytes = 0;
    int xcd_order = -1;          // option "xcd_order": -1 auto, 0 linear tile order, 1 XCD super-tile order
    bool use_graph = false;
    std::vector<GraphEntry> graphs;
    int ls_fft_first_max = 32;   // FFT-first LS kernel up to this Nt (measured: faster at 32, slower at 64); debug knob CSI_LS_FFT_FIRST_MAX
    int force_pair_tile = 0;     // debug knob CSI_FORCE_PAIR_TILE=128|256: forces the row-tile height of every GEMM (tests)
    // profiling
    bool prof_on = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[K_COUNT] = {0};
    int64_t prof_launches[K_COUNT] = {0};
    double prof_flops[K_COUNT] = {0};
    double prof_bytes[K_COUNT] = {0};
};

namespace {

int fail(csi_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, CSI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

struct ProfScope {
    csi_ctx* c;
    bool on;
    ProfSpan sp;
    ProfScope(csi_ctx* ctx, int id, double flops, double bytes) : c(ctx), on(ctx->prof_on) {
        if (!on) return;
        sp.id = id;
        sp.beg = take();
        sp.end = take();
        c->prof_launches[id] += 1;
        c->prof_flops[id] += flops;
        c->prof_bytes[id] += bytes;
        hipEventRecord(sp.beg, c->stream);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(sp.end, c->stream);
        c->spans.push_back(sp);
    }
    hipEvent_t take() {
        if (!c->ev_pool.empty()) {
            hipEvent_t e = c->ev_pool.back();
            c->ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
};

int prof_collect(csi_ctx* c) {
    if (c->spans.empty()) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& sp : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.beg, sp.end) == hipSuccess) c->prof_ms[sp.id] += ms;
        c->ev_pool.push_back(sp.beg);
        c->ev_pool.push_back(sp.end);
    }
    c->spans.clear();
    return CSI_OK;
}

void drop_graphs(csi_ctx* c) {
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

int ensure_bytes(csi_ctx* c, char** buf, size_t* have, size_t need) {
    if (*have >= need) return CSI_OK;
    drop_graphs(c);           // captured launches point into the old buffer
    if (*buf) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    const size_t bytes = need + G_SLACK_FLOATS * sizeof(float);
    if (hipMalloc((void**)buf, bytes) != hipSuccess) {
        *buf = nullptr;
        return fail(c, CSI_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
    }
    HIP_TRY(c, hipMemsetAsync(*buf, 0, bytes, c->stream));     // never-written parts must be finite
    *have = need;
    return CSI_OK;
}

// Every device array a GEMM may read as its A side (or as a per-column vector) is followed by
// G_SLACK_FLOATS zeroed floats: the K tail of the last tile over-reads into finite memory.
int upload(csi_ctx* c, float** dst, const float* src, size_t n) {
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    const size_t bytes = (n + G_SLACK_FLOATS) * sizeof(float);
    if (hipMalloc((void**)dst, bytes) != hipSuccess)
        return fail(c, CSI_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
    HIP_TRY(c, hipMemset(*dst, 0, bytes));
    HIP_TRY(c, hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return CSI_OK;
}

void free_layer(Layer& l) {
    if (l.Wt) hipFree(l.Wt);
    if (l.Wb) hipFree(l.Wb);
    if (l.bias) hipFree(l.bias);
    if (l.scale) hipFree(l.scale);
    if (l.shift) hipFree(l.shift);
    l = Layer();
}

void free_model(Model& m) {
    for (auto& l : m.layers) free_layer(l);
    m.layers.clear();
    if (m.W0p) hipFree(m.W0p);
    if (m.W0rm) hipFree(m.W0rm);
    m.W0rm = nullptr;
    if (m.T) hipFree(m.T);
    m.W0p = m.T = nullptr;
    m.loaded = m.table_ok = false;
}

const csi_tensor* find_tensor(const csi_tensor* t, int n, const std::string& name) {
    for (int i = 0; i < n; ++i)
        if (t[i].name && name == t[i].name) return &t[i];
    return nullptr;
}

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
                       c->P, m.W0p, m.layers[0].bias, m.T, nt, h1);
    HIP_TRY(c, hipGetLastError());
    m.table_ok = true;
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
    const size_t budget = cf.workspace_bytes > 0 ? (size_t)cf.workspace_bytes : ((size_t)3 << 29);   // 1.5 GiB
    const int64_t max_rows = (int64_t)0x7fffffff / ((int64_t)nr * nt * 2);     // M2 must fit an int
    int64_t nchunks = 1, chunk = npkt;
    int splits_max = 1;
    for (;;) {
        chunk = (npkt + nchunks - 1) / nchunks;
        int kps_tmp;
        splits_max = choose_splits((int)std::min<int64_t>(chunk * nr, 1 << 30), h1, cf.len_ltf, &kps_tmp);
        const size_t need = ((size_t)nr * h1 * 4 * (splits_max > 1 ? splits_max + 1 : 1) + hid_pkt) * (size_t)chunk;
        if ((need <= budget && chunk <= max_rows) || chunk == 1) break;
        nchunks = std::max(nchunks + 1, (int64_t)((double)nchunks * (double)need / (double)budget));
    }
    {   // the last chunk can be shorter and may want a different split factor
        int kps_tmp;
        const int64_t tail = npkt - (nchunks - 1) * chunk;
        if (tail > 0 && tail != chunk)
            splits_max = std::max(splits_max, choose_splits((int)(tail * nr), h1, cf.len_ltf, &kps_tmp));
    }
    const size_t slab_floats = (size_t)chunk * nr * h1;
    const size_t per_chunk = slab_floats * 4 * (splits_max > 1 ? splits_max + 1 : 1) + hid_pkt * (size_t)chunk;
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
        hbuf[1] = hbuf[0] + (size_t)chunk * nr * nt * maxh;

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
        const int splits = l0 ? 1 : choose_splits(M1, h1, cf.len_ltf, &kps);
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
        GemmArgs p{};
        p.A = l0; p.lda = h1;
        p.T = m.T; p.s0 = m.layers[0].scale; p.t0 = m.layers[0].shift; p.nt = nt;
        p.M = M2; p.K = h1;
        const Layer& l1 = m.layers[1];
        p.Bt = l1.Wt; p.ldb = l1.ldw; p.N = l1.out;
        p.bias = l1.bias; p.scale = l1.scale; p.shift = l1.shift;
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
            q.bias = l.bias; q.scale = l.scale; q.shift = l.shift;
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

// ---------------------------------------------------------------- bf16 mode
template <int EPI, bool OUT_BF16>
int launch_gemm_bf16(csi_ctx* c, int kid, GemmBf16Args g, int splits) {
    if (g.M <= 0) return CSI_OK;
    if ((g.lda & 7) || (g.ldb % B_BK))
        return fail(c, CSI_ERR_INVALID_ARG, "bf16 gemm: lda must be a multiple of 8 and ldb of 64 (lda=%d ldb=%d)", g.lda, g.ldb);
    const double flops = 2.0 * (double)g.M * g.N * g.K;
    const double bytes = 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (OUT_BF16 ? 2.0 : 4.0) * (double)g.M * g.N * splits;
    ProfScope ps(c, kid, flops, bytes);
    // 256x256 tiles (8 waves) once they fill the 256 CUs, 128x128 tiles (4 waves, 2 per CU) below
    const long big_tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * splits;
    if (big_tiles >= 256) {
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

int cast_bf16(csi_ctx* c, const float* src, bf16_t* dst, size_t n) {
    ProfScope ps(c, K_CAST_BF16, 0.0, 6.0 * n);
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 8192);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, src, dst, n8);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// hidden layers 1.. and the regressor on a bf16 activation matrix hin [M][l.in]; writes d_out fp32
int bf16_tail(csi_ctx* c, Model& m, const bf16_t* hin, int M, bf16_t* hb0, bf16_t* hb1, float* d_out, int first_layer) {
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
        int rc;
        if (li == cf.n_hidden) {
            q.C = d_out; q.ldc = cf.n_out;
            rc = launch_gemm_bf16<EPI_BIAS, false>(c, K_REGRESSOR, q, 1);
        } else {
            q.C = hb[w]; q.ldc = l.out;
            rc = launch_gemm_bf16<EPI_BIAS_RELU_AFFINE, true>(c, li == 1 ? K_PAIR_DENSE : K_DENSE_HIDDEN, q, 1);
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
    const size_t per_pkt = (size_t)nr * cf.len_ltf * 2 + (size_t)nr * h1 * 4 + (size_t)nr * nt * h1 * 2 +
                           (size_t)nr * nt * maxh * 2 * (nh >= 3 ? 2 : (nh >= 2 ? 1 : 0));
    const size_t budget = cf.workspace_bytes > 0 ? (size_t)cf.workspace_bytes : ((size_t)3 << 29);
    int64_t cap = std::max<int64_t>(1, (int64_t)(budget / per_pkt));
    cap = std::min(cap, (int64_t)0x7fffffff / ((int64_t)nr * nt * 2));
    const int64_t nchunks = (npkt + cap - 1) / cap;
    const int64_t chunk = (npkt + nchunks - 1) / nchunks;
    int rc = ensure_bytes(c, &c->ws, &c->ws_bytes, per_pkt * (size_t)chunk + 1024);
    if (rc) return rc;
    char* base = c->ws;
    bf16_t* xb = reinterpret_cast<bf16_t*>(base);             base += (size_t)chunk * nr * cf.len_ltf * 2;
    float* l0 = reinterpret_cast<float*>(base);               base += (size_t)chunk * nr * h1 * 4;
    bf16_t* h1b = reinterpret_cast<bf16_t*>(base);            base += (size_t)chunk * nr * nt * h1 * 2;
    bf16_t* hb0 = reinterpret_cast<bf16_t*>(base);            base += (size_t)chunk * nr * nt * maxh * 2;
    bf16_t* hb1 = reinterpret_cast<bf16_t*>(base);
    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        const int M1 = (int)(np * nr), M2 = (int)(np * nr * nt);
        rc = cast_bf16(c, d_ltf + (size_t)p0 * nr * cf.len_ltf, xb, (size_t)M1 * cf.len_ltf);
        if (rc) return rc;
        GemmBf16Args g{};
        g.A = xb; g.lda = cf.len_ltf;
        g.Bt = m.layers[0].Wb; g.ldb = m.layers[0].ldwb;
        g.C = l0; g.ldc = h1;
        g.M = M1; g.N = h1; g.K = cf.len_ltf;
        g.k_per_split = cf.len_ltf;
        rc = launch_gemm_bf16<EPI_RAW, false>(c, K_LAYER0_LTF, g, 1);
        if (rc) return rc;
        {
            ProfScope ps(c, K_PAIR_H1_BF16, 3.0 * M2 * h1, 2.0 * M2 * h1 + 4.0 * M1 * h1);
            const size_t total = (size_t)M2 * (h1 / 8);
            const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 16384);
            hipLaunchKernelGGL(pair_h1_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, l0, 1, (size_t)0, m.T,
                               m.layers[0].scale, m.layers[0].shift, h1b, M2, nt, h1);
            HIP_TRY(c, hipGetLastError());
        }
        rc = bf16_tail(c, m, h1b, M2, hb0, hb1, d_out + (size_t)p0 * nr * nt * cf.n_out, 1);
        if (rc) return rc;
    }
    return CSI_OK;
}

int check_ready(csi_ctx* c, bool need_models, int model = -1) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "single-input context (nt=0): only csi_predict_samples is available");
    if (!c->pilot_ok) return fail(c, CSI_ERR_NOT_READY, "csi_set_pilot has not been called");
    if (need_models) {
        for (int d = 0; d < 2; ++d) {
            if (model >= 0 && d != model) continue;
            if (!c->model[d].loaded)
                return fail(c, CSI_ERR_NOT_READY, "weights of the %s model are not loaded", d ? "imag" : "real");
            if (!c->model[d].table_ok) {
                int rc = build_pilot_table(c, c->model[d]);
                if (rc) return rc;
            }
        }
    }
    return CSI_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

int csi_abi_version(void) { return CSI_ABI_VERSION; }

const char* csi_last_error(const csi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int csi_create(const csi_config* cfg, csi_ctx** out) {
    if (!cfg || !out) return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: null argument");
    *out = nullptr;
    if (cfg->nt < 0 || cfg->nr < 1 || cfg->n_out < 1 || cfg->n_hidden < 1 || cfg->n_hidden > CSI_MAX_HIDDEN)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: bad shape nt=%d nr=%d n_hidden=%d n_out=%d",
                    cfg->nt, cfg->nr, cfg->n_hidden, cfg->n_out);
    // nt == 0: single-input model without pilot input (DNN.py:180,234); csi_predict_samples only
    if (cfg->nt > 0 && cfg->len_ltf != LS_SYM * cfg->nt)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: len_ltf (%d) must be 320*nt (%d)", cfg->len_ltf,
                    LS_SYM * cfg->nt);
    if (cfg->nt == 0 && (cfg->len_ltf < 4 || cfg->len_ltf % 4))
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: single-input width %d must be a positive multiple of 4", cfg->len_ltf);
    if (cfg->nt % 4)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: nt must be a multiple of 4 (16-byte rows), got %d", cfg->nt);
    for (int i = 0; i < cfg->n_hidden; ++i)
        if (cfg->hidden[i] < 4 || cfg->hidden[i] % 4)
            return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: hidden[%d]=%d must be a positive multiple of 4", i,
                        cfg->hidden[i]);
    if (cfg->dtype != CSI_DTYPE_F32 && cfg->dtype != CSI_DTYPE_BF16)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: unknown dtype %d", cfg->dtype);
    if (cfg->dtype == CSI_DTYPE_BF16) {
        for (int i = 0; i < cfg->n_hidden; ++i)
            if (cfg->hidden[i] % 8)
                return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: bf16 needs hidden widths that are multiples of 8 (hidden[%d]=%d)", i, cfg->hidden[i]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CSI_ERR_NO_DEVICE, "csi_create: no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, CSI_ERR_HIP, "csi_create: hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CSI_ERR_NO_DEVICE, "csi_create: device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);

    csi_ctx* c = new csi_ctx();
    c->cfg = *cfg;
    if (c->cfg.bn_eps <= 0.f) c->cfg.bn_eps = 1e-3f;
    c->d_in = cfg->len_ltf + cfg->nt;
    if (const char* e = std::getenv("CSI_FORCE_PAIR_TILE")) c->force_pair_tile = std::atoi(e);
    if (const char* e = std::getenv("CSI_LS_FFT_FIRST_MAX")) c->ls_fft_first_max = std::min(64, std::max(0, std::atoi(e)));
    auto bail = [&](int code) {
        g_create_error = c->err;
        csi_destroy(c);
        return code;
    };
    if (hipSetDevice(cfg->device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(CSI_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->err = "hipStreamCreate failed";
        return bail(CSI_ERR_HIP);
    }
    // LS constants.  Twiddles in double on the host so the table is correctly rounded.
    std::vector<float> tw(2 * LS_FFT);
    for (int u = 0; u < LS_FFT; ++u) {
        const double ang = -2.0 * M_PI * u / LS_FFT;
        tw[u] = (float)std::cos(ang);
        tw[LS_FFT + u] = (float)std::sin(ang);
    }
    // VHT-LTF literal of helperMIMOChannelEstimate.m:16-23 and the data-bin list of
    // generate_maMIMO_LTF.m:98-102, in fftshift-ed (1-based MATLAB) bin order.
    static const int ltfL[26] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const int ltfR[26] = {1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const int midA[11] = {-1, -1, -1, 1, 1, -1, 1, -1, 1, 1, -1};
    static const int midB[9] = {1, -1, 1, -1, 0, 1, -1, -1, 1};
    std::vector<int> ltf;
    auto push = [&](const int* v, int n) { ltf.insert(ltf.end(), v, v + n); };
    auto seg = [&]() { push(ltfL, 26); ltf.push_back(1); push(ltfR, 26); };
    ltf.assign(7, 0);
    seg(); push(midA, 11); seg(); push(midB, 9); seg(); push(midA, 11); seg();
    ltf.insert(ltf.end(), 6, 0);
    if ((int)ltf.size() != LS_FFT) { c->err = "internal: LTF literal length"; return bail(CSI_ERR_INVALID_ARG); }
    static const int pilots[8] = {26, 54, 90, 118, 140, 168, 204, 232};
    std::vector<int> bin_pos;
    std::vector<float> denom;
    for (int k1 = 1; k1 <= LS_FFT; ++k1) {
        bool skip = (k1 <= 7) || (k1 == 129) || (k1 >= 251);
        for (int pk : pilots) skip = skip || (k1 == pk);
        if (skip) continue;
        bin_pos.push_back((k1 - 1 + LS_FFT / 2) % LS_FFT);      // undo fftshift: shifted index -> FFT bin
        denom.push_back((float)cfg->nt * (float)ltf[k1 - 1]);
    }
    if ((int)bin_pos.size() != LS_NDATA) { c->err = "internal: data-bin count"; return bail(CSI_ERR_INVALID_ARG); }
    if (upload(c, &c->tw, tw.data(), tw.size())) return bail(CSI_ERR_HIP);
    if (upload(c, &c->denom, denom.data(), denom.size())) return bail(CSI_ERR_HIP);
    if (hipMalloc((void**)&c->bin_pos, LS_NDATA * sizeof(int)) != hipSuccess ||
        hipMemcpy(c->bin_pos, bin_pos.data(), LS_NDATA * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        c->err = "bin table upload failed";
        return bail(CSI_ERR_HIP);
    }
    const size_t ls_lds = (size_t)(cfg->nt * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
    if (cfg->nt > 0) {
        const bool fft_first = cfg->nt <= c->ls_fft_first_max;
        const size_t bytes = fft_first ? ls_lds : (size_t)(LSD_ROWS * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
        const void* fn = !fft_first ? (const void*)ls_despread_first_kernel
                         : (cfg->nt <= 32 ? (const void*)ls_estimate_kernel<8> : (const void*)ls_estimate_kernel<16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            c->err = "hipFuncSetAttribute(LS kernel) failed";
            return bail(CSI_ERR_HIP);
        }
    }
    *out = c;
    return CSI_OK;
}

void csi_destroy(csi_ctx* c) {
    if (!c) return;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& sp : c->spans) { hipEventDestroy(sp.beg); hipEventDestroy(sp.end); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    drop_graphs(c);
    free_model(c->model[0]);
    free_model(c->model[1]);
    if (c->P) hipFree(c->P);
    if (c->tw) hipFree(c->tw);
    if (c->bin_pos) hipFree(c->bin_pos);
    if (c->denom) hipFree(c->denom);
    if (c->ws) hipFree(c->ws);
    if (c->stage) hipFree(c->stage);
    if (c->skbuf) hipFree(c->skbuf);
    if (c->l0skinny) hipFree(c->l0skinny);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int csi_load_weights(csi_ctx* c, int model, const csi_tensor* tensors, int n) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (model < 0 || model > 1 || !tensors || n <= 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const csi_config& cf = c->cfg;
    Model& m = c->model[model];
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    free_model(m);
    m.layers.resize(cf.n_hidden + 1);
    int fan_in = c->d_in;
    for (int li = 0; li <= cf.n_hidden; ++li) {
        const bool reg = li == cf.n_hidden;
        const std::string base = reg ? std::string("fc_regressor") : "fc_dense" + std::to_string(li);
        const int out = reg ? cf.n_out : cf.hidden[li];
        const csi_tensor* k = find_tensor(tensors, n, base + ".kernel");
        const csi_tensor* b = find_tensor(tensors, n, base + ".bias");
        if (!k || !b || !k->data || !b->data) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: missing %s.kernel/.bias", base.c_str());
        if (k->rows != fan_in || k->cols != out || b->rows * b->cols != out)
            return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: %s.kernel is [%lld,%lld], expected [%d,%d]", base.c_str(),
                        (long long)k->rows, (long long)k->cols, fan_in, out);
        Layer& L = m.layers[li];
        L.in = fan_in;
        L.out = out;
        // transpose [in][out] -> [out][ldw] (K-major, K zero-padded to a multiple of the k-tile) on
        // the host, blocked for cache friendliness
        L.ldw = (fan_in + G_BK - 1) / G_BK * G_BK;
        std::vector<float> wt((size_t)out * L.ldw, 0.f);
        const int TB = 32;
        for (int i0 = 0; i0 < fan_in; i0 += TB)
            for (int o0 = 0; o0 < out; o0 += TB)
                for (int i = i0; i < std::min(fan_in, i0 + TB); ++i)
                    for (int o = o0; o < std::min(out, o0 + TB); ++o) wt[(size_t)o * L.ldw + i] = k->data[(size_t)i * out + o];
        int rc = CSI_OK;
        const bool bf16 = cf.dtype == CSI_DTYPE_BF16;
        auto rne = [](float f) {                       // fp32 -> bf16, round to nearest even
            uint32_t u;
            std::memcpy(&u, &f, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        if (bf16) {
            L.ldwb = (fan_in + B_BK - 1) / B_BK * B_BK;
            std::vector<uint16_t> wb((size_t)out * L.ldwb, 0);
            for (int o = 0; o < out; ++o)
                for (int i = 0; i < fan_in; ++i) wb[(size_t)o * L.ldwb + i] = rne(wt[(size_t)o * L.ldw + i]);
            const size_t bytes = wb.size() * 2 + 256;
            if (hipMalloc((void**)&L.Wb, bytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
            HIP_TRY(c, hipMemset(L.Wb, 0, bytes));
            HIP_TRY(c, hipMemcpy(L.Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
        } else {
            rc = upload(c, &L.Wt, wt.data(), wt.size());
            if (rc) return rc;
        }
        rc = upload(c, &L.bias, b->data, out);
        if (rc) return rc;
        std::vector<float> sc(out, 1.f), sh(out, 0.f);
        if (!reg && cf.use_bn) {
            const std::string bn = "bn" + std::to_string(li);
            const csi_tensor* ga = find_tensor(tensors, n, bn + ".gamma");
            const csi_tensor* be = find_tensor(tensors, n, bn + ".beta");
            const csi_tensor* mu = find_tensor(tensors, n, bn + ".moving_mean");
            const csi_tensor* va = find_tensor(tensors, n, bn + ".moving_variance");
            if (!ga || !be || !mu || !va) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: missing %s.* (use_bn=1)", bn.c_str());
            for (const csi_tensor* t : {ga, be, mu, va})
                if (!t->data || t->rows * t->cols != out)
                    return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: %s.* must have %d elements", bn.c_str(), out);
            for (int o = 0; o < out; ++o) {
                // keras non-fused inference: inv = rsqrt(var + eps) * gamma; y = x*inv + (beta - mean*inv)
                const float inv = (1.0f / std::sqrt(va->data[o] + cf.bn_eps)) * ga->data[o];
                sc[o] = inv;
                sh[o] = be->data[o] - mu->data[o] * inv;
            }
        }
        if (!reg) {
            rc = upload(c, &L.scale, sc.data(), out);
            if (rc) return rc;
            rc = upload(c, &L.shift, sh.data(), out);
            if (rc) return rc;
        }
        if (li == 0 && cf.nt > 0) {
            // pilot rows of fc_dense0.kernel, [nt][h1] row-major as stored (bf16 mode: rounded like
            // every other weight, the table itself is evaluated in fp32)
            std::vector<float> w0p(k->data + (size_t)cf.len_ltf * out, k->data + (size_t)(cf.len_ltf + cf.nt) * out);
            if (bf16)
                for (float& v : w0p) { const uint32_t u = (uint32_t)rne(v) << 16; std::memcpy(&v, &u, 4); }
            rc = upload(c, &m.W0p, w0p.data(), w0p.size());
            if (rc) return rc;
            if (!bf16) {
                rc = upload(c, &m.W0rm, k->data, (size_t)cf.len_ltf * out);
                if (rc) return rc;
            }
        }
        fan_in = out;
    }
    m.loaded = true;
    m.table_ok = false;
    return build_pilot_table(c, m);
}

int csi_set_pilot(csi_ctx* c, const float* P) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!P || c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_set_pilot: null P or single-input context");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    int rc = upload(c, &c->P, P, (size_t)c->cfg.nt * c->cfg.nt);
    if (rc) return rc;
    c->pilot_ok = true;
    for (int d = 0; d < 2; ++d) {
        c->model[d].table_ok = false;
        rc = build_pilot_table(c, c->model[d]);
        if (rc) return rc;
    }
    return CSI_OK;
}

int csi_predict_device(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_out_re,
                       float* d_out_im) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!d_ltf_re || !d_ltf_im || !d_out_re || !d_out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_device: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto run = [&]() -> int {
        if (c->cfg.dtype == CSI_DTYPE_BF16) {
            int r = predict_plane_bf16(c, c->model[0], d_ltf_re, npkt, d_out_re);
            if (r) return r;
            return predict_plane_bf16(c, c->model[1], d_ltf_im, npkt, d_out_im);
        }
        int r = predict_plane(c, c->model[0], d_ltf_re, npkt, d_out_re);
        if (r) return r;
        return predict_plane(c, c->model[1], d_ltf_im, npkt, d_out_im);
    };
    if (!c->use_graph || c->prof_on) return run();

    // hipGraph replay: 1st call with a key runs eagerly (sizes every buffer), 2nd call captures,
    // later calls replay.  Any reallocation / weight / pilot change drops the cache.
    GraphEntry* ge = nullptr;
    for (auto& g : c->graphs)
        if (g.in_re == d_ltf_re && g.in_im == d_ltf_im && g.out_re == d_out_re && g.out_im == d_out_im && g.npkt == npkt) ge = &g;
    if (!ge) {
        if (c->graphs.size() >= 16) drop_graphs(c);
        c->graphs.push_back(GraphEntry{d_ltf_re, d_ltf_im, d_out_re, d_out_im, npkt, 0, nullptr});
        ge = &c->graphs.back();
    }
    if (ge->exec) {
        HIP_TRY(c, hipGraphLaunch(ge->exec, c->stream));
        return CSI_OK;
    }
    if (ge->seen++ == 0) return run();
    const GraphEntry key = *ge;               // run() may not reallocate now, but keep a copy anyway
    HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    rc = run();
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e_end != hipSuccess) return fail(c, CSI_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e_end));
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e_inst != hipSuccess) return fail(c, CSI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e_inst));
    for (auto& g : c->graphs)
        if (g.in_re == key.in_re && g.in_im == key.in_im && g.out_re == key.out_re && g.out_im == key.out_im && g.npkt == key.npkt) g.exec = exec;
    HIP_TRY(c, hipGraphLaunch(exec, c->stream));
    return CSI_OK;
}

int csi_ls_estimate_device(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_h_re,
                           float* d_h_im) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!d_ltf_re || !d_ltf_im || !d_h_re || !d_h_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_ls_estimate_device: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    const bool fft_first = cf.nt <= c->ls_fft_first_max;   // spectra of all Nt symbols in LDS, else despread first
    const size_t lds = (size_t)((fft_first ? cf.nt : LSD_ROWS) * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
    const int n_jc = (cf.nt + LSD_ROWS - 1) / LSD_ROWS;
    HIP_TRY(c, hipSetDevice(cf.device));
    const int64_t nblk = npkt * cf.nr;
    LsArgs a{};
    a.P = c->P; a.tw = c->tw; a.bin_pos = c->bin_pos; a.denom = c->denom;
    a.nt = cf.nt; a.len_ltf = cf.len_ltf;
    const int64_t max_grid = ((int64_t)1 << 30) / n_jc;      // also keeps nb inside an int
    for (int64_t b0 = 0; b0 < nblk; b0 += max_grid) {
        const int64_t nb = std::min(max_grid, nblk - b0);
        a.ltf_re = d_ltf_re + (size_t)b0 * cf.len_ltf;
        a.ltf_im = d_ltf_im + (size_t)b0 * cf.len_ltf;
        a.h_re = d_h_re + (size_t)b0 * cf.nt * LS_NDATA;
        a.h_im = d_h_im + (size_t)b0 * cf.nt * LS_NDATA;
        const double pairs = (double)nb * cf.nt;
        ProfScope ps(c, K_LS_ESTIMATE, pairs * (10240.0 + 8.0 * LS_NDATA * cf.nt), pairs * (2560.0 + 1872.0));
        if (fft_first) {
            // persistent grid: as many workgroups as the LDS lets reside (x256 CUs)
            const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / lds)));
            const unsigned grid = (unsigned)std::min<int64_t>(nb, (int64_t)256 * per_cu);
            if (cf.nt <= 32) hipLaunchKernelGGL((ls_estimate_kernel<8>), dim3(grid), dim3(LS_THREADS), lds, c->stream, a, (int)nb);
            else hipLaunchKernelGGL((ls_estimate_kernel<16>), dim3(grid), dim3(LS_THREADS), lds, c->stream, a, (int)nb);
        } else
            hipLaunchKernelGGL(ls_despread_first_kernel, dim3((unsigned)(nb * n_jc)), dim3(LS_THREADS), lds, c->stream, a, n_jc);
        HIP_TRY(c, hipGetLastError());
    }
    return CSI_OK;
}

int csi_lmmse_estimate_device(csi_ctx* c, const float* d_h_re, const float* d_h_im, int64_t npkt, const float* d_hvec, int L,
                              const float* d_snr_db, float* d_out_re, float* d_out_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "single-input context (nt=0): no LMMSE estimate");
    if (npkt < 0 || L < 1 || (npkt > 0 && (!d_h_re || !d_h_im || !d_hvec || !d_snr_db || !d_out_re || !d_out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_lmmse_estimate_device: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    const int n_jc = (cf.nt + LM_RHS - 1) / LM_RHS;
    const int64_t nblk = npkt * cf.nr;
    const int64_t max_items = ((int64_t)1 << 30) / n_jc / cf.nr * cf.nr;      // whole packets per launch
    for (int64_t b0 = 0; b0 < nblk; b0 += max_items) {
        const int64_t nb = std::min(max_items, nblk - b0);
        LmmseArgs a{};
        a.h_re = d_h_re + (size_t)b0 * cf.nt * LM_N;
        a.h_im = d_h_im + (size_t)b0 * cf.nt * LM_N;
        a.hvec = d_hvec + (size_t)(b0 / cf.nr) * L;
        a.snr_db = d_snr_db + b0;
        a.o_re = d_out_re + (size_t)b0 * cf.nt * LM_N;
        a.o_im = d_out_im + (size_t)b0 * cf.nt * LM_N;
        a.nt = cf.nt; a.nr = cf.nr; a.L = L;
        // per right-hand side: 2 * n^2 complex multiply-adds = 16 n^2 flop (fp64)
        ProfScope ps(c, K_LMMSE, (double)nb * (cf.nt + n_jc) * 16.0 * LM_N * LM_N, (double)nb * cf.nt * LM_N * 16.0);
        hipLaunchKernelGGL(lmmse_levinson_kernel, dim3((unsigned)(nb * n_jc)), dim3(LM_THREADS), 0, c->stream, a, n_jc);
        HIP_TRY(c, hipGetLastError());
    }
    return CSI_OK;
}

int csi_lmmse_estimate(csi_ctx* c, const float* h_re, const float* h_im, int64_t npkt, const float* hvec, int L,
                       const float* snr_db, float* out_re, float* out_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (npkt < 0 || L < 1 || (npkt > 0 && (!h_re || !h_im || !hvec || !snr_db || !out_re || !out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_lmmse_estimate: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    const size_t pkt_f = (size_t)cf.nr * cf.nt * LM_N;                     // floats per packet and plane
    int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / (int64_t)(4 * pkt_f * sizeof(float)));
    chunk = std::min(chunk, npkt);
    const size_t need = (4 * pkt_f + (size_t)L + cf.nr) * sizeof(float) * (size_t)chunk;
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, need);
    if (rc) return rc;
    float* d_re = reinterpret_cast<float*>(c->stage);
    float* d_im = d_re + pkt_f * chunk;
    float* d_ore = d_im + pkt_f * chunk;
    float* d_oim = d_ore + pkt_f * chunk;
    float* d_hv = d_oim + pkt_f * chunk;
    float* d_snr = d_hv + (size_t)L * chunk;
    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        HIP_TRY(c, hipMemcpyAsync(d_re, h_re + p0 * pkt_f, pkt_f * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_im, h_im + p0 * pkt_f, pkt_f * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_hv, hvec + p0 * L, (size_t)L * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_snr, snr_db + p0 * cf.nr, (size_t)cf.nr * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        rc = csi_lmmse_estimate_device(c, d_re, d_im, np, d_hv, L, d_snr, d_ore, d_oim);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(out_re + p0 * pkt_f, d_ore, pkt_f * np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out_im + p0 * pkt_f, d_oim, pkt_f * np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

int csi_set_option(csi_ctx* c, const char* name, int64_t value) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!name) return fail(c, CSI_ERR_INVALID_ARG, "csi_set_option: null name");
    const std::string n(name);
    if (n == "use_graph") {
        if (!value) drop_graphs(c);
        c->use_graph = value != 0;
    } else if (n == "force_tile") {
        if (value != 0 && value != 128 && value != 256) return fail(c, CSI_ERR_INVALID_ARG, "force_tile must be 0, 128 or 256");
        drop_graphs(c);
        c->force_pair_tile = (int)value;
    } else if (n == "xcd_order") {
        if (value < -1 || value > 1) return fail(c, CSI_ERR_INVALID_ARG, "xcd_order must be -1 (auto), 0 or 1");
        drop_graphs(c);
        c->xcd_order = (int)value;
    } else if (n == "ls_fft_first_max") {
        if (value < 0 || value > 64) return fail(c, CSI_ERR_INVALID_ARG, "ls_fft_first_max must be 0..64");
        c->ls_fft_first_max = (int)value;
        if (c->cfg.nt > 0) {
            const bool fft_first = c->cfg.nt <= c->ls_fft_first_max;
            const size_t bytes = (size_t)((fft_first ? c->cfg.nt : LSD_ROWS) * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
            const void* fn = !fft_first ? (const void*)ls_despread_first_kernel
                             : (c->cfg.nt <= 32 ? (const void*)ls_estimate_kernel<8> : (const void*)ls_estimate_kernel<16>);
            HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        }
    } else {
        return fail(c, CSI_ERR_INVALID_ARG, "csi_set_option: unknown option '%s'", name);
    }
    return CSI_OK;
}

int csi_synchronize(csi_ctx* c) {
    if (!c) return CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

// ---- host-buffer entry points: stage through device memory in packet chunks
static int host_packets(csi_ctx* c, const float* re, const float* im, int64_t npkt, float* o_re, float* o_im,
                        int n_out, bool ls) {
    const csi_config& cf = c->cfg;
    const size_t in_pkt = (size_t)cf.nr * cf.len_ltf * sizeof(float);
    const size_t out_pkt = (size_t)cf.nr * cf.nt * n_out * sizeof(float);
    int64_t chunk = std::max<int64_t>(1, ((int64_t)512 << 20) / (int64_t)(2 * (in_pkt + out_pkt)));
    chunk = std::min(chunk, npkt);
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, 2 * (in_pkt + out_pkt) * (size_t)chunk);
    if (rc) return rc;
    float* d_re = reinterpret_cast<float*>(c->stage);
    float* d_im = reinterpret_cast<float*>(c->stage + in_pkt * chunk);
    float* d_ore = reinterpret_cast<float*>(c->stage + 2 * in_pkt * chunk);
    float* d_oim = reinterpret_cast<float*>(c->stage + 2 * in_pkt * chunk + out_pkt * chunk);
    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        HIP_TRY(c, hipMemcpyAsync(d_re, re + (size_t)p0 * cf.nr * cf.len_ltf, in_pkt * np, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_im, im + (size_t)p0 * cf.nr * cf.len_ltf, in_pkt * np, hipMemcpyHostToDevice, c->stream));
        rc = ls ? csi_ls_estimate_device(c, d_re, d_im, np, d_ore, d_oim) : csi_predict_device(c, d_re, d_im, np, d_ore, d_oim);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(o_re + (size_t)p0 * cf.nr * cf.nt * n_out, d_ore, out_pkt * np, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(o_im + (size_t)p0 * cf.nr * cf.nt * n_out, d_oim, out_pkt * np, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

int csi_predict(csi_ctx* c, const float* ltf_re, const float* ltf_im, int64_t npkt, float* out_re, float* out_im) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_re || !ltf_im || !out_re || !out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_predict: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return host_packets(c, ltf_re, ltf_im, npkt, out_re, out_im, c->cfg.n_out, false);
}

int csi_ls_estimate(csi_ctx* c, const float* ltf_re, const float* ltf_im, int64_t npkt, float* h_re, float* h_im) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_re || !ltf_im || !h_re || !h_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_ls_estimate: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return host_packets(c, ltf_re, ltf_im, npkt, h_re, h_im, LS_NDATA, true);
}

int csi_predict_samples(csi_ctx* c, int model, const float* x, int64_t B, float* y) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (model < 0 || model > 1) return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_samples: model must be 0 or 1");
    Model& m = c->model[model];
    if (!m.loaded) return fail(c, CSI_ERR_NOT_READY, "weights of the %s model are not loaded", model ? "imag" : "real");
    if (B < 0 || (B > 0 && (!x || !y))) return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_samples: bad argument");
    if (B == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    int maxh = 0;
    for (int i = 0; i < cf.n_hidden; ++i) maxh = std::max(maxh, cf.hidden[i]);
    if (cf.dtype == CSI_DTYPE_BF16) {
        const int ldx = (c->d_in + B_BK - 1) / B_BK * B_BK;
        const size_t per_row_b = (size_t)c->d_in * 4 + (size_t)ldx * 2 + 2 * (size_t)maxh * 2 + (size_t)cf.n_out * 4;
        int64_t chunk_b = std::min<int64_t>(B, std::max<int64_t>(1, ((int64_t)512 << 20) / (int64_t)per_row_b));
        int rcb = ensure_bytes(c, &c->stage, &c->stage_bytes, per_row_b * (size_t)chunk_b + 1024);
        if (rcb) return rcb;
        char* base = c->stage;
        float* d_xf = reinterpret_cast<float*>(base);      base += (size_t)chunk_b * c->d_in * 4;
        float* d_yf = reinterpret_cast<float*>(base);      base += (size_t)chunk_b * cf.n_out * 4;
        bf16_t* d_xb = reinterpret_cast<bf16_t*>(base);    base += (size_t)chunk_b * ldx * 2;
        bf16_t* hb0 = reinterpret_cast<bf16_t*>(base);     base += (size_t)chunk_b * maxh * 2;
        bf16_t* hb1 = reinterpret_cast<bf16_t*>(base);
        for (int64_t r0 = 0; r0 < B; r0 += chunk_b) {
            const int nb = (int)std::min(chunk_b, B - r0);
            HIP_TRY(c, hipMemcpyAsync(d_xf, x + (size_t)r0 * c->d_in, (size_t)nb * c->d_in * 4, hipMemcpyHostToDevice, c->stream));
            {
                ProfScope ps(c, K_CAST_BF16, 0.0, 6.0 * nb * c->d_in);
                const unsigned blocks = (unsigned)std::min<size_t>(((size_t)nb * ldx + 255) / 256, 8192);
                hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, d_xf, d_xb, nb, c->d_in, ldx);
                HIP_TRY(c, hipGetLastError());
            }
            const Layer& l0 = m.layers[0];
            GemmBf16Args q{};
            q.A = d_xb; q.lda = ldx;
            q.Bt = l0.Wb; q.ldb = l0.ldwb;
            q.C = hb0; q.ldc = l0.out;
            q.M = nb; q.N = l0.out; q.K = c->d_in;
            q.bias = l0.bias; q.scale = l0.scale; q.shift = l0.shift;
            q.k_per_split = l0.ldwb;
            rcb = launch_gemm_bf16<EPI_BIAS_RELU_AFFINE, true>(c, K_NAIVE_DENSE0, q, 1);
            if (rcb) return rcb;
            // hidden layers 1.. ping-pong starting from hb1, so layer 1 never overwrites its input
            rcb = bf16_tail(c, m, hb0, nb, hb1, hb0, d_yf, 1);
            if (rcb) return rcb;
            HIP_TRY(c, hipMemcpyAsync(y + (size_t)r0 * cf.n_out, d_yf, (size_t)nb * cf.n_out * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        return CSI_OK;
    }
    const size_t per_row = ((size_t)c->d_in + 2 * (size_t)maxh + cf.n_out) * sizeof(float);
    int64_t chunk = std::max<int64_t>(1, ((int64_t)512 << 20) / (int64_t)per_row);
    chunk = std::min(chunk, B);
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, per_row * (size_t)chunk);
    if (rc) return rc;
    float* d_x = reinterpret_cast<float*>(c->stage);
    float* hb[2];
    hb[0] = d_x + (size_t)chunk * c->d_in;
    hb[1] = hb[0] + (size_t)chunk * maxh;
    float* d_y = hb[1] + (size_t)chunk * maxh;
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t nb = std::min(chunk, B - r0);
        HIP_TRY(c, hipMemcpyAsync(d_x, x + (size_t)r0 * c->d_in, (size_t)nb * c->d_in * sizeof(float), hipMemcpyHostToDevice, c->stream));
        const float* cur = d_x;
        int cur_ld = c->d_in, w = 0;
        for (int li = 0; li <= cf.n_hidden; ++li) {
            const Layer& l = m.layers[li];
            GemmArgs q{};
            q.A = cur; q.lda = cur_ld;
            q.Bt = l.Wt; q.ldb = l.ldw;
            q.M = (int)nb; q.N = l.out; q.K = l.in;
            q.bias = l.bias; q.scale = l.scale; q.shift = l.shift;
            q.k_per_split = ((l.in + G_BK - 1) / G_BK) * G_BK;
            if (li == cf.n_hidden) {
                q.C = d_y; q.ldc = cf.n_out;
                rc = launch_gemm<EPI_BIAS>(c, K_REGRESSOR, q, 1);
            } else {
                q.C = hb[w]; q.ldc = l.out;
                rc = launch_gemm<EPI_BIAS_RELU_AFFINE>(c, li == 0 ? K_NAIVE_DENSE0 : K_DENSE_HIDDEN, q, 1);
                cur = hb[w];
                cur_ld = l.out;
                w ^= 1;
            }
            if (rc) return rc;
        }
        HIP_TRY(c, hipMemcpyAsync(y + (size_t)r0 * cf.n_out, d_y, (size_t)nb * cf.n_out * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

// ---- device memory plumbing
int csi_device_malloc(csi_ctx* c, void** dptr, int64_t bytes) {
    if (!c || !dptr || bytes < 0) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_device_malloc: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    *dptr = nullptr;
    if (bytes == 0) return CSI_OK;
    if (hipMalloc(dptr, (size_t)bytes) != hipSuccess) {
        *dptr = nullptr;
        return fail(c, CSI_ERR_NOMEM, "csi_device_malloc: %lld bytes failed", (long long)bytes);
    }
    return CSI_OK;
}

int csi_device_free(csi_ctx* c, void* dptr) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!dptr) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(dptr));
    return CSI_OK;
}

int csi_memcpy_h2d(csi_ctx* c, void* dst_dev, const void* src_host, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst_dev || !src_host))) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_memcpy_h2d: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

int csi_memcpy_d2h(csi_ctx* c, void* dst_host, const void* src_dev, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst_host || !src_dev))) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_memcpy_d2h: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

int csi_synth_white(csi_ctx* c, uint64_t seed, int64_t first_pkt, int64_t npkt, float* d_re, float* d_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (npkt < 0 || first_pkt < 0 || (npkt > 0 && (!d_re || !d_im))) return fail(c, CSI_ERR_INVALID_ARG, "csi_synth_white: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t per_pkt = (size_t)c->cfg.nr * c->cfg.len_ltf;
    const size_t n = per_pkt * (size_t)npkt;
    ProfScope ps(c, K_SYNTH_WHITE, 0.0, 8.0 * n);
    hipLaunchKernelGGL(synth_white_kernel, dim3(2048), dim3(256), 0, c->stream, seed, (uint64_t)first_pkt * per_pkt, n, d_re, d_im);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// ---- profiling
int csi_profile_enable(csi_ctx* c, int on) {
    if (!c) return CSI_ERR_INVALID_ARG;
    int rc = prof_collect(c);
    c->prof_on = on != 0;
    return rc;
}

int csi_profile_reset(csi_ctx* c) {
    if (!c) return CSI_ERR_INVALID_ARG;
    int rc = prof_collect(c);
    for (int i = 0; i < K_COUNT; ++i) {
        c->prof_ms[i] = 0;
        c->prof_launches[i] = 0;
        c->prof_flops[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return rc;
}

int csi_profile_num_kernels(void) { return K_COUNT; }

const char* csi_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }

int csi_profile_query(csi_ctx* c, int id, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (id < 0 || id >= K_COUNT) return fail(c, CSI_ERR_INVALID_ARG, "csi_profile_query: kernel id %d out of range", id);
    int rc = prof_collect(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof_ms[id];
    if (launches) *launches = c->prof_launches[id];
    if (flops) *flops = c->prof_flops[id];
    if (bytes) *bytes = c->prof_bytes[id];
    return CSI_OK;
}

}  // extern "C"
