// csi_train.hpp - host orchestration of the on-box fine-tuning step (csi_train_* of
// include/csi_mamimo.h; reference fit(): massiveMIMO_CSI_prediction_DNN.py:272-316).
//
// A Trainer owns its own copy of one component model in the forward GEMM layout
// (Wt [out][ldw], K-major, zero padded), the Adam moments, the gradients and the activations of
// the last batch.  Every matrix product of the step runs on launch_gemm (fp32 MFMA kernels):
//     forward   a_l  = relu(in_l . Wt_l^T + b_l)            EPI_BIAS_RELU_AFFINE (scale 1, shift 0)
//               out  = h . Wt_reg^T + b_reg                  EPI_BIAS
//     wgrad     gWt_l [out][in] = dz_l^T [out][B] . (in_l^T [in][B])^T       EPI_RAW, K = batch
//     dgrad     dh_{l-1} [B][in] = dz_l [B][out] . (W_l [in][out])^T         EPI_RAW, K = out
// with all leading dimensions rounded up to 32 and the padding kept at zero, so the GEMM kernels'
// over-read rules hold without special cases.  BatchNormalization / Dropout / loss / Adam are the
// kernels of train.hip.h.  csi_train_end(commit) hands the tensors back through csi_load_weights.
#pragma once
#include "csi_context.hpp"
#include "csi_dnn_f32.hpp"
#include "train.hip.h"

struct csi_trainer {
    struct L {
        int in = 0, out = 0, ldw = 0, ldo = 0;         // ldo = round32(out): row pitch of this layer's activations
        float *Wt = nullptr, *b = nullptr, *gWt = nullptr, *gb = nullptr;
        float *mWt = nullptr, *vWt = nullptr, *mb = nullptr, *vb = nullptr;
        float *gamma = nullptr, *beta = nullptr, *mmean = nullptr, *mvar = nullptr;
        float *ggamma = nullptr, *gbeta = nullptr, *mgamma = nullptr, *vgamma = nullptr, *mbeta = nullptr, *vbeta = nullptr;
        float *mu = nullptr, *istd = nullptr, *scale = nullptr, *shift = nullptr;
        float* Wk = nullptr;                           // [in][ldo] keras-layout copy (dgrad operand), layers >= 1
        float *a = nullptr, *h = nullptr, *ht = nullptr;   // [cap][ldo] post-relu, [cap][ldo] layer output, [out][ldb] transpose
    };
    csi_train_config tc{};
    std::vector<L> layers;            // n_hidden + regressor
    std::vector<float*> owned;        // every device allocation, for csi_train_end
    int64_t step = 0;
    int cap = 0, ldb = 0;             // batch capacity of the activation buffers, round32(cap)
    int maxw = 0;                     // widest padded layer output
    int k0 = 0, ldx = 0;              // input width, row pitch of xn
    float *x = nullptr, *y = nullptr, *xn = nullptr, *xt = nullptr;
    float *dz = nullptr, *dzt = nullptr, *dh[2] = {nullptr, nullptr};
    float *out = nullptr, *dout = nullptr, *doutt = nullptr, *partial = nullptr, *loss = nullptr;
    float *ones = nullptr, *zeros = nullptr, *tmp = nullptr;
    // resident dataset (csi_train_set_dataset): preamble table, per-sample table row / tx index, labels
    float *ds_table = nullptr, *ds_y = nullptr;
    int *ds_row = nullptr, *ds_itx = nullptr, *ids = nullptr;
    int64_t ds_n = 0, ds_rows = 0;
    int ids_cap = 0;
    float* gflat = nullptr;           // all gradients, one allocation (data-parallel all-reduce operand)
    int64_t gcount = 0;
    bool grads_pending = false;       // csi_train_backward ran, csi_train_apply has not
    size_t tmp_floats = 0;
};

namespace {

inline int r32(int v) { return (v + 31) / 32 * 32; }

int tr_alloc(csi_ctx* c, csi_trainer* t, float** p, size_t n) {
    const size_t bytes = (n + G_SLACK_FLOATS) * sizeof(float);
    if (hipMalloc((void**)p, bytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "trainer: device allocation of %zu bytes failed", bytes);
    t->owned.push_back(*p);
    HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, c->stream));
    return CSI_OK;
}

void tr_free(csi_trainer* t) {
    if (!t) return;
    for (float* p : t->owned) hipFree(p);
    delete t;
}

int tr_fill(csi_ctx* c, float* p, size_t n, float v) {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, c->stream, p, n, v);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

int tr_transpose(csi_ctx* c, const float* src, float* dst, int R, int C, int lds_, int ldd) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, c->stream, src, dst, R, C, lds_, ldd);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

uint64_t tr_stream(const csi_trainer* t, int tag) {
    uint64_t x = t->tc.seed + 0x9E3779B97F4A7C15ull * (uint64_t)(t->step * 64 + tag + 1);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// (re)allocate the batch-sized buffers
int tr_reserve(csi_ctx* c, csi_trainer* t, int B) {
    if (B <= t->cap) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // batch buffers are re-created from scratch; the old ones stay owned until csi_train_end (rare path)
    const int cap = std::max(B, 256), ldb = r32(cap);
    int rc = 0;
    const int n_out_ld = t->layers.back().ldo;
    int maxw = n_out_ld;
    for (auto& l : t->layers) maxw = std::max(maxw, l.ldo);
    rc |= tr_alloc(c, t, &t->x, (size_t)cap * t->k0);
    rc |= tr_alloc(c, t, &t->y, (size_t)cap * t->layers.back().out);
    rc |= tr_alloc(c, t, &t->xn, (size_t)cap * t->ldx);
    rc |= tr_alloc(c, t, &t->xt, (size_t)t->k0 * ldb);
    rc |= tr_alloc(c, t, &t->dz, (size_t)cap * maxw);
    rc |= tr_alloc(c, t, &t->dzt, (size_t)maxw * ldb);
    rc |= tr_alloc(c, t, &t->dh[0], (size_t)cap * maxw);
    rc |= tr_alloc(c, t, &t->dh[1], (size_t)cap * maxw);
    rc |= tr_alloc(c, t, &t->out, (size_t)cap * n_out_ld);
    rc |= tr_alloc(c, t, &t->dout, (size_t)cap * n_out_ld);
    rc |= tr_alloc(c, t, &t->doutt, (size_t)n_out_ld * ldb);
    for (size_t li = 0; li + 1 < t->layers.size(); ++li) {
        auto& l = t->layers[li];
        rc |= tr_alloc(c, t, &l.a, (size_t)cap * l.ldo);
        rc |= tr_alloc(c, t, &l.h, (size_t)cap * l.ldo);
        rc |= tr_alloc(c, t, &l.ht, (size_t)l.ldo * ldb);
    }
    if (rc) return CSI_ERR_NOMEM;
    t->cap = cap;
    t->ldb = ldb;
    return CSI_OK;
}

template <int EPI>
int tr_gemm(csi_ctx* c, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K, const float* bias,
            const float* scale, const float* shift) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Bt = Bt; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.k_per_split = (K + G_BK - 1) / G_BK * G_BK;
    g.bias = bias; g.scale = scale; g.shift = shift;
    return launch_gemm<EPI>(c, K_TRAIN_GEMM, g, 1);
}

// forward of the hidden stack; training: batch statistics + dropout, else folded running statistics
int tr_forward(csi_ctx* c, csi_trainer* t, int B, bool training) {
    const csi_config& cf = c->cfg;
    const int nh = cf.n_hidden;
    const float* in = t->xn;
    int ld_in = t->ldx, k_in = t->k0;
    for (int li = 0; li < nh; ++li) {
        auto& l = t->layers[li];
        int rc;
        if (training) {
            rc = tr_gemm<EPI_BIAS_RELU_AFFINE>(c, in, ld_in, l.Wt, l.ldw, l.a, l.ldo, B, l.out, k_in, l.b, t->ones, t->zeros);
            if (rc) return rc;
            const float p = li < nh - 1 ? t->tc.dropout : 0.f;
            ProfScope ps(c, K_TRAIN_ELEMWISE, 8.0 * B * l.out, 12.0 * B * l.out);
            hipLaunchKernelGGL(bn_dropout_forward_kernel, dim3((l.out + TRC - 1) / TRC), dim3(TRC, TRC), 0, c->stream, l.a, l.h, B, l.out, l.ldo,
                               cf.use_bn, l.gamma, l.beta, l.mmean, l.mvar, l.mu, l.istd, cf.bn_eps, t->tc.bn_momentum, p, tr_stream(t, li));
            HIP_TRY(c, hipGetLastError());
        } else {
            if (cf.use_bn) {
                hipLaunchKernelGGL(bn_fold_kernel, dim3((l.out + 255) / 256), dim3(256), 0, c->stream, l.gamma, l.beta, l.mmean, l.mvar,
                                   cf.bn_eps, l.scale, l.shift, l.out);
                HIP_TRY(c, hipGetLastError());
            }
            rc = tr_gemm<EPI_BIAS_RELU_AFFINE>(c, in, ld_in, l.Wt, l.ldw, l.h, l.ldo, B, l.out, k_in, l.b,
                                               cf.use_bn ? l.scale : t->ones, cf.use_bn ? l.shift : t->zeros);
            if (rc) return rc;
        }
        in = l.h;
        ld_in = l.ldo;
        k_in = l.ldo;            // padded K: the pad columns of h and of the next Wt are zero
    }
    auto& r = t->layers[nh];
    return tr_gemm<EPI_BIAS>(c, in, ld_in, r.Wt, r.ldw, t->out, r.ldo, B, r.out, k_in, r.b, nullptr, nullptr);
}

int tr_loss(csi_ctx* c, csi_trainer* t, int B, bool want_grad, float* h_loss) {
    auto& r = t->layers.back();
    const int nblk = (r.out + TRC - 1) / TRC;
    {
        ProfScope ps(c, K_TRAIN_ELEMWISE, 4.0 * B * r.out, 16.0 * B * r.out);
        hipLaunchKernelGGL(mse_grad_kernel, dim3(nblk), dim3(TRC, TRC), 0, c->stream, t->out, t->y, t->dout, t->doutt, r.gb, t->partial, B, r.out,
                           r.ldo, t->ldb, want_grad ? 1 : 0);
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, c->stream, t->partial, nblk, 1.0f / ((float)B * (float)r.out), t->loss);
        HIP_TRY(c, hipGetLastError());
    }
    if (h_loss) {
        HIP_TRY(c, hipMemcpyAsync(h_loss, t->loss, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

int tr_adam(csi_ctx* c, csi_trainer* t, float* p, const float* g, float* m, float* v, size_t n, float lr_t) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, p, g, m, v, n, lr_t,
                       t->tc.beta1, t->tc.beta2, t->tc.eps);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

int tr_upload_batch(csi_ctx* c, csi_trainer* t, const float* x, const float* y, int B) {
    HIP_TRY(c, hipMemcpyAsync(t->x, x, (size_t)B * t->k0 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(t->y, y, (size_t)B * t->layers.back().out * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return CSI_OK;
}

// stale rows / columns of a previous, larger batch must not leak into the padded-K products
int tr_clear_batch_padding(csi_ctx* c, csi_trainer* t, int B) {
    // transposed operands: columns B..ldb of every row are the K padding of the wgrad GEMMs
    if (B == t->ldb) return CSI_OK;
    auto clear_cols = [&](float* p, int rows) -> int {
        HIP_TRY(c, hipMemset2DAsync(p + B, (size_t)t->ldb * sizeof(float), 0, (size_t)(t->ldb - B) * sizeof(float), rows, c->stream));
        return CSI_OK;
    };
    int rc = clear_cols(t->xt, t->k0);
    if (!rc) rc = clear_cols(t->dzt, t->maxw);
    if (!rc) rc = clear_cols(t->doutt, t->layers.back().ldo);
    for (size_t li = 0; !rc && li + 1 < t->layers.size(); ++li) rc = clear_cols(t->layers[li].ht, t->layers[li].ldo);
    return rc;
}

// stage a batch: from host rows (ids == nullptr) or from the resident dataset by sample index
int tr_stage(csi_ctx* c, csi_trainer* t, const float* x, const float* y, const int* ids, int B, float noise_std, uint64_t stream) {
    const csi_config& cf = c->cfg;
    int rc = tr_reserve(c, t, B);
    if (rc) return rc;
    rc = tr_clear_batch_padding(c, t, B);
    if (rc) return rc;
    const int n_noisy = cf.nt > 0 ? cf.len_ltf : t->k0;
    ProfScope ps(c, K_TRAIN_ELEMWISE, 0.0, 12.0 * B * t->k0);
    if (!ids) {
        rc = tr_upload_batch(c, t, x, y, B);
        if (rc) return rc;
        hipLaunchKernelGGL(train_input_kernel, dim3((t->k0 + 31) / 32, (B + 31) / 32), dim3(256), 0, c->stream, t->x, t->xn, t->xt, B, t->k0,
                           t->ldx, t->ldb, n_noisy, noise_std, stream);
    } else {
        if (!t->ds_table) return fail(c, CSI_ERR_NOT_READY, "no resident dataset: call csi_train_set_dataset first");
        if (B > t->ids_cap) {
            const int cap = std::max(B, 256);
            float* p = nullptr;
            rc = tr_alloc(c, t, &p, (size_t)cap);
            if (rc) return rc;
            t->ids = reinterpret_cast<int*>(p);
            t->ids_cap = cap;
        }
        for (int b = 0; b < B; ++b)
            if (ids[b] < 0 || ids[b] >= t->ds_n) return fail(c, CSI_ERR_INVALID_ARG, "sample index %d outside the resident dataset (%lld samples)", ids[b], (long long)t->ds_n);
        HIP_TRY(c, hipMemcpyAsync(t->ids, ids, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(train_gather_kernel, dim3((t->k0 + 31) / 32, (B + 31) / 32), dim3(256), 0, c->stream, t->ds_table, t->ds_row, t->ds_itx,
                           c->P, t->ds_y, t->ids, t->xn, t->xt, t->y, B, t->k0, n_noisy, cf.nt, t->layers.back().out, t->ldx, t->ldb, noise_std, stream);
    }
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// forward + backward of one batch: loss and all gradients (no parameter update)
int tr_backward(csi_ctx* c, csi_trainer* t, const float* x, const float* y, const int* ids, int B, float noise_std, float* h_loss) {
    const csi_config& cf = c->cfg;
    const int nh = cf.n_hidden;
    t->step += 1;
    int rc = tr_stage(c, t, x, y, ids, B, noise_std, tr_stream(t, 60));
    if (rc) return rc;
    rc = tr_forward(c, t, B, true);
    if (rc) return rc;
    rc = tr_loss(c, t, B, true, nullptr);
    if (rc) return rc;

    // ---- backward
    auto& r = t->layers[nh];
    auto& last = t->layers[nh - 1];
    rc = tr_transpose(c, last.h, last.ht, B, last.out, last.ldo, t->ldb);
    if (rc) return rc;
    // regressor: gWt [n_out][in] = dout^T . h^T^T ;  dh = dout . W
    rc = tr_gemm<EPI_RAW>(c, t->doutt, t->ldb, last.ht, t->ldb, r.gWt, r.ldw, r.out, r.in, t->ldb, nullptr, nullptr, nullptr);
    if (rc) return rc;
    rc = tr_transpose(c, r.Wt, r.Wk, r.out, r.in, r.ldw, r.ldo);
    if (rc) return rc;
    int cur = 0;
    rc = tr_gemm<EPI_RAW>(c, t->dout, r.ldo, r.Wk, r.ldo, t->dh[cur], last.ldo, B, r.in, r.ldo, nullptr, nullptr, nullptr);
    if (rc) return rc;
    for (int li = nh - 1; li >= 0; --li) {
        auto& l = t->layers[li];
        const float p = li < nh - 1 ? t->tc.dropout : 0.f;
        {
            ProfScope ps(c, K_TRAIN_ELEMWISE, 12.0 * B * l.out, 20.0 * B * l.out);
            hipLaunchKernelGGL(bn_dropout_backward_kernel, dim3((l.out + TRC - 1) / TRC), dim3(TRC, TRC), 0, c->stream, t->dh[cur], l.a, t->dz, t->dzt, B,
                               l.out, l.ldo, t->ldb, cf.use_bn, l.gamma, l.mu, l.istd, l.ggamma, l.gbeta, l.gb, p, tr_stream(t, li));
            HIP_TRY(c, hipGetLastError());
        }
        const float* in_t = li == 0 ? t->xt : t->layers[li - 1].ht;
        if (li > 0) {
            auto& pl = t->layers[li - 1];
            rc = tr_transpose(c, pl.h, pl.ht, B, pl.out, pl.ldo, t->ldb);
            if (rc) return rc;
        }
        rc = tr_gemm<EPI_RAW>(c, t->dzt, t->ldb, in_t, t->ldb, l.gWt, l.ldw, l.out, l.in, t->ldb, nullptr, nullptr, nullptr);
        if (rc) return rc;
        if (li > 0) {
            auto& pl = t->layers[li - 1];
            rc = tr_transpose(c, l.Wt, l.Wk, l.out, l.in, l.ldw, l.ldo);
            if (rc) return rc;
            rc = tr_gemm<EPI_RAW>(c, t->dz, l.ldo, l.Wk, l.ldo, t->dh[cur ^ 1], pl.ldo, B, l.in, l.ldo, nullptr, nullptr, nullptr);
            if (rc) return rc;
            cur ^= 1;
        }
    }

    t->grads_pending = true;
    if (h_loss) {
        HIP_TRY(c, hipMemcpyAsync(h_loss, t->loss, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

// Adam on the gradients of the last backward pass (keras: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t))
int tr_apply(csi_ctx* c, csi_trainer* t) {
    const csi_config& cf = c->cfg;
    const int nh = cf.n_hidden;
    if (!t->grads_pending) return fail(c, CSI_ERR_NOT_READY, "csi_train_apply: no gradients pending (call csi_train_backward first)");
    const double tt = (double)t->step;
    const float lr_t = (float)((double)t->tc.lr * std::sqrt(1.0 - std::pow((double)t->tc.beta2, tt)) / (1.0 - std::pow((double)t->tc.beta1, tt)));
    double elems = 0.0;
    for (auto& l : t->layers) elems += (double)l.out * l.ldw;
    ProfScope ps(c, K_TRAIN_ELEMWISE, 10.0 * elems, 28.0 * elems);
    for (int li = 0; li <= nh; ++li) {
        auto& l = t->layers[li];
        int rc = tr_adam(c, t, l.Wt, l.gWt, l.mWt, l.vWt, (size_t)l.out * l.ldw, lr_t);
        if (!rc) rc = tr_adam(c, t, l.b, l.gb, l.mb, l.vb, l.out, lr_t);
        if (!rc && li < nh && cf.use_bn) {
            rc = tr_adam(c, t, l.gamma, l.ggamma, l.mgamma, l.vgamma, l.out, lr_t);
            if (!rc) rc = tr_adam(c, t, l.beta, l.gbeta, l.mbeta, l.vbeta, l.out, lr_t);
        }
        if (rc) return rc;
    }
    t->grads_pending = false;
    return CSI_OK;
}

int tr_eval(csi_ctx* c, csi_trainer* t, const float* x, const float* y, const int* ids, int B, float* h_loss) {
    int rc = tr_stage(c, t, x, y, ids, B, 0.f, (uint64_t)0);
    if (rc) return rc;
    rc = tr_forward(c, t, B, false);
    if (rc) return rc;
    return tr_loss(c, t, B, false, h_loss);
}

// upload the training set once: it stays in HBM for the whole fit (491 MB + 359 MB at the shipped size)
int tr_set_dataset(csi_ctx* c, csi_trainer* t, const float* table, int64_t n_rows, const int* ltf_row, const int* itx, const float* y, int64_t N) {
    const csi_config& cf = c->cfg;
    if (cf.nt > 0 && !c->pilot_ok) return fail(c, CSI_ERR_NOT_READY, "csi_train_set_dataset: csi_set_pilot first (the pilot columns of a sample are rows of P)");
    const int len = cf.nt > 0 ? cf.len_ltf : t->k0;
    for (int64_t i = 0; i < N; ++i) {
        if (ltf_row[i] < 0 || ltf_row[i] >= n_rows) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_set_dataset: ltf_row[%lld] out of range", (long long)i);
        if (cf.nt > 0 && (itx[i] < 0 || itx[i] >= cf.nt)) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_set_dataset: itx[%lld] out of range", (long long)i);
    }
    float *p_row = nullptr, *p_itx = nullptr;
    int rc = tr_alloc(c, t, &t->ds_table, (size_t)n_rows * len);
    if (!rc) rc = tr_alloc(c, t, &t->ds_y, (size_t)N * t->layers.back().out);
    if (!rc) rc = tr_alloc(c, t, &p_row, (size_t)N);
    if (!rc) rc = tr_alloc(c, t, &p_itx, (size_t)N);
    if (rc) return rc;
    t->ds_row = reinterpret_cast<int*>(p_row);
    t->ds_itx = reinterpret_cast<int*>(p_itx);
    HIP_TRY(c, hipMemcpyAsync(t->ds_table, table, (size_t)n_rows * len * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(t->ds_y, y, (size_t)N * t->layers.back().out * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(t->ds_row, ltf_row, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (cf.nt > 0) HIP_TRY(c, hipMemcpyAsync(t->ds_itx, itx, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    t->ds_n = N;
    t->ds_rows = n_rows;
    return CSI_OK;
}

// name -> device pointer / shape of a trainer tensor.  Kernels are exposed in the keras layout
// [in][out]; "grad:<name>" addresses the gradient of the last step.
struct TrRef {
    const float* p;
    int rows, cols, ld;
    bool kernel;       // stored transposed ([out][ld]) -> needs a transpose on the way out
};
bool tr_find(const csi_ctx* c, const csi_trainer* t, std::string name, TrRef* r) {
    bool grad = false;
    if (name.rfind("grad:", 0) == 0) { grad = true; name = name.substr(5); }
    const int nh = c->cfg.n_hidden;
    for (int li = 0; li <= nh; ++li) {
        const auto& l = t->layers[li];
        const std::string base = li == nh ? std::string("fc_regressor") : "fc_dense" + std::to_string(li);
        if (name == base + ".kernel") { *r = {grad ? l.gWt : l.Wt, l.in, l.out, l.ldw, true}; return true; }
        if (name == base + ".bias") { *r = {grad ? l.gb : l.b, 1, l.out, l.out, false}; return true; }
        if (li < nh && c->cfg.use_bn) {
            const std::string bn = "bn" + std::to_string(li);
            if (name == bn + ".gamma") { *r = {grad ? l.ggamma : l.gamma, 1, l.out, l.out, false}; return true; }
            if (name == bn + ".beta") { *r = {grad ? l.gbeta : l.beta, 1, l.out, l.out, false}; return true; }
            if (!grad && name == bn + ".moving_mean") { *r = {l.mmean, 1, l.out, l.out, false}; return true; }
            if (!grad && name == bn + ".moving_variance") { *r = {l.mvar, 1, l.out, l.out, false}; return true; }
        }
    }
    return false;
}

int tr_get(csi_ctx* c, csi_trainer* t, const char* name, float* out, int64_t count) {
    TrRef r;
    if (!tr_find(c, t, name, &r)) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_get: unknown tensor '%s'", name);
    if (count != (int64_t)r.rows * r.cols)
        return fail(c, CSI_ERR_INVALID_ARG, "csi_train_get: '%s' has %d x %d elements, buffer holds %lld", name, r.rows, r.cols, (long long)count);
    if (!r.kernel) {
        HIP_TRY(c, hipMemcpyAsync(out, r.p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    } else {
        // stored [out][ld] -> keras [in][out]
        if (t->tmp_floats < (size_t)count) {
            int rc = tr_alloc(c, t, &t->tmp, (size_t)count);
            if (rc) return rc;
            t->tmp_floats = (size_t)count;
        }
        int rc = tr_transpose(c, r.p, t->tmp, r.cols, r.rows, r.ld, r.cols);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(out, t->tmp, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

int tr_begin(csi_ctx* c, int model, const csi_train_config* tc, const csi_tensor* tensors, int n) {
    const csi_config& cf = c->cfg;
    if (cf.dtype != CSI_DTYPE_F32) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: training runs in fp32 contexts only");
    if (tc->lr <= 0.f || tc->dropout < 0.f || tc->dropout >= 1.f || tc->bn_momentum < 0.f || tc->bn_momentum > 1.f)
        return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: bad hyper-parameter");
    if (c->d_in & 3) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: input width %d must be a multiple of 4", c->d_in);
    if (c->trainer[model]) { tr_free(c->trainer[model]); c->trainer[model] = nullptr; }
    csi_trainer* t = new csi_trainer();
    c->trainer[model] = t;
    t->tc = *tc;
    t->k0 = c->d_in;
    t->ldx = c->d_in;
    const int nh = cf.n_hidden;
    t->layers.resize(nh + 1);
    int fan_in = c->d_in, maxw = 0;
    int rc = 0;
    // every gradient lives in ONE flat buffer (regressor first, layer 0 last = the order the backward
    // pass produces them): a data-parallel caller all-reduces it with a single collective
    size_t gtotal = 0;
    for (int li = 0; li <= nh; ++li) {
        auto& l = t->layers[li];
        l.in = fan_in;
        l.out = li == nh ? cf.n_out : cf.hidden[li];
        l.ldw = r32(li == 0 ? fan_in : t->layers[li - 1].ldo);      // = the padded K the forward GEMM walks
        l.ldo = r32(l.out);
        maxw = std::max(maxw, l.ldo);
        gtotal += (size_t)l.out * l.ldw + r32(l.out) * (size_t)(li < nh && cf.use_bn ? 3 : 1);
        fan_in = l.out;
    }
    rc = tr_alloc(c, t, &t->gflat, gtotal);
    if (rc) return rc;
    t->gcount = (int64_t)gtotal;
    size_t goff = 0;
    auto carve = [&](float** p, size_t n) { *p = t->gflat + goff; goff += n; };
    for (int li = nh; li >= 0; --li) {
        auto& l = t->layers[li];
        carve(&l.gWt, (size_t)l.out * l.ldw);
        carve(&l.gb, r32(l.out));
        if (li < nh && cf.use_bn) { carve(&l.ggamma, r32(l.out)); carve(&l.gbeta, r32(l.out)); }
    }
    for (int li = 0; li <= nh; ++li) {
        auto& l = t->layers[li];
        const size_t nw = (size_t)l.out * l.ldw;
        rc |= tr_alloc(c, t, &l.Wt, nw);
        rc |= tr_alloc(c, t, &l.mWt, nw); rc |= tr_alloc(c, t, &l.vWt, nw);
        rc |= tr_alloc(c, t, &l.b, l.out);
        rc |= tr_alloc(c, t, &l.mb, l.out); rc |= tr_alloc(c, t, &l.vb, l.out);
        if (li > 0) rc |= tr_alloc(c, t, &l.Wk, (size_t)l.in * l.ldo);
        if (li < nh) {
            for (float** p : {&l.gamma, &l.beta, &l.mmean, &l.mvar, &l.mgamma, &l.vgamma, &l.mbeta, &l.vbeta, &l.mu, &l.istd, &l.scale,
                              &l.shift})
                rc |= tr_alloc(c, t, p, l.out);
            if (!cf.use_bn) { rc |= tr_alloc(c, t, &l.ggamma, l.out); rc |= tr_alloc(c, t, &l.gbeta, l.out); }
        }
        if (rc) return CSI_ERR_NOMEM;
    }
    t->maxw = maxw;
    if ((cf.n_out + TRC - 1) / TRC > 128) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: n_out above 4096 is not supported");
    rc |= tr_alloc(c, t, &t->ones, maxw);
    rc |= tr_alloc(c, t, &t->zeros, maxw);
    rc |= tr_alloc(c, t, &t->partial, 128);
    rc |= tr_alloc(c, t, &t->loss, 4);
    if (rc) return CSI_ERR_NOMEM;
    rc = tr_fill(c, t->ones, maxw, 1.f);
    if (rc) return rc;

    // ---- initial values
    for (int li = 0; li <= nh; ++li) {
        auto& l = t->layers[li];
        const std::string base = li == nh ? std::string("fc_regressor") : "fc_dense" + std::to_string(li);
        if (n > 0) {
            const csi_tensor* k = find_tensor(tensors, n, base + ".kernel");
            const csi_tensor* b = find_tensor(tensors, n, base + ".bias");
            if (!k || !b || !k->data || !b->data || k->rows != l.in || k->cols != l.out || b->rows * b->cols != l.out)
                return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: missing or mis-shaped %s.kernel/.bias", base.c_str());
            // keras [in][out] -> device staging -> Wt [out][ldw]
            float* stage = nullptr;
            if (hipMalloc((void**)&stage, (size_t)l.in * l.out * sizeof(float)) != hipSuccess)
                return fail(c, CSI_ERR_NOMEM, "csi_train_begin: staging allocation failed");
            hipError_t e = hipMemcpyAsync(stage, k->data, (size_t)l.in * l.out * sizeof(float), hipMemcpyHostToDevice, c->stream);
            rc = e == hipSuccess ? tr_transpose(c, stage, l.Wt, l.in, l.out, l.out, l.ldw) : CSI_ERR_HIP;
            if (!rc && hipMemcpyAsync(l.b, b->data, (size_t)l.out * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = CSI_ERR_HIP;
            if (hipStreamSynchronize(c->stream) != hipSuccess) rc = CSI_ERR_HIP;       // host tensors may go away after the call
            hipFree(stage);
            if (rc) return fail(c, rc, "csi_train_begin: upload of %s failed", base.c_str());
        } else {
            hipLaunchKernelGGL(glorot_kernel, dim3(1024), dim3(256), 0, c->stream, l.Wt, l.out, l.in, l.ldw, tr_stream(t, 40 + li));
            HIP_TRY(c, hipGetLastError());
        }
        if (li < nh && cf.use_bn) {
            const std::string bn = "bn" + std::to_string(li);
            if (n > 0) {
                struct { const char* suffix; float* dst; } items[] = {{".gamma", l.gamma}, {".beta", l.beta}, {".moving_mean", l.mmean},
                                                                       {".moving_variance", l.mvar}};
                for (auto& it : items) {
                    const csi_tensor* q = find_tensor(tensors, n, bn + it.suffix);
                    if (!q || !q->data || q->rows * q->cols != l.out)
                        return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: missing or mis-shaped %s%s", bn.c_str(), it.suffix);
                    HIP_TRY(c, hipMemcpyAsync(it.dst, q->data, (size_t)l.out * sizeof(float), hipMemcpyHostToDevice, c->stream));
                }
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            } else {
                rc = tr_fill(c, l.gamma, l.out, 1.f);
                if (!rc) rc = tr_fill(c, l.mvar, l.out, 1.f);
                if (rc) return rc;
            }
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

}  // namespace
