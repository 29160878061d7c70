// train.hip.h - device kernels of the on-box fine-tuning step.
//
// Reference: massiveMIMO_CSI_prediction_DNN.py
//   :191-193  GaussianNoise 'AWGN_layer' on the LTF input only (method default_SNR, training)
//   :92-100   changeNoisePower: per-batch stddev = sqrt(avg_sigPow / 10^(SNR/10)) / sqrt(2)
//   :211-227  Dense(relu) -> BatchNormalization -> Dropout (all hidden layers but the last) -> Dense(linear)
//   :272-276  Adam(lr), loss 'mse'
// Keras semantics restated (TensorFlow is not in this image; "parity unpinned" like the other
// Keras rows): BatchNormalization in training mode normalises with the biased batch variance,
// epsilon 1e-3, and moves its running statistics with momentum 0.99; Dropout scales the kept
// units by 1/(1-rate); mse is the mean over all B*n_out elements; Adam uses
// lr_t = lr*sqrt(1-b2^t)/(1-b1^t), p -= lr_t*m/(sqrt(v)+eps), eps = 1e-7.
//
// The GEMMs of the step (forward, dgrad, wgrad) run on the fp32 MFMA kernels of gemm_f32.hip.h;
// this file holds the element-wise / column-reduction kernels around them.  Batches are small
// (256 rows in full_pipeline_maMIMO_DNNEst.sh:40), so these kernels favour simplicity: one thread
// per feature column walks the rows (coalesced across the 64 lanes of a wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ls_estimate.hip.h"      // splitmix64

namespace csi {

// uniform in (0,1) and standard normal from a counter (stream, index)
__device__ __forceinline__ float tr_uniform(uint64_t stream, uint64_t idx) {
    const uint64_t h = splitmix64(stream ^ splitmix64(idx));
    return ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float tr_normal(uint64_t stream, uint64_t idx) {
    const uint64_t h = splitmix64(stream ^ splitmix64(idx));
    const float u1 = ((float)(uint32_t)(h >> 32) + 0.5f) * (1.0f / 4294967296.0f);
    const float u2 = ((float)(uint32_t)h + 0.5f) * (1.0f / 4294967296.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// xn[b][k] = x[b][k] + noise_std * N(0,1) for k < n_noisy (the LTF columns), copied otherwise;
// xt[k][b] = xn[b][k] (ldt >= B, padding pre-zeroed).  32x32 tiles through LDS.
__global__ __launch_bounds__(256) void train_input_kernel(const float* __restrict__ x, float* __restrict__ xn, float* __restrict__ xt,
                                                          int B, int K, int ldx, int ldt, int n_noisy, float noise_std, uint64_t stream) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, k = k0 + tx;
        float v = 0.f;
        if (b < B && k < K) {
            v = x[(size_t)b * K + k];
            if (k < n_noisy && noise_std != 0.f) v += noise_std * tr_normal(stream, (uint64_t)b * K + k);
            xn[(size_t)b * ldx + k] = v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, b = b0 + tx;
        if (k < K && b < B) xt[(size_t)k * ldt + b] = tile[tx][r];
    }
}

// Resident-dataset form of train_input_kernel: row b of the batch is sample ids[b] of a dataset that
// lives in HBM - its LTF part is row ltf_row[s] of the preamble table (every rx preamble stored once,
// create_massiveMIMO_CSIest_dnn_dataset.py:50-63), its pilot part row itx[s] of P
// (massiveMIMO_dataGenerator.py:309-311).  Also gathers the labels: yb[b] = y[ids[b]].
__global__ __launch_bounds__(256) void train_gather_kernel(const float* __restrict__ table, const int* __restrict__ ltf_row,
                                                           const int* __restrict__ itx, const float* __restrict__ P,
                                                           const float* __restrict__ yall, const int* __restrict__ ids,
                                                           float* __restrict__ xn, float* __restrict__ xt, float* __restrict__ yb, int B, int K,
                                                           int len_ltf, int nt, int n_out, int ldx, int ldt, float noise_std, uint64_t stream) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, k = k0 + tx;
        float v = 0.f;
        if (b < B && k < K) {
            const int s = ids[b];
            if (k < len_ltf) {
                v = table[(size_t)ltf_row[s] * len_ltf + k];
                if (noise_std != 0.f) v += noise_std * tr_normal(stream, (uint64_t)b * K + k);
            } else {
                v = P[(size_t)itx[s] * nt + (k - len_ltf)];
            }
            xn[(size_t)b * ldx + k] = v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, b = b0 + tx;
        if (k < K && b < B) xt[(size_t)k * ldt + b] = tile[tx][r];
    }
    // labels: the first column-blocks of the grid copy them (n_out <= K always holds: 234 vs 321*nt)
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, n = k0 + tx;
        if (b < B && n < n_out) yb[(size_t)b * n_out + n] = yall[(size_t)ids[b] * n_out + n];
    }
}

// dst[c][r] = src[r][c]   (src [R][lds_], dst [C][ldd], ldd >= R, padding untouched)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, int lds_, int ldd) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldd + r] = tile[tx][i];
    }
}

// Column kernels: a workgroup of 32 x 32 threads owns 32 feature columns; thread (tx, ty) walks the
// rows ty, ty+32, ... (a wave reads two 128-byte row segments per step) and column sums are
// combined through LDS in a fixed order (deterministic).
constexpr int TRC = 32;

__device__ __forceinline__ float tr_col_sum(float v, float (*red)[TRC + 1], int tx, int ty) {
    red[ty][tx] = v;
    __syncthreads();
    if (ty == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TRC; ++i) s += red[i][tx];
        red[0][tx] = s;
    }
    __syncthreads();
    const float r = red[0][tx];
    __syncthreads();
    return r;
}

// BatchNormalization (training) + Dropout over a [B][F] post-relu activation a (row pitch ld).
//   mu = mean_b a, var = mean_b (a-mu)^2, xhat = (a-mu)*rsqrt(var+eps), h = keep*(gamma*xhat+beta)/(1-p)
// Saves mu / inv_std for the backward pass, moves the running statistics (keras:
// moving = moving*momentum + batch*(1-momentum); the variance moved is the biased one).
// use_bn == 0: h = keep*a/(1-p).   p == 0: no dropout.
__global__ __launch_bounds__(TRC * TRC) void bn_dropout_forward_kernel(const float* __restrict__ a, float* __restrict__ h, int B, int F, int ld,
                                                                        int use_bn, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, float* __restrict__ mmean,
                                                                        float* __restrict__ mvar, float* __restrict__ mu_out,
                                                                        float* __restrict__ istd_out, float eps, float momentum, float p,
                                                                        uint64_t stream) {
    __shared__ float red[TRC][TRC + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int j = blockIdx.x * TRC + tx;
    const bool ok = j < F;
    float mu = 0.f, is = 1.f, ga = 1.f, be = 0.f;
    if (use_bn) {
        float s = 0.f;
        if (ok) for (int b = ty; b < B; b += TRC) s += a[(size_t)b * ld + j];
        mu = tr_col_sum(s, red, tx, ty) / (float)B;
        float q = 0.f;
        if (ok) for (int b = ty; b < B; b += TRC) { const float d = a[(size_t)b * ld + j] - mu; q = fmaf(d, d, q); }
        const float var = tr_col_sum(q, red, tx, ty) / (float)B;
        is = 1.0f / sqrtf(var + eps);
        if (ok) {
            ga = gamma[j]; be = beta[j];
            if (ty == 0) {
                mmean[j] = mmean[j] * momentum + mu * (1.f - momentum);
                mvar[j] = mvar[j] * momentum + var * (1.f - momentum);
                mu_out[j] = mu;
                istd_out[j] = is;
            }
        }
    }
    if (!ok) return;
    const float keep_scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int b = ty; b < B; b += TRC) {
        float v = a[(size_t)b * ld + j];
        if (use_bn) v = fmaf((v - mu) * is, ga, be);
        if (p > 0.f) v = tr_uniform(stream, (uint64_t)b * F + j) >= p ? v * keep_scale : 0.f;
        h[(size_t)b * ld + j] = v;
    }
}

// Backward of Dropout -> BatchNormalization(training) -> relu for one hidden layer.
//   dy = dh*keep/(1-p);  dgamma = sum dy*xhat, dbeta = sum dy
//   da = gamma*istd/B * (B*dy - dbeta - xhat*dgamma);  dz = da * (a > 0);  dbias = sum dz
// Writes dz [B][F] and its transpose dzt [F][ldt] (wgrad operand; through an LDS tile so that both
// stores are coalesced).
__global__ __launch_bounds__(TRC * TRC) void bn_dropout_backward_kernel(const float* __restrict__ dh, const float* __restrict__ a,
                                                                         float* __restrict__ dz, float* __restrict__ dzt, int B, int F, int ld,
                                                                         int ldt, int use_bn, const float* __restrict__ gamma,
                                                                         const float* __restrict__ mu_in, const float* __restrict__ istd_in,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                         float* __restrict__ dbias, float p, uint64_t stream) {
    __shared__ float red[TRC][TRC + 1];
    __shared__ float tile[TRC][TRC + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int f0 = blockIdx.x * TRC;
    const int j = f0 + tx;
    const bool ok = j < F;
    const float keep_scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    auto dy_at = [&](int b) {
        float dy = dh[(size_t)b * ld + j];
        if (p > 0.f) dy = tr_uniform(stream, (uint64_t)b * F + j) >= p ? dy * keep_scale : 0.f;
        return dy;
    };
    float mu = 0.f, is = 1.f, ga = 1.f, sdy = 0.f, sdyx = 0.f;
    if (use_bn) {
        if (ok) { mu = mu_in[j]; is = istd_in[j]; ga = gamma[j]; }
        float s0 = 0.f, s1 = 0.f;
        if (ok)
            for (int b = ty; b < B; b += TRC) {
                const float dy = dy_at(b);
                s0 += dy;
                s1 = fmaf(dy, (a[(size_t)b * ld + j] - mu) * is, s1);
            }
        sdy = tr_col_sum(s0, red, tx, ty);
        sdyx = tr_col_sum(s1, red, tx, ty);
        if (ok && ty == 0) { dgamma[j] = sdyx; dbeta[j] = sdy; }
    }
    float sb = 0.f;
    const float invB = 1.f / (float)B;
    for (int b0 = 0; b0 < B; b0 += TRC) {
        const int b = b0 + ty;
        float g = 0.f;
        if (ok && b < B) {
            const float dy = dy_at(b);
            const float av = a[(size_t)b * ld + j];
            float da = dy;
            if (use_bn) da = ga * is * (dy - invB * sdy - invB * ((av - mu) * is) * sdyx);
            g = av > 0.f ? da : 0.f;
            dz[(size_t)b * ld + j] = g;
            sb += g;
        }
        tile[ty][tx] = g;                        // [row][feature]
        __syncthreads();
        const int fo = f0 + ty, bo = b0 + tx;    // transposed store: feature = ty, row = tx
        if (fo < F && bo < B) dzt[(size_t)fo * ldt + bo] = tile[tx][ty];
        __syncthreads();
    }
    const float tot = tr_col_sum(sb, red, tx, ty);
    if (ok && ty == 0) dbias[j] = tot;
}

// mse loss and its gradient for the regressor output: loss = mean (out-y)^2 over B*N elements,
// dout = 2 (out-y)/(B*N); also dout^T [N][ldt] and dbias[n] = sum_b dout.  One workgroup per 32
// output columns; the per-workgroup partial losses are combined by loss_finish_kernel (fixed order).
__global__ __launch_bounds__(TRC * TRC) void mse_grad_kernel(const float* __restrict__ out, const float* __restrict__ y, float* __restrict__ dout,
                                                              float* __restrict__ doutt, float* __restrict__ dbias, float* __restrict__ partial,
                                                              int B, int N, int ldo, int ldt, int want_grad) {
    __shared__ float red[TRC][TRC + 1];
    __shared__ float tile[TRC][TRC + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int n0 = blockIdx.x * TRC;
    const int n = n0 + tx;
    const bool ok = n < N;
    const float sc = 2.0f / ((float)B * (float)N);
    float acc = 0.f, sb = 0.f;
    for (int b0 = 0; b0 < B; b0 += TRC) {
        const int b = b0 + ty;
        float g = 0.f;
        if (ok && b < B) {
            const float d = out[(size_t)b * ldo + n] - y[(size_t)b * N + n];
            acc = fmaf(d, d, acc);
            g = sc * d;
            if (want_grad) dout[(size_t)b * ldo + n] = g;
            sb += g;
        }
        if (want_grad) {
            tile[ty][tx] = g;
            __syncthreads();
            const int no = n0 + ty, bo = b0 + tx;
            if (no < N && bo < B) doutt[(size_t)no * ldt + bo] = tile[tx][ty];
            __syncthreads();
        }
    }
    const float col = tr_col_sum(acc, red, tx, ty);
    const float colb = tr_col_sum(sb, red, tx, ty);
    if (want_grad && ok && ty == 0) dbias[n] = colb;
    // sum of the 32 column totals of this workgroup
    red[0][tx] = ok ? col : 0.f;
    __syncthreads();
    if (tx == 0 && ty == 0) {
        float s = 0.f;
        for (int i = 0; i < TRC; ++i) s += red[0][i];
        partial[blockIdx.x] = s;
    }
}
__global__ void loss_finish_kernel(const float* __restrict__ partial, int n, float inv_count, float* __restrict__ loss) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)partial[i];
        *loss = (float)(s * (double)inv_count);
    }
}

// Adam over a flat parameter array (keras: eps outside the sqrt, bias correction folded into lr_t)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr_t, float b1, float b2, float eps) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// inference-mode BatchNormalization folded to scale/shift (as csi_load_weights does on the host)
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mmean,
                               const float* __restrict__ mvar, float eps, float* __restrict__ scale, float* __restrict__ shift, int F) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F) return;
    const float inv = (1.0f / sqrtf(mvar[j] + eps)) * gamma[j];
    scale[j] = inv;
    shift[j] = beta[j] - mmean[j] * inv;
}

__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// Glorot-uniform initialisation of a K-major weight matrix Wt [out][ld] (columns >= in stay zero):
// U(-l, l), l = sqrt(6/(in+out))   (kernel_initializer='glorot_uniform', DNN.py:213,227)
__global__ void glorot_kernel(float* __restrict__ wt, int out, int in, int ld, uint64_t stream) {
    const size_t total = (size_t)out * in;
    const float lim = sqrtf(6.0f / (float)(in + out));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int o = (int)(i / in), k = (int)(i - (size_t)o * in);
        wt[(size_t)o * ld + k] = (2.f * tr_uniform(stream, i) - 1.f) * lim;
    }
}

}  // namespace csi
