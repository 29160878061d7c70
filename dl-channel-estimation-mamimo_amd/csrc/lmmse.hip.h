// lmmse.hip.h - LMMSE smoothing of the LS estimate.
//
// Reference: LMMSE_ce.m:23-39, called per link from helperMIMOChannelEstimate.m:37-39 with
// Nfft = Np = 234 and Nps = 1:
//     H_mmse = Rhp * inv(Rpp) * H_ls,   Rhp[a][b] = 1 / (1 + j 2 pi tau_rms (a-b) / 234),
//                                       Rpp = Rhp + I / snr
// where tau_rms is the rms "delay" of the vector h the caller passes and snr = 10^(SNR/10).
// The reference inverts the 234x234 matrix once per (tx, rx) link (its slowest stage: 1.1 s per
// packet at Nt = 32, timing_cpu_vs_gpu_barplot.eps).  Here:
//   * R is Hermitian Toeplitz and identical for the Nt links of an rx antenna, and
//     R (R + s I)^-1 H = H - s (R + s I)^-1 H, so only ONE Hermitian-Toeplitz system with Nt
//     right-hand sides is solved per (packet, rx) - no inverse, no 234x234 matrix in memory.
//   * Levinson recursion (O(n^2) per right-hand side, O(n) storage), in fp64 (the vector fp64
//     rate of gfx950 makes this cheap; R + s I has condition numbers up to ~1e4).
//   * one workgroup per (packet, rx, 32 tx antennas): 8 lanes per right-hand side, each lane keeps
//     30 solution entries in registers; the shared forward vector lives in LDS (ping-pong, one
//     barrier per recursion step); the LS columns are staged in LDS and reused for the output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

constexpr int LM_N = 234;
constexpr int LM_RHS = 32;                 // right-hand sides per workgroup
constexpr int LM_PART = 8;                 // lanes per right-hand side
constexpr int LM_EPL = (LM_N + LM_PART - 1) / LM_PART;     // 30 entries per lane
constexpr int LM_THREADS = LM_RHS * LM_PART;

struct LmmseArgs {
    const float* h_re;      // LS estimate [nblk][nt][234]
    const float* h_im;
    const float* hvec;      // [npkt][L]  the vector LMMSE_ce receives as 'h'
    const float* snr_db;    // [nblk]     SNR(i) per (packet, rx)
    float* o_re;            // [nblk][nt][234]
    float* o_im;
    int nt, nr, L;
};

struct cd {
    double x, y;
};
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cfma(cd a, cd b, cd c) {          // a*b + c
    return {fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y))};
}
__device__ __forceinline__ cd cconj(cd a) { return {a.x, -a.y}; }
__device__ __forceinline__ cd group_sum(cd v) {                  // over the 8 lanes of a right-hand side
#pragma unroll
    for (int o = 1; o < LM_PART; o <<= 1) {
        v.x += __shfl_xor(v.x, o);
        v.y += __shfl_xor(v.y, o);
    }
    return v;
}

__global__ __launch_bounds__(LM_THREADS) void lmmse_levinson_kernel(const LmmseArgs a, int n_jc) {
    __shared__ cd t[LM_N];                       // first column of the normalised matrix, t[0] = 1
    __shared__ cd f[2][LM_N];                    // forward vector, ping-pong
    __shared__ float2 y[LM_RHS][LM_N];           // LS columns, later the output

    const int tid = threadIdx.x;
    const int jl = tid / LM_PART, e = tid % LM_PART;
    const size_t blk = blockIdx.x / n_jc;
    const int jc = blockIdx.x % n_jc;
    const int p = (int)(blk / a.nr);

    // rms "delay" of h (LMMSE_ce.m:27-30); every thread evaluates it (L is ~100)
    double hh = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < a.L; ++k) {
        const double v = (double)a.hvec[(size_t)p * a.L + k];
        const double w = v * v;
        hh += w;
        s1 += w * k;
        s2 += w * k * (double)k;
    }
    const double r = s1 / hh, r2 = s2 / hh;
    const double tau_rms = sqrt(fmax(r2 - r * r, 0.0));
    const double c = 2.0 * M_PI * tau_rms / LM_N;                       // :31-32, df = 1/Nfft
    const double sig2 = pow(10.0, -0.1 * (double)a.snr_db[blk]);        // 1/snr
    const double t0 = 1.0 + sig2;                                       // diagonal of Rpp

    if (tid < LM_N) {
        const double d = (double)tid;
        const double den = (1.0 + c * c * d * d) * t0;                  // 1/(1 + j c d) = (1 - j c d)/(1 + c^2 d^2)
        t[tid] = tid == 0 ? cd{1.0, 0.0} : cd{1.0 / den, -c * d / den};
    }
    // stage the LS columns of this workgroup's tx antennas
    const int j0 = jc * LM_RHS;
    for (int idx = tid; idx < LM_RHS * LM_N; idx += LM_THREADS) {
        const int jj = idx / LM_N, k = idx - jj * LM_N;
        float2 v = {0.f, 0.f};
        if (j0 + jj < a.nt) {
            const size_t o = (blk * a.nt + j0 + jj) * LM_N + k;
            v = float2{a.h_re[o], a.h_im[o]};
        }
        y[jj][k] = v;
    }
    if (tid == 0) f[0][0] = cd{1.0, 0.0};
    __syncthreads();

    // x[u] <-> solution entry 8u + e of right-hand side jl (normalised system M z = y)
    cd x[LM_EPL];
#pragma unroll
    for (int u = 0; u < LM_EPL; ++u) x[u] = cd{0.0, 0.0};
    if (e == 0) x[0] = cd{(double)y[jl][0].x, (double)y[jl][0].y};

    for (int k = 1; k < LM_N; ++k) {
        const cd* fc = f[(k - 1) & 1];           // length k
        cd* fn = f[k & 1];                       // length k + 1
        // forward error ef = sum_{i<k} t[k-i] fc[i]; error of the solution ex = sum_{i<k} t[k-i] x[i]
        cd ef = {0.0, 0.0}, ex = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < LM_EPL; ++u) {
            const int i = LM_PART * u + e;
            if (i < k) {
                const cd tk = t[k - i];
                ef = cfma(tk, fc[i], ef);
                ex = cfma(tk, x[u], ex);
            }
        }
        ef = group_sum(ef);
        ex = group_sum(ex);
        const double inv = 1.0 / (1.0 - (ef.x * ef.x + ef.y * ef.y));
        // fn = ([fc; 0] - ef [0; conj(reverse(fc))]) / (1 - |ef|^2)
        if (tid <= k) {
            const cd fe = tid < k ? fc[tid] : cd{0.0, 0.0};
            const cd be = tid > 0 ? cconj(fc[k - tid]) : cd{0.0, 0.0};
            const cd m = cmul(ef, be);
            fn[tid] = cd{(fe.x - m.x) * inv, (fe.y - m.y) * inv};
        }
        __syncthreads();
        // x <- [x; 0] + (y_k - ex) * conj(reverse(fn))
        const float2 yk = y[jl][k];
        const cd coef = {(double)yk.x - ex.x, (double)yk.y - ex.y};
#pragma unroll
        for (int u = 0; u < LM_EPL; ++u) {
            const int i = LM_PART * u + e;
            if (i <= k) x[u] = cfma(coef, cconj(fn[k - i]), x[u]);
        }
    }
    __syncthreads();
    // H_mmse = H_ls - (sig2 / t0) z      (R (R + s I)^-1 H = H - s (R + s I)^-1 H, z solves M z = H)
    const double g = sig2 / t0;
#pragma unroll
    for (int u = 0; u < LM_EPL; ++u) {
        const int i = LM_PART * u + e;
        if (i < LM_N) {
            const float2 v = y[jl][i];
            y[jl][i] = float2{(float)((double)v.x - g * x[u].x), (float)((double)v.y - g * x[u].y)};
        }
    }
    __syncthreads();
    for (int idx = tid; idx < LM_RHS * LM_N; idx += LM_THREADS) {
        const int jj = idx / LM_N, k = idx - jj * LM_N;
        if (j0 + jj < a.nt) {
            const size_t o = (blk * a.nt + j0 + jj) * LM_N + k;
            a.o_re[o] = y[jj][k].x;
            a.o_im[o] = y[jj][k].y;
        }
    }
}

}  // namespace csi
