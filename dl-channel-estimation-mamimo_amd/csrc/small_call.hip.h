// small_call.hip.h - the one-packet regime: a call of a handful of rx preambles (massiveMIMO_CSI_prediction_DNN.py:339-346 predicts
// ONE packet per step: Nt*Nr = 128 rows) through BOTH component models in three launches.
//
// Such a call is bound by two things: streaming the weights once (2 x 47 MB at Nt = 32: ~12 us of HBM time) and kernel boundaries
// (each ~5 us on this part).  The general kernels (gemm_f32.hip.h) spend six launches per component model on it - layer 0 as K slabs
// + their sum, split-K GEMM + epilogue for the per-pair layer and again for the regressor - on two streams.  Here:
//
//   small_l0_gemv_kernel     layer 0 of both models (blockIdx.y): L0[m][n] = sum_k x[m][k] Wt[n][k] for m < 8 preambles.  A workgroup owns
//                            4 output columns = 4 contiguous rows of the K-major weights and walks ALL of K: 16-byte loads, every lane its
//                            own k positions, fully coalesced, ~40 loads in flight per lane; the 256 partial sums of a column are combined
//                            in LDS in a fixed order.  No slabs, no second kernel, no atomics: the result is run-to-run identical.
//                            It then writes its 4 columns of the per-pair layer's input for all M * nt pair rows.
//   small_tile_gemm_kernel   every layer behind it, both models (blockIdx.z), on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32
//                            accumulate: the arithmetic of gemm_f32.hip.h) in 16 x 16 output tiles - a one-packet layer of the shipped
//                            model is 1024 of them, one per SIMD of the chip.  A workgroup of 16 waves owns RG row tiles of one column
//                            tile and ALL of K: the waves of a row tile split K between them and add their partial tiles in LDS in k
//                            order - nothing leaves the workgroup half-summed, no slabs, no second kernel.  Operands go
//                            global -> registers directly (32 bytes per lane and operand feed eight MFMA k-steps: lane (i, q) holds
//                            k = 32 j + 8 q + e in step e for BOTH operands, so the k order inside a group of 32 is permuted the same
//                            way on both sides), a batch of groups in flight; bias, relu and the BatchNormalization affine (after the
//                            relu) are fused behind the last MFMA.  The first per-pair layer's input rows
//                            h1 = relu(L0[row / nt] + T[row % nt]) * s0 + t0 (DNN.py:211-219, shared layer 0) are written by the
//                            layer-0 kernel itself - they are element-wise in the column a workgroup of that kernel owns.
//
// Two accumulators per wave (alternating k-steps, added at the end in a fixed order) cover the 40-cycle dependent latency of the
// 32-cycle MFMA.  Reference lines: Dense + relu DNN.py:211-214, BatchNormalization DNN.py:215-219, regressor DNN.py:227.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_f32.hip.h"
#include "ls_estimate.hip.h"

namespace csi {

constexpr int SC_MAX_ROWS0 = 8;          // preambles (packet, rx) a small call may hold

struct SmallL0Args {
    const float* x[2];       // [M][lda] one plane of the preambles per component model
    const float* Wt[2];      // [h1][ldw] K-major layer-0 weights (LTF columns)
    const float* T[2];       // [nt][h1] pilot table incl. the layer-0 bias
    const float* s0[2];      // [h1] BatchNormalization scale of layer 0 (1 without BN)
    const float* t0[2];      // [h1] its shift (a zero vector when the shift lives in the next layer's bias)
    float* h1out[2];         // [M * nt][h1]: the first per-pair layer's input rows relu(L0[m] + T[t]) * s0 + t0, row m * nt + t
    int M, K, lda, ldw, h1, nt;
};

// grid (ceil(h1 / COLS), 2), 256 threads; COLS output columns per workgroup (a multiple of 4), UN k steps of 1024 in flight
template <int MR, int SC_GEMV_COLS, int UN>
__device__ __forceinline__ void small_l0_gemv_body(const SmallL0Args& a, const int wg_x, const int z) {
    __shared__ float red[4][64][MR * SC_GEMV_COLS + 1];
    __shared__ float part[4][MR * SC_GEMV_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wg_x * SC_GEMV_COLS;
    const float* __restrict__ x = a.x[z];
    const float* __restrict__ W = a.Wt[z];
    f32x4 acc[MR][SC_GEMV_COLS];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < SC_GEMV_COLS; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight rows beyond h1 are clamped (their sums are never stored)
    const float* wrow[SC_GEMV_COLS];
#pragma unroll
    for (int c = 0; c < SC_GEMV_COLS; ++c) wrow[c] = W + (size_t)min(n0 + c, a.h1 - 1) * a.ldw;
    for (int k0 = 4 * tid; k0 < a.K; k0 += 1024 * UN) {
        f32x4 w[UN][SC_GEMV_COLS], xv[UN][MR];
        float okfs[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            // branch-free: a quad beyond K (K % 4 == 0: inside or outside as a whole) is fetched from k = 0 and multiplied by 0;
            // rows beyond M repeat row M - 1 (their sums are never stored)
            const int k = k0 + 1024 * u;
            const bool ok = k < a.K;
            const int kc = ok ? k : 0;
            const float okf = ok ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < SC_GEMV_COLS; ++c) w[u][c] = *reinterpret_cast<const f32x4*>(wrow[c] + kc);      // (non-temporal loads measured 20.1 us against 17.0)
            okfs[u] = okf;
#pragma unroll
            for (int m = 0; m < MR; ++m) xv[u][m] = *reinterpret_cast<const f32x4*>(x + (size_t)min(m, a.M - 1) * a.lda + kc);
        }
        __builtin_amdgcn_sched_barrier(0);      // all UN * (4 + MR) loads are requested before the first fma (the scheduler otherwise sinks each to its use)
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int c = 0; c < SC_GEMV_COLS; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][c][e] = fmaf(xv[u][m][e] * okfs[u], w[u][c][e], acc[m][c][e]);
    }
    // fixed-order combination: lane's four k phases, then the 64 lanes of a wave (one thread per value), then the 4 waves
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < SC_GEMV_COLS; ++c)
            red[wave][lane][m * SC_GEMV_COLS + c] = (acc[m][c][0] + acc[m][c][1]) + (acc[m][c][2] + acc[m][c][3]);
    __syncthreads();
    for (int idx = tid; idx < 4 * MR * SC_GEMV_COLS; idx += 256) {
        const int w = idx / (MR * SC_GEMV_COLS), v = idx - w * (MR * SC_GEMV_COLS);
        float s = 0.f;
        for (int l = 0; l < 64; ++l) s += red[w][l][v];
        part[w][v] = s;
    }
    __syncthreads();
    // The layer-0 sum of a column is complete inside this workgroup, and the first per-pair layer's input is element-wise in the
    // column: h1[(m, t)][n] = relu(L0[m][n] + T[t][n]) * s0[n] + t0[n] (DNN.py:211-219 on the shared layer 0).  So the workgroup writes
    // its 4 columns of all M * nt pair rows itself - 16 bytes per row - and the per-pair layer reads a plain matrix: generated inside
    // that layer's tiles instead, every one of its 64 column tiles fetched L0, T and both BatchNormalization vectors again (4 x the
    // operand traffic through the vector memory path: 31 us per one-packet layer against 17 for this whole kernel).
    __shared__ float l0s[MR][SC_GEMV_COLS];
    if (tid < MR * SC_GEMV_COLS) l0s[tid / SC_GEMV_COLS][tid % SC_GEMV_COLS] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    const int rows = a.M * a.nt;
    constexpr int Q = SC_GEMV_COLS / 4;          // 16-byte quads per row of this workgroup's columns
    for (int idx = tid; idx < rows * Q; idx += 256) {
        const int r = idx / Q, qd = idx - r * Q, n = n0 + 4 * qd;
        if (n >= a.h1) continue;                 // (h1 % 4 == 0 by csi_create: a quad is inside or outside as a whole)
        const int m = r / a.nt, t = r - m * a.nt;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.s0[z] + n), sh = *reinterpret_cast<const f32x4*>(a.t0[z] + n);
        const f32x4 tv = *reinterpret_cast<const f32x4*>(a.T[z] + (size_t)t * a.h1 + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(fmaxf(l0s[m][4 * qd + e] + tv[e], 0.f), sc[e], sh[e]);
        *reinterpret_cast<f32x4*>(a.h1out[z] + (size_t)r * a.h1 + n) = v;
    }
}

template <int MR, int SC_GEMV_COLS, int UN>
__global__ __launch_bounds__(256) void small_l0_gemv_kernel(SmallL0Args a) {
    small_l0_gemv_body<MR, SC_GEMV_COLS, UN>(a, blockIdx.x, blockIdx.y);
}

// Round 6: the LS estimate of a one-packet call IN THE SAME LAUNCH as layer 0.  The two are independent (the same preambles feed both,
// generate_maMIMO_LTF.m:336-349), the LS kernel of such a call is 4 workgroups of latency (8.9 us) and the weight stream leaves the CUs
// idle enough: workgroups (x < ls_blocks, y = 0) run the Walsh-Hadamard LS body (ls_estimate.hip.h), the others layer 0 of component model y.
// One launch and one boundary less per call; both bodies are the functions the separate kernels call, so the results are the same bits.
template <int NT, int MR>
__global__ __launch_bounds__(256) void small_l0_ls_kernel(SmallL0Args a, LsArgs la, int nblk, int ls_blocks) {
    if ((int)blockIdx.x < ls_blocks) {
        if (blockIdx.y == 0) ls_fwht2_body<NT, 1, 8, (NT == 64 ? 3 : 1), false, false, true>(la, nblk, blockIdx.x, (unsigned)ls_blocks);
        return;
    }
    small_l0_gemv_body<MR, 4, 2>(a, (int)blockIdx.x - ls_blocks, blockIdx.y);
}

struct SmallGemmArgs {
    const float* A[2];       // [M][lda] (the first per-pair layer reads the rows small_l0_gemv_kernel wrote)
    const float* Bt[2];      // [N][ldb] K-major weights, zero padded in K to a multiple of 32
    const float* bias[2];    // [N]
    const float* scale[2];   // [N] EPI_BIAS_RELU_AFFINE
    const float* shift[2];   // [N]
    float* C[2];             // [M][ldc]; EPI_H1: [M * nt][ldc]
    int M, N, K, lda, ldb, ldc;
    // EPI_H1 (layer 0 of a call of 9 ... 64 preambles): the tile is the layer-0 LTF product L0[m][n]; what is written is the first
    // per-pair layer's input, rows m * nt + t = relu(L0[m][n] + T[t][n]) * s0[n] + t0[n] for every tx antenna t (DNN.py:211-219)
    const float* T[2];       // [nt][N] pilot table incl. the layer-0 bias
    const float* s0[2];      // [N]
    const float* t0[2];      // [N]
    int nt;
};
constexpr int EPI_H1 = 3;

// grid (ceil(N / 16), ceil(M / (16 RG)), 2), 1024 threads = 16 waves: wave w -> row group w % RG (16 rows), k part w / RG of KS = 16 / RG.
// Every wave walks its part of K with its loads for several groups of 16 k in flight; four waves share a SIMD, so one wave's MFMAs run
// under the others' load latency (a single wave per tile over the whole K measured 32 us per one-packet layer: each batch of loads was a
// full L2 / HBM round trip with nothing beside it).  The KS partial tiles of a row group meet in LDS and are added in k order by the
// wave that holds part 0 - inside the workgroup, in a fixed order, no global slabs.  RG is chosen by the host so that a layer is
// ~256 workgroups: 4 for the 64 column tiles of the shipped per-pair layer, 1 for the regressor's 15.
template <int EPI, int RG, int UB>
__global__ __launch_bounds__(1024) void small_tile_gemm_kernel(SmallGemmArgs g) {
    constexpr int KS = 16 / RG;
    __shared__ f32x4 part[16][64];
    const int z = blockIdx.z, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: scalar k offsets below)
    const int rg = wave % RG, kq = wave / RG;
    const int i = lane & 15, q = lane >> 4;
    const int row0 = ((int)blockIdx.y * RG + rg) * 16;
    const int m = min(row0 + i, g.M - 1);                                        // this lane's A row (clamped; clamped rows are never stored)
    const int n = min((int)blockIdx.x * 16 + i, g.N - 1);                        // this lane's B column
    // k layout: groups of 32 k; lane (i, q) owns the 8 consecutive k = 32 j + 8 q .. + 7 of its row / column (two 16-byte loads per operand
    // and group) and supplies element e of them in MFMA step e - the same permutation on both operands, so every k meets its partner.
    // The four q lanes of a row read 128 contiguous bytes: whole cache lines per request (with 16-k groups a request took 64 bytes of
    // each of 16 lines and left the other halves to a later iteration, by when 16 waves' worth of requests had evicted them).
    const float* __restrict__ bp = g.Bt[z] + (size_t)n * g.ldb + 8 * q;
    const float* __restrict__ ap = g.A[z] + (size_t)m * g.lda + 8 * q;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // A batch = UB groups: all of its 4 UB loads are requested before its first MFMA (no branch inside a batch), 8 UB MFMAs follow;
    // four waves share a SIMD, so one wave's MFMAs run under the others' round trips.  The host picks UB = 4 / 2 / 1 from the groups a
    // wave owns; the few groups behind the last whole batch go one at a time.
    const int ngroups = (g.K + 31) >> 5;        // the K tail multiplies zero-padded weight columns (ldb % 32 == 0); A-side buffers carry zeroed slack
    const int per = (ngroups + KS - 1) / KS;
    const int jb = kq * per, je = min(ngroups, jb + per);
    auto steps = [&](const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc0, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc0, 0, 0, 0);
        }
    };
    if (row0 < g.M) {                           // (a wave without rows still meets the barrier below)
        int j0 = jb;
        for (; j0 + UB <= je; j0 += UB) {
            const float* __restrict__ bq = bp + 32 * j0;
            const float* __restrict__ aq = ap + 32 * j0;
            f32x4 a[UB][2], b[UB][2];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                b[u][0] = *reinterpret_cast<const f32x4*>(bq + 32 * u);
                b[u][1] = *reinterpret_cast<const f32x4*>(bq + 32 * u + 4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                a[u][0] = *reinterpret_cast<const f32x4*>(aq + 32 * u);
                a[u][1] = *reinterpret_cast<const f32x4*>(aq + 32 * u + 4);
            }
            // nothing crosses: without it the scheduler sinks every load to just in front of its MFMAs (34 registers, two loads in
            // flight, a vmcnt wait per four MFMAs) - the opposite of what a latency-bound wave needs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UB; ++u) steps(a[u][0], a[u][1], b[u][0], b[u][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; j0 < je; ++j0) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + 32 * j0), b1 = *reinterpret_cast<const f32x4*>(bp + 32 * j0 + 4);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + 32 * j0), a1 = *reinterpret_cast<const f32x4*>(ap + 32 * j0 + 4);
            steps(a0, a1, b0, b1);
        }
    }
    part[wave][lane] = acc0 + acc1;
    __syncthreads();
    if (kq != 0 || row0 >= g.M) return;
    f32x4 sum = part[rg][lane];
#pragma unroll
    for (int k = 1; k < KS; ++k) sum += part[k * RG + rg][lane];                  // k order: fixed
    // C/D layout: column = lane & 15, row = 4 (lane >> 4) + r
    const int col = (int)blockIdx.x * 16 + i;
    if (col >= g.N) return;
    if constexpr (EPI == EPI_H1) {
        const float sc = g.s0[z][col], sh = g.t0[z][col];
        for (int t = 0; t < g.nt; ++t) {
            const float tv = g.T[z][(size_t)t * g.N + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * q + r;
                if (row < g.M) g.C[z][((size_t)row * g.nt + t) * g.ldc + col] = fmaf(fmaxf(sum[r] + tv, 0.f), sc, sh);
            }
        }
        return;
    }
    const float bias = g.bias[z][col];
    float sc = 1.f, sh = 0.f;
    if constexpr (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[z][col]; sh = g.shift[z][col]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * q + r;
        if (row >= g.M) continue;
        float v = sum[r] + bias;
        if constexpr (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v, 0.f), sc, sh);
        g.C[z][(size_t)row * g.ldc + col] = v;
    }
}

// The same scheme on 32 x 32 output tiles (v_mfma_f32_32x32x2_f32): twice the flops per operand byte.  The 16 x 16 form of a one-packet
// per-pair layer moves 512 KB per CU from the L2 (every activation row is fetched by all 64 column tiles, every weight row by both row
// halves) and measures 17 us, of which 6 each are the A and the B loads and none the MFMAs (CSI_SMALL_DBG ablation) -
// it is bound by the L2 -> CU path at ~13 TB/s for the chip.  Here: grid (ceil(N / 32), ceil(M / (32 RG)), 2), 1024 threads; wave w ->
// row tile w % RG, k part w / RG of KS = 16 / RG; groups of 32 k, lane (i, q) owns k = 32 j + 16 q .. + 15 of its row / column (four
// 16-byte loads per operand: the two q lanes of a row read 128 contiguous bytes) and supplies element e in MFMA step e of 16.  The KS
// partial tiles meet in LDS (64 KB) and every wave of a row tile adds and finishes 16 / KS of its 16 accumulator registers, k order fixed.
template <int EPI, int RG, int UB>
__global__ __launch_bounds__(1024) void small_tile32_gemm_kernel(SmallGemmArgs g) {
    constexpr int KS = 16 / RG;
    __shared__ float part[16][16][64];
    const int z = blockIdx.z, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave % RG, kq = wave / RG;
    const int i = lane & 31, q = lane >> 5;
    const int row0 = ((int)blockIdx.y * RG + rg) * 32;
    const int m = min(row0 + i, g.M - 1);
    const int n = min((int)blockIdx.x * 32 + i, g.N - 1);
    const float* __restrict__ bp = g.Bt[z] + (size_t)n * g.ldb + 16 * q;
    const float* __restrict__ ap = g.A[z] + (size_t)m * g.lda + 16 * q;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ngroups = (g.K + 31) >> 5;
    const int per = (ngroups + KS - 1) / KS;
    const int jb = kq * per, je = min(ngroups, jb + per);
    auto steps = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v][e], b[v][e], acc, 0, 0, 0);
    };
    if (row0 < g.M) {
        int j0 = jb;
        for (; j0 + UB <= je; j0 += UB) {
            f32x4 a[UB][4], b[UB][4];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) b[u][v] = *reinterpret_cast<const f32x4*>(bp + 32 * (j0 + u) + 4 * v);
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) a[u][v] = *reinterpret_cast<const f32x4*>(ap + 32 * (j0 + u) + 4 * v);
            __builtin_amdgcn_sched_barrier(0);          // every load of the batch is requested before its first MFMA (see the 16 x 16 form)
#pragma unroll
            for (int u = 0; u < UB; ++u) steps(a[u], b[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; j0 < je; ++j0) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { b[v] = *reinterpret_cast<const f32x4*>(bp + 32 * j0 + 4 * v); a[v] = *reinterpret_cast<const f32x4*>(ap + 32 * j0 + 4 * v); }
            steps(a, b);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    if (row0 >= g.M) return;
    // C/D layout of the 32 x 32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  This wave finishes registers
    // kq * RG .. + RG - 1 of its row tile.
    const int col = (int)blockIdx.x * 32 + i;
    if (col >= g.N) return;
    const float bias = g.bias[z][col];
    float sc = 1.f, sh = 0.f;
    if constexpr (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[z][col]; sh = g.shift[z][col]; }
#pragma unroll
    for (int t = 0; t < RG; ++t) {
        const int r = kq * RG + t;
        float sum = part[rg][r][lane];
#pragma unroll
        for (int k = 1; k < KS; ++k) sum += part[k * RG + rg][r][lane];          // k order: fixed
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * q;
        if (row >= g.M) continue;
        float v = sum + bias;
        if constexpr (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v, 0.f), sc, sh);
        g.C[z][(size_t)row * g.ldc + col] = v;
    }
}

}  // namespace csi
