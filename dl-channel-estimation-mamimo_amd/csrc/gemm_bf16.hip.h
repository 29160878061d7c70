// gemm_bf16.hip.h - bf16-operand / fp32-accumulate variant of the dense layers
// (csi_dtype CSI_DTYPE_BF16, BASELINE.json config 3).  Same reference layers as gemm_f32.hip.h
// (massiveMIMO_CSI_prediction_DNN.py:211-227); operands (preambles, weights, hidden activations)
// are rounded to bf16 (round-to-nearest-even), products are exact, accumulation and the
// bias / relu / BatchNormalization epilogue are fp32, the regressor output is fp32.  It cannot
// meet the 1e-5 fp32 contract (8-bit mantissa operands); its error is reported separately.
//
//   * arithmetic: v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate)
//   * operands K-major bf16; LDS stage rows are 64 bf16 = 128 B (8 chunks of 16 B), one
//     ds_read_b128 = one MFMA operand (lane: row l&31, k = 8*(l>>5)..+7); chunk index XOR
//     (row>>1)&7 on the DMA source address and on the read address -> conflict-free
//   * HBM/L2 -> LDS by LDS-DMA into a ring of NS stages, counted vmcnt + raw s_barrier hand-over
//   * tile geometry is a template parameter (waves WM x WN, MI x NJ MFMA tiles per wave)
//   * at bf16 MFMA rates the per-pair h1 cannot be generated per fragment any more (VALU- and
//     LDS-bound); it is materialised once in bf16 by pair_h1_bf16_kernel instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_f32.hip.h"

namespace csi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;

constexpr int B_BK = 64;                      // bf16 k-columns per stage (128-byte rows)
constexpr int B_ROWF = 32;                    // floats per image row (128 B)

__device__ __forceinline__ int b_swz(int row) { return (row >> 1) & 7; }

// round-to-nearest-even fp32 -> bf16 (inputs are finite)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

struct GemmBf16Args {
    const bf16_t* A;       // [M][lda] bf16
    const bf16_t* Bt;      // [N][ldb] bf16, K contiguous, ldb % 64 == 0, columns >= K are zero
    void* C;               // fp32 [M][ldc] (EPI_RAW slabs / final output) or bf16 [M][ldc]
    int M, N, K;
    int lda, ldb, ldc;
    int k_per_split;       // multiple of B_BK
    int tiles_n;
    const float* bias;
    const float* scale;
    const float* shift;
};

template <int P, int NS>
__device__ __forceinline__ void bring_handover(int groups) {
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    if (NS >= 4 && groups >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * P) : "memory");
    else if (NS >= 3 && groups >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int EPI, bool OUT_BF16, int WM, int WN, int MI, int NJ, int NS>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 ? 2 : 1)) void gemm_bf16_kernel(const GemmBf16Args g) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MI * 32, BN = WN * NJ * 32;
    constexpr int STAGE = (BM + BN) * B_ROWF;                  // floats
    constexpr int PT = (BM + BN) / 8;                          // 1-KiB DMA pieces per stage
    constexpr int PW = PT / NW;                                // per wave
    constexpr int D = NS - 1;
    static_assert(PT % NW == 0, "pieces must divide over the waves");
    __shared__ __attribute__((aligned(16))) float lds[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + B_BK - 1) / B_BK;

    // DMA: piece p of a stage = image rows 8p..8p+7 (A rows first, then B rows); lane -> row
    // 8p + lane/8, physical chunk lane%8 holding logical chunk (lane%8) ^ swz(row)
    const bf16_t* src[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int piece = wave * PW + u;
        const int row = 8 * piece + (lane >> 3);
        const int clog = (lane & 7) ^ b_swz(row);
        if (piece < BM / 8) src[u] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + clog * 8 + kbeg;
        else src[u] = g.Bt + (size_t)min(n0 + row - BM, g.N - 1) * g.ldb + clog * 8 + kbeg;
    }
    auto issue = [&](int kt, int u) {
        float* st = lds + (kt % NS) * STAGE + (wave * PW + u) * 256;
        dma16(reinterpret_cast<const float*>(src[u] + kt * B_BK), st);
    };

    // fragment addressing: every fragment row is (multiple of 32) + l31 -> one swizzle value
    const int fswz = b_swz(l31);
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;            // floats
    const int abase = (wm * MI * 32 + l31) * B_ROWF;
    const int bbase = (BM + wn * NJ * 32 + l31) * B_ROWF;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int npro = min(nkt, D);
    for (int t = 0; t < npro; ++t)
#pragma unroll
        for (int u = 0; u < PW; ++u) issue(t, u);

    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const float* st = lds + (kt % NS) * STAGE;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + abase + i * 32 * B_ROWF + xo[c]));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + bbase + j * 32 * B_ROWF + xo[c]));
            if (MORE) {
#pragma unroll
                for (int u = 0; u < PW; ++u)
                    if (u * 4 / PW == c) issue(kt + D, u);          // spread the pieces over the 4 chunks
            }
            if (MI * NJ >= 8) __builtin_amdgcn_s_setprio(1);     // +5 % on the big tiles, -2 % on 2x2
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (MI * NJ >= 8) __builtin_amdgcn_s_setprio(0);
        }
    };
    int kt = 0;
    for (; kt < nkt - D; ++kt) {
        bring_handover<PW, NS>(D - 1);
        ktile(kt, std::true_type{});
    }
    for (; kt < nkt; ++kt) {
        bring_handover<PW, NS>(nkt - 1 - kt);
        ktile(kt, std::false_type{});
    }

    // epilogue (C/D layout as the fp32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const bool full_rows = (m0 + BM) <= g.M;
    const int wrow = m0 + wm * MI * 32 + 4 * hi;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int col = n0 + wn * NJ * 32 + nj * 32 + l31;
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 1);
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (EPI != EPI_RAW) bias = g.bias[colc];
        if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc]; sh = g.shift[colc]; }
        float* cf = reinterpret_cast<float*>(g.C) + (EPI == EPI_RAW ? (size_t)blockIdx.z * g.M * g.ldc : (size_t)0) +
                    (size_t)wrow * g.ldc + col;
        bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + (size_t)wrow * g.ldc + col;
        auto put = [&](int rr, float v) {
            if (EPI == EPI_BIAS) v += bias;
            if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
            if (OUT_BF16) cb[(size_t)rr * g.ldc] = f2bf(v);
            else cf[(size_t)rr * g.ldc] = v;
        };
        if (full_rows) {
            if (cok) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) put(mi * 32 + (r & 3) + 8 * (r >> 2), acc[mi][nj][r]);
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mi * 32 + (r & 3) + 8 * (r >> 2);
                    if (cok && (wrow + rr) < g.M) put(rr, acc[mi][nj][r]);
                }
        }
    }
}

// dst[i] = bf16(src[i]); n8 = number of 8-element groups
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        uint4 o;
        o.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16);
        o.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
        o.z = (uint32_t)f2bf(b[0]) | ((uint32_t)f2bf(b[1]) << 16);
        o.w = (uint32_t)f2bf(b[2]) | ((uint32_t)f2bf(b[3]) << 16);
        reinterpret_cast<uint4*>(dst)[i] = o;
    }
}

// dst[r][0..cols) = bf16(src[r][0..cols)), dst[r][cols..ldd) = 0   (row pitch change for the literal
// network, whose input width 321*nt need not be a multiple of 8)
__global__ void f32_to_bf16_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols, int ldd) {
    const size_t total = (size_t)rows * ldd;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int r = (int)(i / ldd), cidx = (int)(i - (size_t)r * ldd);
        dst[i] = cidx < cols ? f2bf(src[(size_t)r * cols + cidx]) : (bf16_t)0;
    }
}

// h1[(pr,t)][k] = bf16( bn0( relu( L0[pr][k] + T[t][k] ) ) ), 8 columns per thread.
// L0 fp32 [M1][h1] (sum of S slabs), T fp32 [nt][h1], out bf16 [M1*nt][h1]
__global__ void pair_h1_bf16_kernel(const float* __restrict__ L0, int S, size_t slab, const float* __restrict__ T,
                                    const float* __restrict__ s0, const float* __restrict__ t0,
                                    bf16_t* __restrict__ out, int M2, int nt, int h1) {
    const int c8 = h1 >> 3;
    const size_t total = (size_t)M2 * c8;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int m = (int)(i / c8);
        const int k = (int)(i - (size_t)m * c8) * 8;
        const int pr = m / nt, t = m - pr * nt;
        uint32_t o[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 l = *reinterpret_cast<const f32x4*>(L0 + (size_t)pr * h1 + k + 4 * h);
            for (int z = 1; z < S; ++z) l += *reinterpret_cast<const f32x4*>(L0 + z * slab + (size_t)pr * h1 + k + 4 * h);
            const f32x4 tt = *reinterpret_cast<const f32x4*>(T + (size_t)t * h1 + k + 4 * h);
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s0 + k + 4 * h);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(t0 + k + 4 * h);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(fmaxf(l[e] + tt[e], 0.f), sv[e], hv[e]);
            o[2 * h] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            o[2 * h + 1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        }
        *reinterpret_cast<uint4*>(out + (size_t)m * h1 + k) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace csi
