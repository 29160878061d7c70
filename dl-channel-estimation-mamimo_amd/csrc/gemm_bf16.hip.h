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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
    const unsigned long long* stamps = nullptr;   // timing probe (tools/bf16_pair_probe.hip), null otherwise: pp_stamp
};

// Optional per-workgroup time stamps (timing probes; null in the library): wave 0 writes (shader cycles, 10-ns wall ticks)
// at kernel entry, after the prologue, after the main loop and after the epilogue - slots 0..3 of 6 x 16 bytes per workgroup.
__device__ __forceinline__ void pp_stamp(const unsigned long long* base, int i) {
    if (!base) return;
    if (threadIdx.x == 0) {
        unsigned long long* p = const_cast<unsigned long long*>(base) + ((size_t)blockIdx.x * 6 + i) * 2;
        p[0] = __builtin_readcyclecounter();
        p[1] = wall_clock64();
    }
}

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

// ---------------------------------------------------------------------------------------------
// 256x256 "ping-pong" kernel for the large bf16 grids.
//
// At bf16 rates the lock-step kernel above loses the matrix pipe whenever its waves issue LDS-DMA
// pieces (60-185 cycles each, 8 per wave and 64-column k-tile) or wait at the one barrier per
// k-tile.  Here the 8 waves form two groups of four (wm = 0 / 1, one wave of each group on every
// SIMD) that run the SAME instruction stream half a phase apart:
//
//     group 0:  load | MFMA | load | MFMA | ...         a phase = 16 k-columns of the wave's
//     group 1:       | load | MFMA | load | MFMA ...     128x64 output (6 ds_read_b128, 8 MFMAs)
//
// Every segment ends in an s_barrier, group 1 executes one extra barrier up front (and skips its
// last one), so a wave's fragment reads and DMA issue always run beside the other group's 8
// MFMAs (256 pipe cycles) on the same SIMD.
//
// LDS: ring of NSUB sub-tiles of 32 k-columns ((256 + 256) rows x 64 B = 32 KiB each, XOR chunk
// swizzle (row>>2)&3 on the DMA source and the read address).  Sub-tile u + NSUB - 1 is issued
// while sub-tile u is read (2 pieces per wave and phase).  Ordering (local step = segment index
// of the wave, group 1 one step behind group 0):
//   RAW  every wave counts its own pieces with vmcnt in the SECOND load segment of sub-tile u
//        (sub-tile u + 1 complete, the 4 (NSUB - 2) younger pieces stay in flight) and two
//        barriers lie between that wait and the first read by either group;
//   WAR  slot (u - 1) % NSUB is re-filled from the first load segment of sub-tile u; its last
//        reads (second load segment of sub-tile u - 1) were retired by the lgkmcnt(0) in front
//        of that segment's barrier, one (group 0) or two (group 1) barriers earlier.
constexpr int PP_BM = 256, PP_BN = 256;
constexpr int PP_BK = 32;                            // bf16 k-columns per sub-tile
constexpr int PP_ROWF = 16;                          // floats per image row (64 B)
constexpr int PP_SUBF = (PP_BM + PP_BN) * PP_ROWF;   // floats per sub-tile
constexpr int PP_PW = 4;                             // 1-KiB DMA pieces per wave and sub-tile
constexpr int PP_THREADS = 512;

// s_waitcnt through the builtin (gfx9 encoding: vmcnt = [15:14|3:0], expcnt = [6:4], lgkmcnt = [11:8]) so
// that the compiler's own wait-count bookkeeping sees it; an asm statement is opaque to it and it
// then re-waits for every fragment read in front of the MFMAs that were meant to cover them
template <int N>
__device__ __forceinline__ void pp_wait_vm_lgkm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_wait_lgkm() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_wait_vm_lgkm_rt(int n) {
    if (n >= 12) pp_wait_vm_lgkm<12>();
    else if (n >= 8) pp_wait_vm_lgkm<8>();
    else if (n >= 4) pp_wait_vm_lgkm<4>();
    else pp_wait_vm_lgkm<0>();
}
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// grid: 8 * tiles_n * ceil(tiles_m / 8) workgroups; workgroup b runs on XCD b % 8, and the tiles_n
// column tiles of a row tile are consecutive on ONE XCD (A rows enter that L2 once)
__host__ __forceinline__ unsigned pp_grid(int tiles_m, int tiles_n) { return 8u * tiles_n * ((tiles_m + 7) / 8); }

// Epilogue of the 256x256 ping-pong kernels: 8 waves as 2 (rows) x 4 (columns), 128x64 per wave.
// bf16 output goes through a wave-private LDS image so that every lane stores 16 contiguous bytes
// (2-byte stores of the C/D layout cost 8x the store instructions).  The ring is idle by then: every
// DMA has landed and every fragment read was retired in front of a barrier the wave has passed.
// Image: [64 row pairs][64 columns] words, word = (row 2P | row 2P+1 << 16) of one column - the C/D
// layout holds rows 2P, 2P+1 of a column in adjacent accumulator registers, so packing needs no
// cross-lane traffic.  fp32 output is stored straight from the C/D layout (col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)): 32 lanes x 4 B = one 128-byte line per row; EPI_RAW writes
// split-K slab blockIdx.z.
template <int EPI, bool OUT_BF16>
__device__ __forceinline__ void pp_epilogue(f32x16 (&acc)[4][2], const GemmBf16Args& g, float* lds, int m0, int n0, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool full_rows = (m0 + PP_BM) <= g.M;
    if constexpr (OUT_BF16) {
        uint32_t* ep = reinterpret_cast<uint32_t*>(lds) + wave * 4096;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int colc = min(n0 + wn * 64 + nj * 32 + l31, g.N - 1);
            float bias = 0.f, sc = 1.f, sh = 0.f;
            if (EPI != EPI_RAW) bias = g.bias[colc];
            if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc]; sh = g.shift[colc]; }
            auto fin = [&](float v) {
                if (EPI == EPI_BIAS) v += bias;
                if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                return v;
            };
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const f32x2 v = {fin(acc[mi][nj][2 * r2]), fin(acc[mi][nj][2 * r2 + 1])};
                    const int P = mi * 16 + 4 * (r2 >> 1) + 2 * hi + (r2 & 1);
                    ep[P * 64 + nj * 32 + l31] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cg = lane & 7;
        const int col8 = n0 + wn * 64 + cg * 8;
        bf16_t* cb = reinterpret_cast<bf16_t*>(g.C) + col8;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int P = pass * 8 + (lane >> 3);
            const uint4 w0 = *reinterpret_cast<const uint4*>(ep + P * 64 + cg * 8);
            const uint4 w1 = *reinterpret_cast<const uint4*>(ep + P * 64 + cg * 8 + 4);
            uint4 e, o;
            e.x = (w0.x & 0xffffu) | (w0.y << 16);  o.x = (w0.x >> 16) | (w0.y & 0xffff0000u);
            e.y = (w0.z & 0xffffu) | (w0.w << 16);  o.y = (w0.z >> 16) | (w0.w & 0xffff0000u);
            e.z = (w1.x & 0xffffu) | (w1.y << 16);  o.z = (w1.x >> 16) | (w1.y & 0xffff0000u);
            e.w = (w1.z & 0xffffu) | (w1.w << 16);  o.w = (w1.z >> 16) | (w1.w & 0xffff0000u);
            const int row = m0 + wm * 128 + 2 * P;
            if (full_rows) {
                if (col8 < g.N) {
                    *reinterpret_cast<uint4*>(cb + (size_t)row * g.ldc) = e;
                    *reinterpret_cast<uint4*>(cb + (size_t)(row + 1) * g.ldc) = o;
                }
            } else {
                if (col8 < g.N && row < g.M) *reinterpret_cast<uint4*>(cb + (size_t)row * g.ldc) = e;
                if (col8 < g.N && row + 1 < g.M) *reinterpret_cast<uint4*>(cb + (size_t)(row + 1) * g.ldc) = o;
            }
        }
    } else {
        // fp32 output straight from the C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)):
        // 32 lanes x 4 B = one 128-byte line per row
        const int wrow = m0 + wm * 128 + 4 * hi;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int col = n0 + wn * 64 + nj * 32 + l31;
            const bool cok = col < g.N;
            const int colc = min(col, g.N - 1);
            float bias = 0.f, sc = 1.f, sh = 0.f;
            if (EPI != EPI_RAW) bias = g.bias[colc];
            if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc]; sh = g.shift[colc]; }
            float* cf = reinterpret_cast<float*>(g.C) + (EPI == EPI_RAW ? (size_t)blockIdx.z * g.M * g.ldc : (size_t)0) +
                        (size_t)wrow * g.ldc + col;
            auto put = [&](int rr, float v) {
                if (EPI == EPI_BIAS) v += bias;
                if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                cf[(size_t)rr * g.ldc] = v;
            };
            if (full_rows) {
                if (cok) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) put(mi * 32 + (r & 3) + 8 * (r >> 2), acc[mi][nj][r]);
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mi * 32 + (r & 3) + 8 * (r >> 2);
                        if (cok && (wrow + rr) < g.M) put(rr, acc[mi][nj][r]);
                    }
            }
        }
    }
}

template <int EPI, bool OUT_BF16, int NSUB, int DBG = 0>
__global__ __launch_bounds__(PP_THREADS, 1) void gemm_bf16_pp_kernel(const GemmBf16Args g) {
    static_assert(NSUB == 4 || NSUB == 5, "ring depth");
    constexpr int D = NSUB - 1;
    __shared__ __attribute__((aligned(16))) float lds[NSUB * PP_SUBF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tm = (idx / g.tiles_n) * 8 + xcd, tn = idx % g.tiles_n;
    if (tm * PP_BM >= g.M) return;
    const int m0 = tm * PP_BM, n0 = tn * PP_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nsub = (kend - kbeg + PP_BK - 1) / PP_BK;

    // DMA piece = 16 image rows; lane -> row 16*piece + lane/4, physical chunk lane%4.  Waves 0-3
    // carry the A rows, waves 4-7 the B rows: one buffer descriptor per wave (tile base, SGPRs),
    // a 32-bit byte offset per lane and piece, the k advance in the scalar offset - an issue is
    // s_mov m0 + buffer_load ... lds, no vector address arithmetic in the loop.
    const bool a_side = wave < 4;
    const bf16_t* tile_base = a_side ? g.A + (size_t)m0 * g.lda + kbeg : g.Bt + (size_t)n0 * g.ldb + kbeg;
    const int ld = a_side ? g.lda : g.ldb;
    const int rmax = a_side ? g.M - 1 - m0 : g.N - 1 - n0;          // last valid row of this tile
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tile_base, 0, 0x7fffffff, 0x00020000);
    int voff[PP_PW];
#pragma unroll
    for (int u = 0; u < PP_PW; ++u) {
        const int row = 16 * ((wave & 3) * PP_PW + u) + (lane >> 2);
        const int clog = (lane & 3) ^ ((row >> 2) & 3);
        voff[u] = (min(row, rmax) * ld + clog * 8) * 2;
    }
    auto issue = [&](int sub, int slot, int u) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + slot * PP_SUBF + (wave * PP_PW + u) * 256),
                                                 16, voff[u], sub * (PP_BK * 2), 0, 0);
    };

    const int fswz = (l31 >> 2) & 3;
    int xo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;            // floats
    const int abase = (wm * 128 + l31) * PP_ROWF;
    const int bbase = (PP_BM + wn * 64 + l31) * PP_ROWF;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragments of the phase being multiplied / of the next phase (read during the MFMAs)
    bf16x8 fa[2][4], fb[2][2];
    auto read_frags = [&](int slot, int c, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
        const float* st = lds + slot * PP_SUBF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + abase + i * 32 * PP_ROWF + xo[c]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
            b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + bbase + j * 32 * PP_ROWF + xo[c]));
    };

    const int npro = min(nsub, D);
    for (int t = 0; t < npro; ++t)
#pragma unroll
        for (int u = 0; u < PP_PW; ++u) issue(t, t, u);
    pp_wait_vm_lgkm_rt(PP_PW * (npro - 1));
    pp_barrier();                       // sub-tile 0 visible to every wave
    read_frags(0, 0, fa[0], fb[0]);
    pp_wait_lgkm();
    if (wm == 1) pp_barrier();          // group 1 runs one segment behind

    // MFMA segment: 8 MFMAs on the current fragments with the 6 fragment reads of the next phase
    // issued in their shadow; the reads are retired before the closing barrier (WAR rule above)
    auto mfma_seg = [&](int cur, bool more, int nslot, int nc, bool closing_barrier) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (more && !(DBG & 2)) read_frags(nslot, nc, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        if (more && !(DBG & 2)) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        pp_wait_lgkm();
        __builtin_amdgcn_sched_barrier(0);
        if (closing_barrier) pp_barrier();
    };

    int slot = 0, fill = D % NSUB;       // slot read now / slot that sub-tile u + D goes to
    int u = 0;
    for (; u < nsub - D; ++u) {
        const int nslot = slot + 1 == NSUB ? 0 : slot + 1;
        // phase (u, 0): sub-tile u + 1 must have landed before any wave reads it in the next MFMA segment but one
        if (!(DBG & 1)) { issue(u + D, fill, 0); issue(u + D, fill, 1); }
        if (!(DBG & 4)) pp_wait_vm_lgkm<(DBG & 1) ? 0 : PP_PW * (D - 2) + 2>();
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        mfma_seg(0, true, slot, 1, true);
        // phase (u, 1)
        if (!(DBG & 1)) { issue(u + D, fill, 2); issue(u + D, fill, 3); }
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        mfma_seg(1, true, nslot, 0, true);
        slot = nslot;
        fill = fill + 1 == NSUB ? 0 : fill + 1;
    }
    for (; u < nsub; ++u) {
        const int r = nsub - 1 - u;      // sub-tiles still to come after this one
        const int nslot = slot + 1 == NSUB ? 0 : slot + 1;
        if (r >= 1) pp_wait_vm_lgkm_rt(PP_PW * (r - 1));
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        mfma_seg(0, true, slot, 1, true);
        pp_barrier();
        mfma_seg(1, r >= 1, nslot, 0, !(wm == 1 && r == 0));
        slot = nslot;
    }

    if ((DBG & 8) && g.M > 0) return;
    pp_epilogue<EPI, OUT_BF16>(acc, g, lds, m0, n0, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel of the FIRST per-pair layer with the A operand generated in the kernel:
//     A[(pr,t)][k] = bf16( relu( L0[pr][k] + T[t][k] ) )        (fp32 in, rounded once; bn0 sits in the weights)
// so h1 never exists in HBM (pair_h1_bf16_kernel wrote 2.6 GB per launch at config 3 and this
// kernel read it back).  Same phase structure as gemm_bf16_pp_kernel; differences:
//   * every wave owns two A pieces and two B pieces of a sub-tile.  B pieces are LDS-DMA'd 3
//     sub-tiles ahead (one per phase).  For the A pieces the wave loads the L0 / T values of
//     sub-tile u+2 into registers in the first load segment of sub-tile u (8 x 16 B, L2-resident
//     operands) and, in the second, turns them into 2 x 16 B of bf16 and writes them into the A
//     image with ds_write_b128 (lane-linear, i.e. the layout the DMA would have produced).
//   * one vmcnt per sub-tile: vmcnt(1) in front of the generation (the loads it needs are older
//     than the single DMA issued behind them).  Memory operations retire in order, so that wait
//     also retires every older B piece - which is what the hand-over of B sub-tile u+1 needs.
// RAW: the A image of sub-tile u+2 is written (lgkmcnt(0), barrier) in load segment (u,1) and first
// read in MFMA segment (u+1,1) - four barriers later for either group.  WAR: its slot last held
// sub-tile u-2, whose final reads were retired two sub-tiles ago.
// Vector-memory operations of the fused kernel are inline asm, counted by hand.  hipcc puts an s_waitcnt vmcnt in front
// of every ds_write (and every asm statement) that follows a BUILTIN LDS-DMA still in flight - it cannot prove the two
// do not alias - which made every phase wait for the B piece issued one phase earlier: the three-sub-tile prefetch
// distance did not exist and a phase lasted one L2 round trip (0.65 us against 0.15 us of MFMA work).  With the DMA
// hidden from the compiler its own count for the A-side loads would be short by the hidden pieces (an s_waitcnt is a
// threshold on ONE counter), so those loads are asm as well and every use is preceded by a counted wait that names the
// registers it releases ("+v": the compiler cannot move the use in front of it).
__device__ __forceinline__ f32x4 pp_gload16(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
__device__ __forceinline__ void pp_gdma16(const void* g_lane_src, uint32_t lds_byte_off) {
    lds_byte_off = __builtin_amdgcn_readfirstlane(lds_byte_off);       // wave-uniform by construction; the "s" operand needs an SGPR
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_lane_src), "s"(lds_byte_off) : "memory");
}
template <int N>
__device__ __forceinline__ void pp_wait_vm_dep(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void pp_wait_vm_dep(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ void pp_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int PPP_RING_FLOATS = 4 * PP_SUBF;           // LDS ring of the fused kernel (4 sub-tiles)

struct PairSrc {
    const float* L0;       // [M / nt][ldl] fp32 (single slab)
    const float* T;        // [nt][ldl] fp32 pilot table incl. bias
    int ldl, nt;
};

// CAST = true: the plain variant of the same machinery for layer 0 - A[m][k] = bf16(X[m][k]) from the
// fp32 preambles (ps.L0 = X, ps.ldl = its row pitch; no T / bn0), split-K over blockIdx.z - which
// removes the separate cast pass over the inputs.
template <int EPI, bool OUT_BF16, bool CAST = false>
__global__ __launch_bounds__(PP_THREADS, 1) void gemm_bf16_pp_pair_kernel(const GemmBf16Args g, const PairSrc ps) {
    constexpr int NSUB = 4, D = 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // NSUB * PP_SUBF ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // pair mode: a COLUMN tile per XCD - its slice of the weights stays resident in that L2 (the A side is 32 KB per row
    // tile); layer 0 (CAST, 10+ MB of preambles per row tile): row tiles per XCD (gemm_hs.hip.h: hs_tile_of_block)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const bool cols = !CAST && (8 % g.tiles_n) == 0;
    const int tm = cols ? idx * (8 / g.tiles_n) + xcd / g.tiles_n : (idx / g.tiles_n) * 8 + xcd;
    const int tn = cols ? xcd % g.tiles_n : idx % g.tiles_n;
    if (tm * PP_BM >= g.M) return;
    pp_stamp(g.stamps, 0);
    const int m0 = tm * PP_BM, n0 = tn * PP_BN;
    const int kbeg = CAST ? blockIdx.z * g.k_per_split : 0;
    const int kend = CAST ? min(g.K, kbeg + g.k_per_split) : g.K;
    const int nsub = (kend - kbeg + PP_BK - 1) / PP_BK;

    // ---- B side: pieces 2w, 2w+1 of the 16 B pieces of a sub-tile
    const bf16_t* bsrc[2];                              // this lane's 16 bytes of B piece u, sub-tile 0
    const uint32_t lds_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds);
    // ---- A side: pieces 2w, 2w+1 of the 16 A pieces; this lane's row and 8-column group
    const float* lrow[2];
    const float* trow[2];
    const int clog = (lane & 3) ^ ((lane >> 4) & 3);          // (row >> 2) & 3 == (lane >> 4) & 3 for every piece
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = 16 * (2 * wave + u) + (lane >> 2);
        bsrc[u] = g.Bt + (size_t)(n0 + min(row, g.N - 1 - n0)) * g.ldb + kbeg + clog * 8;
        const int m = min(m0 + row, g.M - 1);
        if (CAST) {
            lrow[u] = ps.L0 + (size_t)m * ps.ldl + kbeg + clog * 8;
            trow[u] = lrow[u];
        } else {
            const int pr = m / ps.nt, t = m - pr * ps.nt;
            lrow[u] = ps.L0 + (size_t)pr * ps.ldl + clog * 8;
            trow[u] = ps.T + (size_t)t * ps.ldl + clog * 8;
        }
    }
    auto issue_b = [&](int sub, int slot, int u) {
        pp_gdma16(bsrc[u] + sub * PP_BK, lds_off + (uint32_t)(slot * PP_SUBF + (PP_BM / 16 + 2 * wave + u) * 256) * 4u);
    };
    f32x4 lv[2][2], tv[2][2];
    auto load_a = [&](int sub, int u) {                 // L0 / T values of A piece u of sub-tile sub
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lv[u][h] = pp_gload16(lrow[u] + sub * PP_BK + 4 * h);
            if (!CAST) tv[u][h] = pp_gload16(trow[u] + sub * PP_BK + 4 * h);
        }
    };
    auto wait_a = [&](int u, auto n_tag) {              // at most N younger operations stay in flight; releases piece u's values
        constexpr int N = decltype(n_tag)::value;
        if (CAST) pp_wait_vm_dep<N>(lv[u][0], lv[u][1]);
        else pp_wait_vm_dep<N>(lv[u][0], lv[u][1], tv[u][0], tv[u][1]);
    };
    auto gen_a = [&](int sub, int slot, int u) {        // -> 16 B of bf16 per lane, lane-linear in the A image
        if (CAST) {
            uint4 o;
            o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{lv[u][0][0], lv[u][0][1]}), bf16x2));
            o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{lv[u][0][2], lv[u][0][3]}), bf16x2));
            o.z = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{lv[u][1][0], lv[u][1][1]}), bf16x2));
            o.w = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{lv[u][1][2], lv[u][1][3]}), bf16x2));
            *reinterpret_cast<uint4*>(lds + slot * PP_SUBF + (2 * wave + u) * 256 + lane * 4) = o;
            return;
        }
        // bn0 is folded into the next layer's weights at load time (csi_load_weights): A = bf16(relu(L0 + T))
        const f32x4 a0 = lv[u][0] + tv[u][0], a1 = lv[u][1] + tv[u][1];
        uint4 o;
        o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{fmaxf(a0[0], 0.f), fmaxf(a0[1], 0.f)}), bf16x2));
        o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{fmaxf(a0[2], 0.f), fmaxf(a0[3], 0.f)}), bf16x2));
        o.z = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{fmaxf(a1[0], 0.f), fmaxf(a1[1], 0.f)}), bf16x2));
        o.w = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{fmaxf(a1[2], 0.f), fmaxf(a1[3], 0.f)}), bf16x2));
        *reinterpret_cast<uint4*>(lds + slot * PP_SUBF + (2 * wave + u) * 256 + lane * 4) = o;
    };

    const int fswz = (l31 >> 2) & 3;
    int xo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;
    const int abase = (wm * 128 + l31) * PP_ROWF;
    const int bbase = (PP_BM + wn * 64 + l31) * PP_ROWF;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16x8 fa[2][4], fb[2][2];
    auto read_frags = [&](int slot, int c, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
        const float* st = lds + slot * PP_SUBF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + abase + i * 32 * PP_ROWF + xo[c]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
            b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(st + bbase + j * 32 * PP_ROWF + xo[c]));
    };
    auto mfma_seg = [&](int cur, bool more, int nslot, int nc, bool closing_barrier) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (more) read_frags(nslot, nc, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        pp_wait_lgkm();
        __builtin_amdgcn_sched_barrier(0);
        if (closing_barrier) pp_barrier();
    };

    // ---- prologue: B sub-tiles 0..2 in flight, A sub-tiles 0 and 1 generated synchronously
    const int npro = min(nsub, D);
    for (int t = 0; t < npro; ++t) { issue_b(t, t, 0); issue_b(t, t, 1); }
    for (int t = 0; t < min(nsub, 2); ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load_a(t, p);
            wait_a(p, std::integral_constant<int, 0>{});
            gen_a(t, t, p);
        }
    pp_wait_vm0();
    pp_wait_lgkm();
    if (nsub > 2) { load_a(2, 0); load_a(2, 1); }      // consumed in the two load segments of sub-tile 0
    pp_barrier();
    read_frags(0, 0, fa[0], fb[0]);
    pp_wait_lgkm();
    if (wm == 1) pp_barrier();          // group 1 runs one segment behind

    pp_stamp(g.stamps, 1);
    int slot = 0;
    // one sub-tile (two phases).  STEADY: sub-tiles u+2 and u+3 exist and u is not the last one -
    // straight-line code, so that the MFMA / fragment-read interleave pattern applies
    auto subtile = [&](int u, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int nslot = (slot + 1) & 3;
        const bool has_a = STEADY || u + 2 < nsub, has_b = STEADY || u + D < nsub, last = !STEADY && u == nsub - 1;
        // Each phase p (0, 1): turn the values requested one sub-tile ago into A piece p of sub-tile u+2
        // and store it, request piece p of sub-tile u+3 into the same registers, issue B piece p of
        // sub-tile u+3.  The values waited for are older than 4 loads + 2 DMAs -> vmcnt(6).
        const bool nxt_a = STEADY || u + 3 < nsub;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (has_a) {
                // operations younger than this piece's loads: one DMA + the other piece's loads + one DMA (4 loads per
                // piece, 2 when CAST); in the first sub-tile only what the prologue and phase 0 issued behind them
                if (!STEADY) wait_a(p, std::integral_constant<int, 0>{});
                else if (u == 0) wait_a(p, std::integral_constant<int, (CAST ? 2 : 4)>{});
                else wait_a(p, std::integral_constant<int, (CAST ? 4 : 6)>{});
                pp_wait_lgkm();
                gen_a(u + 2, (slot + 2) & 3, p);
                if (nxt_a) load_a(u + 3, p);
            } else if (p == 1) {
                pp_wait_vm0();          // tail: every outstanding B piece has landed
                pp_wait_lgkm();
            }
            if (has_b) issue_b(u + D, (slot + D) & 3, p);
            pp_wait_lgkm();             // the A image is written before the barrier publishes it
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            if (p == 0) mfma_seg(0, true, slot, 1, true);
        }
        mfma_seg(1, !last, nslot, 0, !(wm == 1 && last));
        slot = nslot;
    };
    int u = 0;
    for (; u + D < nsub; ++u) subtile(u, std::true_type{});
    for (; u < nsub; ++u) subtile(u, std::false_type{});

    pp_stamp(g.stamps, 2);
    pp_epilogue<EPI, OUT_BF16>(acc, g, lds, m0, n0, wave, lane);
    pp_stamp(g.stamps, 3);
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

// h1[(pr,t)][k] = bf16( bn0( relu( L0[pr][k] + T[t][k] ) ) ).
// L0 fp32 [S slabs][M1][h1] (split-K partial sums of layer 0), T fp32 [nt][h1], out bf16 [M1*nt][h1].
// One workgroup = one (packet, rx) row of L0 and 2048 columns at most: a thread owns 8 columns, sums
// the slabs and keeps scale/shift in registers, then walks the nt pilot rows - per output row one
// 32-byte read of T (L2-resident) and one 16-byte store, both fully coalesced.  HBM-write-bound.
__global__ __launch_bounds__(256) void pair_h1_bf16_kernel(const float* __restrict__ L0, int S, size_t slab, const float* __restrict__ T,
                                                           const float* __restrict__ s0, const float* __restrict__ t0,
                                                           bf16_t* __restrict__ out, int M1, int nt, int h1) {
    const int c8 = h1 >> 3;                                   // 8-column groups per row
    const int gpb = min(c8, 256);                             // groups per workgroup pass
    const int tsplit = 256 / gpb;                             // pilot rows handled side by side
    const int g = threadIdx.x % gpb, tl = threadIdx.x / gpb;
    for (int pr = blockIdx.x; pr < M1; pr += gridDim.x) {
        for (int g0 = 0; g0 < c8; g0 += gpb) {
            const int k = (g0 + g) * 8;
            if (g0 + g >= c8 || tl >= tsplit) continue;
            f32x4 la = *reinterpret_cast<const f32x4*>(L0 + (size_t)pr * h1 + k);
            f32x4 lb = *reinterpret_cast<const f32x4*>(L0 + (size_t)pr * h1 + k + 4);
            for (int z = 1; z < S; ++z) {
                la += *reinterpret_cast<const f32x4*>(L0 + z * slab + (size_t)pr * h1 + k);
                lb += *reinterpret_cast<const f32x4*>(L0 + z * slab + (size_t)pr * h1 + k + 4);
            }
            const f32x4 sa = *reinterpret_cast<const f32x4*>(s0 + k), sb = *reinterpret_cast<const f32x4*>(s0 + k + 4);
            const f32x4 ha = *reinterpret_cast<const f32x4*>(t0 + k), hb = *reinterpret_cast<const f32x4*>(t0 + k + 4);
            bf16_t* orow = out + (size_t)pr * nt * h1 + k;
#pragma unroll 4
            for (int t = tl; t < nt; t += tsplit) {
                const f32x4 ta = *reinterpret_cast<const f32x4*>(T + (size_t)t * h1 + k);
                const f32x4 tb = *reinterpret_cast<const f32x4*>(T + (size_t)t * h1 + k + 4);
                uint4 o;
                f32x2 v;
                v = f32x2{fmaf(fmaxf(la[0] + ta[0], 0.f), sa[0], ha[0]), fmaf(fmaxf(la[1] + ta[1], 0.f), sa[1], ha[1])};
                o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                v = f32x2{fmaf(fmaxf(la[2] + ta[2], 0.f), sa[2], ha[2]), fmaf(fmaxf(la[3] + ta[3], 0.f), sa[3], ha[3])};
                o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                v = f32x2{fmaf(fmaxf(lb[0] + tb[0], 0.f), sb[0], hb[0]), fmaf(fmaxf(lb[1] + tb[1], 0.f), sb[1], hb[1])};
                o.z = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                v = f32x2{fmaf(fmaxf(lb[2] + tb[2], 0.f), sb[2], hb[2]), fmaf(fmaxf(lb[3] + tb[3], 0.f), sb[3], hb[3])};
                o.w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                *reinterpret_cast<uint4*>(orow + (size_t)t * h1) = o;
            }
        }
    }
}

}  // namespace csi
