// gemm_f32.hip.h - fp32 dense layers of the per-pair CSI regressor on CDNA4 matrix cores.
//
// Two kernels cover every Dense layer of the reference model
// (massiveMIMO_CSI_prediction_DNN.py:211-227):   C[M,N] = epilogue( A[M,K] * W[K,N] )
//
//   gemm_f32_kernel       A is a plain row-major matrix (layer 0 over the rx preambles, hidden
//                         layers 2.., fc_regressor, and the literal un-shared network).
//   pair_gemm_f32_kernel  first per-pair layer; its A operand
//                             h1[(p,r,t), k] = bn0( relu( L0[(p,r), k] + T[t, k] ) )
//                         is generated at fragment-read time from two small LDS images (the
//                         layer-0 LTF product L0, computed once per (packet, rx) and shared by the
//                         Nt pairs, and the pilot table T = P * W0[lenLTF:, :] + b0).  h1 never
//                         exists in HBM, nor as a tile in LDS.
//
// Common structure
//   * W is held K-major on the device (Bt[N][ldb], transposed and zero-padded in K to a
//     multiple of 32 once at load time); both operands are read from LDS with one ds_read_b128
//     per four k-steps.
//   * arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (a k-ordered
//     fmaf chain), i.e. the number format of the reference's TF-CPU float32 kernels.
//   * block tile 128x128, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles,
//     2 workgroups per CU.
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into
//     a ring of R_NS = 4 stages of R_BK = 16 k-columns.  The DMA of k-tile t+3 is issued while
//     k-tile t is multiplied; a counted s_waitcnt vmcnt(N) + one raw s_barrier per k-tile hand a
//     stage over, so neither the DMA issue cost nor its landing latency (measured: -10 % when
//     exposed) sits on the matrix pipe's critical path.
//   * the LDS images are lane-linear (64-B rows, no padding); bank conflicts of the fragment
//     reads are removed by XOR-ing the 16-B chunk index with (row>>2)&3, applied on the per-lane
//     SOURCE address of the DMA and on the ds_read address.
//   * epilogue: + bias, relu, BatchNormalization affine applied AFTER the relu as in
//     DNN.py:211-219 (y = relu(z) * inv + (beta - mean * inv)).
//
// Bounds: rows are clamped to the last valid row (results of clamped rows are never stored);
// the K tail of a tile multiplies zero-padded weight columns, and every A-side buffer carries
// >= 256 B of zeroed slack so that the over-read stays in bounds and finite.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace csi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128;
constexpr int G_BN = 128;
constexpr int G_BK = 32;                       // K granularity the host pads / splits to
constexpr int G_THREADS = 256;
constexpr int G_SLACK_FLOATS = 64;             // zeroed slack behind every A-side buffer

constexpr int R_BK = 16;                       // k-columns per ring stage
constexpr int R_NS = 4;                        // ring stages
constexpr int R_D = R_NS - 1;                  // prefetch distance in k-tiles
constexpr int R_CH = R_BK / 4;                 // 16-B chunks per image row (4)
constexpr int R_RPP = 64 / R_CH;               // image rows per 1-KiB DMA piece (16)

enum Epi { EPI_RAW = 0, EPI_BIAS = 1, EPI_BIAS_RELU_AFFINE = 2 };

struct GemmArgs {
    const float* A;        // plain: [M][lda].  pair: L0 [M/nt][lda] (pre-bias layer-0 LTF product)
    const float* Bt;       // [N][ldb], K contiguous, ldb % 32 == 0, columns >= K are zero
    float* C;              // [M][ldc]; EPI_RAW split z writes slab C + z*M*ldc
    int M, N, K;
    int lda, ldb, ldc;
    int k_per_split;       // multiple of G_BK
    int tiles_n;
    int tiles_m;           // plain kernels: row tiles; > 0 selects the XCD super-tile order, 0 the linear one
    // pair kernel only
    const float* T;        // [nt][lda] pilot table, includes the layer-0 bias
    const float* s0;       // [K] layer-0 BN scale   (1 when the model has no BN)
    const float* t0;       // [K] layer-0 BN shift   (0 when the model has no BN)
    int nt;
    // epilogue
    const float* bias;     // [N]
    const float* scale;    // [N]
    const float* shift;    // [N]
};

// XCD-aware tile order of the plain GEMMs.  Workgroup b runs on XCD b % 8 (observed dispatch
// order; used for speed only).  Each XCD works through 'super-tiles' of 64 output tiles - exactly
// its 32 CUs x 2 resident workgroups - shaped SR row tiles x SC column tiles (SC = min(tiles_n, 8)):
// the 64 workgroups march through K together, so every A k-slice is fetched into that XCD's L2
// once for SC column tiles and every W k-slice once for SR row tiles (layer 0: 2.8 -> ~0.5 GB of
// fabric traffic per launch).  Ragged edges map to tiles outside the matrix; those workgroups exit.
struct TileMap {
    int sc, sr, ncg, nsuper;
};
__host__ __device__ __forceinline__ TileMap make_tile_map(int tiles_m, int tiles_n) {
    TileMap t;
    t.sc = tiles_n >= 8 ? 8 : (tiles_n >= 4 ? 4 : (tiles_n >= 2 ? 2 : 1));
    t.sr = 64 / t.sc;
    t.ncg = (tiles_n + t.sc - 1) / t.sc;
    t.nsuper = ((tiles_m + t.sr - 1) / t.sr) * t.ncg;
    return t;
}
// The super-tile order is used only where it balances the 8 XCDs at least as well as the linear
// order fills the 512 workgroup slots (few super-tiles would leave whole XCDs idle).
__host__ __forceinline__ bool tile_map_pays(int tiles_m, int tiles_n, int splits) {
    const TileMap t = make_tile_map(tiles_m, tiles_n);
    const double tiles = (double)tiles_m * tiles_n;
    const double eff_super = (tiles / 512.0) / (double)((t.nsuper + 7) / 8);        // every split repeats the order
    const long blocks = (long)tiles_m * tiles_n * splits;
    const double eff_linear = ((double)blocks / 512.0) / (double)((blocks + 511) / 512);
    return eff_super >= eff_linear - 0.01;
}
__host__ __forceinline__ unsigned tile_map_grid(int tiles_m, int tiles_n) {
    const TileMap t = make_tile_map(tiles_m, tiles_n);
    return (unsigned)(8 * 64 * ((t.nsuper + 7) / 8));
}
__device__ __forceinline__ bool tile_map(int b, int tiles_m, int tiles_n, int& tm, int& tn) {
    if (tiles_m <= 0) {                      // linear order, column tile fastest
        tn = b % tiles_n;
        tm = b / tiles_n;
        return true;
    }
    const TileMap t = make_tile_map(tiles_m, tiles_n);
    const int xcd = b & 7, idx = b >> 3;
    const int S = (idx >> 6) * 8 + xcd, w = idx & 63;
    const int rg = S / t.ncg, cg = S - rg * t.ncg;
    tm = rg * t.sr + w / t.sc;
    tn = cg * t.sc + w % t.sc;
    return tm < tiles_m && tn < tiles_n;
}

// 16 bytes per lane, HBM/L2 -> LDS, destination = wave-uniform base + lane*16
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int r_swz(int row) { return (row >> 2) & (R_CH - 1); }

// Stage hand-over: wait until at most `groups` of this wave's DMA groups (P instructions each)
// are still in flight, retire this wave's LDS reads, then meet the other waves.  The counted
// vmcnt keeps the younger k-tiles' DMA in flight ACROSS the barrier; __syncthreads() would
// drain them (vmcnt(0)).
template <int P>
__device__ __forceinline__ void ring_handover(int groups) {
    if (groups >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * P) : "memory");
    else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
static_assert(R_D == 3, "ring_handover covers 0..2 groups in flight");

// Epilogue shared by the GEMM kernels.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Kept free of per-element branches: the per-column
// vectors are loaded unconditionally from a clamped index and interior row tiles store
// straight-line (a branch per store makes hipcc wait vmcnt(0) before every store, which
// serialises the whole tail).
template <int EPI, int MI = 2, int BMT = G_BM>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[MI][2], const GemmArgs& g, int m0, int n0,
                                              int wm, int wn, int l31, int hi) {
    float* Cz = g.C + (EPI == EPI_RAW ? (size_t)blockIdx.z * g.M * g.ldc : (size_t)0);
    const bool full_rows = (m0 + BMT) <= g.M;           // block-uniform
    const int wrow = m0 + wm * (MI * 32) + 4 * hi;      // first row this lane stores
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int col = n0 + wn * 64 + nj * 32 + l31;
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 1);
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (EPI != EPI_RAW) bias = g.bias[colc];
        if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc]; sh = g.shift[colc]; }
        float* cbase = Cz + (size_t)wrow * g.ldc + col;
        if (full_rows) {
            if (cok) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[mi][nj][r];
                        if (EPI == EPI_BIAS) v += bias;
                        if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                        cbase[(size_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mi * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[mi][nj][r];
                    if (EPI == EPI_BIAS) v += bias;
                    if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                    if (cok && (wrow + rr) < g.M) cbase[(size_t)rr * g.ldc] = v;
                }
            }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// 16 MFMAs: one 8-deep k-chunk of the wave's 64x64 tile.  k-permutation inside the chunk:
// lanes 0-31 hold k = 0..3, lanes 32-63 k = 4..7; step s consumes component s of both
// operands, so A and B agree on k.
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[2][2], const f32x4& a0, const f32x4& a1, const f32x4& b0,
                                           const f32x4& b1) {
    // no s_setprio here: measured -4 % on the 128x128 kernels (16-MFMA clusters), +4 % on the
    // 256-row pair kernel (32-MFMA clusters), which raises its priority itself
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
    }
}

// =============================================================================================
// plain GEMM.  Ring stage = [A image 128x16 | B image 128x16]; each wave issues 2 + 2 DMA
// pieces (16 rows x 64 B each) per k-tile.
// =============================================================================================
constexpr int GP_STAGE = 2 * 128 * R_BK;       // floats per stage (16 KiB)
constexpr int GP_P = 4;                        // DMA instructions per wave per k-tile

template <int EPI>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[R_NS * GP_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    int tm, tn;
    if (!tile_map(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn)) return;      // block-uniform
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + R_BK - 1) / R_BK;

    // DMA map: wave w issues pieces 2w, 2w+1 of each image; piece j = image rows 16j..16j+15;
    // lane -> row 16j + lane/4, physical chunk lane%4 holding logical chunk (lane%4) ^ swz(row)
    const float* asrc[2];
    const float* bsrc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = R_RPP * (2 * wave + u) + (lane >> 2);
        const int clog = (lane & 3) ^ r_swz(row);
        asrc[u] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + clog * 4 + kbeg;
        bsrc[u] = g.Bt + (size_t)min(n0 + row, g.N - 1) * g.ldb + clog * 4 + kbeg;
    }
    auto issue = [&](int kt, int u, bool b_side) {
        float* st = lds + (kt & (R_NS - 1)) * GP_STAGE + (b_side ? 128 * R_BK : 0) + (2 * wave + u) * 256;
        dma16((b_side ? bsrc[u] : asrc[u]) + kt * R_BK, st);
    };

    // fragment read offsets inside a stage
    const int arow = wm * 64 + l31, brow = wn * 64 + l31;
    int aoff[2], boff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        aoff[c] = arow * R_BK + (((2 * c + hi) ^ r_swz(arow)) << 2);
        boff[c] = 128 * R_BK + brow * R_BK + (((2 * c + hi) ^ r_swz(brow)) << 2);
    }

    f32x16 acc[2][2];
    zero_acc(acc);

    const int npro = min(nkt, R_D);
    for (int t = 0; t < npro; ++t) {
        issue(t, 0, false); issue(t, 1, false); issue(t, 0, true); issue(t, 1, true);
    }
    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const float* st = lds + (kt & (R_NS - 1)) * GP_STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + aoff[c]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + aoff[c] + 32 * R_BK);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(st + boff[c]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(st + boff[c] + 32 * R_BK);
            if (MORE) { issue(kt + R_D, c, false); issue(kt + R_D, c, true); }
            mfma_chunk(acc, a0, a1, b0, b1);
        }
    };
    int kt = 0;
    for (; kt < nkt - R_D; ++kt) {          // steady state: 2 younger groups stay in flight
        ring_handover<GP_P>(R_D - 1);
        ktile(kt, std::true_type{});
    }
    for (; kt < nkt; ++kt) {                // drain
        ring_handover<GP_P>(nkt - 1 - kt);
        ktile(kt, std::false_type{});
    }
    gemm_epilogue<EPI>(acc, g, m0, n0, wm, wn, l31, hi);
}

// =============================================================================================
// plain GEMM, 256x128 block tile: 4 waves (2x2), wave tile 128x64 = 4x2 MFMA tiles, ring of 3
// stages [A image 256x16 | B image 128x16] (72 KiB, 2 workgroups per CU), 6 DMA pieces per wave
// per k-tile of 64 MFMAs (0.094 per MFMA against 0.125 for the 128x128 kernel).  No s_setprio:
// with the A operand streaming from HBM it starves the other workgroup's DMA (measured -9 %).
// Used when the grid fills the chip (+1.5-2 %); the 128x128 kernel otherwise.
// =============================================================================================
constexpr int G2_BM = 256;

template <int EPI>
__global__ __launch_bounds__(G_THREADS, 2) void gemm256_f32_kernel(const GemmArgs g) {
    constexpr int NS = 3, D = NS - 1, P = 6;
    constexpr int STAGE = (G2_BM + 128) * R_BK;                    // floats (24 KiB)
    constexpr int BOFF = G2_BM * R_BK;
    __shared__ __attribute__((aligned(16))) float lds[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    int tm, tn;
    if (!tile_map(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn)) return;      // block-uniform
    const int m0 = tm * G2_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + R_BK - 1) / R_BK;

    // DMA: 24 pieces of 16 image rows per stage (16 A, 8 B); wave w issues pieces 6w .. 6w+5
    const float* src[P];
    int dst[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int piece = P * wave + u;
        const int row = R_RPP * piece + (lane >> 2);               // image row (A rows 0..255, then B rows)
        const int clog = (lane & 3) ^ r_swz(row);
        if (piece < G2_BM / R_RPP) src[u] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + clog * 4 + kbeg;
        else src[u] = g.Bt + (size_t)min(n0 + row - G2_BM, g.N - 1) * g.ldb + clog * 4 + kbeg;
        dst[u] = piece * 256;
    }
    auto issue = [&](int kt, int u) { dma16(src[u] + kt * R_BK, lds + (kt % NS) * STAGE + dst[u]); };

    int aoff[2], boff[2];
    const int arow = wm * 128 + l31, brow = wn * 64 + l31;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        aoff[c] = arow * R_BK + (((2 * c + hi) ^ r_swz(arow)) << 2);
        boff[c] = BOFF + brow * R_BK + (((2 * c + hi) ^ r_swz(brow)) << 2);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int npro = min(nkt, D);
    for (int t = 0; t < npro; ++t)
#pragma unroll
        for (int u = 0; u < P; ++u) issue(t, u);

    auto handover = [&](int groups) {
        if (groups >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const float* st = lds + (kt % NS) * STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 a[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(st + aoff[c] + mi * 32 * R_BK);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(st + boff[c]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(st + boff[c] + 32 * R_BK);
            if (MORE) {
#pragma unroll
                for (int u = 0; u < 3; ++u) issue(kt + D, 3 * c + u);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b0[s], acc[mi][0], 0, 0, 0);
                    acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b1[s], acc[mi][1], 0, 0, 0);
                }
            }
        }
    };
    int kt = 0;
    for (; kt < nkt - D; ++kt) {
        handover(D - 1);
        ktile(kt, std::true_type{});
    }
    for (; kt < nkt; ++kt) {
        handover(nkt - 1 - kt);
        ktile(kt, std::false_type{});
    }
    gemm_epilogue<EPI, 4, G2_BM>(acc, g, m0, n0, wm, wn, l31, hi);
}

// =============================================================================================
// pair GEMM (first per-pair layer).  Ring stage = three images, all filled by LDS-DMA:
//   Bs [128][16]   weight tile
//   Ts [128][16]   pilot-table slice, row t (rows >= nt are unused duplicates)
//   Ls [48][16]    rows 0..NL-1 = the L0 rows of the (packet, rx) pairs this block touches,
//                  row NL = bn0 scale slice, row NL+1 = bn0 shift slice
// Each lane builds its A fragment as fma(max(L + T, 0), s, t) right before the MFMAs (24 VALU
// per 16 MFMA, hidden under the matrix pipe).  Every wave issues exactly 2 B + TPW T + 1 L DMA
// pieces per k-tile so that the counted vmcnt is the same constant for all waves (TPW = 1
// covers nt <= 64, TPW = 2 nt <= 128; a wave whose piece lies beyond the image repeats the
// last one - same bytes, same destination).  Requires 4 <= nt <= 128.
// =============================================================================================
constexpr int PL_LROWS = 48;                                   // 128/4 + 1 L0 rows + 2 vector rows -> 3 pieces
constexpr int PL_STAGE = (128 + 128 + PL_LROWS) * R_BK;        // floats per stage (19 KiB)
template <int EPI, int TPW>
__global__ __launch_bounds__(G_THREADS, 2) void pair_gemm_f32_kernel(const GemmArgs g) {
    constexpr int PL_P = 2 + TPW + 1;
    __shared__ __attribute__((aligned(16))) float lds[R_NS * PL_STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_per_split;           // split-K (small-batch latency path)
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + R_BK - 1) / R_BK;
    const int nt = g.nt;
    const int pr_base = m0 / nt;
    const int NL = min(m0 + G_BM - 1, g.M - 1) / nt - pr_base + 1;       // <= 33

    // ---- DMA sources (piece = 16 image rows; lane -> row 16j + lane/4, chunk lane%4)
    const int prow = lane >> 2, pch = lane & 3;
    const float* bsrc[2];
    const float* tsrc[TPW];
    const float* lsrc;
    const int nT = (nt + R_RPP - 1) / R_RPP;                   // pieces of the T image that exist
    int tpiece[TPW];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rb = R_RPP * (2 * wave + u) + prow;
        bsrc[u] = g.Bt + (size_t)min(n0 + rb, g.N - 1) * g.ldb + ((pch ^ r_swz(rb)) << 2) + kbeg;
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        tpiece[u] = min(wave + 4 * u, nT - 1);                 // T pieces w (and w+4)
        const int rt = R_RPP * tpiece[u] + prow;
        tsrc[u] = g.T + (size_t)min(rt, nt - 1) * g.lda + ((pch ^ r_swz(rt)) << 2) + kbeg;
    }
    const int lpiece = min(wave, PL_LROWS / R_RPP - 1);        // wave 3 repeats piece 2 (same bytes)
    {
        const int rl = R_RPP * lpiece + prow;
        const float* p = g.A + (size_t)(pr_base + min(rl, NL - 1)) * g.lda;
        if (rl == NL) p = g.s0;
        if (rl == NL + 1) p = g.t0;
        lsrc = p + ((pch ^ r_swz(rl)) << 2) + kbeg;
    }
    // u = 0/1: {B piece, T piece};  u = 2: the L piece
    auto issue = [&](int kt, int u) {
        float* st = lds + (kt & (R_NS - 1)) * PL_STAGE;
        const int k = kt * R_BK;
        if (u < 2) {
            dma16(bsrc[u] + k, st + (2 * wave + u) * 256);
            if (u < TPW) dma16(tsrc[u] + k, st + 128 * R_BK + tpiece[u] * 256);
        } else {
            dma16(lsrc + k, st + 256 * R_BK + lpiece * 256);
        }
    };

    // ---- per-lane fragment addressing (rows are fixed for the whole kernel)
    int loff[2][2], toff[2][2], boff[2], soff[2], hoff[2];
    const int brow = wn * 64 + l31;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = 2 * c + hi;                       // logical 16-B chunk of this lane
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = min(m0 + wm * 64 + mi * 32 + l31, g.M - 1);
            const int pr = m / nt;
            const int t = m - pr * nt;
            const int lr = pr - pr_base;
            loff[mi][c] = 256 * R_BK + lr * R_BK + ((ch ^ r_swz(lr)) << 2);
            toff[mi][c] = 128 * R_BK + t * R_BK + ((ch ^ r_swz(t)) << 2);
        }
        boff[c] = brow * R_BK + ((ch ^ r_swz(brow)) << 2);
        soff[c] = 256 * R_BK + NL * R_BK + ((ch ^ r_swz(NL)) << 2);
        hoff[c] = 256 * R_BK + (NL + 1) * R_BK + ((ch ^ r_swz(NL + 1)) << 2);
    }

    f32x16 acc[2][2];
    zero_acc(acc);

    const int npro = min(nkt, R_D);
    for (int t = 0; t < npro; ++t) {
        issue(t, 0); issue(t, 1); issue(t, 2);
    }
    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const float* st = lds + (kt & (R_NS - 1)) * PL_STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(st + soff[c]);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(st + hoff[c]);
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(st + loff[0][c]);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(st + loff[1][c]);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(st + toff[0][c]);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(st + toff[1][c]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(st + boff[c]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(st + boff[c] + 32 * R_BK);
            if (MORE) {
                issue(kt + R_D, c);
                if (c == 1) issue(kt + R_D, 2);
            }
            f32x4 a0, a1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = fmaf(fmaxf(l0[e] + t0[e], 0.f), sv[e], hv[e]);
                a1[e] = fmaf(fmaxf(l1[e] + t1[e], 0.f), sv[e], hv[e]);
            }
            mfma_chunk(acc, a0, a1, b0, b1);
        }
    };
    int kt = 0;
    for (; kt < nkt - R_D; ++kt) {          // steady state: 2 younger groups stay in flight
        ring_handover<PL_P>(R_D - 1);
        ktile(kt, std::true_type{});
    }
    for (; kt < nkt; ++kt) {                // drain
        ring_handover<PL_P>(nkt - 1 - kt);
        ktile(kt, std::false_type{});
    }
    gemm_epilogue<EPI>(acc, g, m0, n0, wm, wn, l31, hi);
}

// =============================================================================================
// pair GEMM, 256x128 block tile.  The A side of this layer costs (almost) no DMA - the pilot-table
// slice is the same for every row tile and the L0 rows are 1/Nt of the rows - so doubling the row
// tile halves the LDS-DMA instructions per MFMA, which is what limits the 128x128 kernel
// (tools/gemm_probe.hip).  4 waves (2x2), wave tile 128x64 = 4x2 MFMA tiles (128 accumulator
// registers), 2 workgroups per CU.  Ring stage = Bs [128][16] | Ts [64*TPW][16] | Ls [64*LPW][16];
// every wave issues 2 B + TPW T + LPW L pieces per k-tile (TPW = 1: nt <= 64, LPW = 1: nt >= 8).
// =============================================================================================
constexpr int P2_BM = 256;

template <int EPI, int TPW, int LPW, int DBG = 0>     // DBG: timing ablations of tools/gemm_probe only (1 no in-loop DMA, 2 no h1 generation, 4 no setprio, 8 no stores)
__global__ __launch_bounds__(G_THREADS, 2) void pair_gemm256_f32_kernel(const GemmArgs g) {
    constexpr int TROWS = 64 * TPW, LROWS = 64 * LPW;
    constexpr int STAGE = (128 + TROWS + LROWS) * R_BK;             // floats
    constexpr int NS = 4;                                           // 64 KiB / 80 KiB of LDS: two workgroups per CU either way
    constexpr int D = NS - 1;
    constexpr int P = 2 + TPW + LPW;
    constexpr int TOFF = 128 * R_BK, LOFF = (128 + TROWS) * R_BK;
    __shared__ __attribute__((aligned(16))) float lds[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * P2_BM, n0 = tn * G_BN;
    const int nkt = (g.K + R_BK - 1) / R_BK;
    const int nt = g.nt;
    const int pr_base = m0 / nt;
    const int NL = min(m0 + P2_BM - 1, g.M - 1) / nt - pr_base + 1;      // <= 65
    const int nT = (nt + R_RPP - 1) / R_RPP;
    const int nL = (NL + 2 + R_RPP - 1) / R_RPP;

    // ---- DMA sources (piece = 16 image rows; lane -> row 16j + lane/4, chunk lane%4)
    const int prow = lane >> 2, pch = lane & 3;
    const float* bsrc[2];
    const float* tsrc[TPW];
    const float* lsrc[LPW];
    int tpiece[TPW], lpiece[LPW];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rb = R_RPP * (2 * wave + u) + prow;
        bsrc[u] = g.Bt + (size_t)min(n0 + rb, g.N - 1) * g.ldb + ((pch ^ r_swz(rb)) << 2);
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        tpiece[u] = min(wave + 4 * u, nT - 1);
        const int rt = R_RPP * tpiece[u] + prow;
        tsrc[u] = g.T + (size_t)min(rt, nt - 1) * g.lda + ((pch ^ r_swz(rt)) << 2);
    }
#pragma unroll
    for (int u = 0; u < LPW; ++u) {
        lpiece[u] = min(wave + 4 * u, nL - 1);
        const int rl = R_RPP * lpiece[u] + prow;
        const float* p = g.A + (size_t)(pr_base + min(rl, NL - 1)) * g.lda;
        if (rl == NL) p = g.s0;
        if (rl == NL + 1) p = g.t0;
        lsrc[u] = p + ((pch ^ r_swz(rl)) << 2);
    }
    // one call = this wave's pieces of k-tile kt, part u (0, 1): {B piece u, T piece u, L piece u}
    auto issue = [&](int kt, int u) {
        float* st = lds + (kt % NS) * STAGE;
        const int k = kt * R_BK;
        dma16(bsrc[u] + k, st + (2 * wave + u) * 256);
        if (u < TPW) dma16(tsrc[u] + k, st + TOFF + tpiece[u] * 256);
        if (u < LPW) dma16(lsrc[u] + k, st + LOFF + lpiece[u] * 256);
    };

    // ---- per-lane fragment addressing
    int loff[4][2], toff[4][2], boff[2], soff[2], hoff[2];
    const int brow = wn * 64 + l31;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = 2 * c + hi;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = min(m0 + wm * 128 + mi * 32 + l31, g.M - 1);
            const int pr = m / nt;
            const int t = m - pr * nt;
            const int lr = pr - pr_base;
            loff[mi][c] = LOFF + lr * R_BK + ((ch ^ r_swz(lr)) << 2);
            toff[mi][c] = TOFF + t * R_BK + ((ch ^ r_swz(t)) << 2);
        }
        boff[c] = brow * R_BK + ((ch ^ r_swz(brow)) << 2);
        soff[c] = LOFF + NL * R_BK + ((ch ^ r_swz(NL)) << 2);
        hoff[c] = LOFF + (NL + 1) * R_BK + ((ch ^ r_swz(NL + 1)) << 2);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int npro = min(nkt, D);
    for (int t = 0; t < npro; ++t) { issue(t, 0); issue(t, 1); }

    auto handover = [&](int groups) {
        if (groups >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * P) : "memory");
        else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto ktile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const float* st = lds + (kt % NS) * STAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(st + soff[c]);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(st + hoff[c]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(st + boff[c]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(st + boff[c] + 32 * R_BK);
            f32x4 a[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f32x4 l = *reinterpret_cast<const f32x4*>(st + loff[mi][c]);
                const f32x4 t = *reinterpret_cast<const f32x4*>(st + toff[mi][c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[mi][e] = (DBG & 2) ? l[e] : fmaf(fmaxf(l[e] + t[e], 0.f), sv[e], hv[e]);
            }
            if (MORE && !(DBG & 1)) issue(kt + D, c);
            if (!(DBG & 4)) __builtin_amdgcn_s_setprio(1);      // the MFMA cluster outranks the other workgroup's loads/VALU (+4 %)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b0[s], acc[mi][0], 0, 0, 0);
                    acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b1[s], acc[mi][1], 0, 0, 0);
                }
            }
            if (!(DBG & 4)) __builtin_amdgcn_s_setprio(0);
        }
    };
    int kt = 0;
    for (; kt < nkt - D; ++kt) {
        handover((DBG & 1) ? 0 : D - 1);
        ktile(kt, std::true_type{});
    }
    for (; kt < nkt; ++kt) {
        handover((DBG & 1) ? 0 : nkt - 1 - kt);
        ktile(kt, std::false_type{});
    }
    if ((DBG & 8) && g.M > 0) return;
    gemm_epilogue<EPI, 4, P2_BM>(acc, g, m0, n0, wm, wn, l31, hi);
}

// Layer 0 for a handful of preambles (the reference's literal one-packet call has M1 = Nr = 4
// rows): a weight-streaming kernel instead of a GEMM.  W0 is read ONCE, row-major [K][h1] as keras
// stores it (n contiguous -> 16 B per lane, coalesced).  Workgroup (z, y) owns 4*SK_KS consecutive
// k rows and 256 columns: wave w streams rows w*SK_KS.., lane 4 columns, MR accumulator rows;
// the four waves are combined in LDS (fixed order) and the workgroup writes slab z, which
// splitk_reduce_kernel sums in fixed order.  HBM-bound: 4*K*h1 bytes.
constexpr int SK_KS = 40;                      // k rows per wave; 160 per workgroup (320*nt / 160 = 2*nt slabs)

template <int MR>
__global__ __launch_bounds__(256) void layer0_skinny_kernel(const float* __restrict__ x, int lda, int M,
                                                            const float* __restrict__ W, int h1, int K,
                                                            float* __restrict__ slabs) {
    __shared__ float xs[4][MR * SK_KS];
    __shared__ __attribute__((aligned(16))) float red[4][MR][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wave) * SK_KS;
    const int kn = max(0, min(SK_KS, K - k0));
    for (int i = lane; i < MR * SK_KS; i += 64) {
        const int m = i / SK_KS, k = i - m * SK_KS;
        xs[wave][i] = (m < M && k < kn) ? x[(size_t)m * lda + k0 + k] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = blockIdx.y * 256 + lane * 4;
    const bool nok = n < h1;
    f32x4 acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = W + (size_t)k0 * h1 + (nok ? n : 0);
#pragma unroll 8
    for (int k = 0; k < SK_KS; ++k) {
        if (k < kn) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wp + (size_t)k * h1);
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float xv = xs[wave][m * SK_KS + k];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][e] = fmaf(xv, w[e], acc[m][e]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) *reinterpret_cast<f32x4*>(&red[wave][m][lane * 4]) = acc[m];
    __syncthreads();
    // 256 threads: thread t sums column t of every accumulator row over the four waves
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col < h1) {
        float* out = slabs + (size_t)blockIdx.x * M * h1 + col;
#pragma unroll
        for (int m = 0; m < MR; ++m)
            if (m < M) out[(size_t)m * h1] = (red[0][m][threadIdx.x] + red[1][m][threadIdx.x]) + (red[2][m][threadIdx.x] + red[3][m][threadIdx.x]);
    }
}

// out[i] = sum_z slab_z[i]  (deterministic order z = 0..S-1); n4 = number of float4
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                     size_t n4, int S) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        // loads of 8 slabs in flight at a time; the sum itself stays in z order (deterministic)
        f32x4 v = reinterpret_cast<const f32x4*>(slabs)[i];
        int z = 1;
        for (; z + 8 <= S; z += 8) {
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = reinterpret_cast<const f32x4*>(slabs)[i + (size_t)(z + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += w[u];
        }
        for (; z < S; ++z) v += reinterpret_cast<const f32x4*>(slabs)[i + (size_t)z * n4];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// out[m][n] = epilogue( sum_z slab_z[m][n] ), slabs compact [S][M][N]; one thread per element.
// Used by the small-batch path, where split-K spreads a short GEMM over the whole chip.
template <int EPI>
__global__ void splitk_epilogue_kernel(const float* __restrict__ slabs, int S, int M, int N, float* __restrict__ out,
                                       int ldc, const float* __restrict__ bias, const float* __restrict__ scale,
                                       const float* __restrict__ shift) {
    const size_t total = (size_t)M * N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
        float v = slabs[i];
        int z = 1;
        for (; z + 8 <= S; z += 8) {            // 8 loads in flight, summed in z order
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = slabs[(size_t)(z + u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += w[u];
        }
        for (; z < S; ++z) v += slabs[(size_t)z * total + i];
        if (EPI == EPI_BIAS) v += bias[n];
        if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias[n], 0.f), scale[n], shift[n]);
        out[(size_t)m * ldc + n] = v;
    }
}

// T[t][n] = scale * (b0[n] + sum_i P[t][i] * W0p[i][n])   (W0p = rows lenLTF.. of fc_dense0.kernel; scale = 1, or the
// power of two the split-f16 pair kernel carries its A operand at - exact, so that copy is bit for bit scale * T)
__global__ void pilot_table_kernel(const float* __restrict__ P, const float* __restrict__ W0p,
                                   const float* __restrict__ b0, float* __restrict__ T,
                                   int nt, int h1, float scale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (n >= h1) return;
    float acc = 0.f;
    for (int i = 0; i < nt; ++i) acc = fmaf(P[t * nt + i], W0p[(size_t)i * h1 + n], acc);
    T[(size_t)t * h1 + n] = (acc + b0[n]) * scale;
}

}  // namespace csi
