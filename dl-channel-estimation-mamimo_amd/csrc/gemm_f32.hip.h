// gemm_f32.hip.h - fp32 dense layers of the per-pair CSI regressor on CDNA4 matrix cores.
//
// One kernel template covers every Dense layer of the reference model
// (massiveMIMO_CSI_prediction_DNN.py:211-227):   C[M,N] = epilogue( A[M,K] * W[K,N] )
//   * W is held K-major on the device (Bt[N][ldb], transposed and zero-padded in K to a
//     multiple of 32 once at load time) so that both operands are read from LDS with one
//     ds_read_b128 per four k-steps.
//   * arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (a k-ordered
//     fmaf chain), i.e. the same number format the reference's TF-CPU float32 kernels use.
//   * block tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles,
//     LDS double-buffered (2 x 32 KiB), one barrier per k-tile, 2 workgroups per CU.
//   * operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip).
//     The LDS image is lane-linear (128-B rows, no padding); bank conflicts of the fragment
//     reads are removed by an XOR swizzle of the 16-B chunk index with (row>>1)&7, applied on
//     the per-lane SOURCE address of the DMA and on the ds_read address.
//   * A_PAIR mode builds the layer-1 activations on the fly in the A-operand prologue:
//        h1[(p,r,t), k] = bn0( relu( L0[(p,r), k] + T[t, k] ) )
//     where L0 = LTF part of layer 0 (computed once per (packet, rx) and shared by the Nt
//     pairs) and T = P * W0[lenLTF:, :] + b0 (Nt x H1 table).  h1 never exists in HBM.
//   * epilogue: + bias, relu, BatchNormalization affine applied AFTER the relu as in
//     DNN.py:211-219 (y = relu(z) * inv + (beta - mean * inv)).
//
// Bounds: rows are clamped to the last valid row (results of clamped rows are never stored);
// the K tail of a tile (K % 32 != 0) multiplies zero-padded weight columns, and every A-side
// buffer carries >= 128 B of zeroed slack so that the over-read stays in bounds and finite.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128;
constexpr int G_BN = 128;
constexpr int G_BK = 32;                       // K granularity the host pads / splits to
constexpr int G_THREADS = 256;
constexpr int G_SLACK_FLOATS = 64;             // zeroed slack behind every A-side buffer

enum AMode { A_PLAIN = 0, A_PAIR = 1 };
enum Epi { EPI_RAW = 0, EPI_BIAS = 1, EPI_BIAS_RELU_AFFINE = 2 };

struct GemmArgs {
    const float* A;        // A_PLAIN: [M][lda].  A_PAIR: L0 [M/nt][lda] (pre-bias layer-0 LTF product)
    const float* Bt;       // [N][ldb], K contiguous, ldb % 32 == 0, columns >= K are zero
    float* C;              // [M][ldc]; EPI_RAW split z writes slab C + z*M*ldc
    int M, N, K;
    int lda, ldb, ldc;
    int k_per_split;       // multiple of G_BK
    int tiles_n;
    // A_PAIR only
    const float* T;        // [nt][lda] pilot table, includes the layer-0 bias
    const float* s0;       // [K] layer-0 BN scale   (1 when the model has no BN)
    const float* t0;       // [K] layer-0 BN shift   (0 when the model has no BN)
    int nt;
    // epilogue
    const float* bias;     // [N]
    const float* scale;    // [N]
    const float* shift;    // [N]
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// 16 bytes per lane, HBM/L2 -> LDS, destination = wave-uniform base + lane*16
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Epilogue shared by the GEMM kernels.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Kept free of per-element branches: the per-column
// vectors are loaded unconditionally from a clamped index and interior row tiles store
// straight-line (a branch per store makes hipcc wait vmcnt(0) before every store, which
// serialises the whole tail).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[2][2], const GemmArgs& g, int m0, int n0,
                                              int wm, int wn, int l31, int hi) {
    float* Cz = g.C + (EPI == EPI_RAW ? (size_t)blockIdx.z * g.M * g.ldc : (size_t)0);
    const bool full_rows = (m0 + G_BM) <= g.M;          // block-uniform
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int col = n0 + wn * 64 + nj * 32 + l31;
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 1);
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (EPI != EPI_RAW) bias = g.bias[colc];
        if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc]; sh = g.shift[colc]; }
        float* cbase = Cz + (size_t)(m0 + wm * 64 + 4 * hi) * g.ldc + col;
        if (full_rows) {
            if (cok) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
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
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mi * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[mi][nj][r];
                    if (EPI == EPI_BIAS) v += bias;
                    if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                    if (cok && (m0 + wm * 64 + 4 * hi + rr) < g.M) cbase[(size_t)rr * g.ldc] = v;
                }
            }
        }
    }
}

// Tile geometry for a k-tile of BK floats (16 or 32):
//   CH   = 16-byte chunks per tile row                       (BK/4)
//   RPP  = tile rows covered by one 1-KiB DMA piece          (64/CH)
//   NPW  = DMA pieces each wave issues per operand tile      (128/RPP/4)
//   swizzle of the chunk index: f(row) = (row / (16/CH)) % CH  - 16 consecutive rows then hit
//   16 distinct 16-B slots of the 256-B LDS bank row, so ds_read_b128 is conflict-free.
template <int BK>
struct TileGeom {
    static constexpr int CH = BK / 4;
    static constexpr int RPP = 64 / CH;
    static constexpr int NPW = 128 / RPP / 4;
    static constexpr int TILE = 128 * BK;
    static constexpr int RSH = (CH == 8) ? 1 : 2;       // log2(16/CH)
    __device__ static __forceinline__ int swz(int row) { return (row >> RSH) & (CH - 1); }
};

template <int AMODE, int EPI, int BK, int STAGES, int MINW>
__global__ __launch_bounds__(G_THREADS, MINW) void gemm_f32_kernel(const GemmArgs g) {
    using TG = TileGeom<BK>;
    constexpr int TILE = TG::TILE;
    __shared__ __attribute__((aligned(16))) float lds[STAGES * 2 * TILE];   // [stage][A|B][128][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    // ---- LDS-DMA map: wave w issues pieces j = NPW*w .. NPW*w+NPW-1; piece j = tile rows
    // RPP*j .. RPP*j+RPP-1; lane -> row RPP*j + lane/CH, physical chunk lane%CH, which holds
    // logical chunk (lane%CH) ^ swz(row) of that row.
    const float* bsrc[TG::NPW];
    const float* asrc[TG::NPW];
#pragma unroll
    for (int u = 0; u < TG::NPW; ++u) {
        const int row = TG::RPP * (TG::NPW * wave + u) + lane / TG::CH;
        const int clog = (lane % TG::CH) ^ TG::swz(row);
        bsrc[u] = g.Bt + (size_t)min(n0 + row, g.N - 1) * g.ldb + clog * 4 + kbeg;
        if (AMODE == A_PLAIN) asrc[u] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + clog * 4 + kbeg;
        else asrc[u] = nullptr;
    }

    // ---- A_PAIR register-staged map: thread -> chunk c4 of rows r0 + (256/CH)*i, i < CH/2
    constexpr int PR = TG::CH / 2;                // rows per thread
    constexpr int RSTEP = G_THREADS / TG::CH;     // row stride between a thread's rows
    const int c4 = tid % TG::CH;
    const int r0 = tid / TG::CH;
    const float* lptr[PR];
    const float* tptr[PR];
    int awoff[PR];
    if (AMODE == A_PAIR) {
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const int row = r0 + RSTEP * i;
            const int m = min(m0 + row, g.M - 1);
            const int pr = m / g.nt;
            const int t = m - pr * g.nt;
            lptr[i] = g.A + (size_t)pr * g.lda + c4 * 4 + kbeg;
            tptr[i] = g.T + (size_t)t * g.lda + c4 * 4 + kbeg;
            awoff[i] = row * BK + ((c4 ^ TG::swz(row)) << 2);
        }
    }
    const float* sptr = (AMODE == A_PAIR) ? g.s0 + c4 * 4 + kbeg : nullptr;
    const float* hptr = (AMODE == A_PAIR) ? g.t0 + c4 * 4 + kbeg : nullptr;

    f32x4 pa[PR], pt[PR], ps, psh;

    // one DMA piece pair (B, and A when plain) of k-tile kt into stage buffer `buf`
    auto issue_dma_piece = [&](int kt, int buf, int u) {
        float* As = lds + buf * (2 * TILE);
        float* Bs = As + TILE;
        const int k = kt * BK;
        const int piece = (TG::NPW * wave + u) * 256;     // floats per 1-KiB piece
        dma16(bsrc[u] + k, Bs + piece);
        if (AMODE == A_PLAIN) dma16(asrc[u] + k, As + piece);
    };
    auto load_pair = [&](int kt) {
        const int k = kt * BK;
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            pa[i] = ldg4(lptr[i] + k);
            pt[i] = ldg4(tptr[i] + k);
        }
        ps = ldg4(sptr + k);
        psh = ldg4(hptr + k);
    };
    auto store_pair = [&](int buf) {
        float* As = lds + buf * (2 * TILE);
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(fmaxf(pa[i][e] + pt[i][e], 0.f), ps[e], psh[e]);
            *reinterpret_cast<f32x4*>(As + awoff[i]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets (floats): row*BK + ((2c+hi) ^ swz)*4, swz identical for row and row+32
    const int arow = wm * 64 + l31, brow = wn * 64 + l31;
    int aoff[BK / 8], boff[BK / 8];
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
        aoff[c] = arow * BK + (((2 * c + hi) ^ TG::swz(arow)) << 2);
        boff[c] = brow * BK + (((2 * c + hi) ^ TG::swz(brow)) << 2);
    }

    // multiply k-tile in stage `buf`; the DMA pieces of k-tile `kt_next` (if >= 0) are issued
    // between the MFMA groups so that their issue cost hides under the matrix pipe.
    auto compute_tile = [&](int buf, int kt_next, int buf_next) {
        const float* As = lds + buf * (2 * TILE);
        const float* Bs = As + TILE;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            // k-permutation inside each 8-chunk: lanes 0-31 take k = 0..3, lanes 32-63 k = 4..7;
            // MFMA step s consumes component s of both operands, so A and B agree on k.
            f32x4 a0 = *reinterpret_cast<const f32x4*>(As + aoff[c]);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(As + aoff[c] + 32 * BK);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + boff[c]);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + boff[c] + 32 * BK);
            if (kt_next >= 0 && c < TG::NPW) issue_dma_piece(kt_next, buf_next, c);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
            }
        }
    };
    static_assert(TG::NPW <= BK / 8, "one DMA piece per MFMA group");

    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < TG::NPW; ++u) issue_dma_piece(0, 0, u);
        if (AMODE == A_PAIR) {
            load_pair(0);
            store_pair(0);
        }
    }
    __syncthreads();          // drains the LDS-DMA (vmcnt) and publishes the ds_writes
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more && AMODE == A_PAIR) load_pair(kt + 1);
        compute_tile(kt & 1, more ? kt + 1 : -1, (kt + 1) & 1);
        if (more && AMODE == A_PAIR) store_pair((kt + 1) & 1);
        __syncthreads();
    }

    gemm_epilogue<EPI>(acc, g, m0, n0, wm, wn, l31, hi);
}

// ---------------------------------------------------------------------------------------------
// First per-pair layer with the layer-1 activations generated AT FRAGMENT-READ TIME.
//   C[(p,r,t), n] = epi( sum_k h1[(p,r,t), k] * W[k, n] ),
//   h1[(p,r,t), k] = bn0( relu( L0[(p,r), k] + T[t, k] ) )
// Instead of a 128-row A tile the k-tile stage holds three small images, all filled by LDS-DMA:
//   Ls  [PL_LROWS][32]  rows 0..NL-1 = the L0 rows of the (packet, rx) pairs this block touches,
//                       row NL = bn0 scale slice, row NL+1 = bn0 shift slice
//   Ts  [nt (<=128)][32] the pilot table slice (every t of the block is a row of it)
//   Bs  [128][32]        the weight tile
// and each lane builds its A fragment as fma(max(L + T, 0), s, t) right before the MFMAs
// (24 VALU per 16 MFMA, hidden under the matrix pipe).  No h1 tile is ever written anywhere,
// not even to LDS.  Requires 4 <= nt <= 128.
constexpr int PL_LROWS = 40;                   // 128/4 + 1 L0 rows + 2 vector rows, rounded to 8
constexpr int PL_STAGE = 128 * 32 + 128 * 32 + PL_LROWS * 32;    // floats per stage

template <int EPI>
__global__ __launch_bounds__(G_THREADS, 2) void pair_gemm_f32_kernel(const GemmArgs g) {
    constexpr int BK = 32;
    using TG = TileGeom<BK>;
    __shared__ __attribute__((aligned(16))) float lds[2 * PL_STAGE];     // [stage][Bs | Ts | Ls]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int nkt = (g.K + BK - 1) / BK;
    const int nt = g.nt;
    const int pr_base = m0 / nt;
    const int pr_last = min(m0 + G_BM - 1, g.M - 1) / nt;
    const int NL = pr_last - pr_base + 1;                // <= 33
    const int nL = (NL + 2 + 7) >> 3;                    // DMA pieces of the L image
    const int nT = (nt + 7) >> 3;                        // DMA pieces of the T image

    // ---- DMA sources.  Piece j covers image rows 8j..8j+7; lane -> row 8j + lane/8, physical
    // chunk lane%8 holding logical chunk (lane%8) ^ swz(row).  Wave w issues B pieces 4w..4w+3,
    // T pieces w, w+4, w+8, w+12 and L pieces w, w+4 (when they exist).
    const int prow = lane >> 3, pch = lane & 7;
    const float* bsrc[4];
    const float* tsrc[4];
    const float* lsrc[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rb = 8 * (4 * wave + u) + prow;
        bsrc[u] = g.Bt + (size_t)min(n0 + rb, g.N - 1) * g.ldb + ((pch ^ TG::swz(rb)) << 2);
        const int rt = 8 * (wave + 4 * u) + prow;
        tsrc[u] = g.T + (size_t)min(rt, nt - 1) * g.lda + ((pch ^ TG::swz(rt)) << 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rl = 8 * (wave + 4 * u) + prow;
        const int cl = (pch ^ TG::swz(rl)) << 2;
        const float* p = g.A + (size_t)(pr_base + min(rl, NL - 1)) * g.lda;
        if (rl == NL) p = g.s0;
        if (rl == NL + 1) p = g.t0;
        lsrc[u] = p + cl;
    }

    auto issue_piece = [&](int kt, int buf, int u) {      // u in 0..3: the u-th piece group of this wave
        float* Bs = lds + buf * PL_STAGE;
        float* Ts = Bs + 128 * BK;
        float* Ls = Ts + 128 * BK;
        const int k = kt * BK;
        dma16(bsrc[u] + k, Bs + (4 * wave + u) * 256);
        if (wave + 4 * u < nT) dma16(tsrc[u] + k, Ts + (wave + 4 * u) * 256);
        if (u < 2 && wave + 4 * u < nL) dma16(lsrc[u] + k, Ls + (wave + 4 * u) * 256);
    };

    // ---- per-lane fragment addressing (rows are fixed for the whole kernel)
    int loff[2], lswz[2], toff[2], tswz[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = min(m0 + wm * 64 + mi * 32 + l31, g.M - 1);
        const int pr = m / nt;
        const int t = m - pr * nt;
        const int lr = pr - pr_base;
        loff[mi] = lr * BK; lswz[mi] = TG::swz(lr);
        toff[mi] = t * BK;  tswz[mi] = TG::swz(t);
    }
    const int brow = wn * 64 + l31;
    const int bswz = TG::swz(brow);
    const int soff = NL * BK, hoff = (NL + 1) * BK;
    const int sswz = TG::swz(NL), hswz = TG::swz(NL + 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute_tile = [&](int buf, int kt_next, int buf_next) {
        const float* Bs = lds + buf * PL_STAGE;
        const float* Ts = Bs + 128 * BK;
        const float* Ls = Ts + 128 * BK;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int ch = 2 * c + hi;                    // logical 16-B chunk of this lane
            const f32x4 sv = *reinterpret_cast<const f32x4*>(Ls + soff + ((ch ^ sswz) << 2));
            const f32x4 hv = *reinterpret_cast<const f32x4*>(Ls + hoff + ((ch ^ hswz) << 2));
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(Ls + loff[0] + ((ch ^ lswz[0]) << 2));
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(Ls + loff[1] + ((ch ^ lswz[1]) << 2));
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(Ts + toff[0] + ((ch ^ tswz[0]) << 2));
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(Ts + toff[1] + ((ch ^ tswz[1]) << 2));
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + brow * BK + ((ch ^ bswz) << 2));
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + (brow + 32) * BK + ((ch ^ bswz) << 2));
            if (kt_next >= 0) issue_piece(kt_next, buf_next, c);
            f32x4 a0, a1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = fmaf(fmaxf(l0[e] + t0[e], 0.f), sv[e], hv[e]);
                a1[e] = fmaf(fmaxf(l1[e] + t1[e], 0.f), sv[e], hv[e]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
            }
        }
    };

    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) issue_piece(0, 0, u);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        compute_tile(kt & 1, more ? kt + 1 : -1, (kt + 1) & 1);
        __syncthreads();
    }
    gemm_epilogue<EPI>(acc, g, m0, n0, wm, wn, l31, hi);
}

// out[i] = sum_z slab_z[i]  (deterministic order z = 0..S-1); n4 = number of float4
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                     size_t n4, int S) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(slabs)[i];
        for (int z = 1; z < S; ++z) {
            f32x4 w = reinterpret_cast<const f32x4*>(slabs)[i + (size_t)z * n4];
            v += w;
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// T[t][n] = b0[n] + sum_i P[t][i] * W0p[i][n]   (W0p = rows lenLTF.. of fc_dense0.kernel)
__global__ void pilot_table_kernel(const float* __restrict__ P, const float* __restrict__ W0p,
                                   const float* __restrict__ b0, float* __restrict__ T,
                                   int nt, int h1) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (n >= h1) return;
    float acc = 0.f;
    for (int i = 0; i < nt; ++i) acc = fmaf(P[t * nt + i], W0p[(size_t)i * h1 + n], acc);
    T[(size_t)t * h1 + n] = acc + b0[n];
}

}  // namespace csi
