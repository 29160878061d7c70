// gemm_f32.hip.h - fp32 dense layers of the per-pair CSI regressor on CDNA4 matrix cores.
//
// One kernel template covers every Dense layer of the reference model
// (massiveMIMO_CSI_prediction_DNN.py:211-227):   C[M,N] = epilogue( A[M,K] * W[K,N] )
//   * W is held K-major on the device (Bt[N][K], transposed once at load time) so that both
//     operands are read from LDS with one ds_read_b128 per four k-steps.
//   * arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (a k-ordered
//     fmaf chain), i.e. the same number format the reference's TF-CPU float32 kernels use.
//   * block tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles,
//     LDS double-buffered, next tile prefetched into registers while the current one is
//     multiplied (one barrier per k-tile), 2 workgroups per CU.
//   * A_PAIR mode builds the layer-1 activations on the fly in the A-operand prologue:
//        h1[(p,r,t), k] = bn0( relu( L0[(p,r), k] + T[t, k] ) )
//     where L0 = LTF part of layer 0 (computed once per (packet, rx) and shared by the Nt
//     pairs) and T = P * W0[lenLTF:, :] + b0 (Nt x H1 table).  h1 never exists in HBM.
//   * epilogue: + bias, relu, BatchNormalization affine applied AFTER the relu as in
//     DNN.py:211-219 (y = relu(z) * inv + (beta - mean * inv)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128;
constexpr int G_BN = 128;
constexpr int G_BK = 32;
constexpr int G_PITCH = G_BK + 4;              // 36 floats = 144 B: ds_read_b128 conflict-free
constexpr int G_TILE = 128 * G_PITCH;          // floats per operand tile
constexpr int G_THREADS = 256;

enum AMode { A_PLAIN = 0, A_PAIR = 1 };
enum Epi { EPI_RAW = 0, EPI_BIAS = 1, EPI_BIAS_RELU_AFFINE = 2 };

struct GemmArgs {
    const float* A;        // A_PLAIN: [M][lda].  A_PAIR: L0 [M/nt][lda] (pre-bias layer-0 LTF product)
    const float* Bt;       // [N][ldb], K contiguous
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

template <int AMODE, int EPI>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * G_TILE];   // [buf][A|B][128][36]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile = blockIdx.x;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + G_BK - 1) / G_BK;

    // staging map: thread -> 16-byte chunk c4 of rows r0 + 32*i
    const int c4 = tid & 7;
    const int r0 = tid >> 3;

    // per-thread row pointers (k offset added per tile)
    const float* aptr[4];
    const float* tptr[4];
    bool arow_ok[4];
    const float* bptr[4];
    bool brow_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        arow_ok[i] = m < g.M;
        const int mc = arow_ok[i] ? m : 0;
        if (AMODE == A_PAIR) {
            const int pr = mc / g.nt;
            const int t = mc - pr * g.nt;
            aptr[i] = g.A + (size_t)pr * g.lda + c4 * 4;
            tptr[i] = g.T + (size_t)t * g.lda + c4 * 4;
        } else {
            aptr[i] = g.A + (size_t)mc * g.lda + c4 * 4;
            tptr[i] = nullptr;
        }
        const int n = n0 + r0 + 32 * i;
        brow_ok[i] = n < g.N;
        bptr[i] = g.Bt + (size_t)(brow_ok[i] ? n : 0) * g.ldb + c4 * 4;
    }

    f32x4 pa[4], pt[4], pb[4], ps, psh;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int k = kbeg + kt * G_BK;
        const bool kok = (k + c4 * 4) < kend;        // K % 4 == 0 is a precondition
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] = (kok && arow_ok[i]) ? ldg4(aptr[i] + k) : zero4;
            if (AMODE == A_PAIR) pt[i] = (kok && arow_ok[i]) ? ldg4(tptr[i] + k) : zero4;
            pb[i] = (kok && brow_ok[i]) ? ldg4(bptr[i] + k) : zero4;
        }
        if (AMODE == A_PAIR) {
            ps = kok ? ldg4(g.s0 + k + c4 * 4) : zero4;
            psh = kok ? ldg4(g.t0 + k + c4 * 4) : zero4;
        }
    };

    auto store_tile = [&](int buf) {
        float* As = lds + buf * (2 * G_TILE);
        float* Bs = As + G_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = pa[i];
            if (AMODE == A_PAIR) {
                // h1 = bn0(relu(L0 + T)); rows beyond M and columns beyond K stay zero only if
                // shift is zero there - they are masked at the epilogue / by zero B columns.
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(fmaxf(pa[i][e] + pt[i][e], 0.f), ps[e], psh[e]);
            }
            *reinterpret_cast<f32x4*>(As + (r0 + 32 * i) * G_PITCH + c4 * 4) = v;
            *reinterpret_cast<f32x4*>(Bs + (r0 + 32 * i) * G_PITCH + c4 * 4) = pb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute_tile = [&](int buf) {
        const float* As = lds + buf * (2 * G_TILE) + (wm * 64 + l31) * G_PITCH + hi * 4;
        const float* Bs = lds + buf * (2 * G_TILE) + G_TILE + (wn * 64 + l31) * G_PITCH + hi * 4;
#pragma unroll
        for (int c = 0; c < G_BK / 8; ++c) {
            // k-permutation inside each 8-chunk: lanes 0-31 take k = 0..3, lanes 32-63 k = 4..7;
            // MFMA step s consumes component s of both operands, so A and B agree on k.
            f32x4 a0 = *reinterpret_cast<const f32x4*>(As + c * 8);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 32 * G_PITCH + c * 8);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + c * 8);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + 32 * G_PITCH + c * 8);
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
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) load_tile(kt + 1);
        compute_tile(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = g.C + (EPI == EPI_RAW ? (size_t)blockIdx.z * g.M * g.ldc : (size_t)0);
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int col = n0 + wn * 64 + nj * 32 + l31;
        const bool cok = col < g.N;
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (EPI != EPI_RAW && cok) bias = g.bias[col];
        if (EPI == EPI_BIAS_RELU_AFFINE && cok) { sc = g.scale[col]; sh = g.shift[col]; }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float v = acc[mi][nj][r];
                if (EPI == EPI_BIAS) v += bias;
                if (EPI == EPI_BIAS_RELU_AFFINE) v = fmaf(fmaxf(v + bias, 0.f), sc, sh);
                if (cok && row < g.M) Cz[(size_t)row * g.ldc + col] = v;
            }
        }
    }
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
