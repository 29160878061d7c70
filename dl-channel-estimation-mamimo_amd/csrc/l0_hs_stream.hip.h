// l0_hs_stream.hip.h - layer 0 (the shared LTF product, massiveMIMO_CSI_prediction_DNN.py:211-214 on the columns of the flattened
// preamble) of a MID-SIZE call - 9 ... 1280 rx preambles, i.e. 3 ... 320 packets of the shipped shape - as a weight-streaming kernel on
// the split-f16 matrix path.
//
// Such a call's layer 0 is bound by streaming the weights once (42 MB per component model at Nt = 32: ~8 us of HBM time); its
// arithmetic is small.  The general kernels do not get there: the fp32 MFMA GEMM (gemm_f32.hip.h) pads the rows to its 128-row tile
// and is bound by the fp32 matrix rate (157 TFLOP/s: 34 us for the two models whatever M <= 128), the 256 x 256 ping-pong kernel of
// the split engine (gemm_hs.hip.h) has a handful of workgroups at these sizes.  Here:
//
//   grid (ceil(N / 128), KS): a workgroup of 4 waves owns 128 output columns (32 per wave) and one k range of `kps` columns, ALL M rows.
//   B operand = the split weights as csi_load_weights stores them (hs rows: groups of 16 k as 16 hi | 16 lo halves), global ->
//   registers, four 16-byte loads per lane and 32-k chunk (the two k halves of a column read 128 contiguous bytes), four chunks ahead.
//   A operand = the fp32 preamble rows, scaled, split into hi + lo halves once per workgroup and chunk and parked in LDS
//   ([row][32 hi | 32 lo | pad]: 144-byte rows, conflict-free for the 16-byte fragment reads).
//   Products: a_hi b_lo + a_hi b_hi + a_lo b_hi on v_mfma_f32_32x32x16_f16, fp32 accumulators, like every kernel of the engine.
//
// Input scale.  The engine needs |s x| inside the f16 range with room below for the lo half.  The 256 x 256 kernel takes ONE scale per
// launch from a sampled maximum (hs_absmax_sample_kernel: a memset and a kernel in front) and guards both ends on the device.  This kernel
// scales every ROW of its k range by its own power of two - 2^(14 - exponent of the row's largest magnitude in the range), found in a
// first pass over the range - and undoes it exactly in the epilogue (a row of the partial product depends on that row of x only).  The
// largest value of a row lands in [2^13, 2^14): no overflow by construction, and every value down to 2^-17 of its own row's maximum keeps a
// normal lo half - no sampling, no guard, no second launch.
//
// Output: partial products slabs[s][m][n] (fp32), summed over s in k order by splitk_reduce_kernel - deterministic, run-to-run identical.
#pragma once
#include "gemm_hs.hip.h"

namespace csi {

struct L0StreamArgs {
    const float* x;        // [M][lda] fp32: one component plane of the preambles
    const uint16_t* Wh;    // hs [N][ldwh halves]: layer-0 weights (LTF columns), K-major, stored times 2^wshift
    float* slabs;          // [gridDim.y][M][N] fp32
    int M, N, K, lda, ldwh;
    int kps;               // k-columns per blockIdx.y, a multiple of 32
    int wshift;
    const float* row_max;  // [M] largest magnitude of every row over all K (l0_row_max_kernel), or null: the kernel finds its k range's own
};

constexpr int L0S_COLS = 128;      // output columns per workgroup
constexpr int L0S_KC = 32;         // k-columns per chunk
constexpr int L0S_ROWB = 144;      // LDS bytes per row and chunk: 32 hi halves | 32 lo halves | 16 bytes of padding
constexpr int L0S_AHEAD = 4;       // chunks of weight loads in flight (even)

inline size_t l0s_lds_bytes(int rt) { return (size_t)2 * 32 * rt * L0S_ROWB + (size_t)2 * 32 * rt * sizeof(float); }

// row_max[m] = max_k |x[m][k]|: one workgroup per row.  Calls of more than 64 preambles take their row scales from here - the first pass
// of the kernel below re-reads every row once per column group (8 x the preambles through the L2: a third of that kernel's traffic at 256 rows)
__global__ __launch_bounds__(256) void l0_row_max_kernel(const float* __restrict__ x, int lda, int K, float* __restrict__ row_max) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * lda;
    float mx = 0.f;
    const int nq = K / 4;
    for (int base = 0; base < nq; base += 256 * 10) {
        f32x4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = *reinterpret_cast<const f32x4*>(xr + 4 * min(base + (int)threadIdx.x + 256 * u, nq - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 10; ++u) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) row_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// scale 2^n of a row whose largest magnitude is m: m 2^n in [2^13, 2^14) (1 for an all-zero row; inf / nan rows: 1)
__device__ __forceinline__ int l0s_row_shift(float m) {
    const unsigned bits = __builtin_bit_cast(unsigned, m);
    const int ex = (int)((bits >> 23) & 0xffu);
    return (bits == 0 || ex == 0xff) ? 0 : max(-100, min(100, 14 - (ex - 126)));      // frexp exponent e = ex - 126: m in [2^(e-1), 2^e)
}

// RT = row tiles of 32 per workgroup (M <= 32 RT gridDim.z)
template <int RT>
__global__ __launch_bounds__(256) void l0_hs_stream_kernel(L0StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char l0s_lds[];
    unsigned char* xs = l0s_lds;                                                   // [2][32 RT][144]
    float* sc = reinterpret_cast<float*>(l0s_lds + (size_t)2 * 32 * RT * L0S_ROWB);   // [32 RT] row scale 2^n
    float* inv = sc + 32 * RT;                                                     // [32 RT] 2^-(n + wshift)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, g = lane >> 5;
    const int kbeg = (int)blockIdx.y * a.kps;
    const int klen = min(a.K, kbeg + a.kps) - kbeg;                                // a multiple of 32 (host)
    const int nchunk = klen / L0S_KC;
    const int n0 = (int)blockIdx.x * L0S_COLS + 32 * wave;
    // row block blockIdx.z of 32 RT rows (calls of more than 256 preambles: the weights of a (column group, k range) are read once per
    // row block, from the L2 after the first)
    const int mrow0 = (int)blockIdx.z * 32 * RT;
    a.x += (size_t)mrow0 * a.lda;
    if (a.row_max) a.row_max += mrow0;
    const int Mtot = a.M;
    a.M = min(a.M - mrow0, 32 * RT);

    // ---- first pass: the largest magnitude of every row inside this k range -> the row's scale.  8 consecutive lanes own a row; the
    // loads of a batch (10 per lane: the 320 k of the shipped split) are all requested before the first is used
    if (a.row_max) {
        for (int row = tid; row < 32 * RT; row += 256) {
            const int n = l0s_row_shift(a.row_max[min(row, a.M - 1)]);
            sc[row] = row < a.M ? __builtin_bit_cast(float, (unsigned)(n + 127) << 23) : 0.f;
            inv[row] = __builtin_bit_cast(float, (unsigned)(max(-126, min(126, -n - a.wshift)) + 127) << 23);
        }
    } else {
        constexpr int PB = 10, RB = RT >= 2 ? 2 : 1;             // row blocks of 32 per pass: 10 RB loads in flight per lane
        const int sub = tid & 7, nq = klen / 4;
#pragma unroll 1
        for (int r0 = 0; r0 < 32 * RT; r0 += 32 * RB) {
            float mx[RB];
            const float* xr[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                mx[b] = 0.f;
                xr[b] = a.x + (size_t)min(r0 + 32 * b + (tid >> 3), a.M - 1) * a.lda + kbeg;
            }
            for (int base = 0; base < nq; base += 8 * PB) {
                f32x4 v[RB][PB];
#pragma unroll
                for (int b = 0; b < RB; ++b)
#pragma unroll
                    for (int u = 0; u < PB; ++u) v[b][u] = *reinterpret_cast<const f32x4*>(xr[b] + 4 * min(base + sub + 8 * u, nq - 1));      // (clamped: a repeated quad does not change the maximum)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < RB; ++b)
#pragma unroll
                    for (int u = 0; u < PB; ++u) mx[b] = fmaxf(mx[b], fmaxf(fmaxf(fabsf(v[b][u][0]), fabsf(v[b][u][1])), fmaxf(fabsf(v[b][u][2]), fabsf(v[b][u][3]))));
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int row = r0 + 32 * b + (tid >> 3);
                float m = mx[b];
                m = fmaxf(m, __shfl_xor(m, 1));
                m = fmaxf(m, __shfl_xor(m, 2));
                m = fmaxf(m, __shfl_xor(m, 4));
                if (sub == 0 && row < 32 * RT) {
                    const int n = l0s_row_shift(m);
                    // (rows beyond M are staged as copies of row M - 1 times 0: the chunk loop below has no branch)
                    sc[row] = row < a.M ? __builtin_bit_cast(float, (unsigned)(n + 127) << 23) : 0.f;
                    inv[row] = __builtin_bit_cast(float, (unsigned)(max(-126, min(126, -n - a.wshift)) + 127) << 23);
                }
            }
        }
    }
    __syncthreads();

    // ---- staging of the A chunks: thread -> rows 32 t + (tid >> 3), 16-byte quad tid & 7 of the chunk's 32 k
    const int srow = tid >> 3, sq = tid & 7;
    f32x4 xv[2][RT];                      // chunks c + 1 and c + 2 in flight (set = chunk & 1)
    // The chunk loop is branch-free - rows beyond M read row M - 1 (scale 0), chunks beyond the range repeat the last one (never used): with a
    // branch inside it the compiler drains the vector-memory queue (s_waitcnt vmcnt(0)) at every chunk instead of counting
    const float* xrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) xrow[t] = a.x + (size_t)min(32 * t + srow, a.M - 1) * a.lda + kbeg + 4 * sq;
    auto load_x = [&](int c, f32x4 (&v)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; ++t) v[t] = *reinterpret_cast<const f32x4*>(xrow[t] + L0S_KC * min(c, nchunk - 1));
    };
    auto store_x = [&](int buf, const f32x4 (&v)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int row = 32 * t + srow;
            const float s = sc[row];
            uint2 hi, lo;
            hs_split2(v[t][0] * s, v[t][1] * s, hi.x, lo.x);
            hs_split2(v[t][2] * s, v[t][3] * s, hi.y, lo.y);
            unsigned char* d = xs + ((size_t)buf * 32 * RT + row) * L0S_ROWB + 8 * sq;
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + 64) = lo;
        }
    };

    // ---- B side: this lane's column (clamped: columns beyond N are never stored), k half g of every 16-k group
    const uint16_t* wp = a.Wh + (size_t)min(n0 + j, a.N - 1) * a.ldwh + 2 * (size_t)kbeg + 8 * g;
    uint4 wq[L0S_AHEAD][4];               // [chunk in flight][group 0 hi, group 0 lo, group 1 hi, group 1 lo]
    auto load_w = [&](int c, uint4 (&w)[4]) {
        const uint16_t* p = wp + (size_t)min(c, nchunk - 1) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const uint4*>(p + 16 * u);
    };

    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    load_x(0, xv[0]);
    load_x(1, xv[1]);
#pragma unroll
    for (int u = 0; u < L0S_AHEAD; ++u) load_w(u, wq[u]);
    store_x(0, xv[0]);
    __syncthreads();

    const unsigned char* arow = xs + (size_t)j * L0S_ROWB + 16 * g;
    // chunk c: MFMAs on LDS buffer c & 1 and weight set c % L0S_AHEAD; requests the preamble quads of chunk c + 2 (into the set chunk c's
    // came from) and the weights of chunk c + L0S_AHEAD (into its own set); converts chunk c + 1 - requested a whole chunk earlier - into
    // the other LDS buffer
    auto chunk = [&](int c, uint4 (&w)[4], f32x4 (&xc)[RT], const f32x4 (&xn)[RT]) {
        const int buf = c & 1;
        load_x(c + 2, xc);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const f16x8 bh = __builtin_bit_cast(f16x8, w[2 * grp]), bl = __builtin_bit_cast(f16x8, w[2 * grp + 1]);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const unsigned char* p = arow + ((size_t)buf * 32 * RT + 32 * t) * L0S_ROWB + 32 * grp;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(p), al = *reinterpret_cast<const f16x8*>(p + 64);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            }
        }
        load_w(c + L0S_AHEAD, w);                                      // this chunk's registers are free again
        store_x(buf ^ 1, xn);
        __syncthreads();
    };
    // the loop is unrolled by L0S_AHEAD (even) so that the register sets stay registers
    static_assert(L0S_AHEAD % 2 == 0, "weight sets and preamble sets rotate together");
    int c = 0;
    for (; c + L0S_AHEAD <= nchunk; c += L0S_AHEAD) {
#pragma unroll
        for (int u = 0; u < L0S_AHEAD; ++u) chunk(c + u, wq[u], xv[u & 1], xv[(u & 1) ^ 1]);
    }
#pragma unroll
    for (int u = 0; u < L0S_AHEAD - 1; ++u)
        if (c + u < nchunk) chunk(c + u, wq[u], xv[u & 1], xv[(u & 1) ^ 1]);

    // ---- epilogue: D column = lane & 31 (this lane's weight column), D row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3); undo the row's scale
    const int col = n0 + j;
    if (col >= a.N) return;
    float* out = a.slabs + ((size_t)blockIdx.y * Mtot + mrow0) * a.N + col;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * t + 8 * (r >> 2) + 4 * g + (r & 3);
            if (row < a.M) out[(size_t)row * a.N] = acc[t][r] * inv[row];
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel for bf16 contexts (csi_dtype = CSI_DTYPE_BF16, BASELINE configs[2]'s arithmetic: bf16 operands, fp32 accumulation):
// one v_mfma_f32_32x32x16_bf16 per 16 k and row tile, the preambles rounded to bf16 exactly as f32_to_bf16_kernel rounds them, no input
// scale (bf16 carries the fp32 exponent).  Before it a bf16 call of fewer than 256 row tiles of 256 x 256 ran layer 0 as eight
// workgroups of the 128 x 128 kernel over the whole K: 223 us whatever the call's size (545 us per one-packet call at Nt = 64).
struct L0StreamBf16Args {
    const float* x;        // [M][lda] fp32
    const bf16_t* Wb;      // [N][ldwb] bf16 layer-0 weights (LTF columns), K-major
    float* slabs;          // [gridDim.y][M][N] fp32
    int M, N, K, lda, ldwb;
    int kps;               // k-columns per blockIdx.y, a multiple of 32
};
constexpr int L0B_ROWB = 80;       // LDS bytes per row and chunk: 32 bf16 | 16 bytes of padding (conflict-free 16-byte reads)
inline size_t l0b_lds_bytes(int rt) { return (size_t)2 * 32 * rt * L0B_ROWB; }

template <int RT>
__global__ __launch_bounds__(256) void l0_bf16_stream_kernel(L0StreamBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char l0s_lds[];
    unsigned char* xs = l0s_lds;                                                   // [2][32 RT][80]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, g = lane >> 5;
    const int kbeg = (int)blockIdx.y * a.kps;
    const int klen = min(a.K, kbeg + a.kps) - kbeg;
    const int nchunk = klen / L0S_KC;
    const int n0 = (int)blockIdx.x * L0S_COLS + 32 * wave;
    const int mrow0 = (int)blockIdx.z * 32 * RT;
    a.x += (size_t)mrow0 * a.lda;
    const int Mtot = a.M;
    a.M = min(a.M - mrow0, 32 * RT);

    const int srow = tid >> 3, sq = tid & 7;
    f32x4 xv[2][RT];
    const float* xrow[RT];
    float live[RT];                       // rows beyond M read row M - 1 and are staged as zeros (the chunk loop has no branch)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        xrow[t] = a.x + (size_t)min(32 * t + srow, a.M - 1) * a.lda + kbeg + 4 * sq;
        live[t] = 32 * t + srow < a.M ? 1.f : 0.f;
    }
    auto load_x = [&](int c, f32x4 (&v)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; ++t) v[t] = *reinterpret_cast<const f32x4*>(xrow[t] + L0S_KC * min(c, nchunk - 1));
    };
    auto store_x = [&](int buf, const f32x4 (&v)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            uint2 o;
            o.x = (uint32_t)f2bf(v[t][0] * live[t]) | ((uint32_t)f2bf(v[t][1] * live[t]) << 16);
            o.y = (uint32_t)f2bf(v[t][2] * live[t]) | ((uint32_t)f2bf(v[t][3] * live[t]) << 16);
            *reinterpret_cast<uint2*>(xs + ((size_t)buf * 32 * RT + 32 * t + srow) * L0B_ROWB + 8 * sq) = o;
        }
    };
    // B side: this lane's column, k half g of every 16-k group: two 16-byte loads per 32-k chunk
    const bf16_t* wp = a.Wb + (size_t)min(n0 + j, a.N - 1) * a.ldwb + (size_t)kbeg + 8 * g;
    uint4 wq[L0S_AHEAD][2];
    auto load_w = [&](int c, uint4 (&w)[2]) {
        const bf16_t* p = wp + (size_t)min(c, nchunk - 1) * 32;
        w[0] = *reinterpret_cast<const uint4*>(p);
        w[1] = *reinterpret_cast<const uint4*>(p + 16);
    };
    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    load_x(0, xv[0]);
    load_x(1, xv[1]);
#pragma unroll
    for (int u = 0; u < L0S_AHEAD; ++u) load_w(u, wq[u]);
    store_x(0, xv[0]);
    __syncthreads();
    const unsigned char* arow = xs + (size_t)j * L0B_ROWB + 16 * g;
    auto chunk = [&](int c, uint4 (&w)[2], f32x4 (&xc)[RT], const f32x4 (&xn)[RT]) {
        const int buf = c & 1;
        load_x(c + 2, xc);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, w[grp]);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + ((size_t)buf * 32 * RT + 32 * t) * L0B_ROWB + 32 * grp);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[t], 0, 0, 0);
            }
        }
        load_w(c + L0S_AHEAD, w);
        store_x(buf ^ 1, xn);
        __syncthreads();
    };
    int c = 0;
    for (; c + L0S_AHEAD <= nchunk; c += L0S_AHEAD) {
#pragma unroll
        for (int u = 0; u < L0S_AHEAD; ++u) chunk(c + u, wq[u], xv[u & 1], xv[(u & 1) ^ 1]);
    }
#pragma unroll
    for (int u = 0; u < L0S_AHEAD - 1; ++u)
        if (c + u < nchunk) chunk(c + u, wq[u], xv[u & 1], xv[(u & 1) ^ 1]);
    const int col = n0 + j;
    if (col >= a.N) return;
    float* out = a.slabs + ((size_t)blockIdx.y * Mtot + mrow0) * a.N + col;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * t + 8 * (r >> 2) + 4 * g + (r & 3);
            if (row < a.M) out[(size_t)row * a.N] = acc[t][r];
        }
}

}  // namespace csi
