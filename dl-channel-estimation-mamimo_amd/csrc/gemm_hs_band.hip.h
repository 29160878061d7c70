// gemm_hs_band.hip.h - host / device pieces the library needs for the fused band kernel (first per-pair layer + regressor of the
// shipped network, massiveMIMO_CSI_prediction_DNN.py:211-227, as ONE kernel with h2 in registers): its argument records, the
// shape predicates, the k permutation of the regressor weights, the weight converter and the slab-ordered pilot table.
// The kernel the library launches is generated assembly (band_kernel_gen.py: 8 waves, two per SIMD); the first form of the
// idea - C++, 4 waves x 512 registers - lives in tools/gemm_hs_band4.hip.h (probe only: it is slower).
#pragma once
#include "gemm_hs.hip.h"
#include <type_traits>
#include <utility>

namespace csi {

constexpr int BAND_ROWS = 128;
constexpr int BAND_THREADS = 256;
constexpr int BAND_SLOT_BYTES = 16384;           // one weight sub-tile: 256 rows x 64 B
constexpr int BAND_NSLOT = 4;

struct BandArgs {
    // stage 1
    const float* L0;       // [M / nt][ldl] fp32: layer-0 product per (packet, rx)
    const float* Ts;       // [nt][ldl] fp32: in_scale * (pilot table incl. bias)
    int ldl, nt;
    float in_scale;        // 2^sa1
    const uint16_t* W1;    // hs [N1][ldb1 halves], K-major, bn0's scale folded into its columns
    int ldb1;
    const float* bias1;    // [N1] (incl. bn0's shift through W1)
    int M, K1, N1;         // K1 % 64 == 0, N1 % 256 == 0
    float acc_scale1;      // 2^-(sa1 + sw1)
    float out_scale;       // 2^sa2: h2 is carried as out_scale * relu(z1)
    // stage 2
    const uint16_t* W2p;   // hs [256][ldb2 halves]: regressor weights, rows >= n2 zero, k permuted (hs_band_kperm), bn1's scale folded
    int ldb2;
    const float* bias2;    // [n2] incl. bn1's shift through W2
    int n2;                // <= 256
    float acc_scale2;      // 2^-(sa2 + sw2)
    float* out;            // [M][ldo] fp32
    int ldo;
    unsigned* peak;        // range guard (may be null), see hs_report_peak
    const unsigned long long* stamps;   // timing probe, null in the library
};

// Kernel arguments of the assembly form (band_kernel_gen.py: "csi_band8", 8 waves, two per SIMD): 128 bytes, loaded with two
// s_load_dwordx16 - the layout is part of that kernel.
struct Band8Args {
    const float* L0;
    const float* Ts;
    const uint16_t* W1;
    const float* bias1;
    const uint16_t* W2p;
    const float* bias2;
    float* out;
    unsigned* peak;
    const unsigned long long* stamps;
    int ldl, nt, ldb1, M, K1, N1, ldb2, n2, ldo;
    float in_scale, as1os, out_scale, acc_scale2;
    unsigned nt_magic;           // floor(2^32 / nt): row -> (pair row, tx antenna) by multiplication
};
static_assert(sizeof(Band8Args) == 128, "Band8Args is read by s_load_dwordx16 x 2");
// the column-split form ("csi_band8_cs", grid (bands, splits)): the same record describing split 0 - N1 = the hidden features ONE split
// computes - followed by the base of the partial outputs of splits 1 .. ([splits - 1][M][ldo] fp32)
struct Band8ArgsCs {
    Band8Args a;
    float* part;
    unsigned long long pad;
};
static_assert(sizeof(Band8ArgsCs) == 144, "Band8ArgsCs: s_load_dwordx16 x 2 + s_load_dwordx2 at 0x80");
constexpr int BAND8_THREADS = 512;
constexpr int BAND8_MAX_N1 = 4096;

inline Band8Args band8_args(const BandArgs& g) {
    Band8Args a{};
    a.L0 = g.L0; a.Ts = g.Ts; a.W1 = g.W1; a.bias1 = g.bias1; a.W2p = g.W2p; a.bias2 = g.bias2; a.out = g.out; a.peak = g.peak;
    a.stamps = g.stamps;
    a.ldl = g.ldl; a.nt = g.nt; a.ldb1 = g.ldb1; a.M = g.M; a.K1 = g.K1; a.N1 = g.N1; a.ldb2 = g.ldb2; a.n2 = g.n2; a.ldo = g.ldo;
    a.in_scale = g.in_scale; a.as1os = g.acc_scale1 * g.out_scale; a.out_scale = g.out_scale; a.acc_scale2 = g.acc_scale2;
    a.nt_magic = g.nt <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned long long)g.nt);
    return a;
}
// shapes the assembly kernel serves (the caller falls back to the separate kernels otherwise); bf16: sub-tiles of 32 k
inline bool band8_serves(const BandArgs& g, bool bf16 = false) {
    return g.K1 >= (bf16 ? 256 : 128) && (g.K1 % (bf16 ? 128 : 64)) == 0 && g.N1 >= 256 && (g.N1 % 256) == 0 && g.N1 <= BAND8_MAX_N1 && g.n2 >= 1 && g.n2 <= 256 &&
           g.nt >= 1 && (unsigned long long)g.M * (unsigned long long)g.nt < 0xffffffffull && (unsigned long long)g.M * g.ldo * 4ull < 0xffffffffull &&
           (unsigned long long)((g.M + g.nt - 1) / g.nt) * g.ldl * 4ull < 0x7fffffffull;
}

// out[i] += part[0][i] + part[1][i] + ... in split order (the regressor sums of the column-split band kernel; split 0 wrote out with the bias)
__global__ void band_split_sum_kernel(float* __restrict__ out, const float* __restrict__ part, size_t n, size_t stride, int extra) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = out[i];
        for (int s = 0; s < extra; ++s) v += part[(size_t)s * stride + i];
        out[i] = v;
    }
}

// position p of a 16-k group of the permuted regressor weights holds original k-column hs_band_kperm(p)
__host__ __device__ __forceinline__ constexpr int hs_band_kperm(int p) { return (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1); }

// dst (hs [256][ldh halves]) = split(scale * W2[n2][cols]) with the k permutation of the band kernel inside every 16-group,
// rows >= n2 and columns >= cols zero.  One thread = 8 positions of a row.
__global__ void f32_to_hs_band_w2_kernel(const float* __restrict__ src, int ld_src, int n2, int cols, uint16_t* __restrict__ dst, int ldh, float scale) {
    const int c8 = ldh >> 4;
    const size_t total = (size_t)256 * c8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c8), c = (int)(i - (size_t)r * c8) * 8;      // positions c .. c + 7 of row r
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = c + e, k = (p & ~15) + hs_band_kperm(p & 15);
            v[e] = (r < n2 && k < cols) ? src[(size_t)r * ld_src + k] : 0.f;
        }
        uint4 oh, ol;
        hs_split2(v[0] * scale, v[1] * scale, oh.x, ol.x);
        hs_split2(v[2] * scale, v[3] * scale, oh.y, ol.y);
        hs_split2(v[4] * scale, v[5] * scale, oh.z, ol.z);
        hs_split2(v[6] * scale, v[7] * scale, oh.w, ol.w);
        uint16_t* d = dst + (size_t)r * ldh + (c >> 4) * 32 + (c & 8);
        *reinterpret_cast<uint4*>(d) = oh;
        *reinterpret_cast<uint4*>(d + 16) = ol;
    }
}

// The pilot table in the order the staged band kernels stream it (band_kernel_gen.py, TS_OFF): slab u = k-columns SK u .. + SK - 1
// of all nt rows, [nt][SK] floats (SK = 32: bf16 form, 16: split-f16 form), the 16-byte units of a row XOR-swizzled - by (row >> 1) & 7
// for 8 units per row, (row >> 2) & 3 for 4 - so that the ds_read_b128 of 16 consecutive rows touches every bank once; K1 / SK slabs
// plus one the kernel requests past the end of a column step and never reads.
template <int SK>
__global__ void band_tsw_kernel(const float* __restrict__ T, int ldt, int nt, int K1, float* __restrict__ dst) {
    constexpr int U = SK / 4;
    const size_t total = (size_t)(K1 / SK + 1) * nt * U;              // 16-byte units
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pos = (int)(i % U);
        const size_t rt = i / U;
        const int t = (int)(rt % nt), u = (int)(rt / nt);
        const int k = SK * u + 4 * (pos ^ ((t >> (U == 8 ? 1 : 2)) & (U - 1)));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K1) v = *reinterpret_cast<const float4*>(T + (size_t)t * ldt + k);
        *reinterpret_cast<float4*>(dst + i * 4) = v;
    }
}
// The weight streams of the register-blocked band kernel (band4_kernel_gen.py) PRE-TILED: every 16-KiB sub-tile [256 rows][32 k] bf16 stored
// contiguously, in the order the kernel streams them, in the swizzled image the ring slot holds (unit pos of row r = k-columns
// 8 (pos ^ ((r >> 2) & 3)) .. + 7 of the sub-tile) - so that an LDS-DMA piece is 1 KiB of whole 128-byte lines (the K-major matrix gives it
// sixteen 64-byte row segments) and the four pieces of a sub-tile differ by an immediate.
//   mode 0, pair layer W1 [N1][ld]: tile (c, u) = rows 256 c .., k-columns 32 u ..; tiles c-major; `spare` zero tiles behind the last.
//   mode 1, regressor W2p [256][ld]: tile (c, q) = all 256 rows, k-columns 256 c + 128 (q & 1) + 32 (q >> 1) .. (fragment order of the kernel:
//   the h2 fragments alternate between the halves).
//   mode 2, the split-f16 form's regressor (hs [256][ld halves], 16 k = 32 halves as hi | lo per group): tile (c, q), q = 0 .. 15, = the group of
//   k-columns 256 c + 128 (q & 1) + 32 (q >> 2) + 16 ((q >> 1) & 1) ..; its pair layer is mode 0 with nsub = K1 / 16 (32 halves per sub-tile row too).
__global__ void band4_tile_kernel(const uint16_t* __restrict__ src, int ld, int ncol, int nsub, int mode, int spare, uint16_t* __restrict__ dst) {
    const size_t units = (size_t)(ncol * nsub + spare) * 1024;            // 16-byte units
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (size_t)gridDim.x * blockDim.x) {
        const int pos = (int)(i & 3), r = (int)((i >> 2) & 255);
        const size_t t = i >> 10;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t < (size_t)ncol * nsub) {
            const int c = (int)(t / nsub), u = (int)(t % nsub);
            const int row = mode == 0 ? 256 * c + r : r;
            const int k0 = mode == 0 ? 32 * u : (mode == 1 ? 256 * c + 128 * (u & 1) + 32 * (u >> 1) : 512 * c + 256 * (u & 1) + 64 * (u >> 2) + 32 * ((u >> 1) & 1));
            const int k = k0 + 8 * (pos ^ ((r >> 2) & 3));      // (in 2-byte elements: bf16 values / f16 halves)
            v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + k);
        }
        *reinterpret_cast<uint4*>(dst + i * 8) = v;
    }
}

// which form serves a call: the staged ones need the band's L0 rows in one 1-KiB slab (8 rows of 32 k / 16 rows of 16 k) and the T slab in 8 KiB
inline bool band8_staged(const BandArgs& g, bool bf16 = true) { return bf16 ? (g.nt >= 32 && g.nt <= 64) : (g.nt >= 16 && g.nt <= 128); }

}  // namespace csi
