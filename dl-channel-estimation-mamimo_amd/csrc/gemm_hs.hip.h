// gemm_hs.hip.h - the fp32 dense layers (massiveMIMO_CSI_prediction_DNN.py:211-227) on the f16
// matrix cores with SPLIT operands: fp32-grade results at a multiple of the fp32 MFMA rate.
//
// Arithmetic.  Every fp32 operand x (times a power of two s chosen per matrix, exact) is carried as
//     hi = f16(s*x)   (round to nearest even),      lo = f16(s*x - hi)
// so that s*x = hi + lo + e with |e| <= 2^-23 |s*x| (two 11-bit significands; below the f16 normal
// range the absolute error is 2^-25).  A product of two such operands is evaluated as
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped a_lo*b_lo is <= 2^-22 |a*b|)
// = three v_mfma_f32_32x32x16_f16 per 16 k-columns.  f16 x f16 products are exact in fp32 and the
// accumulation is fp32, exactly as in the native fp32 MFMA path; the per-term operand error
// (~2^-22) is below the fp32 accumulation error of a K = 1024 reduction, so the result meets the
// same 1e-5 norm-relative contract as gemm_f32.hip.h (measured: DESIGN.md 4.6).  The accumulator is
// multiplied by 2^-(sa+sw) in the epilogue (exact) before bias / relu / BatchNormalization.
// Range: |s*x| must stay below 65504 and well above the f16 denormals - weights are scaled at load so
// that their maximum sits at 2^12..2^13, the preambles per launch from a sampled maximum
// (hs_absmax_sample_kernel), hidden activations per layer from the BatchNormalization vectors; both
// ends are guarded on the device (hs_report_peak).
//
// Storage ("hs" matrices, 4 bytes per element like fp32).  Row-major, K contiguous, in groups of 16
// k-columns: 16 hi halves (32 B) followed by the 16 lo halves (32 B).  One 64-byte group of a row is
// one image row of a ping-pong sub-tile, i.e. byte for byte the layout gemm_bf16_pp_kernel stages
// (its "k-columns 0-15" are the hi plane, "16-31" the lo plane), so the LDS-DMA addressing, the
// XOR swizzle and the fragment reads are shared with that kernel.
//
// Kernels: gemm_hs_pp_kernel (both operands hs in HBM), gemm_hs_pp_pair_kernel (A generated in the
// kernel from L0 + T - the first per-pair layer - or converted from fp32 rows - layer 0), both on
// the 256x256 / 8-wave ping-pong schedule of gemm_bf16.hip.h with THREE 8-MFMA phases per
// sub-tile:    P0: a_hi x b_lo    P1: a_hi x b_hi    P2: a_lo x b_hi
// so that each phase prefetches into fragment registers the previous phase has released (b_hi in
// P0, a_lo in P1, next a_hi + b_lo in P2) - 48 fragment registers, no double buffering.
#pragma once
#include "gemm_bf16.hip.h"

namespace csi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

constexpr int HS_G = 16;                       // k-columns per group (= per ping-pong sub-tile)

struct GemmHsArgs {
    const uint16_t* A;     // hs [M][lda]   (lda in halves = 2 * K rounded up to 16)
    const uint16_t* Bt;    // hs [N][ldb]   weights, K-major
    void* C;               // fp32 [M][ldc] (EPI_RAW slabs / EPI_BIAS) or hs [M][ldc halves]
    int M, N, K;           // K = reduction length actually walked (multiple of 16 as stored)
    int lda, ldb, ldc;
    int k_per_split;       // real k-columns per blockIdx.z, multiple of 16
    int tiles_n;
    float acc_scale;       // 2^-(sa + sw): undoes the operand scaling (exact)
    float out_scale;       // hs output: 2^sa of the layer that reads it
    const float* bias;
    const float* scale;    // BN scale / shift
    const float* shift;
    unsigned* peak;        // range guard (may be null): two words, see hs_report_peak
    const unsigned* dyn_max;   // CAST mode, automatic input scale: bits of a (sampled) max |x| of this launch's rows,
    int wshift;                //   written earlier on the stream by hs_absmax_sample_kernel; scale = 2^(14 - exponent)
    const unsigned long long* stamps;   // timing probe (tools/hs_probe.hip), null otherwise: hs_stamp
    int xcd_cols;              // pair kernel: 1 = a column tile stays on one XCD (weights tile resident in that L2), see hs_tile_of_block
    int a_blk, c_blk;          // hs ACTIVATION matrices (A input / C output) in the blocked layout, see hs_blk_offset
};

// Optional per-workgroup time stamps (timing probes; null in the library): wave 0 writes (shader cycles, 10-ns
// wall ticks) at kernel entry, after the prologue, after the main loop and after the epilogue (slots 0..3; the fused
// regressor stage adds 4 = both halves computed, 5 = flag wait over; 6 slots of 16 bytes per workgroup).
__device__ __forceinline__ void hs_stamp(const unsigned long long* base, int i) {
    if (!base) return;
    if (threadIdx.x == 0) {
        unsigned long long* p = const_cast<unsigned long long*>(base) + ((size_t)blockIdx.x * 6 + i) * 2;
        p[0] = __builtin_readcyclecounter();
        p[1] = wall_clock64();
    }
}

// Blocked layout of hs activation matrices: [16-row block][k-group of 16][16 rows][64 B], i.e. the 16 rows x 64 B
// that ONE LDS-DMA piece of the consuming GEMM fetches are 1 KiB of contiguous memory instead of sixteen 64-byte
// runs 4 KiB apart (row-major).  A layer that reads its A operand once from HBM - the regressor: one column tile, no
// reuse through L2 - is bound by how HBM likes its requests: 2.6 TB/s row-major at config 2.  The producer's stores
// stay whole lines: the 8 rows a store instruction covers are adjacent 64-byte runs of one k-group.
// Offset in halves of (row, k-group) for a matrix of ld halves per row (ld = 2 K):
__device__ __host__ __forceinline__ size_t hs_blk_offset(int row, int kgroup, int ld) {
    return ((size_t)(row >> 4) * (ld >> 5) + kgroup) * 512 + (size_t)(row & 15) * 32;
}

// Workgroup -> tile.  Workgroup b runs on XCD b % 8 (round-robin dispatch; an affinity for speed, nothing depends on it).
//   rows-per-XCD (default): XCD x takes row tiles == x (mod 8) and all their column tiles - the A rows of a row tile enter
//     one L2 once.  Right when A is the big operand (layer 0: 10 MB of preambles per row tile).
//   columns-per-XCD (xcd_cols, needs 8 % tiles_n == 0): XCD x takes column tile x % tiles_n only - its 1 MiB slice of the
//     weights stays resident in that 4 MiB L2 instead of the whole 4 MiB matrix competing with the output stream for
//     it.  Right for the first per-pair layer, whose A side is 32 KB per row tile.
__device__ __forceinline__ void hs_tile_of_block(const GemmHsArgs& g, int& tm, int& tn) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (g.xcd_cols && (8 % g.tiles_n) == 0) {
        const int per = 8 / g.tiles_n;                 // row tiles per 8 consecutive workgroups
        tn = xcd % g.tiles_n;
        tm = idx * per + xcd / g.tiles_n;
    } else {
        tm = (idx / g.tiles_n) * 8 + xcd;
        tn = idx % g.tiles_n;
    }
}

// (a, b) -> packed hi halves, packed lo halves
__device__ __forceinline__ void hs_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2 x = {a, b};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// lo halves of two values whose packed hi halves are `hi`:  f16(a - f32(hi.lo)) | f16(b - f32(hi.hi)) << 16.
// v_fma_mixlo/hi_f16 take the f16 operand as it is (op_sel_hi), evaluate hi * (-1) + x in fp32 - exact, the
// difference of x and its own 11-bit rounding is an fp32 number - and round once to f16: one VALU per value
// instead of v_cvt_f32_f16 + v_sub_f32 + half a v_cvt_pk_f16_f32, and bit-identical to hs_split2.
__device__ __forceinline__ uint32_t hs_lo_pair(float a, float b, uint32_t hi) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t hs_hi_pair(float a, float b) {
    const f32x2 x = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
}

// Range guard.  |s*x| above 65504 becomes inf in the hi half; below the f16 normal range the lo half is
// a denormal with an absolute error of 2^-25, i.e. a row whose scaled magnitudes are all tiny loses
// relative accuracy (rms 0.016 -> 2e-6, rms 0.003 -> 1e-5).  Kernels that convert operands keep a
// running maximum per lane (one v_max3 per two values; in the A-generating kernels a lane converts ONE
// matrix row, so its maximum is a row maximum) and report once at their end: peak[0] = atomicMax of
// the bits of any magnitude above HS_PEAK_REPORT, peak[1] |= 1 if a lane's non-zero maximum stayed
// below HS_LOW_REPORT.  The host then tells the caller / repeats the call on the fp32 MFMA kernels
// (csi_mamimo.hip).
constexpr float HS_PEAK_REPORT = 60000.f;
constexpr float HS_LOW_REPORT = 0.0625f;
__device__ __forceinline__ float hs_absmax(float m, float a, float b) { return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))); }
__device__ __forceinline__ void hs_report_peak(unsigned* peak, float m, bool row_maximum) {
    if (!peak) return;
    if (m > HS_PEAK_REPORT) atomicMax(peak, __builtin_bit_cast(unsigned, m));
    if (row_maximum && m > 0.f && m < HS_LOW_REPORT) atomicOr(peak + 1, 1u);
}

// Epilogue of the hs ping-pong kernels (8 waves as 2 x 4, 128 x 64 per wave).  fp32 output straight
// from the C/D layout.  hs output: per half of the wave's rows, (hi | lo << 16) words go through a
// wave-private LDS image [64 rows][64 columns]; a lane then owns 8 columns of a row, separates the
// planes with two v_perm_b32 per word pair and stores 16 B of hi and 16 B of lo (the 8 lanes of a
// row write 256 contiguous bytes).  The ring is idle by then (see pp_epilogue).
// NOSTORE (timing probe): everything but the global stores (they sit behind a condition that is false at run time).
template <int EPI, bool OUT_HS, bool NOSTORE = false>
__device__ __forceinline__ void hs_epilogue(f32x16 (&acc)[4][2], const GemmHsArgs& g, float* lds, int m0, int n0, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool full_rows = (m0 + PP_BM) <= g.M;
    const float as = g.acc_scale;
    if constexpr (OUT_HS) {
        const float os = g.out_scale;
        float pk = 0.f;
        uint32_t* ep = reinterpret_cast<uint32_t*>(lds) + wave * 4096;
        const int cg = lane & 7;
        const int col8 = n0 + wn * 64 + cg * 8;
        // halves offset of this lane's 8 columns inside a row: group (col8 / 16), hi part
        uint16_t* cb = reinterpret_cast<uint16_t*>(g.C) + (g.c_blk ? 0 : (col8 >> 4) * 32) + (col8 & 8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int colc = min(n0 + wn * 64 + nj * 32 + l31, g.N - 1);
                // out = os * (bn(relu(as * acc + bias)))  with os folded into the per-column constants
                float bias = 0.f, sc = os, sh = 0.f;
                if (EPI != EPI_RAW) bias = g.bias[colc];
                if (EPI == EPI_BIAS_RELU_AFFINE) { sc = g.scale[colc] * os; sh = g.shift[colc] * os; }
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float v0 = fmaf(acc[2 * half + m2][nj][r], as, bias), v1 = fmaf(acc[2 * half + m2][nj][r + 1], as, bias);
                        if (EPI == EPI_BIAS_RELU_AFFINE) { v0 = fmaf(fmaxf(v0, 0.f), sc, sh); v1 = fmaf(fmaxf(v1, 0.f), sc, sh); }
                        else { v0 *= sc; v1 *= sc; }
                        pk = hs_absmax(pk, v0, v1);
                        const uint32_t h = hs_hi_pair(v0, v1), l = hs_lo_pair(v0, v1, h);
                        const int row = m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;      // rows row, row + 1
                        ep[row * 64 + nj * 32 + l31] = (h & 0xffffu) | (l << 16);
                        ep[(row + 1) * 64 + nj * 32 + l31] = (h >> 16) | (l & 0xffff0000u);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                const uint4 w0 = *reinterpret_cast<const uint4*>(ep + rl * 64 + cg * 8);
                const uint4 w1 = *reinterpret_cast<const uint4*>(ep + rl * 64 + cg * 8 + 4);
                uint4 vh, vl;
                vh.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x05040100u);  vl.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x07060302u);
                vh.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x05040100u);  vl.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x07060302u);
                vh.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x05040100u);  vl.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x07060302u);
                vh.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x05040100u);  vl.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x07060302u);
                const int row = m0 + wm * 128 + half * 64 + rl;
                if (col8 < g.N && (full_rows || row < g.M) && (!NOSTORE || g.M < 0)) {
                    uint16_t* d = cb + (g.c_blk ? hs_blk_offset(row, col8 >> 4, g.ldc) : (size_t)row * g.ldc);
                    *reinterpret_cast<uint4*>(d) = vh;
                    *reinterpret_cast<uint4*>(d + 16) = vl;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        hs_report_peak(g.peak, pk, false);       // a lane holds two COLUMNS here (a dead feature may be tiny everywhere)
    } else {
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
                v *= as;
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

// The three MFMA phases of a sub-tile, shared by both kernels.  Fragment registers: a_hi, a_lo (4
// row tiles each), b_hi, b_lo (2 column tiles each).  rd(slot, c, ...) reads chunk pair c (0 = hi
// plane, 1 = lo plane) of a sub-tile.
struct HsFrags {
    f16x8 a_hi[4], a_lo[4], b_hi[2], b_lo[2];
};

// SWAP: the operands change places, acc[i][j] then holds the TRANSPOSED 32x32 tile - lanes run over the rows of A
// (a lane owns one activation row), registers over the columns (four consecutive ones per register quad): the
// layout the fused regressor stage needs to turn the tile into an A image without a transposition.
template <int PH, bool SWAP = false, typename RdA, typename RdB>
__device__ __forceinline__ void hs_mfma_seg(f32x16 (&acc)[4][2], HsFrags& f, RdA&& read_a, RdB&& read_b, bool more, int slot, int nslot,
                                            bool closing_barrier) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (PH == 0) read_b(slot, 0, f.b_hi);
    if (PH == 1) read_a(slot, 1, f.a_lo);
    if (PH == 2 && more) {
        read_a(nslot, 0, f.a_hi);
        read_b(nslot, 1, f.b_lo);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16x8 a = PH == 2 ? f.a_lo[i] : f.a_hi[i];
            const f16x8 b = PH == 0 ? f.b_lo[j] : f.b_hi[j];
            acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
        }
    constexpr int NRD = PH == 0 ? 2 : (PH == 1 ? 4 : 6);
    if (PH != 2 || more) {
#pragma unroll
        for (int q = 0; q < NRD; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8 - NRD, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    pp_wait_lgkm();
    __builtin_amdgcn_sched_barrier(0);
    if (closing_barrier) pp_barrier();
}

// ---------------------------------------------------------------------------------------------
// Both operands hs in HBM.  Ring of NSUB sub-tiles of 16 k-columns (32 KiB each: 256 A rows + 256 B
// rows x 64 B), each wave DMAs 4 one-KiB pieces per sub-tile (2 + 1 + 1 over the three phases), D =
// NSUB - 1 sub-tiles ahead.  Sub-tile u+1 is first read by the prefetch in P2(u); its pieces are
// waited for in the load segment of P1(u) (vmcnt leaves the 4(D-2)+3 younger pieces in flight) - a
// full phase earlier, so that the barrier pair in between publishes the other group's pieces too.
// Slot of sub-tile u+D = slot of u-1, whose last fragment read (a_lo, P1(u-1)) was retired two
// barriers before either group issues into it.
template <int EPI, bool OUT_HS, int NSUB = 5, int DBG = 0>
__global__ __launch_bounds__(PP_THREADS, 1) void gemm_hs_pp_kernel(const GemmHsArgs g) {
    constexpr int D = NSUB - 1;
    __shared__ __attribute__((aligned(16))) float lds[NSUB * PP_SUBF];

    auto bar = [] { if (!(DBG & 16)) pp_barrier(); };        // DBG 16: timing probe without barriers (results invalid)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    int tm, tn;
    hs_tile_of_block(g, tm, tn);
    if (tm * PP_BM >= g.M) return;
    hs_stamp(g.stamps, 0);
    const int m0 = tm * PP_BM, n0 = tn * PP_BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nsub = (kend - kbeg + HS_G - 1) / HS_G;

    const bool a_side = wave < 4;
    const bool ablk = a_side && g.a_blk;
    const uint16_t* tile_base = a_side ? (ablk ? g.A + hs_blk_offset(m0, kbeg >> 4, g.lda) : g.A + (size_t)m0 * g.lda + 2 * kbeg)
                                       : g.Bt + (size_t)n0 * g.ldb + 2 * kbeg;
    const int ld = a_side ? g.lda : g.ldb;
    const int rmax = a_side ? g.M - 1 - m0 : g.N - 1 - n0;
    const int sstep = ablk ? 1024 : 64;              // bytes from one k-group of a row (block) to the next
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tile_base, 0, 0x7fffffff, 0x00020000);
    int voff[PP_PW];
#pragma unroll
    for (int u = 0; u < PP_PW; ++u) {
        const int row = 16 * ((wave & 3) * PP_PW + u) + (lane >> 2);
        const int clog = (lane & 3) ^ ((row >> 2) & 3);
        const int rc = min(row, rmax);
        voff[u] = ablk ? (int)(hs_blk_offset(rc, 0, ld) + clog * 8) * 2 : (rc * ld + clog * 8) * 2;
    }
    auto issue = [&](int sub, int slot, int u) {
        if (DBG & 1) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + slot * PP_SUBF + (wave * PP_PW + u) * 256),
                                                 16, voff[u], sub * sstep, 0, 0);
    };

    const int fswz = (l31 >> 2) & 3;
    int xo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;            // floats
    const int abase = (wm * 128 + l31) * PP_ROWF;
    const int bbase = (PP_BM + wn * 64 + l31) * PP_ROWF;
    bool skip_reads = false;           // DBG 32: timing probe, fragments read once (results invalid)
    auto read_a = [&](int slot, int c, f16x8 (&a)[4]) {
        if ((DBG & 32) && skip_reads) return;
        const float* st = lds + slot * PP_SUBF + abase + xo[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(st + i * 32 * PP_ROWF));
    };
    auto read_b = [&](int slot, int c, f16x8 (&b)[2]) {
        if ((DBG & 32) && skip_reads) return;
        const float* st = lds + slot * PP_SUBF + bbase + xo[c];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(st + j * 32 * PP_ROWF));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    HsFrags f;

    const int npro = min(nsub, D);
    for (int t = 0; t < npro; ++t)
#pragma unroll
        for (int u = 0; u < PP_PW; ++u) issue(t, t, u);
    pp_wait_vm_lgkm_rt(PP_PW * (npro - 1));
    bar();                       // sub-tile 0 visible to every wave
    read_a(0, 0, f.a_hi);
    read_b(0, 1, f.b_lo);
    if (DBG & 32) { read_a(0, 1, f.a_lo); read_b(0, 0, f.b_hi); skip_reads = true; }
    pp_wait_lgkm();
    if (wm == 1) bar();          // group 1 runs one segment behind

    hs_stamp(g.stamps, 1);
    int slot = 0, fill = D % NSUB;
    int u = 0;
    for (; u < nsub - D; ++u) {
        const int nslot = slot + 1 == NSUB ? 0 : slot + 1;
        issue(u + D, fill, 0);
        issue(u + D, fill, 1);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        hs_mfma_seg<0>(acc, f, read_a, read_b, true, slot, nslot, !(DBG & 16));
        issue(u + D, fill, 2);
        pp_wait_vm_lgkm<(DBG & 1) ? 0 : PP_PW * (D - 2) + 3>();
        __builtin_amdgcn_sched_barrier(0);
        bar();
        hs_mfma_seg<1>(acc, f, read_a, read_b, true, slot, nslot, !(DBG & 16));
        issue(u + D, fill, 3);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        hs_mfma_seg<2>(acc, f, read_a, read_b, true, slot, nslot, !(DBG & 16));
        slot = nslot;
        fill = fill + 1 == NSUB ? 0 : fill + 1;
    }
    for (; u < nsub; ++u) {
        const int r = nsub - 1 - u;      // sub-tiles still to come after this one
        const int nslot = slot + 1 == NSUB ? 0 : slot + 1;
        bar();
        hs_mfma_seg<0>(acc, f, read_a, read_b, true, slot, nslot, !(DBG & 16));
        if (r >= 1) pp_wait_vm_lgkm_rt(PP_PW * (r - 1));
        __builtin_amdgcn_sched_barrier(0);
        bar();
        hs_mfma_seg<1>(acc, f, read_a, read_b, true, slot, nslot, !(DBG & 16));
        bar();
        hs_mfma_seg<2>(acc, f, read_a, read_b, r >= 1, slot, nslot, !(wm == 1 && r == 0) && !(DBG & 16));
        slot = nslot;
    }

    hs_stamp(g.stamps, 2);
    if ((DBG & 8) && g.M > 0) return;
    hs_epilogue<EPI, OUT_HS, (DBG & 64) != 0>(acc, g, lds, m0, n0, wave, lane);   // DBG 64: epilogue without its global stores
    hs_stamp(g.stamps, 3);
}

// ---------------------------------------------------------------------------------------------
// Regressor fused behind the first per-pair layer (two hidden layers, n_out <= 256 - the shipped network): the
// 256 x 256 tile of h2 = relu(z1) a workgroup has just accumulated never leaves the CU.  It becomes the A operand of
// a second product  partial[256][n2] = h2_tile[256][256] . W2[k-slice of this column tile][n2]  on the same matrix
// pipe, and only that partial result (n2 = 234 columns instead of 1024, fp32) goes to memory: the 2.1 GB of h2 a
// config-2 launch used to write - and the regressor kernel used to read back - disappear, with them the hs
// conversion for memory, the regressor's A-side DMA and its own prologue / epilogue.
//   * The first stage runs with SWAPPED MFMA operands, so a lane owns an activation ROW and each register quad four
//     consecutive columns: relu / scale / split of a quad is two ds_write_b64 (hi | lo) straight into an A image
//     [16 sub-tiles][128 rows][64 B] in the (now idle) ring - the sub-tile row format of the main loop, same
//     XOR swizzle, so the fragment reads are the main loop's.
//   * 128 accumulator registers per lane: no room for a second accumulator set beside the first.  The two wave
//     groups therefore take turns: group g turns ITS half of the tile (128 rows) into the A image, zeroes its
//     accumulators and computes its 128 x 256 of the second product (one wave = 32 rows x 8 column tiles, the SIMD's
//     matrix pipe to itself, 12 MFMAs per quad of column tiles with the next quad's fragments requested under them);
//     the other group's waves are its DMA engines for the W2 sub-tiles (2 slots of 16 KiB behind the ring; one
//     workgroup barrier per sub-tile hands a slot back and publishes the next).  160 KiB of LDS in all.
//   * BatchNormalization of the pair layer: scale folded into the rows of W2, shift into the regressor bias, at load.
//   * The column tiles of a row tile hold different k-slices of the second product.  Tiles 0 .. tiles_n-2 store their
//     partial sums in slabs and count themselves in flags[row tile] (release); the LAST column tile waits for the
//     count (acquire; it is dispatched after the others, so they are running or done), adds slab 0, 1, ... and its own
//     part in that fixed order, the bias, and writes the output - deterministic, no atomics on data.
struct PairRegArgs {
    const uint16_t* B2;    // hs [n2][ldb2 halves]: regressor weights, K-major, rows scaled by the pair layer's BN scale
    const float* bias2;    // [n2] regressor bias incl. the folded BN shift
    float* out;            // [M][n2]
    float* slabs;          // [tiles_n - 1][M][n2]
    unsigned* flags;       // [row tiles], zero before the launch
    unsigned* err;         // set to 1 if a wait for the other column tiles timed out (never in a healthy run)
    int ldb2, n2;
    float acc_scale2;      // 2^-(activation shift of h2 + weight shift of B2)
};

constexpr int PR_A_FLOATS = 16 * 128 * PP_ROWF;     // A image of one half tile: 16 sub-tiles x 128 rows x 64 B = 128 KiB
constexpr int PR_B_FLOATS = 256 * PP_ROWF;          // one W2 sub-tile: 256 rows x 64 B = 16 KiB
constexpr int PR_LDS_FLOATS = PR_A_FLOATS + 2 * PR_B_FLOATS;

// acc: the SWAPPED first-stage accumulators on entry, the second-stage result (plain layout: lane = column) on exit
__device__ __forceinline__ void hs_fused_regressor(f32x16 (&acc)[4][2], const GemmHsArgs& g, const PairRegArgs& rg, float* lds, int m0, int n0,
                                                   int tn, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    float* bring = lds + PR_A_FLOATS;
    const float as1 = g.acc_scale * g.out_scale;
    const int fswz = (l31 >> 2) & 3;
    int xo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;

    // W2 sub-tile t of this column tile's k-slice -> slot: 16 one-KiB pieces, 4 per DMA wave
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(rg.B2 + (size_t)2 * n0), 0, 0x7fffffff, 0x00020000);
    int voff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = 16 * (wn * 4 + u) + (lane >> 2);
        const int clog = (lane & 3) ^ ((row >> 2) & 3);
        voff[u] = (min(row, rg.n2 - 1) * rg.ldb2 + clog * 8) * 2;
    }
    auto issue = [&](int t) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(bring + (t & 1) * PR_B_FLOATS + (wn * 4 + u) * 256),
                                                     16, voff[u], t * 64, 0, 0);
    };
    u16x2 pk16 = {0, 0};
    pp_wait_vm_lgkm<0>();
    pp_barrier();                                   // every wave is out of the main loop: the ring is idle

#pragma unroll 1
    for (int gsel = 0; gsel < 2; ++gsel) {
        if (wm == gsel) {
            // ---- this group's half of the h2 tile -> A image.  acc[i][j][4q + e]: row i*32 + l31, column wn*64 + j*32 + 8q + 4hi + e
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nloc = wn * 64 + j * 32 + 8 * q + 4 * hi;
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + n0 + nloc);
                    b4 *= g.out_scale;
                    const int kk = nloc >> 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[i][j][4 * q + e], as1, b4[e]), 0.f);
                        uint2 oh, ol;
                        oh.x = hs_hi_pair(v[0], v[1]);
                        oh.y = hs_hi_pair(v[2], v[3]);
                        ol.x = hs_lo_pair(v[0], v[1], oh.x);
                        ol.y = hs_lo_pair(v[2], v[3], oh.y);
                        pk16 = __builtin_elementwise_max(__builtin_elementwise_max(pk16, __builtin_bit_cast(u16x2, oh.x)), __builtin_bit_cast(u16x2, oh.y));
                        const int row = i * 32 + l31;
                        float* img = lds + kk * (128 * PP_ROWF) + row * PP_ROWF + 2 * hi;
                        *reinterpret_cast<uint2*>(img + (((q & 1) ^ fswz) << 2)) = oh;
                        *reinterpret_cast<uint2*>(img + (((2 + (q & 1)) ^ fswz) << 2)) = ol;
                    }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        } else {
            issue(0);
            issue(1);
        }
        pp_wait_vm_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();                               // A image written, W2 sub-tiles 0 and 1 landed

        if (wm == gsel) {
            // ---- 32 rows (row tile wn of the half) x 8 column tiles; acc[jt >> 1][jt & 1] = column tile jt
            f16x8 a_cur[2], a_nxt[2], b_cur[4][2], b_nxt[4][2];
            const float* abase = lds + (wn * 32 + l31) * PP_ROWF;
            auto read_a2 = [&](int t, f16x8 (&a)[2]) {
#pragma unroll
                for (int c = 0; c < 2; ++c) a[c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(abase + t * (128 * PP_ROWF) + xo[c]));
            };
            auto read_b2 = [&](int t, int quad, f16x8 (&b)[4][2]) {
                const float* st = bring + (t & 1) * PR_B_FLOATS + (quad * 128 + l31) * PP_ROWF;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq)
#pragma unroll
                    for (int c = 0; c < 2; ++c) b[jq][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(st + jq * 32 * PP_ROWF + xo[c]));
            };
            auto mfma_quad = [&](int quad, const f16x8 (&a)[2], const f16x8 (&b)[4][2]) {
                // product-major: four independent accumulators between two MFMAs on the same one
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        f32x16& d = acc[(quad * 4 + jq) >> 1][(quad * 4 + jq) & 1];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[pr == 2 ? 1 : 0], b[jq][pr == 0 ? 1 : 0], d, 0, 0, 0);   // a_hi b_lo, a_hi b_hi, a_lo b_hi
                    }
            };
            read_a2(0, a_cur);
            read_b2(0, 0, b_cur);
            pp_wait_lgkm();
#pragma unroll 1
            for (int t = 0; t < 16; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                read_b2(t, 1, b_nxt);                       // quad 1 of this sub-tile, under the MFMAs of quad 0
                mfma_quad(0, a_cur, b_cur);
                __builtin_amdgcn_s_setprio(0);
                pp_wait_lgkm();                             // the last reads of slot t & 1 are done
                __builtin_amdgcn_sched_barrier(0);
                pp_barrier();                               // X_t: slot t & 1 may be refilled, sub-tile t + 1 is visible
                __builtin_amdgcn_s_setprio(1);
                if (t + 1 < 16) {
                    read_a2(t + 1, a_nxt);
                    read_b2(t + 1, 0, b_cur);               // b_cur is dead once quad 0 has issued (in-order issue)
                }
                mfma_quad(1, a_cur, b_nxt);
                __builtin_amdgcn_s_setprio(0);
                pp_wait_lgkm();
#pragma unroll
                for (int c = 0; c < 2; ++c) a_cur[c] = a_nxt[c];
            }
        } else {
#pragma unroll 1
            for (int t = 0; t < 16; ++t) {
                pp_wait_vm_lgkm<0>();                       // sub-tile t + 1 has landed
                __builtin_amdgcn_sched_barrier(0);
                pp_barrier();                               // X_t
                if (t + 2 < 16) issue(t + 2);
            }
        }
    }
    {
        const uint16_t top = pk16[0] > pk16[1] ? pk16[0] : pk16[1];
        hs_report_peak(g.peak, top >= 0x7c00u ? __builtin_inff() : (float)__builtin_bit_cast(_Float16, top), false);
    }
    hs_stamp(g.stamps, 4);

    // ---- partial sums out: wave = rows m0 + wm*128 + wn*32 .. +31, lane = column jt*32 + l31
    const int tiles_n = g.tiles_n;
    const bool last = tn == tiles_n - 1;
    const int tm = m0 / PP_BM;
    if (last && tiles_n > 1) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(rg.flags + tm, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(tiles_n - 1)) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > 200000000ll) { atomicOr(rg.err, 1u); break; }       // 2 s: never in a healthy run
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    hs_stamp(g.stamps, 5);
    const int wrow = m0 + wm * 128 + wn * 32 + 4 * hi;
    const size_t plane = (size_t)g.M * rg.n2;
    float* dst = last ? rg.out : rg.slabs + (size_t)tn * plane;
    const bool full_rows = m0 + PP_BM <= g.M;
#pragma unroll                                      // (static accumulator indices: a dynamic one would put acc in scratch)
    for (int jt = 0; jt < 8; ++jt) {
        const int col = jt * 32 + l31;
        if (col >= rg.n2) continue;
        const f32x16 d = acc[jt >> 1][jt & 1] * rg.acc_scale2;
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f;
        if (last) {
            // slab 0, slab 1, ... in that order, then this tile's own part, then the bias; all 16 loads of a slab are
            // requested before the first one is used
#pragma unroll 1
            for (int s2 = 0; s2 < tiles_n - 1; ++s2) {
                const float* sp = rg.slabs + (size_t)s2 * plane + col;
                f32x16 v;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wrow + (r & 3) + 8 * (r >> 2);
                    v[r] = (full_rows || row < g.M) ? sp[(size_t)row * rg.n2] : 0.f;
                }
                sum = s2 == 0 ? v : sum + v;
            }
        }
        const float b2 = last ? rg.bias2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow + (r & 3) + 8 * (r >> 2);
            const float v = last ? ((tiles_n > 1 ? sum[r] + d[r] : d[r]) + b2) : d[r];
            if (full_rows || row < g.M) dst[(size_t)row * rg.n2 + col] = v;
        }
    }
    if (!last) {
        pp_wait_vm_lgkm<0>();                       // this wave's slab stores have left
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(rg.flags + tm, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// A operand produced in the kernel (PairSrc as in gemm_bf16.hip.h):
//   pair mode:  A[(pr,t)][k] = split( relu( in_scale * L0[pr][k] + Ts[t][k] ) ),   Ts = in_scale * T (a pre-scaled
//               copy of the pilot table, in_scale a power of two): one fma and one max per value.  Neither half of
//               that layer's BatchNormalization is applied here - the library folds the SCALE into the rows of the
//               next layer's split weights and the SHIFT into its bias at load (csi_load_weights: Layer::Wh of
//               layer 1, Layer::bias_hs), which keeps the exact zeros of the relu in A and takes a vector read,
//               four packed multiplies and their power out of every sub-tile (measured -7 % on the kernel; the
//               matrix pipe is power-bound, so what counts is energy, not issue slots - DESIGN.md 4.6);
//   CAST mode:  A[m][k] = split( in_scale * X[m][k] )  from fp32 rows (layer 0), split-K over z.
// Every wave owns 32 A rows (lane -> row, 8-column half of the 16-column sub-tile).  The conversion of a
// sub-tile is spread over the load segments of TWO phases so that neither outlasts the partner group's
// 8-MFMA segment (256 cycles of the SIMD's matrix pipe; a load segment that is longer stalls both groups
// at the next barrier - round 1 did all of it in P0: ~85 instructions, ~60 of them VALU beside the
// partner's MFMAs, and measured 1.60 ms against 1.44 for the kernel without generation):
//   P0: wait for the 8 (+8) values requested one sub-tile earlier, relu / scale, 4 x v_cvt_pk_f16_f32,
//       ds_write_b128 of the hi chunk
//   P1: 8 x v_fma_mix{lo,hi}_f16 (lo halves), ds_write_b128 of the lo chunk, range-guard maximum, THEN the
//       requests for the next values
//   P2: both B pieces (LDS-DMA, D = 3 sub-tiles ahead)
// Both chunks go to their swizzled places (rows are permuted over the lanes so that each 8-lane
// ds_write_b128 group covers all 32 banks).  All vector-memory operations of a sub-tile are issued behind
// its last ds_write: hipcc inserts s_waitcnt vmcnt(0) in front of any ds_write that follows an LDS-DMA in
// flight (possible alias), which would put the L2 latency of the fresh requests on the critical path.
// One vmcnt(0) per sub-tile (top of P0) retires the values and the B pieces issued behind them four
// segments earlier - including sub-tile u+1, which P2(u) prefetches two barriers later.  RAW: the hi chunk
// of sub-tile u+2 is written in P0(u) and first read in P2(u+1), the lo chunk in P1(u) and first read in
// P1(u+2); every MFMA segment ends with lgkmcnt(0) in front of its closing barrier, which also retires the
// chunk written in front of it.  WAR: B slot of sub-tile u+3 = slot of u-1, last read in P0(u-1).
// DBG (timing probes of tools/hs_probe.hip, results invalid): 1 = no L0 / T / X requests, 2 = no conversion (VALU, ds_write)
// FUSE: the regressor runs behind this layer inside the kernel (hs_fused_regressor above; EPI / OUT_HS are then unused)
// VM (round 2): 0 = the schedule above (builtin LDS-DMA, one vmcnt(0) per sub-tile).  1 / 2 = every vector-memory
// operation as inline asm counted by hand (helpers of gemm_bf16.hip.h): the B pieces then stay in flight across the
// conversion's ds_writes and the A-side values are awaited with a counted vmcnt that names their registers; 2 also requests
// the A-side values one sub-tile earlier (two register sets), which is what the fp32 rows of layer 0 need - they come
// from HBM, and two segments of look-ahead are shorter than that latency.
template <int EPI, bool OUT_HS, bool CAST = false, int DBG = 0, bool FUSE = false, int VM = 0>
__global__ __launch_bounds__(PP_THREADS, 1) void gemm_hs_pp_pair_kernel(const GemmHsArgs g, const PairSrc ps, const float in_scale, const PairRegArgs rg) {
    // VM 3: VM 2 with ONE load segment and ONE 24-MFMA segment per sub-tile instead of three of each (two barriers per
    // sub-tile and wave instead of six: an 8-MFMA segment is 256 cycles of the matrix pipe, and the skew of a barrier is
    // a fair fraction of that).  The load segment then overlaps the partner group's whole previous MFMA segment, so the
    // B pieces go into a fifth ring slot (the slot of sub-tile u - 2, which the partner has left a segment ago).
    constexpr bool MERGE = VM == 3;
    constexpr int NSUB = MERGE ? 5 : 4, D = 3;
    constexpr int LA = VM >= 2 ? 4 : 3;            // sub-tiles between the request of A-side values and the sub-tile they belong to
    constexpr bool TWOSETS = VM >= 2;
    constexpr int NAL = CAST ? 2 : 4;               // A-side loads per sub-tile and wave
    extern __shared__ __attribute__((aligned(16))) float lds[];      // NSUB * PP_SUBF ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    int tm, tn;
    hs_tile_of_block(g, tm, tn);
    if (tm * PP_BM >= g.M) return;
    hs_stamp(g.stamps, 0);
    const int m0 = tm * PP_BM, n0 = tn * PP_BN;
    const int kbeg = CAST ? blockIdx.z * g.k_per_split : 0;
    const int kend = CAST ? min(g.K, kbeg + g.k_per_split) : g.K;
    const int nsub = (kend - kbeg + HS_G - 1) / HS_G;

    // automatic input scale (layer 0): the sampled maximum lands in [2^13, 2^14) - a factor 4 of head room
    // for samples the estimate did not see, every value down to 2^-17 of the maximum with a normal lo half
    float a_scale = in_scale, acc_scale = g.acc_scale;
    if (CAST && g.dyn_max) {
        const unsigned bits = __builtin_amdgcn_readfirstlane(*g.dyn_max);
        const int e = (int)((bits >> 23) & 0xffu) - 126;                       // frexp exponent of the maximum (0 -> -126)
        const int n = bits == 0 ? 0 : max(-100, min(100, 14 - e));
        a_scale = __builtin_bit_cast(float, (unsigned)(n + 127) << 23);
        acc_scale = __builtin_bit_cast(float, (unsigned)(max(-126, min(126, -n - g.wshift)) + 127) << 23);
    }

    // ---- B side: pieces 2w, 2w+1 of the 16 B pieces of a sub-tile
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.Bt + (size_t)n0 * g.ldb + 2 * kbeg), 0, 0x7fffffff, 0x00020000);
    int voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = 16 * (2 * wave + u) + (lane >> 2);
        const int clog = (lane & 3) ^ ((row >> 2) & 3);
        voff[u] = (min(row, g.N - 1 - n0) * g.ldb + clog * 8) * 2;
    }
    const uint16_t* bsrc[2];                         // VM: this lane's 16 bytes of B piece u, sub-tile 0
#pragma unroll
    for (int u = 0; u < 2; ++u) bsrc[u] = g.Bt + (size_t)n0 * g.ldb + 2 * kbeg + voff[u] / 2;
    const uint32_t lds_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds);
    auto issue_b = [&](int sub, int slot, int u) {
        if (VM)
            pp_gdma16(bsrc[u] + sub * 32, lds_off + (uint32_t)(slot * PP_SUBF + (PP_BM / 16 + 2 * wave + u) * 256) * 4u);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + slot * PP_SUBF + (PP_BM / 16 + 2 * wave + u) * 256),
                                                     16, voff[u], sub * 64, 0, 0);
    };
    // ---- A side: row 32w + perm(lane >> 1), k half lane & 1.  ds_write_b128 is serviced in groups of 8
    // CONTIGUOUS lanes against 32 banks ((a / 4) mod 32, MI355X_MICROARCH.md "LDS"): the 8 lanes of a group
    // must hit 8 different (row & 1, 16-byte chunk) pairs.  The chunk is (k half or 2 + k half) ^ ((row >> 2) & 3),
    // so a group's four rows have to differ in bit 0 and in bit 3: lane bit 1 -> row bit 0, lane bit 2 ->
    // row bit 3, lane bits 3, 4, 5 -> row bits 1, 2, 4.  (Round 1 mapped lane bit 2 to row bit 1: the four
    // rows of a group shared the swizzle and every store was a 2-way conflict, SQ_LDS_BANK_CONFLICT 7.4e7
    // per launch at config 2.)
    const int ridx = lane >> 1, kh = lane & 1;
    const int arow = 32 * wave + ((ridx & 1) | ((ridx & 2) << 2) | ((ridx & 4) >> 1) | ((ridx & 8) >> 1) | (ridx & 16));
    const int aswz = (arow >> 2) & 3;
    const int a_hi_off = arow * PP_ROWF + ((kh ^ aswz) << 2);            // floats inside a sub-tile
    const int a_lo_off = arow * PP_ROWF + (((2 + kh) ^ aswz) << 2);
    const float* lrow;
    const float* trow;
    {
        const int m = min(m0 + arow, g.M - 1);
        if (CAST) {
            lrow = ps.L0 + (size_t)m * ps.ldl + kbeg + 8 * kh;
            trow = lrow;
        } else {
            const int pr = m / ps.nt, t = m - pr * ps.nt;
            lrow = ps.L0 + (size_t)pr * ps.ldl + 8 * kh;
            trow = ps.T + (size_t)t * ps.ldl + 8 * kh;
        }
    }
    f32x4 lvs[2][2] = {}, tvs[2][2] = {};            // two register sets (sub-tile parity); VM 0 / 1 use set 0 only
    float apk = 0.f;
    u16x2 apk16 = {0, 0};              // pair mode: running maximum of the hi halves (bit patterns)                    // largest |scaled A operand| this lane converted
    auto load_a = [&](int sub, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        if (DBG & 1) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (VM) {
                lvs[SET][h] = pp_gload16(lrow + sub * HS_G + 4 * h);
                if (!CAST) tvs[SET][h] = pp_gload16(trow + sub * HS_G + 4 * h);
            } else {
                lvs[SET][h] = *reinterpret_cast<const f32x4*>(lrow + sub * HS_G + 4 * h);
                if (!CAST) tvs[SET][h] = *reinterpret_cast<const f32x4*>(trow + sub * HS_G + 4 * h);
            }
        }
    };
    // VM: at most N younger operations stay in flight; releases the values of register set SET
    auto wait_a = [&](auto set_tag, auto n_tag) {
        constexpr int SET = decltype(set_tag)::value, N = decltype(n_tag)::value;
        if (CAST) pp_wait_vm_dep<N>(lvs[SET][0], lvs[SET][1]);
        else pp_wait_vm_dep<N>(lvs[SET][0], lvs[SET][1], tvs[SET][0], tvs[SET][1]);
    };
    f32x4 gv[2];                        // the 8 scaled A values of the sub-tile in conversion (P0 -> P1)
    uint4 gh;                           // ... and their packed hi halves
    auto gen_hi = [&](int sub, int slot, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        if (DBG & 2) return;
        if (CAST) {
            gv[0] = lvs[SET][0] * a_scale;
            gv[1] = lvs[SET][1] * a_scale;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[h][e] = fmaxf(fmaf(lvs[SET][h][e], in_scale, tvs[SET][h][e]), 0.f);
        }
        gh.x = hs_hi_pair(gv[0][0], gv[0][1]);
        gh.y = hs_hi_pair(gv[0][2], gv[0][3]);
        gh.z = hs_hi_pair(gv[1][0], gv[1][1]);
        gh.w = hs_hi_pair(gv[1][2], gv[1][3]);
        *reinterpret_cast<uint4*>(lds + slot * PP_SUBF + a_hi_off) = gh;
    };
    auto gen_lo = [&](int slot) {
        if (DBG & 2) return;
        uint4 ol;
        ol.x = hs_lo_pair(gv[0][0], gv[0][1], gh.x);
        ol.y = hs_lo_pair(gv[0][2], gv[0][3], gh.y);
        ol.z = hs_lo_pair(gv[1][0], gv[1][1], gh.z);
        ol.w = hs_lo_pair(gv[1][2], gv[1][3], gh.w);
        *reinterpret_cast<uint4*>(lds + slot * PP_SUBF + a_lo_off) = ol;
        if (CAST) {
            apk = hs_absmax(hs_absmax(hs_absmax(hs_absmax(apk, gv[0][0], gv[0][1]), gv[0][2], gv[0][3]), gv[1][0], gv[1][1]), gv[1][2], gv[1][3]);
        } else {
            // relu output: no sign, so the f16 bit patterns of the hi halves order like unsigned integers (inf on top) -
            // four packed 16-bit maxima instead of six fp32 ones
            apk16 = __builtin_elementwise_max(__builtin_elementwise_max(apk16, __builtin_bit_cast(u16x2, gh.x)), __builtin_bit_cast(u16x2, gh.y));
            apk16 = __builtin_elementwise_max(__builtin_elementwise_max(apk16, __builtin_bit_cast(u16x2, gh.z)), __builtin_bit_cast(u16x2, gh.w));
        }
    };

    const int fswz = (l31 >> 2) & 3;
    int xo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) xo[c] = ((2 * c + hi) ^ fswz) << 2;
    const int abase = (wm * 128 + l31) * PP_ROWF;
    const int bbase = (PP_BM + wn * 64 + l31) * PP_ROWF;
    auto read_a = [&](int slot, int c, f16x8 (&a)[4]) {
        const float* st = lds + slot * PP_SUBF + abase + xo[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(st + i * 32 * PP_ROWF));
    };
    auto read_b = [&](int slot, int c, f16x8 (&b)[2]) {
        const float* st = lds + slot * PP_SUBF + bbase + xo[c];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(st + j * 32 * PP_ROWF));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    HsFrags f;

    // ---- prologue: B sub-tiles 0..2 in flight, A sub-tiles 0 and 1 generated synchronously
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using N0 = std::integral_constant<int, 0>;
    const int npro = min(nsub, D);
    for (int t = 0; t < npro; ++t) { issue_b(t, t, 0); issue_b(t, t, 1); }
    for (int t = 0; t < min(nsub, 2); ++t) {
        load_a(t, S0{});
        if (VM) wait_a(S0{}, N0{});     // VM 0: the compiler waits for the loads it just issued
        gen_hi(t, t, S0{});
        gen_lo(t);
    }
    if (VM) pp_wait_vm0();
    pp_wait_vm_lgkm<0>();
    if (nsub > 2) load_a(2, S0{});      // consumed in P0 of sub-tile 0
    if (TWOSETS && nsub > 3) load_a(3, S1{});
    pp_barrier();
    read_a(0, 0, f.a_hi);
    read_b(0, 1, f.b_lo);
    pp_wait_lgkm();
    if (wm == 1) pp_barrier();          // group 1 runs one segment behind

    hs_stamp(g.stamps, 1);
    int slot = 0;
    // PAR = parity of u + 2, the sub-tile whose A image this sub-tile generates (VM 2: its register set)
    auto subtile = [&](int u, auto steady_tag, auto par_tag) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr int GS = TWOSETS ? decltype(par_tag)::value : 0;                 // set holding the values of sub-tile u + 2
        constexpr int LS = TWOSETS ? (decltype(par_tag)::value ^ (LA & 1)) : 0;    // set receiving the values of sub-tile u + LA
        using GST = std::integral_constant<int, GS>;
        using LST = std::integral_constant<int, LS>;
        const int nslot = (slot + 1) % NSUB;
        const bool has_a = STEADY || u + 2 < nsub, nxt_a = STEADY || u + LA < nsub, has_b = STEADY || u + D < nsub;
        const bool last = !STEADY && u == nsub - 1;
        if (VM == 0) {
            // P0: every memory operation of this wave has landed (the values requested in P1 of the previous
            // sub-tile, and behind them its two B pieces); no new one is issued before the lo chunk is written -
            // hipcc puts s_waitcnt vmcnt(0) in front of a ds_write that follows an LDS-DMA still in flight
            pp_wait_vm_lgkm<0>();
        } else if (STEADY) {
            // the values of sub-tile u + 2 and the B pieces of sub-tile u + 1 (first read in P2 below) have landed; younger
            // and still in flight: the two B pieces of P2(u - 1), and with VM 2 the values and pieces behind them
            // MERGE issues the B pieces in front of the value requests and leaves only the youngest requests in flight: a piece
            // must have landed a whole segment before the partner group reads it (the groups are one segment apart and
            // synchronise only at segment boundaries)
            wait_a(GST{}, std::integral_constant<int, (MERGE ? NAL : (TWOSETS ? NAL + 2 : 2))>{});
            pp_wait_lgkm();
        } else {
            if (has_a) wait_a(GST{}, N0{}); else pp_wait_vm0();
            pp_wait_lgkm();
        }
        if (has_a) gen_hi(u + 2, (slot + 2) % NSUB, GST{});     // its ds_write retires with the lgkmcnt(0) that closes the MFMA segment
        if (MERGE) {
            if (has_a) gen_lo((slot + 2) % NSUB);
            if (has_b) {
                issue_b(u + D, (slot + D) % NSUB, 0);
                issue_b(u + D, (slot + D) % NSUB, 1);
            }
            if (nxt_a) load_a(u + LA, LST{});
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            hs_mfma_seg<0, FUSE>(acc, f, read_a, read_b, true, slot, nslot, false);
            hs_mfma_seg<1, FUSE>(acc, f, read_a, read_b, true, slot, nslot, false);
            hs_mfma_seg<2, FUSE>(acc, f, read_a, read_b, !last, slot, nslot, !(wm == 1 && last));
            slot = nslot;
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        hs_mfma_seg<0, FUSE>(acc, f, read_a, read_b, true, slot, nslot, true);
        // P1
        if (has_a) gen_lo((slot + 2) % NSUB);
        if (nxt_a) load_a(u + LA, LST{});
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        hs_mfma_seg<1, FUSE>(acc, f, read_a, read_b, true, slot, nslot, true);
        // P2
        if (has_b) {
            issue_b(u + D, (slot + D) % NSUB, 0);
            issue_b(u + D, (slot + D) % NSUB, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        hs_mfma_seg<2, FUSE>(acc, f, read_a, read_b, !last, slot, nslot, !(wm == 1 && last));
        slot = nslot;
    };
    int u = 0;
    if (VM) {
        // the counted waits assume the issue pattern of the steady state behind them: the first two sub-tiles drain instead
        if (nsub > 0) subtile(0, std::false_type{}, S0{});
        if (nsub > 1) subtile(1, std::false_type{}, S1{});
        u = min(2, nsub);
    }
    if (TWOSETS) {
        // two sub-tiles per trip: the register set of a sub-tile's values is its parity, a compile-time constant here
        for (; u + LA + 1 < nsub; u += 2) {
            subtile(u, std::true_type{}, S0{});
            subtile(u + 1, std::true_type{}, S1{});
        }
        for (; u < nsub; u += 2) {
            subtile(u, std::false_type{}, S0{});
            if (u + 1 < nsub) subtile(u + 1, std::false_type{}, S1{});
        }
    } else {
        for (; u + LA < nsub; ++u) subtile(u, std::true_type{}, S0{});
        for (; u < nsub; ++u) subtile(u, std::false_type{}, S0{});
    }
    if (VM) pp_wait_vm0();

    hs_stamp(g.stamps, 2);
    if (!CAST) {
        const uint16_t top = apk16[0] > apk16[1] ? apk16[0] : apk16[1];
        apk = top >= 0x7c00u ? __builtin_inff() : (float)__builtin_bit_cast(_Float16, top);
    }
    hs_report_peak(g.peak, apk, true);
    GemmHsArgs ge = g;
    ge.acc_scale = acc_scale;
    if constexpr (FUSE) hs_fused_regressor(acc, ge, rg, lds, m0, n0, tn, wave, lane);
    else hs_epilogue<EPI, OUT_HS>(acc, ge, lds, m0, n0, wave, lane);
    hs_stamp(g.stamps, 3);
}

// out = atomicMax(bits of |x|) over a sample of x: one 1-KiB block (64 float4, read coalesced by a
// wave) out of every `step` blocks - everything when the array is small: the magnitude estimate behind
// the automatic input scale.  out is zeroed on the stream before the launch; positive floats order like
// their bit patterns.  inf / nan are skipped (they would poison the scale; the data itself then trips
// the range guard or propagates as nan).
__global__ __launch_bounds__(256) void hs_absmax_sample_kernel(const float* __restrict__ x, size_t n4, size_t step, unsigned* __restrict__ out) {
    __shared__ unsigned smax;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    float m = 0.f;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t blk = wave * step; blk * 64 < n4; blk += nwaves * step) {
        const size_t i = blk * 64 + lane;
        if (i >= n4) continue;
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = __builtin_fabsf(v[e]);
            if (a < 3.0e38f) m = __builtin_fmaxf(m, a);
        }
    }
    atomicMax(&smax, __builtin_bit_cast(unsigned, m));
    __syncthreads();
    if (threadIdx.x == 0 && smax) atomicMax(out, smax);
}

// dst (hs [rows][ldh]) = split(scale * src[rows][cols]) with zero padding up to ldh / 2 columns.
// One thread = 8 columns of a row: 32 B in, 16 B of hi + 16 B of lo out.
__global__ void f32_to_hs_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, uint16_t* __restrict__ dst, int ldh, float scale,
                                 int blocked = 0) {
    const int c8 = ldh >> 4;                                  // 8-column groups per row
    const size_t total = (size_t)rows * c8;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool vec = (ld_src & 3) == 0 && (((uintptr_t)src) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int r = (int)(i / c8), c = (int)(i - (size_t)r * c8) * 8;
        float v[8];
        const float* s = src + (size_t)r * ld_src + c;
        if (vec && c + 8 <= cols) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c + e < cols) ? s[e] : 0.f;
        }
        uint4 oh, ol;
        hs_split2(v[0] * scale, v[1] * scale, oh.x, ol.x);
        hs_split2(v[2] * scale, v[3] * scale, oh.y, ol.y);
        hs_split2(v[4] * scale, v[5] * scale, oh.z, ol.z);
        hs_split2(v[6] * scale, v[7] * scale, oh.w, ol.w);
        uint16_t* d = dst + (blocked ? hs_blk_offset(r, c >> 4, ldh) : (size_t)r * ldh + (c >> 4) * 32) + (c & 8);
        *reinterpret_cast<uint4*>(d) = oh;
        *reinterpret_cast<uint4*>(d + 16) = ol;
    }
}

}  // namespace csi
