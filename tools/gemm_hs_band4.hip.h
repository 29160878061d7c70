// gemm_hs_band.hip.h - the two per-pair products of the shipped network (massiveMIMO_CSI_prediction_DNN.py:211-227:
// Dense(relu) -> BatchNormalization -> Dense(linear, 234)) as ONE kernel on the split-f16 matrix path of gemm_hs.hip.h,
// without h2 ever leaving the registers and without partial sums being exchanged between workgroups.
//
// Work split.  One workgroup = 4 waves (one per SIMD, up to 512 registers each) owns a BAND of 128 pair rows; wave w owns
// rows 32 w .. 32 w + 31 of it for BOTH products.  Every MFMA runs with SWAPPED operands - the instruction's A operand is
// a weight fragment (32 output features x 16 k), its B operand the activation fragment (16 k x 32 rows) - so the C/D
// layout has lane = activation ROW, registers = output features ((r & 3) + 8 (r >> 2) + 4 (lane >> 5) inside a tile of
// 32).  That is also the layout of an activation fragment: lane (row, k half) holds 8 k-values.  Consequences:
//   * stage 1 (first per-pair layer): a lane GENERATES its own operand - split(relu(s * L0[pair row][k] + Ts[t][k])) for its
//     row and its 8 k-columns - in registers; the A operand never touches the LDS (the 8-wave kernel writes an A image
//     with ds_write_b128 and reads it back four times);
//   * stage 2 (regressor): the 8 accumulator registers a lane holds for 16 consecutive features of h2 ARE, after bias /
//     relu / split, the B operand of the second product for those 16 k-values - in the order {0-3, 8-11 | 4-7, 12-15} of
//     the C/D layout.  The k order of an MFMA is free as long as both operands agree, so the regressor weights are
//     stored with that permutation inside every group of 16 k (hs_band_kperm, applied once at load) and h2 goes from
//     the accumulators of stage 1 into the MFMAs of stage 2 through ~30 VALU operations per 24 MFMAs and nothing else:
//     no LDS image, no transposition, no HBM round trip (the separate kernels write and re-read 2 x 2.1 GB per launch
//     at config 2), no hi / lo store epilogue.
//   * the regressor accumulators (32 rows x 256 columns per wave = 128 registers) live across the column steps of the
//     band, beside the 128 accumulators of the running column step: no partial sums leave the CU.
// The LDS only carries WEIGHTS.  The band walks N1 / 256 column steps; each is nsub1 = K1 / 16 stage-1 sub-steps followed
// by 16 stage-2 sub-steps, and every sub-step consumes ONE 16 KiB weight sub-tile [256 rows][16 k as hi | lo] (the
// sub-tile image of gemm_hs.hip.h: 64-byte rows, XOR chunk swizzle) and issues 24 MFMAs per wave:
//     P0: w_lo x a_hi    P1: w_hi x a_hi    P2: w_hi x a_lo       (8 feature tiles each)
// so the weight stream of a band is one uniform sequence of sub-tiles through a ring of 4 slots, 4 sub-steps ahead, and
// the pipeline never drains between the two products or between column steps.
//
// Sub-step s (slot s & 3), one wave, hand-placed between its own MFMAs (no partner wave to hide behind):
//     top      s_waitcnt vmcnt(4): the A-side values of sub-step s + 1 and (older) this wave's pieces of sub-tiles s + 1, s + 2
//     P0       8 MFMA | ds_read w_hi(s) x 8 | request A-side values of sub-step s + 2 (4 x 16 B per lane) | convert
//     P1       2 MFMA | lgkmcnt(0), s_barrier | 6 MFMA | 4 LDS-DMA pieces of sub-tile s + 4 -> slot s & 3 | convert
//     P2       8 MFMA | ds_read w_lo(s + 1) x 8 | convert (lo halves, range guard)
// The one barrier per sub-step sits behind the wave's last read of sub-tile s (its w_hi fragments are in registers), so
// it both frees slot s & 3 for the refill issued right behind it and publishes every wave's pieces of sub-tile s + 1
// (awaited at the top) for the w_lo(s + 1) reads of P2.  Every sub-step issues exactly 4 loads and 4 pieces - stage-2
// sub-steps request values they do not need - so every wait is a compile-time count on the in-order vmcnt.
// All vector-memory operations are inline asm (see gemm_bf16.hip.h: hipcc cannot count across LDS-DMA it does not see, and
// serialises what it does see).
//
// (round 4: moved out of the product build - only tools/band_probe.hip launches this form; the library runs the generated
// assembly kernel of csrc/band_kernel_gen.py.  Argument records and converters: csrc/gemm_hs_band.hip.h.)
#pragma once
#include "../dl-channel-estimation-mamimo_amd/csrc/gemm_hs_band.hip.h"

namespace csi {

template <int I, int N, typename F>
__device__ __forceinline__ void band_sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        band_sfor<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x4 band_gload16(uint32_t voff, const void* sbase) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase));
    return v;
}
__device__ __forceinline__ f32x4 band_gload16_o16(uint32_t voff, const void* sbase) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(v) : "v"(voff), "s"(sbase));
    return v;
}
__device__ __forceinline__ void band_gdma16(uint32_t voff, const void* sbase, uint32_t lds_byte_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory");
}
template <int N>
__device__ __forceinline__ void band_wait_vm_dep(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

__device__ __forceinline__ void band_pkmax(uint32_t& pk, uint32_t v) { asm("v_pk_max_u16 %0, %0, %1" : "+v"(pk) : "v"(v)); }

// The MFMA as inline asm with the accumulator PINNED to the AGPR half of the register file ("a").  With the builtin
// hipcc selects the AGPR-only form of the instruction for a kernel of more than 256 registers, the two accumulator sets
// then fill the 256 AGPRs exactly, and the allocator - splitting live ranges around the loops - copies whole sets
// between register ranges and spills one of them around the stage-1 loop (1300 v_accvgpr moves, 400 B of scratch).
// Hazards the compiler no longer sees: a dependent MFMA on the same accumulator is 8 MFMAs away (in-order issue, more
// than the 18 wait states of the longest XDL write -> SrcC rule); VALU reads of accumulators (stage-2 conversion, output)
// come at least 7 MFMAs behind their last write except where band_mfma_settle() is called.
template <bool ARCH>
__device__ __forceinline__ void band_mfma(f32x16& d, const f16x8& w, const f16x8& a) {
    if constexpr (ARCH) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(w), "v"(a));
    else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(w), "v"(a));
}
constexpr int BAND_ARCH_TILES = 2;     // stage-1 accumulator tiles kept in the VGPR half: the two sets would otherwise fill all 256 AGPRs
__device__ __forceinline__ void band_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }

#define BAND_FENCE() __builtin_amdgcn_sched_barrier(0)

// DBG (timing probes, results invalid): 1 = no A-side requests, 2 = no conversion, 4 = no LDS-DMA, 8 = no barriers,
// 16 = no fragment reads, 32 = no output stores
template <int DBG = 0>
__global__ __launch_bounds__(BAND_THREADS, 1) void gemm_hs_band_kernel(const BandArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // ring | bias1 * out_scale [N1] | bias2 [256]
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * BAND_ROWS;
    if (m0 >= g.M) return;
    hs_stamp(g.stamps, 0);

    float* bias1s = reinterpret_cast<float*>(ldsb + BAND_NSLOT * BAND_SLOT_BYTES);
    float* bias2s = bias1s + g.N1;
    for (int i = tid; i < g.N1; i += BAND_THREADS) bias1s[i] = g.bias1[i] * g.out_scale;
    if (tid < 256) bias2s[tid] = tid < g.n2 ? g.bias2[tid] : 0.f;

    const int nsub1 = g.K1 >> 4;
    const int ncol = g.N1 >> 8;

    // ---- weight stream: piece p of a sub-tile = image rows 16 (4 wave + p) .. + 15, lane -> (row, 16-byte chunk)
    // ((row >> 2) & 3 = (lane >> 4) & 3 for every piece: the pieces of a wave differ by 16 rows)
    uint32_t voff1, voff2;
    {
        const int row = 64 * wave + (lane >> 2);
        const int clog = (lane & 3) ^ ((row >> 2) & 3);
        voff1 = (uint32_t)(row * g.ldb1 + clog * 8) * 2u;
        voff2 = (uint32_t)(row * g.ldb2 + clog * 8) * 2u;
    }
    const uint32_t pstep1 = (uint32_t)g.ldb1 * 32u, pstep2 = (uint32_t)g.ldb2 * 32u;       // bytes from one piece to the next (16 rows)
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds);
    const uint32_t dma_dst = lds_base + (uint32_t)wave * 4096u;          // + slot * 16384 + p * 1024
    // sub-tile i (position inside a column step) of column step c: stage 1 for i < nsub1, else stage 2.  Plain values
    // and arithmetic only: a ?: on captured variables makes hipcc keep them in scratch and select their ADDRESSES.
    struct Piece { const void* base; uint32_t voff, pstep; };
    const uint32_t dvoff = voff1 - voff2, dpstep = pstep1 - pstep2;
    const uint16_t* const w1p = g.W1;
    const uint16_t* const w2p = g.W2p;
    const size_t colstep1 = (size_t)256 * g.ldb1;
    auto piece_of = [=](int c, int i) {
        c = c >= ncol ? 0 : c;                      // past the end of the band: harmless re-fetch of its head
        const uint32_t m1 = i < nsub1 ? 0xffffffffu : 0u;
        Piece pc;
        pc.voff = voff2 + (dvoff & m1);
        pc.pstep = pstep2 + (dpstep & m1);
        const uintptr_t a1 = (uintptr_t)(w1p + (size_t)c * colstep1 + (size_t)i * 32);
        const uintptr_t a2 = (uintptr_t)(w2p + ((size_t)c * 16 + (size_t)(i - nsub1)) * 32);
        const uintptr_t mm = (uintptr_t)0 - (uintptr_t)(m1 & 1u);
        pc.base = (const void*)((a1 & mm) | (a2 & ~mm));
        return pc;
    };
    auto issue_piece = [=](const Piece pc, int slot, int p) {
        if (DBG & 4) return;
        band_gdma16(pc.voff + (uint32_t)p * pc.pstep, pc.base, dma_dst + (uint32_t)slot * BAND_SLOT_BYTES + (uint32_t)p * 1024u);
    };

    // ---- A side: this lane's row and k half
    uint32_t loff, toff;
    {
        const int m = min(m0 + 32 * wave + l31, g.M - 1);
        const int pr = m / g.nt, t = m - pr * g.nt;
        loff = (uint32_t)(((size_t)pr * g.ldl + 8 * hi) * 4);
        toff = (uint32_t)(((size_t)t * g.ldl + 8 * hi) * 4);
    }
    f32x4 lv[2][2] = {}, tv[2][2] = {};            // two register sets (sub-step parity): A-side values of sub-steps s + 1, s + 2
    auto request_a = [&](int kg, auto set_tag, int which) {
        constexpr int SET = decltype(set_tag)::value;
        if (DBG & 1) return;
        const float* lb = g.L0 + (size_t)kg * 16;
        const float* tb = g.Ts + (size_t)kg * 16;
        if (which == 0) lv[SET][0] = band_gload16(loff, lb);
        if (which == 1) lv[SET][1] = band_gload16_o16(loff, lb);
        if (which == 2) tv[SET][0] = band_gload16(toff, tb);
        if (which == 3) tv[SET][1] = band_gload16_o16(toff, tb);
    };

    // ---- weight fragments: tile j (32 features), plane c (0 = hi, 1 = lo)
    const int fswz = (l31 >> 2) & 3;
    const uint32_t rd_hi = (uint32_t)(l31 * 64 + (((0 + hi) ^ fswz) << 4));
    const uint32_t rd_lo = (uint32_t)(l31 * 64 + (((2 + hi) ^ fswz) << 4));
    f16x8 w_hi[8], w_lo[8];
    auto read_w = [&](auto slot_tag, auto j_tag, auto plane_tag) {
        constexpr int SLOT = decltype(slot_tag)::value, J = decltype(j_tag)::value, PL = decltype(plane_tag)::value;
        if (DBG & 16) return;
        const char* p = ldsb + (PL ? rd_lo : rd_hi) + SLOT * BAND_SLOT_BYTES + J * 2048;
        const f16x8 v = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(p));
        if (PL) w_lo[J] = v; else w_hi[J] = v;
    };

    f32x16 acc1[8], acc2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[j][e] = 0.f; acc2[j][e] = 0.f; }

    // activation fragments: [parity of the sub-step]
    uint4 a_hi[2] = {}, a_lo[2] = {};
    float gv[8];
    uint4 gh = {}, gl = {};
    uint32_t pk1 = 0, pk2 = 0;                     // range guard: running maxima of the hi halves (bit patterns; relu output has no sign)
    const float as1os = g.acc_scale1 * g.out_scale;
    const float in_scale = g.in_scale;
    f32x4 bq[2];                                   // stage 2: the 8 bias values of the fragment in conversion

    // conversion, in pieces that the sub-step places between its MFMAs.  KIND 1: from A-side values of register set SET;
    // KIND 2: from acc1[J][8 G ..] and bq
    auto conv_f = [&](auto kind_tag, auto set_tag, auto j_tag, auto g_tag, int e) {
        constexpr int KIND = decltype(kind_tag)::value, SET = decltype(set_tag)::value, J = decltype(j_tag)::value, G = decltype(g_tag)::value;
        if (DBG & 2) return;
        if (KIND == 1) gv[e] = fmaf(lv[SET][e >> 2][e & 3], in_scale, tv[SET][e >> 2][e & 3]);
        else gv[e] = fmaf(acc1[J][8 * G + e], as1os, bq[e >> 2][e & 3]);
    };
    auto conv_x = [&](int e) { if (!(DBG & 2)) gv[e] = fmaxf(gv[e], 0.f); };
    auto conv_c = [&](int p) {
        if (DBG & 2) return;
        const uint32_t h = hs_hi_pair(gv[2 * p], gv[2 * p + 1]);
        if (p == 0) gh.x = h; else if (p == 1) gh.y = h; else if (p == 2) gh.z = h; else gh.w = h;
    };
    auto conv_l = [&](int p) {
        if (DBG & 2) return;
        const uint32_t h = p == 0 ? gh.x : (p == 1 ? gh.y : (p == 2 ? gh.z : gh.w));
        const uint32_t l = hs_lo_pair(gv[2 * p], gv[2 * p + 1], h);
        if (p == 0) gl.x = l; else if (p == 1) gl.y = l; else if (p == 2) gl.z = l; else gl.w = l;
    };
    auto conv_g = [&](auto kind_tag, int half) {
        constexpr int KIND = decltype(kind_tag)::value;
        if (DBG & 2) return;
        // (asm: a builtin max chain is reassociated by the optimiser into one tree at the end of the unrolled block,
        // with every gh of the block kept alive until then)
        if constexpr (KIND == 1) {
            if (half == 0) { band_pkmax(pk1, gh.x); band_pkmax(pk1, gh.y); } else { band_pkmax(pk1, gh.z); band_pkmax(pk1, gh.w); }
        } else {
            if (half == 0) { band_pkmax(pk2, gh.x); band_pkmax(pk2, gh.y); } else { band_pkmax(pk2, gh.z); band_pkmax(pk2, gh.w); }
        }
    };

    // One sub-step.  SLOT = s & 3 (compile time), PAR = parity of s (fragment buffers), KIND = 1 / 2 (which product this
    // sub-step's MFMAs belong to), J2 / G2 = tile and half of stage 2 (KIND 2), NK = kind of the fragment to produce for
    // sub-step s + 1 (0 none, 1 from A-side values of set (SLOT + 1) & 3, 2 from acc1[NJ][8 NG ..]), ZJ = stage-1
    // accumulator tile to clear (-1 none).  pc4 = sub-tile s + 4 of the stream, kg2 = k-group of the values to request (those of sub-step s + 2).
    auto substep = [&](auto slot_tag, auto par_tag, auto kind_tag, auto nk_tag, auto nj_tag, auto ng_tag, auto zj_tag, const Piece pc4, int kg2) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_tag)::value, PAR = decltype(par_tag)::value, KIND = decltype(kind_tag)::value;
        constexpr int NK = decltype(nk_tag)::value, NJ = decltype(nj_tag)::value, NG = decltype(ng_tag)::value, ZJ = decltype(zj_tag)::value;
        constexpr int NSLOT = (SLOT + 1) & 3;
        using CSET = std::integral_constant<int, PAR ^ 1>;        // values of sub-step s + 1
        using LSET = std::integral_constant<int, PAR>;            // receives the values of sub-step s + 2
        using NKT = std::integral_constant<int, NK>;
        using NJT = std::integral_constant<int, NJ>;
        using NGT = std::integral_constant<int, NG>;
        BAND_FENCE();
        if (!(DBG & 1)) band_wait_vm_dep<4>(lv[PAR ^ 1][0], lv[PAR ^ 1][1], tv[PAR ^ 1][0], tv[PAR ^ 1][1]);
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BAND_FENCE();
        const f16x8 ah = __builtin_bit_cast(f16x8, a_hi[PAR]), al = __builtin_bit_cast(f16x8, a_lo[PAR]);
        band_sfor<0, 24>([&](auto it) {
            constexpr int I = decltype(it)::value;
            constexpr int PH = I >> 3, J = I & 7;
            f32x16& d = KIND == 1 ? acc1[J] : acc2[J];
            const f16x8 w = PH == 0 ? w_lo[J] : w_hi[J];
            const f16x8 a = PH == 2 ? al : ah;
            band_mfma<(KIND == 1 && J < BAND_ARCH_TILES)>(d, w, a);
            // ---- what rides behind this MFMA
            if constexpr (I < 4) {                                 // w_hi of this sub-tile (slot free of DMA: landed before the last barrier)
                read_w(slot_tag, std::integral_constant<int, 2 * I>{}, std::integral_constant<int, 0>{});
                read_w(slot_tag, std::integral_constant<int, 2 * I + 1>{}, std::integral_constant<int, 0>{});
                request_a(kg2, LSET{}, I);
            }
            if constexpr (NK != 0 && I >= 2 && I < 10) conv_f(NKT{}, CSET{}, NJT{}, NGT{}, I - 2);
            if constexpr (I == 9) {
                BAND_FENCE();
                pp_wait_lgkm();
                if (!(DBG & 8)) pp_barrier();
            }
            if constexpr (I >= 10 && I < 14) issue_piece(pc4, SLOT, I - 10);
            if constexpr (NK != 0 && I >= 10 && I < 14) { conv_x(2 * (I - 10)); conv_x(2 * (I - 10) + 1); }
            if constexpr (NK != 0 && (I == 14 || I == 15)) { conv_c(2 * (I - 14)); conv_c(2 * (I - 14) + 1); }
            if constexpr (I >= 12 && I < 20) read_w(std::integral_constant<int, NSLOT>{}, std::integral_constant<int, I - 12>{}, std::integral_constant<int, 1>{});
            if constexpr (NK != 0 && I >= 16 && I < 20) conv_l(I - 16);
            if constexpr (NK != 0 && (I == 20 || I == 21)) conv_g(NKT{}, I - 20);
            if constexpr (ZJ >= 0 && I >= 16) {
#pragma unroll
                for (int e = 0; e < 2; ++e) acc1[ZJ < 0 ? 0 : ZJ][2 * (I - 16) + e] = 0.f;
            }
            BAND_FENCE();
        });
        if constexpr (NK != 0) { a_hi[PAR ^ 1] = gh; a_lo[PAR ^ 1] = gl; }
        BAND_FENCE();
    };

    // stage-2 bias values for fragment (J, G) of column step c: features n0 + 32 J + 16 G + {0..3, 8..11} + 4 hi
    auto read_bias = [&](int n0, int J, int G) {
        const float* bp = bias1s + n0 + 32 * J + 16 * G + 4 * hi;
        bq[0] = *reinterpret_cast<const f32x4*>(bp);
        bq[1] = *reinterpret_cast<const f32x4*>(bp + 8);
    };
    // a whole fragment outside of any MFMA cover (pipeline head; first fragment of stage 2)
    auto convert_now = [&](auto kind_tag, auto set_tag, auto j_tag, auto g_tag, int par) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { conv_f(kind_tag, set_tag, j_tag, g_tag, e); conv_x(e); }
#pragma unroll
        for (int p = 0; p < 4; ++p) conv_c(p);
#pragma unroll
        for (int p = 0; p < 4; ++p) conv_l(p);
        conv_g(kind_tag, 0);
        conv_g(kind_tag, 1);
        a_hi[par] = gh;
        a_lo[par] = gl;
    };

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    using TM = std::integral_constant<int, -1>;

    // ---- pipeline head: values and pieces of sub-steps 0..3, in the steady state's order (values, then pieces)
    __syncthreads();                               // bias tables written (and no LDS-DMA before the compiler's own ds_writes)
    // order: values 0, pieces 0, values 1, pieces 1, pieces 2, pieces 3
    band_sfor<0, 4>([&](auto tt) {
        constexpr int T = decltype(tt)::value;
        if constexpr (T < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) request_a(T, tt, q);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) issue_piece(piece_of(0, T), T, p);
    });
    if (!(DBG & 1)) band_wait_vm_dep<16>(lv[0][0], lv[0][1], tv[0][0], tv[0][1]);
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    pp_barrier();
    convert_now(T1{}, T0{}, T0{}, T0{}, 0);
    band_sfor<0, 8>([&](auto j) { read_w(T0{}, j, T1{}); });
    pp_wait_lgkm();
    hs_stamp(g.stamps, 1);

    // ---- the band
#pragma unroll 1
    for (int c = 0; c < ncol; ++c) {
        const int n0 = c << 8;
        // stage 1: four sub-steps per trip (slot and register set = sub-step & 3)
#pragma unroll 1
        for (int u = 0; u < nsub1; u += 4) {
            const bool last4 = u + 4 >= nsub1;
            // values requested by sub-step u + i are those of position u + i + 2 of this column step
            substep(T0{}, T0{}, T1{}, T1{}, T0{}, T0{}, TM{}, piece_of(c, u + 4), u + 2);
            substep(T1{}, T1{}, T1{}, T1{}, T0{}, T0{}, TM{}, piece_of(c, u + 5), u + 3);
            substep(T2{}, T0{}, T1{}, T1{}, T0{}, T0{}, TM{}, piece_of(c, u + 6), last4 ? 0 : u + 4);
            if (!last4) substep(T3{}, T1{}, T1{}, T1{}, T0{}, T0{}, TM{}, piece_of(c, u + 7), u + 5);
            else substep(T3{}, T1{}, T1{}, T0{}, T0{}, T0{}, TM{}, piece_of(c, u + 7), 0);       // the next fragment needs the finished accumulators
        }
        // first fragment of stage 2: features n0 .. n0 + 15 of h2, from the finished accumulators of tile 0
        read_bias(n0, 0, 0);
        band_mfma_settle();
        convert_now(T2{}, T0{}, T0{}, T0{}, 0);
        // stage 2: sub-step q = (tile q >> 1, half q & 1); its values requests are those of the next column step's head
        {
            band_sfor<0, 16>([&](auto qq) {
                constexpr int Q = decltype(qq)::value;
                constexpr int NQ = Q + 1;
                constexpr int NJ = (NQ >> 1) & 7, NG = NQ & 1;
                constexpr int NK = Q < 15 ? 2 : 1;
                constexpr int ZJ = (Q & 1) ? (Q >> 1) : -1;            // tile Q >> 1 was fully converted one sub-step ago
                if constexpr (Q < 15) read_bias(n0, NJ, NG);
                substep(std::integral_constant<int, Q & 3>{}, std::integral_constant<int, Q & 1>{}, T2{}, std::integral_constant<int, NK>{},
                        std::integral_constant<int, NJ>{}, std::integral_constant<int, NG>{}, std::integral_constant<int, ZJ>{}, Q < 12 ? piece_of(c, nsub1 + Q + 4) : piece_of(c + 1, Q - 12), Q >= 14 ? Q - 14 : 0);
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the re-fetched head of the stream has landed: the ring may go
    hs_stamp(g.stamps, 2);

    // ---- range guard
    {
        const uint16_t t1 = (uint16_t)max(pk1 & 0xffffu, pk1 >> 16);
        hs_report_peak(g.peak, t1 >= 0x7c00u ? __builtin_inff() : (float)__builtin_bit_cast(_Float16, t1), true);
        const uint16_t t2 = (uint16_t)max(pk2 & 0xffffu, pk2 >> 16);
        hs_report_peak(g.peak, t2 >= 0x7c00u ? __builtin_inff() : (float)__builtin_bit_cast(_Float16, t2), false);
    }

    // ---- output: lane = row, register quad = 4 consecutive outputs
    band_mfma_settle();
    {
        const int m = m0 + 32 * wave + l31;
        const bool rok = m < g.M && !((DBG & 32) && g.M > 0);
        float* orow = g.out + (size_t)min(m, g.M - 1) * g.ldo;
        const float as2 = g.acc_scale2;
        band_sfor<0, 8>([&](auto jj) {
            constexpr int J = decltype(jj)::value;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int c0 = 32 * J + 8 * rq + 4 * hi;
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias2s + c0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc2[J][4 * rq + e], as2, b[e]);
                if (rok) {
                    if (c0 + 3 < g.n2) {
                        *reinterpret_cast<float2*>(orow + c0) = make_float2(v[0], v[1]);
                        *reinterpret_cast<float2*>(orow + c0 + 2) = make_float2(v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < g.n2) orow[c0 + e] = v[e];
                    }
                }
            }
        });
    }
    hs_stamp(g.stamps, 3);
}

}  // namespace csi
