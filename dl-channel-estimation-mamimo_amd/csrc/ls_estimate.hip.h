// ls_estimate.hip.h - least-squares pilot estimate of one rx preamble per workgroup.
//
// Restates, for the GPU, ofdmdemod + helperMIMOChannelEstimate
// (packet_generation/phased_arr/generate_maMIMO_LTF.m:336-342, helperMIMOChannelEstimate.m:24-36):
//   F[s][f]  = sum_n x[s*320 + 64 + n] exp(-2 pi i f n / 256)            (CP dropped, unscaled FFT)
//   H[j][q]  = sum_s F[s][f(q)] * P[j][s]  /  (Nt * ltf[q])              (P real; Puse = P')
// for the 234 data bins q (fftshift-ed bin order, nulls and pilots removed).
//
// Data flow per workgroup (one (packet, rx) pair, 256 threads):
//   HBM --float4, coalesced--> registers --digit-reversed scatter--> LDS  [Nt][re|im][256(+pad)]
//   in-place radix-4 DIT FFT, one wave per LTF symbol (4 stages, natural-order output)
//   despread = real [Nt x Nt] x [Nt x 234] product per re/im plane on v_mfma_f32_32x32x2_f32
//   (B operand read straight from the LDS spectrum through the bin table), scaled and written
//   coalesced to H[(p,r)][j][q].
// HBM-bound: 2560 B in + 1872 B out per pair; everything else stays in LDS/registers.
// Nt <= 64 uses this FFT-first kernel; larger Nt the despread-first kernel further down.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LS_FFT = 256;
constexpr int LS_CP = 64;
constexpr int LS_SYM = 320;
constexpr int LS_NDATA = 234;
constexpr int LS_PLANE = LS_FFT + 32;        // 4 pad floats per 32 -> stage-2 butterflies conflict-free
constexpr int LS_THREADS = 256;

struct LsArgs {
    const float* ltf_re;     // [nblk][len_ltf]
    const float* ltf_im;
    const float* P;          // [nt][nt] row j = pilot sequence of tx j
    const float* Ppad;       // [ceil32(nt)][ldp] zero-padded copy of P (chunked kernel)
    const uint16_t* Pbf;     // [ceil16(nt)/16][3 pieces][ceil32(nt)/32][2][32][8] bf16 pieces of P in MFMA operand order (ls_estimate_ringb_kernel)
    int ldp;                 // ceil32(nt)
    int dbg;                 // timing experiments only ("ls_debug" option): 1 skip FFT, 2 skip despread, 4 skip stores, 8 skip scatter, 64 WITH the MFMA drain of round 3 (ringb)
    const float* tw;         // [2][256] cos / -sin table, exp(-2 pi i u / 256)
    const int* bin_pos;      // [234] natural-order FFT index f(q) of data bin q
    const float* denom;      // [234] nt * ltf[q]
    float* h_re;             // [nblk][nt][234]
    float* h_im;
    int nt;
    int len_ltf;
    const int* perm;         // Hadamard-equivalent pilot (ls_estimate_fwht2_kernel<..., PERM>): [4][nt] = source symbol of transform input u,
                             // its sign (float bits), byte offset (antenna * 234 * 4) of the output row of transform row r, its sign; null for the Sylvester matrix itself
};

__device__ __forceinline__ int ls_phys(int p) { return p + ((p >> 5) << 2); }


// In-place 256-point radix-4 DIT FFT of NS LDS row pairs (re, im) by ONE wave, the NS transforms
// interleaved in one instruction stream (independent chains hide the LDS round-trip latency of
// each stage).  Inputs sit at base-4 digit-reversed positions, outputs are in natural bin order.
// lane = one radix-4 butterfly per stage.  Positions go through ls_phys() (4 pad floats per 32).
template <int NS, int ST_BEGIN = 0>
__device__ __forceinline__ void ls_fft256_wave(float* const (&fr)[NS], const float* tw_re, const float* tw_im, int lane) {
#pragma unroll
    for (int st = ST_BEGIN; st < 4; ++st) {
        const int L = 1 << (2 * st);
        const int j = lane & (L - 1);
        const int base = (lane >> (2 * st)) * 4 * L + j;
        const int tstep = 64 >> (2 * st);               // 256 / (4L)
        int p[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) p[m] = ls_phys(base + m * L);
        float xr[NS][4], xi[NS][4];
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                xr[n][m] = fr[n][p[m]];
                xi[n][m] = fr[n][LS_PLANE + p[m]];
            }
        float yr[NS][4], yi[NS][4];
        float twc[4], tws[4];
        if (st > 0) {
#pragma unroll
            for (int m = 1; m < 4; ++m) {
                const int u = (j * m * tstep) & 255;
                twc[m] = tw_re[u];
                tws[m] = tw_im[u];
            }
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            if (st > 0) {
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    const float r = xr[n][m] * twc[m] - xi[n][m] * tws[m];
                    const float i = xr[n][m] * tws[m] + xi[n][m] * twc[m];
                    xr[n][m] = r;
                    xi[n][m] = i;
                }
            }
            // 4-point DFT: y_q = sum_m (-i)^(m q) x_m
            const float ar = xr[n][0] + xr[n][2], ai = xi[n][0] + xi[n][2];
            const float br = xr[n][0] - xr[n][2], bi = xi[n][0] - xi[n][2];
            const float cr = xr[n][1] + xr[n][3], ci = xi[n][1] + xi[n][3];
            const float dr = xr[n][1] - xr[n][3], di = xi[n][1] - xi[n][3];
            yr[n][0] = ar + cr; yi[n][0] = ai + ci;
            yr[n][1] = br + di; yi[n][1] = bi - dr;       // x0 - i x1 - x2 + i x3
            yr[n][2] = ar - cr; yi[n][2] = ai - ci;
            yr[n][3] = br - di; yi[n][3] = bi + dr;       // x0 + i x1 - x2 - i x3
        }
        // all lanes of this wave must have read before anyone overwrites
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                fr[n][p[m]] = yr[n][m];
                fr[n][LS_PLANE + p[m]] = yi[n][m];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// FFT of rows first, first+step, ... < n of an image [n][2][LS_PLANE] by this wave, two at a time
__device__ __forceinline__ void ls_fft_rows(float* F, int first, int n, const float* tw_re, const float* tw_im, int lane,
                                            int step = 4) {
    int s = first;
    for (; s + step < n; s += 2 * step) {
        float* const pr[2] = {F + (size_t)s * 2 * LS_PLANE, F + (size_t)(s + step) * 2 * LS_PLANE};
        ls_fft256_wave<2>(pr, tw_re, tw_im, lane);
    }
    if (s < n) {
        float* const pr[1] = {F + (size_t)s * 2 * LS_PLANE};
        ls_fft256_wave<1>(pr, tw_re, tw_im, lane);
    }
}

// Persistent: gridDim.x workgroups walk the (packet, rx) items; while item i is transformed and
// despread, the samples of item i + gridDim.x are already in flight into registers (SPW symbols
// per wave, 2 x float4 each), so the HBM stream does not stop during the compute phases.
template <int SPW>     // symbols per wave held in registers: 8 covers nt <= 32, 16 covers nt <= 64
__global__ __launch_bounds__(LS_THREADS) void ls_estimate_kernel(const LsArgs a, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw_re = smem;                      // [256]
    float* tw_im = smem + LS_FFT;             // [256]
    float* F = smem + 2 * LS_FFT;             // [nt][2][LS_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = a.nt;
    const int n_jt = (nt + 31) >> 5;
    const int ksteps = (nt + 1) >> 1;
    const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);      // lane = d3 d2 d1 -> d1 d2 d3

    tw_re[tid] = a.tw[tid];
    tw_im[tid] = a.tw[LS_FFT + tid];

    // wave w owns symbols w, w+4, ...; lane holds samples 4*lane..4*lane+3 of each FFT window
    f32x4 vr[SPW], vi[SPW];
    auto fetch = [&](size_t blk) {
        const float* gre = a.ltf_re + blk * a.len_ltf + LS_CP + 4 * lane;
        const float* gim = a.ltf_im + blk * a.len_ltf + LS_CP + 4 * lane;
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int s = min(wave + 4 * u, nt - 1);          // clamp: surplus slots re-read a valid symbol
            vr[u] = *reinterpret_cast<const f32x4*>(gre + (size_t)s * LS_SYM);
            vi[u] = *reinterpret_cast<const f32x4*>(gim + (size_t)s * LS_SYM);
        }
    };

    size_t blk = blockIdx.x;
    if (blk < (size_t)nblk) fetch(blk);
    for (; blk < (size_t)nblk; blk += gridDim.x) {
        // ---- registers -> LDS at base-4 digit-reversed positions
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int s = wave + 4 * u;
            if (s < nt) {
                float* fr = F + (size_t)s * 2 * LS_PLANE;
                float* fi = fr + LS_PLANE;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int p = ls_phys(c * 64 + rev3);
                    fr[p] = vr[u][c];
                    fi[p] = vi[u][c];
                }
            }
        }
        __syncthreads();
        // ---- next item's samples start streaming now
        const size_t nxt = blk + gridDim.x;
        if (nxt < (size_t)nblk) fetch(nxt);

        // ---- FFT: wave w transforms symbols w, w+4, ...
        ls_fft_rows(F, wave, nt, tw_re, tw_im, lane);
        __syncthreads();

        // ---- despread on the matrix core: D[j][q] = sum_s P[j][s] * F[s][f(q)]
        for (int qt = wave; qt < 8; qt += 4) {
            const int q = qt * 32 + l31;
            const bool qok = q < LS_NDATA;
            const int pos = ls_phys(a.bin_pos[qok ? q : 0]);
            const float den = a.denom[qok ? q : 0];
            for (int jt = 0; jt < n_jt; ++jt) {
                const int ja = jt * 32 + l31;                   // A-operand row of this lane
                const bool jok = ja < nt;
                f32x16 dre, dim;
#pragma unroll
                for (int e = 0; e < 16; ++e) { dre[e] = 0.f; dim[e] = 0.f; }
                // this lane's pilot entries P[ja][hi], P[ja][2+hi], ... fetched up front (independent
                // loads) instead of one L1 round trip per MFMA step
                float pvs[2 * SPW];
#pragma unroll
                for (int ks = 0; ks < 2 * SPW; ++ks) {
                    const int s = 2 * ks + hi;
                    pvs[ks] = (jok && s < nt) ? a.P[ja * nt + s] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 2 * SPW; ++ks) {
                    if (ks >= ksteps) break;
                    const int s = 2 * ks + hi;
                    const bool sok = s < nt;
                    const float pv = pvs[ks];
                    const float* fr = F + (size_t)(sok ? s : 0) * 2 * LS_PLANE;
                    const float bre = sok ? fr[pos] : 0.f;
                    const float bim = sok ? fr[LS_PLANE + pos] : 0.f;
                    dre = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, bre, dre, 0, 0, 0);
                    dim = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, bim, dim, 0, 0, 0);
                }
                if (qok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (j < nt) {
                            const size_t o = (blk * nt + j) * LS_NDATA + q;
                            a.h_re[o] = dre[r] / den;
                            a.h_im[o] = dim[r] / den;
                        }
                    }
                }
            }
        }
        __syncthreads();          // spectra consumed; LDS may be overwritten by the next item
    }
}

// ---------------------------------------------------------------------------------------------
// Chunked FFT-first kernel for 32 < Nt <= 32*JT: the same data flow, but the LTF symbols pass
// through LDS in chunks of CH (16: 36.9 KiB, 32: 73.7 KiB, so two workgroups still share a CU) and the despread
// products D[j][q] += sum_{s in chunk} P[j][s] F[s][q] stay in the MFMA accumulators across
// the chunks (NW waves x QW bin tiles x JT antenna tiles x {re, im} x 16 registers).  The input
// is read exactly once; the next chunk's samples are requested as soon as the transforms are
// done and stream in beside the MFMAs.
template <int JT, int NW, int CH, int MINW = 2>
__global__ __launch_bounds__(64 * NW, MINW) void ls_estimate_chunked_kernel(const LsArgs a, int nblk) {
    constexpr bool EARLY = JT == 1 || (JT == 2 && NW == 8);      // accumulators small enough to hold the prefetch across the transforms
    constexpr int SPW = CH / NW;               // symbols per wave and chunk
    constexpr int QW = 8 / NW;                 // bin tiles (32 bins) per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw_re = smem;
    float* tw_im = smem + LS_FFT;
    float* F = smem + 2 * LS_FFT;              // [CH][2][LS_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = a.nt;
    const int nchunk = (nt + CH - 1) / CH;
    const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);

    for (int i = tid; i < LS_FFT; i += 64 * NW) {
        tw_re[i] = a.tw[i];
        tw_im[i] = a.tw[LS_FFT + i];
    }

    f32x4 vr[SPW], vi[SPW];
    auto fetch = [&](size_t blk, int ch) {
        const float* gre = a.ltf_re + blk * a.len_ltf + LS_CP + 4 * lane;
        const float* gim = a.ltf_im + blk * a.len_ltf + LS_CP + 4 * lane;
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int s = min(ch * CH + wave + NW * u, nt - 1);      // clamp: surplus slots re-read a valid symbol
            vr[u] = *reinterpret_cast<const f32x4*>(gre + (size_t)s * LS_SYM);
            vi[u] = *reinterpret_cast<const f32x4*>(gim + (size_t)s * LS_SYM);
        }
    };

    // bin positions / scale of this wave's bin tiles
    int pos[QW];
    float rden[QW];
    bool qok[QW];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        const int q = (wave + NW * qi) * 32 + l31;
        qok[qi] = q < LS_NDATA;
        pos[qi] = ls_phys(a.bin_pos[qok[qi] ? q : 0]);
        rden[qi] = a.denom[qok[qi] ? q : 0];
    }

    size_t blk = blockIdx.x;
    if (blk < (size_t)nblk) fetch(blk, 0);
    for (; blk < (size_t)nblk; blk += gridDim.x) {
        f32x16 acc[QW][JT][2];
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[qi][jt][0][e] = 0.f; acc[qi][jt][1][e] = 0.f; }

        for (int ch = 0; ch < nchunk; ++ch) {
            const int ns = min(CH, nt - ch * CH);                 // symbols in this chunk
            // ---- registers -> LDS rows wave, wave+NW, ... at base-4 digit-reversed positions
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
                const int r = wave + NW * u;
                if (r < ns && !(a.dbg & 8)) {
                    float* fr = F + (size_t)r * 2 * LS_PLANE;
                    float* fi = fr + LS_PLANE;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int p = ls_phys(c * 64 + rev3);
                        fr[p] = vr[u][c];
                        fi[p] = vi[u][c];
                    }
                }
            }
            __syncthreads();
            // ---- the next chunk (or the next item's first chunk) starts streaming: before the
            // transforms where the registers allow it (JT == 1), behind them otherwise
            if (EARLY) {
                if (ch + 1 < nchunk) fetch(blk, ch + 1);
                else if (blk + gridDim.x < (size_t)nblk) fetch(blk + gridDim.x, 0);
            }
            // two interleaved transforms per wave where the registers allow it (JT == 1), else one
            if (JT == 1) {
                if (!(a.dbg & 1)) ls_fft_rows(F, wave, ns, tw_re, tw_im, lane, NW);
            } else {
                for (int r = wave; r < ns && !(a.dbg & 1); r += NW) {
                    float* const pr[1] = {F + (size_t)r * 2 * LS_PLANE};
                    ls_fft256_wave<1>(pr, tw_re, tw_im, lane);
                }
            }
            __syncthreads();
            if (!EARLY) {
                if (ch + 1 < nchunk) fetch(blk, ch + 1);
                else if (blk + gridDim.x < (size_t)nblk) fetch(blk + gridDim.x, 0);
            }

            // ---- despread on the matrix core: D[j][q] += sum_s P[j][s] * F[s][f(q)].  Branch-free: the
            // pilot entries come from the zero-padded copy Ppad (rows / columns >= nt are 0), so rows
            // of F beyond this chunk's symbols only ever meet a zero (they hold finite spectra of
            // the previous chunk: chunk 0 is always full when nt > 32).  Operands of step ks + 1
            // are requested before the MFMAs of step ks.
            const int ksteps = (a.dbg & 2) ? 0 : (ns + 1) >> 1;
            const float* prow = a.Ppad + (size_t)l31 * a.ldp + ch * CH + hi;
            const float* frow = F + (size_t)hi * 2 * LS_PLANE;
            float pvn[JT], bren[QW], bimn[QW];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) pvn[jt] = prow[(size_t)jt * 32 * a.ldp];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) { bren[qi] = frow[pos[qi]]; bimn[qi] = frow[LS_PLANE + pos[qi]]; }
            for (int ks = 0; ks < ksteps; ++ks) {
                float pv[JT], bre[QW], bim[QW];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) pv[jt] = pvn[jt];
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) { bre[qi] = bren[qi]; bim[qi] = bimn[qi]; }
                const int kn = min(ks + 1, CH / 2 - 1);
                const float* fn = frow + (size_t)kn * 4 * LS_PLANE;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) pvn[jt] = prow[(size_t)jt * 32 * a.ldp + 2 * kn];
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) { bren[qi] = fn[pos[qi]]; bimn[qi] = fn[LS_PLANE + pos[qi]]; }
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        acc[qi][jt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[jt], bre[qi], acc[qi][jt][0], 0, 0, 0);
                        acc[qi][jt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[jt], bim[qi], acc[qi][jt][1], 0, 0, 0);
                    }
            }
            __syncthreads();          // spectra consumed; LDS may be overwritten by the next chunk
        }
        // ---- scale and store: rows j = jt*32 + (r&3) + 8*(r>>2) + 4*hi, bins q coalesced over the lanes.
        // One base pointer per plane and compile-time row offsets; full antenna tiles store without
        // per-row tests (a branch per store splits the epilogue into ~128 blocks whose hoisted
        // addresses spill).
        const bool full = (nt & 31) == 0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            if (!qok[qi] || (a.dbg & 4)) continue;
            const size_t o = (blk * nt + 4 * hi) * LS_NDATA + (size_t)((wave + NW * qi) * 32 + l31);
            float* pre = a.h_re + o;
            float* pim = a.h_im + o;
            const float inv = rden[qi];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt * 32 >= nt) break;
                if (full || (jt + 1) * 32 <= nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        pre[jo * LS_NDATA] = acc[qi][jt][0][r] / inv;
                        pim[jo * LS_NDATA] = acc[qi][jt][1][r] / inv;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        if (jo + 4 * hi < nt) {
                            pre[jo * LS_NDATA] = acc[qi][jt][0][r] / inv;
                            pim[jo * LS_NDATA] = acc[qi][jt][1][r] / inv;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Walsh-Hadamard despread for the pilot matrix the 802.11-style sounding actually uses beyond 8
// streams: the Sylvester Hadamard matrix P[j][s] = (-1)^popcount(j & s)  (Nt a power of two).
// The despread  H[j] = sum_s P[j][s] F[s]  is then a fast Walsh-Hadamard transform over the symbol
// index: Nt log2 Nt additions per bin instead of Nt^2 multiply-adds on the matrix pipe (which at
// Nt = 64 is 0.5 ms of pure MFMA time per config-3 launch).  Same chunked data flow as
// ls_estimate_chunked_kernel (16 symbols per chunk, one thread per bin):
//   per chunk c   w[0..15] = FWHT16 of this bin's 16 spectra (registers, 64 additions per plane)
//                 acc[16 a + j'] += (-1)^popcount(a & c) * w[j']      (the cross-chunk stages:
//                 H_N = H_{N/16} (x) H_16 in the Sylvester order)
// The host selects this kernel only when csi_set_pilot saw exactly that matrix; any other P takes
// the MFMA despread.  Results differ from it only in summation order.
// SPLIT = 2 (Nt = 128): 512 threads, two threads per bin, each owning half of the output blocks (64
// antennas = 128 accumulator registers); both run the same FWHT16 of the chunk.
template <int NT, int SPLIT = 1>
__global__ __launch_bounds__(256 * SPLIT, 2) void ls_estimate_fwht_kernel(const LsArgs a, int nblk) {
    static_assert(NT == 16 || NT == 32 || NT == 64 || NT == 128, "power-of-two antenna counts up to 128");
    static_assert(NT / SPLIT <= 64, "at most 64 antennas (128 accumulators) per thread");
    constexpr int CH = 16, NW = 4 * SPLIT, SPW = CH / NW, NCH = NT / CH, NOWN = NCH / SPLIT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw_re = smem;
    float* tw_im = smem + LS_FFT;
    float* F = smem + 2 * LS_FFT;              // [CH][2][LS_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);
    if (tid < LS_FFT) {
        tw_re[tid] = a.tw[tid];
        tw_im[tid] = a.tw[LS_FFT + tid];
    }

    f32x4 vr[SPW], vi[SPW];
    auto fetch = [&](size_t blk, int ch) {
        const float* gre = a.ltf_re + blk * a.len_ltf + LS_CP + 4 * lane;
        const float* gim = a.ltf_im + blk * a.len_ltf + LS_CP + 4 * lane;
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int s = ch * CH + wave + NW * u;
            vr[u] = *reinterpret_cast<const f32x4*>(gre + (size_t)s * LS_SYM);
            vi[u] = *reinterpret_cast<const f32x4*>(gim + (size_t)s * LS_SYM);
        }
    };

    const int q = tid & 255;                    // this thread's data bin
    const int own = (tid >> 8) * NOWN;          // first output block (of 16 antennas) this thread accumulates
    const bool qok = q < LS_NDATA;
    const int pos = ls_phys(a.bin_pos[qok ? q : 0]);
    const float den = a.denom[qok ? q : 0];

    size_t blk = blockIdx.x;
    if (blk < (size_t)nblk) fetch(blk, 0);
    for (; blk < (size_t)nblk; blk += gridDim.x) {
        float hre[NOWN * CH], him[NOWN * CH];
#pragma unroll
        for (int j = 0; j < NOWN * CH; ++j) { hre[j] = 0.f; him[j] = 0.f; }
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
                float* fr = F + (size_t)(wave + NW * u) * 2 * LS_PLANE;
                float* fi = fr + LS_PLANE;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int p = ls_phys(c * 64 + rev3);
                    fr[p] = vr[u][c];
                    fi[p] = vi[u][c];
                }
            }
            __syncthreads();
            // next samples requested before the transforms (two interleaved per wave) while the
            // registers allow it, behind single transforms at NT = 64 (128 accumulators)
            if (NT / SPLIT <= 32) {
                if (ch + 1 < NCH) fetch(blk, ch + 1);
                else if (blk + gridDim.x < (size_t)nblk) fetch(blk + gridDim.x, 0);
                ls_fft_rows(F, wave, CH, tw_re, tw_im, lane, NW);
            } else {
                for (int r = wave; r < CH; r += NW) {
                    float* const pr[1] = {F + (size_t)r * 2 * LS_PLANE};
                    ls_fft256_wave<1>(pr, tw_re, tw_im, lane);
                }
            }
            __syncthreads();
            if (NT / SPLIT > 32) {
                if (ch + 1 < NCH) fetch(blk, ch + 1);
                else if (blk + gridDim.x < (size_t)nblk) fetch(blk + gridDim.x, 0);
            }
            // ---- this bin's 16 spectra -> registers, FWHT16 per plane
            float wr[CH], wi[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                wr[r] = F[(size_t)r * 2 * LS_PLANE + pos];
                wi[r] = F[(size_t)r * 2 * LS_PLANE + LS_PLANE + pos];
            }
#pragma unroll
            for (int h = 1; h < CH; h <<= 1)
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (!(i & h)) {
                        const float xr = wr[i], yr = wr[i + h], xi = wi[i], yi = wi[i + h];
                        wr[i] = xr + yr; wr[i + h] = xr - yr;
                        wi[i] = xi + yi; wi[i + h] = xi - yi;
                    }
            // ---- cross-chunk stages: block a of the output takes +-w by the sign of H_{NT/16}[a][ch]
#pragma unroll
            for (int ab = 0; ab < NOWN; ++ab) {
                const float sgn = (__builtin_popcount((own + ab) & ch) & 1) ? -1.f : 1.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    hre[ab * CH + j] = fmaf(sgn, wr[j], hre[ab * CH + j]);
                    him[ab * CH + j] = fmaf(sgn, wi[j], him[ab * CH + j]);
                }
            }
            __syncthreads();          // spectra consumed
        }
        if (qok) {
            float* pre = a.h_re + (blk * NT + own * CH) * LS_NDATA + q;
            float* pim = a.h_im + (blk * NT + own * CH) * LS_NDATA + q;
#pragma unroll
            for (int j = 0; j < NOWN * CH; ++j) {
                pre[j * LS_NDATA] = hre[j] / den;
                pim[j * LS_NDATA] = him[j] / den;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Walsh-Hadamard kernel, second generation: the same arithmetic as ls_estimate_fwht_kernel, fed by
// LDS-DMA instead of register prefetch.
//   * every LTF symbol plane (1 KiB) goes HBM -> LDS by one global_load_lds_dwordx4 per wave into a
//     ring of NSTG raw chunk slots [CH][re|im][256] (natural sample order, unpadded).  No VGPRs carry
//     samples, so NSTG - 1 whole chunks stay in flight behind the one being transformed however many
//     accumulators the despread owns (Nt = 64 / 128: 128 per thread).
//   * the base-4 digit reversal of the DIT transform is folded into stage 0: butterfly `lane` of that
//     stage needs elements rev3(lane) + 64 m of the natural-order row - across the wave a permutation
//     of 64 consecutive floats, conflict-free - and writes its four outputs as one ds_write_b128 into
//     the padded image F.  Stages 1-3 run in F as before.  The separate scatter pass is gone.
//   * a wave transforms exactly the rows it fetched, so a slot row is recycled (next DMA issued) as
//     soon as that wave's stage-0 reads have returned: no workgroup barrier on the load path, two
//     per chunk on the spectrum path (spectra complete / spectra consumed).
//   * the DMA is inline asm: hipcc would put s_waitcnt vmcnt(0) in front of every ds_write that follows a
//     builtin LDS-DMA.  Landing is awaited with a counted s_waitcnt vmcnt(younger chunks * R); stores
//     issued in between only make that wait conservative (loads complete in order among themselves).
__device__ __forceinline__ void ls_dma16(const float* g_lane_src, uint32_t lds_byte_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_lane_src), "s"(lds_byte_off) : "memory");
}
template <int N>
__device__ __forceinline__ void ls_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ls_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// ---- complex-interleaved spectra image of the two ring kernels: rows of 256 (re, im) pairs, 4 pad elements per 16
// (stage-1 butterflies of 16 lanes then fall on 16 distinct 8-byte bank pairs; stages 2-3 and the per-bin gathers
// read consecutive elements).  Every LDS access of stages 1-3 is one ds_*_b64 per complex value and the arithmetic
// runs on the packed-fp32 pipe: v_pk_add_f32 for the butterflies, v_pk_mul_f32 + v_pk_fma_f32 per twiddle product - half the VALU and LDS
// instructions of the planar transform (ls_fft256_wave), which is what these kernels are bound by once the ring hides the HBM latency.
// The +-i ROTATIONS (outputs 1 and 3 of a butterfly) are two single adds each (round 6): as `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] neg_*` - the second
// source's halves swapped - they returned D.lo = A.lo in lanes 48-63 whenever another wave of the SIMD was issuing MFMAs (a bf16 GEMM workgroup sharing the CU):
// on gfx950 a packed-fp32 instruction with op_sel = 1 on src1 and 0 on src0 loses that operand in the last 16-lane pass (tools/pk_opsel_probe.hip reproduces it
// without any LS code; profiles/r06_pk_opsel_probe.txt).  Same kernel time.  The twiddle product's v_pk_fma_f32 has op_sel = 1 on src0 AND src1:
// not affected (0 of 1e10 in the probe, never seen in a census of bad items).  tests/test_host_round4.py checks the library's machine code for the form.
// CSI_LS_VAR_DEFAULT = the form of the transform when nothing else is asked for (the VAR bits listed at lsc_stage0_write): 128 in the product;
// tools/ls_opsel_hunt.sh builds the library with other values (0 = the packed rotations) for the reproducible case of profiles/r06_small_calls.txt (4)
#ifndef CSI_LS_VAR_DEFAULT
#define CSI_LS_VAR_DEFAULT 128
#endif
constexpr int LSC_ROW = LS_FFT + LS_FFT / 4;            // 320 complex elements per padded row
constexpr int LSC_NTW = 256;                            // twiddle table: stage 1 [3][4], stage 2 [3][16], stage 3 [3][64] (252 used)
__device__ __forceinline__ int lsc_phys(int e) { return e + ((e >> 4) << 2); }

// a - i b  and  a + i b
__device__ __forceinline__ f32x2 pk_add_mi(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add_pi(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// race-hunt forms (ls_estimate_ringb_kernel<..., VAR & 2>): the destination never shares registers with a source
__device__ __forceinline__ f32x2 pk_add_mi_ec(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add_pi_ec(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_cmul_ec(f32x2 x, f32x2 w) {
    f32x2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(d) : "v"(x), "v"(w), "v"(t));
    return d;
}
// VAR & 32: two idle cycles behind every op_sel operation (the next VALU instruction cannot follow it back to back)
__device__ __forceinline__ f32x2 pk_add_mi_np(f32x2 a, f32x2 b) {
    f32x2 d;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\ts_nop 1" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add_pi_np(f32x2 a, f32x2 b) {
    f32x2 d;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\ts_nop 1" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_cmul_np(f32x2 x, f32x2 w) {
    f32x2 t, d;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\ts_nop 1" : "=v"(t) : "v"(x), "v"(w));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\ts_nop 1" : "=v"(d) : "v"(x), "v"(w), "v"(t));
    return d;
}
// VAR & 64: the same three operations without any op_sel / packed instruction (scalar fp32 adds and fmas; same rounding: one
// rounding per add, the product term of the complex multiply rounded once before the fma as in the packed form)
__device__ __forceinline__ f32x2 sc_add_mi(f32x2 a, f32x2 b) {
    float dx, dy;
    asm("v_add_f32 %0, %1, %2" : "=v"(dx) : "v"(a[0]), "v"(b[1]));
    asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(a[1]), "v"(b[0]));
    return f32x2{dx, dy};
}
__device__ __forceinline__ f32x2 sc_add_pi(f32x2 a, f32x2 b) {
    float dx, dy;
    asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(a[0]), "v"(b[1]));
    asm("v_add_f32 %0, %1, %2" : "=v"(dy) : "v"(a[1]), "v"(b[0]));
    return f32x2{dx, dy};
}
__device__ __forceinline__ f32x2 sc_cmul(f32x2 x, f32x2 w) {
    float t0, t1, dx, dy;
    asm("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(w[0]));
    asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(w[0]));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(dx) : "v"(x[1]), "v"(w[1]), "v"(t0));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(dy) : "v"(x[0]), "v"(w[1]), "v"(t1));
    return f32x2{dx, dy};
}
// x * w for w = (c, s):  (xr c - xi s, xi c + xr s)
__device__ __forceinline__ f32x2 pk_cmul(f32x2 x, f32x2 w) {
    f32x2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(x), "v"(w), "v"(t));
    return d;
}
// VAR & 96 == 96: the scalar forms with the same two idle cycles behind each operation group (volatile like the _np forms)
__device__ __forceinline__ f32x2 sc_add_mi_np(f32x2 a, f32x2 b) {
    float dx, dy;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(dx) : "v"(a[0]), "v"(b[1]));
    asm volatile("v_sub_f32 %0, %1, %2\n\ts_nop 1" : "=v"(dy) : "v"(a[1]), "v"(b[0]));
    return f32x2{dx, dy};
}
__device__ __forceinline__ f32x2 sc_add_pi_np(f32x2 a, f32x2 b) {
    float dx, dy;
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(a[0]), "v"(b[1]));
    asm volatile("v_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(dy) : "v"(a[1]), "v"(b[0]));
    return f32x2{dx, dy};
}
__device__ __forceinline__ f32x2 sc_cmul_np(f32x2 x, f32x2 w) {
    float t0, t1, dx, dy;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(w[0]));
    asm volatile("v_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(t1) : "v"(x[1]), "v"(w[0]));
    asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(dx) : "v"(x[1]), "v"(w[1]), "v"(t0));
    asm volatile("v_fma_f32 %0, %1, %2, %3\n\ts_nop 1" : "=v"(dy) : "v"(x[0]), "v"(w[1]), "v"(t1));
    return f32x2{dx, dy};
}
#if (CSI_LS_VAR_DEFAULT) & 512
// 512 (tools/ls_opsel_hunt.sh, never in the product): every +-i rotation computed BOTH ways; a packed result that differs from the scalar one is logged
// (operands, result, thread, workgroup) into a device buffer that csi_debug_opsel_log() copies out
__device__ unsigned g_opsel_log[4 + 16 * 4096];
__device__ __forceinline__ float opsel_copy(float x) { float y; asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ f32x2 opsel_both(unsigned kind, f32x2 a, f32x2 b) {
    // every value that goes into the log is taken with an asm v_mov_b32 (the compiler can neither pack nor merge them): the operands BEFORE the packed
    // operation, its result, the operands again AFTER it, and the result of the same instruction executed once more a few cycles later
    const float a0 = opsel_copy(a[0]), a1 = opsel_copy(a[1]), b0 = opsel_copy(b[0]), b1 = opsel_copy(b[1]);
    f32x2 p, q;
    if (kind == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
    else asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
    const float p0 = opsel_copy(p[0]), p1 = opsel_copy(p[1]);
    const float a0l = opsel_copy(a[0]), a1l = opsel_copy(a[1]), b0l = opsel_copy(b[0]), b1l = opsel_copy(b[1]);
    if (kind == 0) asm volatile("s_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(q) : "v"(a), "v"(b));
    else asm volatile("s_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(q) : "v"(a), "v"(b));
    const float q0 = opsel_copy(q[0]), q1 = opsel_copy(q[1]);
    float s0, s1;
    if (kind == 0) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(a0), "v"(b1)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(a1), "v"(b0)); }
    else { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s0) : "v"(a0), "v"(b1)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(a1), "v"(b0)); }
    const auto u = [](float x) { return __builtin_bit_cast(unsigned, x); };
    if (u(p0) != u(s0) || u(p1) != u(s1) || u(q0) != u(s0) || u(q1) != u(s1)) {
        const unsigned i = atomicAdd(&g_opsel_log[0], 1u);
        if (i < 4096) {
            unsigned* e = g_opsel_log + 4 + 16 * i;
            e[0] = kind | (threadIdx.x << 8); e[1] = blockIdx.x;
            e[2] = u(a0); e[3] = u(a1); e[4] = u(b0); e[5] = u(b1); e[6] = u(p0); e[7] = u(p1);
            e[8] = u(a0l); e[9] = u(a1l); e[10] = u(b0l); e[11] = u(b1l); e[12] = u(q0); e[13] = u(q1); e[14] = u(s0); e[15] = u(s1);
        }
    }
    return p;
}
#endif
template <int VAR>
__device__ __forceinline__ f32x2 lsc_cmul_v(f32x2 x, f32x2 w) {
    if (VAR & 256) return sc_cmul(x, w);          // 256: only the twiddle products in scalar operations
    if ((VAR & 96) == 96) return sc_cmul_np(x, w);
    return (VAR & 64) ? sc_cmul(x, w) : ((VAR & 32) ? pk_cmul_np(x, w) : ((VAR & 2) ? pk_cmul_ec(x, w) : pk_cmul(x, w)));
}
template <int VAR>
__device__ __forceinline__ f32x2 lsc_add_mi_v(f32x2 a, f32x2 b) {
#if (CSI_LS_VAR_DEFAULT) & 512
    if (VAR & 512) return opsel_both(0, a, b);
#endif
    if (VAR & 128) return sc_add_mi(a, b);        // 128: only the +-i rotations (outputs 1 and 3 of a butterfly) in scalar operations
    if ((VAR & 96) == 96) return sc_add_mi_np(a, b);
    return (VAR & 64) ? sc_add_mi(a, b) : ((VAR & 32) ? pk_add_mi_np(a, b) : ((VAR & 2) ? pk_add_mi_ec(a, b) : pk_add_mi(a, b)));
}
template <int VAR>
__device__ __forceinline__ f32x2 lsc_add_pi_v(f32x2 a, f32x2 b) {
#if (CSI_LS_VAR_DEFAULT) & 512
    if (VAR & 512) return opsel_both(1, a, b);
#endif
    if (VAR & 128) return sc_add_pi(a, b);
    if ((VAR & 96) == 96) return sc_add_pi_np(a, b);
    return (VAR & 64) ? sc_add_pi(a, b) : ((VAR & 32) ? pk_add_pi_np(a, b) : ((VAR & 2) ? pk_add_pi_ec(a, b) : pk_add_pi(a, b)));
}

// twc[off(st) + (m - 1) L + j] = exp(-2 pi i j m / (4 L)),  L = 4^st, off = 0 / 12 / 60: the three twiddles of butterfly
// j of stage st, contiguous in j (conflict-free, and no index arithmetic in the transform)
__device__ __forceinline__ void lsc_build_twiddles(f32x2* twc, const float* tw, int tid, int nthreads) {
    for (int i = tid; i < 252; i += nthreads) {
        const int st = i < 12 ? 1 : (i < 60 ? 2 : 3);
        const int r = i - (st == 1 ? 0 : (st == 2 ? 12 : 60));
        const int L = 1 << (2 * st), m = r / L + 1, j = r % L;
        const int u = (j * m * (64 >> (2 * st))) & 255;
        twc[i] = f32x2{tw[u], tw[LS_FFT + u]};
    }
}

// stage 0 of the DIT transform for this wave's SPW rows of a raw (planar, natural-order) chunk slot: butterfly `lane`
// takes samples rev3(lane) + 64 m; no twiddles
template <int SPW, int NW, int VAR = CSI_LS_VAR_DEFAULT>
__device__ __forceinline__ void lsc_stage0_read(const float* srow, int rev3, f32x2 (&y)[SPW][4]) {
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
        const float* sr = srow + (size_t)u * NW * 2 * LS_FFT + rev3;
        f32x2 x[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) x[m] = f32x2{sr[64 * m], sr[LS_FFT + 64 * m]};
        const f32x2 a = x[0] + x[2], b = x[0] - x[2], c = x[1] + x[3], d = x[1] - x[3];
        y[u][0] = a + c;
        y[u][1] = lsc_add_mi_v<VAR>(b, d);
        y[u][2] = a - c;
        y[u][3] = lsc_add_pi_v<VAR>(b, d);
    }
}
// VAR (race hunt, tools/ls_race_fast.py; 0 in every product instantiation): 1 = s_waitcnt lgkmcnt(0) behind every stage's writes
// (LDS write -> read order inside the wave), 2 = op_sel operations never in place, 4 = sixteen idle cycles per stage (the compiler
// places them between the stage's VALU operations: a schedule perturbation, measured to RAISE the event rate ~70 x), 8 (kernel) =
// every LDS-DMA of the wave landed before the "spectra complete" barrier, 16 = the sources of the op_sel adds stay live until the
// stage's writes are out, 32 = two idle cycles behind every op_sel operation, 64 = no op_sel / packed operation at all (scalar forms)
template <int SPW, int NW, int VAR = CSI_LS_VAR_DEFAULT>
__device__ __forceinline__ void lsc_stage0_write(f32x2* Fc, int wave, int lane, const f32x2 (&y)[SPW][4]) {
    const int p0 = 4 * lane + 4 * (lane >> 2);           // lsc_phys(4 lane): elements 4 lane .. 4 lane + 3, 32-byte aligned
    if (VAR & 4) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
        f32x2* fc = Fc + (size_t)(wave + NW * u) * LSC_ROW + p0;
        *reinterpret_cast<f32x4*>(fc) = f32x4{y[u][0][0], y[u][0][1], y[u][1][0], y[u][1][1]};
        *reinterpret_cast<f32x4*>(fc + 2) = f32x4{y[u][2][0], y[u][2][1], y[u][3][0], y[u][3][1]};
    }
    if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// stages 1-3, NS rows interleaved in one instruction stream
template <int NS, int VAR = CSI_LS_VAR_DEFAULT>
__device__ __forceinline__ void lsc_fft_stages(f32x2* const (&fr)[NS], const f32x2* twc, int lane) {
#pragma unroll
    for (int st = 1; st < 4; ++st) {
        const int L = 1 << (2 * st);
        const int j = lane & (L - 1);
        const int base = (lane >> (2 * st)) * 4 * L + j;
        const f32x2* tws = twc + (st == 1 ? 0 : (st == 2 ? 12 : 60)) + j;
        int p[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) p[m] = lsc_phys(base + m * L);
        f32x2 x[NS][4], y[NS][4], w[4], bb[NS], dd[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) x[n][m] = fr[n][p[m]];
#pragma unroll
        for (int m = 1; m < 4; ++m) w[m] = tws[(m - 1) * L];
#pragma unroll
        for (int n = 0; n < NS; ++n) {
#pragma unroll
            for (int m = 1; m < 4; ++m) x[n][m] = lsc_cmul_v<VAR>(x[n][m], w[m]);
            const f32x2 a = x[n][0] + x[n][2], b = x[n][0] - x[n][2], c = x[n][1] + x[n][3], d = x[n][1] - x[n][3];
            y[n][0] = a + c;
            y[n][1] = lsc_add_mi_v<VAR>(b, d);
            y[n][2] = a - c;
            y[n][3] = lsc_add_pi_v<VAR>(b, d);
            if (VAR & 16) { bb[n] = b; dd[n] = d; }
        }
        __builtin_amdgcn_wave_barrier();          // every lane has read before anyone overwrites
        if (VAR & 4) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) fr[n][p[m]] = y[n][m];
        if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (VAR & 16) {          // the sources of the op_sel adds stay live (nothing may be allocated over them) until the stage's writes are out
#pragma unroll
            for (int n = 0; n < NS; ++n) asm volatile("" ::"v"(bb[n]), "v"(dd[n]) : "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
// stages 1-3 of this wave's rows wave, wave + NW, ...: pairs interleaved when PAIR
template <int SPW, int NW, bool PAIR, int VAR = CSI_LS_VAR_DEFAULT>
__device__ __forceinline__ void lsc_fft_rows(f32x2* Fc, int wave, const f32x2* twc, int lane) {
    if (PAIR && SPW >= 2) {
#pragma unroll
        for (int u = 0; u + 1 < SPW; u += 2) {
            f32x2* const pr[2] = {Fc + (size_t)(wave + NW * u) * LSC_ROW, Fc + (size_t)(wave + NW * (u + 1)) * LSC_ROW};
            lsc_fft_stages<2, VAR>(pr, twc, lane);
        }
        if (SPW & 1) {
            f32x2* const pr[1] = {Fc + (size_t)(wave + NW * (SPW - 1)) * LSC_ROW};
            lsc_fft_stages<1, VAR>(pr, twc, lane);
        }
    } else {
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            f32x2* const pr[1] = {Fc + (size_t)(wave + NW * u) * LSC_ROW};
            lsc_fft_stages<1, VAR>(pr, twc, lane);
        }
    }
}

// PERM: the pilot matrix is P[j][s] = rs[j] H[sigma(j)][tau(s)] cs[s] (H Sylvester; csi_set_pilot finds sigma, tau and the signs), so
//     sum_s P[j][s] F[s] = rs[j] FWHT(G)[sigma(j)],   G[u] = cs[s] F[s] with s = tau^-1(u):
// transform input u is FETCHED from symbol tau^-1(u) (the ring's DMA source address comes from a table), its sign enters the first
// butterfly level of the Walsh-Hadamard transform as an fma operand, and transform row r is STORED to antenna sigma^-1(r) with the sign
// folded into the 1 / (Nt ltf) factor.  The four tables are wave-uniform reads from the constant address space (scalar loads: they
// return on lgkmcnt and do not queue behind the ring's vector-memory operations).
// SST: every store as `global_store_dword v_off, v_data, s[base]` - the row base (item, antenna) is wave-uniform and lives in scalar
// registers (scalar address arithmetic), the lane contributes 4 q: no vector address arithmetic per store.  (Inline asm: a dword store
// has no data hazard, and the kernel's counted vmcnt waits are conservative in stores already.)
__device__ __forceinline__ void ls_store_sbase(float* base_uniform, uint32_t byte_off, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(byte_off), "v"(v), "s"(base_uniform) : "memory");
}
template <int NT, int SPLIT, int CH, int NSTG, bool DBF = false, int MINB = (SPLIT == 1 ? 2 : 1), bool PERM = false, bool SST = PERM>
__global__ __launch_bounds__(256 * SPLIT, MINB) void ls_estimate_fwht2_kernel(const LsArgs a, int nblk) {
#define LS_WG_X blockIdx.x
#define LS_WG_N gridDim.x
#include "ls_fwht2_body.inc"
#undef LS_WG_X
#undef LS_WG_N
}

// The same body as a device function of (workgroup index, workgroup count): the one-packet path runs it in the SAME launch as layer 0 of the
// DNN (small_call.hip.h: small_l0_ls_kernel - the LS workgroups beside the weight-streaming ones, round 6), one item per workgroup
template <int NT, int SPLIT, int CH, int NSTG, bool DBF, bool PERM, bool SST>
__device__ __forceinline__ void ls_fwht2_body(const LsArgs& a, const int nblk, const unsigned wg_x, const unsigned wg_n) {
#define LS_WG_X wg_x
#define LS_WG_N wg_n
#include "ls_fwht2_body.inc"
#undef LS_WG_X
#undef LS_WG_N
}

// ---------------------------------------------------------------------------------------------
// Generic-P kernel on the same LDS-DMA ring: the front end of ls_estimate_fwht2_kernel (ring of raw chunk slots,
// digit reversal folded into stage 0, a wave transforms the rows it fetched) with the matrix-core despread of
// ls_estimate_chunked_kernel behind it (D[j][q] += sum_{s in chunk} P[j][s] F[s][f(q)] on v_mfma_f32_32x32x2_f32,
// accumulators persist over the chunks).  Any real P, any Nt from 16 to 32 JT: symbols beyond Nt in the last chunk
// are fetched clamped (a valid symbol again) and meet the zero columns of the padded P (kept in LDS).
template <int JT, int NW, int CH, int NSTG, int MINB = (NW == 4 ? 2 : 1)>
__global__ __launch_bounds__(64 * NW, MINB) void ls_estimate_ring_kernel(const LsArgs a, int nblk) {
    constexpr int SPW = CH / NW, QW = 8 / NW, R = 2 * SPW;
    static_assert(SPW >= 1 && QW >= 1 && NSTG >= 1 && NSTG <= 4 && (NSTG - 1) * R <= 63, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x2* twc = reinterpret_cast<f32x2*>(smem);                       // [LSC_NTW]
    f32x2* Fc = twc + LSC_NTW;                                         // [CH][LSC_ROW]
    float* S = reinterpret_cast<float*>(Fc + CH * LSC_ROW);           // [NSTG][CH][2][256]
    float* Pl = S + NSTG * CH * 2 * LS_FFT;    // [32 JT][ldp + 1] the padded pilot matrix: no global load may sit in the
                                               // steady-state loop (loads return in order - it would wait for the ring)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = a.nt;
    const int nchunk = (nt + CH - 1) / CH;
    const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);
    lsc_build_twiddles(twc, a.tw, tid, 64 * NW);
    const int ldl = a.ldp + 1;
    for (int i = tid; i < 32 * JT * a.ldp; i += 64 * NW) Pl[(i / a.ldp) * ldl + (i % a.ldp)] = a.Ppad[i];
    int pos[QW];
    float rden[QW];
    bool qok[QW];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        const int q = (wave + NW * qi) * 32 + l31;
        qok[qi] = q < LS_NDATA;
        pos[qi] = lsc_phys(a.bin_pos[qok[qi] ? q : 0]);
        rden[qi] = 1.0f / a.denom[qok[qi] ? q : 0];
    }
    __syncthreads();

    const uint32_t s_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)S);
    const int nitems = blockIdx.x < (unsigned)nblk ? (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int T = nitems * nchunk;
    int ti = 0, ich = 0;
    size_t iblk = blockIdx.x;
    auto issue_next = [&]() {
        if (ti >= T) return;
        const size_t o = iblk * a.len_ltf + LS_CP + 4 * lane;
        const uint32_t d = s_off + (uint32_t)((((ti % NSTG) * CH + wave) * 2) * LS_FFT * sizeof(float));
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int sidx = min(ich * CH + wave + NW * u, nt - 1);
            ls_dma16(a.ltf_re + o + (size_t)sidx * LS_SYM, d + u * NW * 2 * LS_FFT * sizeof(float));
            ls_dma16(a.ltf_im + o + (size_t)sidx * LS_SYM, d + (u * NW * 2 + 1) * LS_FFT * sizeof(float));
        }
        ++ti;
        if (++ich == nchunk) { ich = 0; iblk += gridDim.x; }
    };
#pragma unroll
    for (int k = 0; k < NSTG; ++k) issue_next();

    f32x16 acc[QW][JT][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[qi][jt][0][e] = 0.f; acc[qi][jt][1][e] = 0.f; }
    };
    // scale and store: rows j = jt*32 + (r&3) + 8*(r>>2) + 4*hi, bins coalesced over the lanes (as in the chunked kernel)
    auto store_item = [&](size_t blk) {
        const bool full = (nt & 31) == 0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            if (!qok[qi] || (a.dbg & 4)) continue;
            const size_t o = (blk * nt + 4 * hi) * LS_NDATA + (size_t)((wave + NW * qi) * 32 + l31);
            float* pre = a.h_re + o;
            float* pim = a.h_im + o;
            const float inv = rden[qi];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt * 32 >= nt) break;
                if (full || (jt + 1) * 32 <= nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        pre[jo * LS_NDATA] = acc[qi][jt][0][r] * inv;
                        pim[jo * LS_NDATA] = acc[qi][jt][1][r] * inv;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        if (jo + 4 * hi < nt) {
                            pre[jo * LS_NDATA] = acc[qi][jt][0][r] * inv;
                            pim[jo * LS_NDATA] = acc[qi][jt][1][r] * inv;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zero_acc();
    };
    zero_acc();

    int t = 0;
    for (size_t blk = blockIdx.x; blk < (size_t)nblk; blk += gridDim.x) {
#pragma unroll 1
        for (int ch = 0; ch < nchunk; ++ch, ++t) {
            const int ns = min(CH, nt - ch * CH);                 // symbols in this chunk
            const int younger = ti - t - 1;
            if (NSTG == 1 || younger <= 0) ls_wait_vm<0>();
            else if (NSTG == 2 || younger == 1) ls_wait_vm<R>();
            else if (NSTG == 3 || younger == 2) ls_wait_vm<2 * R>();
            else ls_wait_vm<3 * R>();
            f32x2 y0[SPW][4];
            lsc_stage0_read<SPW, NW>(S + (size_t)(((t % NSTG) * CH + wave) * 2) * LS_FFT, rev3, y0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_next();
            if (ch == 0 && t > 0) store_item(blk - gridDim.x);
            if (t > 0) ls_lds_barrier();          // spectra of chunk t - 1 consumed
            lsc_stage0_write<SPW, NW>(Fc, wave, lane, y0);
            if (!(a.dbg & 1)) lsc_fft_rows<SPW, NW, (JT == 1)>(Fc, wave, twc, lane);
            ls_lds_barrier();                     // spectra complete

            // ---- despread on the matrix core (operands of step ks + 1 requested before the MFMAs of step ks)
            const int ksteps = (a.dbg & 2) ? 0 : (ns + 1) >> 1;
            const float* prow = Pl + (size_t)l31 * ldl + ch * CH + hi;
            const f32x2* frow = Fc + (size_t)hi * LSC_ROW;
            float pvn[JT];
            f32x2 bn[QW];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) pvn[jt] = prow[(size_t)jt * 32 * ldl];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) bn[qi] = frow[pos[qi]];
            for (int ks = 0; ks < ksteps; ++ks) {
                float pv[JT];
                f32x2 bc[QW];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) pv[jt] = pvn[jt];
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) bc[qi] = bn[qi];
                const int kn = min(ks + 1, CH / 2 - 1);
                const f32x2* fn = frow + (size_t)kn * 2 * LSC_ROW;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) pvn[jt] = prow[(size_t)jt * 32 * ldl + 2 * kn];
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) bn[qi] = fn[pos[qi]];
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        acc[qi][jt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[jt], bc[qi][0], acc[qi][jt][0], 0, 0, 0);
                        acc[qi][jt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[jt], bc[qi][1], acc[qi][jt][1], 0, 0, 0);
                    }
            }
        }
    }
    if (nitems > 0) store_item(blockIdx.x + (size_t)(nitems - 1) * gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Generic-P kernel, bf16-split despread: the ring kernel above with the despread moved from v_mfma_f32_32x32x2_f32
// (64 cycles per 2 symbols) to v_mfma_f32_32x32x16_bf16 (32 cycles per 16 symbols).  Every fp32 value is cut EXACTLY
// into three bf16 pieces (8 significand bits each: x = b1 + b2 + b3, truncation, each residual is exact), bf16 has
// fp32's exponent range - no scaling, no range guard - and the products kept are the pairs (P piece i, F piece j)
// with i + j <= 4: what is dropped is below 2^-24 of |P||F|, accumulation is fp32 in the matrix core.
// NPP = how many pieces the pilot matrix needs (found on the host when it is set): 1 for +-1 / small-integer
// pilots (3 MFMAs per product), 2 for 16-bit entries (5), 3 for arbitrary floats (6): 6, 10 and 12 cycles per symbol
// and 32x32 tile against 32.  A chunk is one K = 16 MFMA step.  The P pieces of a chunk - [piece][antenna tile] blocks
// of 1 KiB, cut and laid out on the host in MFMA operand order (ls_pilot_pieces_layout) - ride the ring with the samples:
// one more LDS-DMA per block and chunk, from L2, into a ring of NSTG + 1 slots (the slot refilled during chunk t held the
// pieces of chunk t - 1, and the refill is issued behind the barrier that ends that chunk's despread).  The spectra are
// gathered per bin from the fp32 image and cut in registers.
typedef __bf16 ls_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ls_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ls_bf_top(float x) { return __builtin_bit_cast(uint32_t, x) & 0xffff0000u; }
// the three bf16 pieces of two floats, packed (low half = x0)
__device__ __forceinline__ void ls_bf_split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t a0 = ls_bf_top(x0), a1 = ls_bf_top(x1);
    const float r0 = x0 - __builtin_bit_cast(float, a0), r1 = x1 - __builtin_bit_cast(float, a1);
    const uint32_t b0 = ls_bf_top(r0), b1 = ls_bf_top(r1);
    const float q0 = r0 - __builtin_bit_cast(float, b0), q1 = r1 - __builtin_bit_cast(float, b1);
    p1 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, q1), __builtin_bit_cast(uint32_t, q0), 0x07060302u);
}

constexpr int LSB_BLOCK = 512;      // bf16 elements of one (chunk, piece, antenna tile) block: [2 k-halves][32 rows][8 symbols]

template <int JT, int NW, int NSTG, int NPP, int MINB = 1, bool DBF = false, int VAR = CSI_LS_VAR_DEFAULT>
__global__ __launch_bounds__(64 * NW, MINB) void ls_estimate_ringb_kernel(const LsArgs a, int nblk) {
    constexpr int CH = 16, SPW = CH / NW, QW = 8 / NW;
    constexpr int NB = NPP * JT, NPD = (NB + NW - 1) / NW;      // P blocks per chunk, LDS-DMAs per wave for them
    constexpr int R = 2 * SPW + NPD, NPS = NSTG + 1;
    static_assert(SPW >= 1 && QW >= 1 && NSTG >= 1 && NSTG <= 4 && (NSTG - 1) * R <= 63 && NPP >= 1 && NPP <= 3, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x2* twc = reinterpret_cast<f32x2*>(smem);                       // [LSC_NTW]
    f32x2* Fc = twc + LSC_NTW;                                         // [CH][LSC_ROW]
    float* S = reinterpret_cast<float*>(Fc + (DBF ? 2 : 1) * CH * LSC_ROW);   // [NSTG][CH][2][256]
    uint16_t* Pb = reinterpret_cast<uint16_t*>(S + NSTG * CH * 2 * LS_FFT);   // [NPS][NB][LSB_BLOCK] bf16
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = a.nt;
    const int nchunk = (nt + CH - 1) / CH;
    const int rev3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);
    if (a.dbg & 256) {        // race hunt (tools/ls_race_repro.py): the whole LDS of the workgroup starts as NaN - any read of a location this
                              // workgroup has not written yet turns into NaN results instead of plausible leftovers
        constexpr int NF = 2 * LSC_NTW + (DBF ? 2 : 1) * CH * 2 * LSC_ROW + NSTG * CH * 2 * LS_FFT + (NSTG + 1) * NPP * JT * LSB_BLOCK / 2;
        for (int i = tid; i < NF; i += 64 * NW) smem[i] = __builtin_bit_cast(float, 0x7fc00000u);
        __syncthreads();
    }
    lsc_build_twiddles(twc, a.tw, tid, 64 * NW);
    int pos[QW];
    float rden[QW];
    bool qok[QW];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        const int q = (wave + NW * qi) * 32 + l31;
        qok[qi] = q < LS_NDATA;
        pos[qi] = lsc_phys(a.bin_pos[qok[qi] ? q : 0]);
        rden[qi] = 1.0f / a.denom[qok[qi] ? q : 0];
    }
    __syncthreads();

    const uint32_t s_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)S);
    const uint32_t p_off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)Pb);
    const int nitems = blockIdx.x < (unsigned)nblk ? (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int T = nitems * nchunk;
    int ti = 0, ich = 0;
    size_t iblk = blockIdx.x;
    auto issue_next = [&]() {                   // the samples of chunk ti
        if (ti >= T) return;
        const size_t o = iblk * a.len_ltf + LS_CP + 4 * lane;
        const uint32_t d = s_off + (uint32_t)((((ti % NSTG) * CH + wave) * 2) * LS_FFT * sizeof(float));
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
            const int sidx = min(ich * CH + wave + NW * u, nt - 1);
            ls_dma16(a.ltf_re + o + (size_t)sidx * LS_SYM, d + u * NW * 2 * LS_FFT * sizeof(float));
            ls_dma16(a.ltf_im + o + (size_t)sidx * LS_SYM, d + (u * NW * 2 + 1) * LS_FFT * sizeof(float));
        }
        ++ti;
        if (++ich == nchunk) { ich = 0; iblk += gridDim.x; }
    };
    int tp = 0, ichp = 0;
    auto issue_pieces = [&]() {                 // the P pieces of chunk tp: block x by wave x mod NW (the last one again where NB is no multiple of NW)
        if (tp >= T) return;
        const float* src = reinterpret_cast<const float*>(a.Pbf + (size_t)ichp * 3 * JT * LSB_BLOCK) + 4 * lane;
        const uint32_t d = p_off + (uint32_t)((tp % NPS) * NB * LSB_BLOCK * 2);
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int x = min(wave + NW * u, NB - 1);
            ls_dma16(src + (size_t)x * (LSB_BLOCK / 2), d + (uint32_t)x * LSB_BLOCK * 2);
        }
        ++tp;
        if (++ichp == nchunk) ichp = 0;
    };
#pragma unroll
    for (int k = 0; k < NSTG; ++k) { issue_next(); issue_pieces(); }

    f32x16 acc[QW][JT][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[qi][jt][0][e] = 0.f; acc[qi][jt][1][e] = 0.f; }
    };
    // rows j = jt*32 + (r&3) + 8*(r>>2) + 4*hi, bins coalesced over the lanes (the accumulator layout of the 32x32 MFMAs)
    auto store_item = [&](size_t blk) {
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            if (!qok[qi] || (a.dbg & 4)) continue;
            const size_t o = (blk * nt + 4 * hi) * LS_NDATA + (size_t)((wave + NW * qi) * 32 + l31);
            float* pre = a.h_re + o;
            float* pim = a.h_im + o;
            const float inv = rden[qi];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt * 32 >= nt) break;
                if ((jt + 1) * 32 <= nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        pre[jo * LS_NDATA] = acc[qi][jt][0][r] * inv;
                        pim[jo * LS_NDATA] = acc[qi][jt][1][r] * inv;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jo = jt * 32 + (r & 3) + 8 * (r >> 2);
                        if (jo + 4 * hi < nt) {
                            pre[jo * LS_NDATA] = acc[qi][jt][0][r] * inv;
                            pim[jo * LS_NDATA] = acc[qi][jt][1][r] * inv;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zero_acc();
    };
    zero_acc();

    int t = 0;
    for (size_t blk = blockIdx.x; blk < (size_t)nblk; blk += gridDim.x) {
#pragma unroll 1
        for (int ch = 0; ch < nchunk; ++ch, ++t) {
            // the DMAs of this wave for chunk t (its sample rows, its P blocks) have landed?  Younger chunks: R each.
            const int younger = ti - t - 1;
            if (NSTG == 1 || younger <= 0) ls_wait_vm<0>();
            else if (NSTG == 2 || younger == 1) ls_wait_vm<R>();
            else if (NSTG == 3 || younger == 2) ls_wait_vm<2 * R>();
            else ls_wait_vm<3 * R>();
            f32x2 y0[SPW][4];
            lsc_stage0_read<SPW, NW, VAR>(S + (size_t)(((t % NSTG) * CH + wave) * 2) * LS_FFT, rev3, y0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_next();
            if (ch == 0 && t > 0) store_item(blk - gridDim.x);
            // DBF: two spectra images alternate - the image written now was last read two chunks ago, and every wave had finished that
            // despread before it arrived at the previous "spectra complete" barrier: one barrier per chunk instead of two, a fast wave
            // transforms chunk t + 1 while a slow one still despreads chunk t
            f32x2* Fb = Fc + (DBF ? (t & 1) * CH * LSC_ROW : 0);
            if (!DBF) {
                if (t > 0) ls_lds_barrier();      // spectra and P pieces of chunk t - 1 consumed
                issue_pieces();                   // ... so the slot of those pieces takes chunk t + NSTG
            }
            lsc_stage0_write<SPW, NW, VAR>(Fb, wave, lane, y0);
            if (!(a.dbg & 1)) lsc_fft_rows<SPW, NW, (JT == 1), VAR>(Fb, wave, twc, lane);
            if (VAR & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // race hunt: no LDS-DMA of this wave in flight across the barrier
            ls_lds_barrier();                     // spectra complete; every wave has seen its P blocks of chunk t land
            if (DBF) issue_pieces();              // every wave is past the despread of chunk t - 1: its P slot takes chunk t + NSTG

            // ---- despread: one K = 16 step.  A = P pieces (row j = l31 of antenna tile jt, symbols 8 hi .. 8 hi + 7),
            // B = this lane's bin of the same 8 symbols, cut into pieces here.
            if (a.dbg & 2) continue;
            const uint16_t* pbs = Pb + (size_t)(t % NPS) * NB * LSB_BLOCK + lane * 8;
            ls_bf16x8 pa[JT][NPP];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int k = 0; k < NPP; ++k)
                    pa[jt][k] = __builtin_bit_cast(ls_bf16x8, *reinterpret_cast<const ls_u32x4*>(pbs + (k * JT + jt) * LSB_BLOCK));
            const f32x2* frow = Fb + (size_t)(8 * hi) * LSC_ROW;
            ls_u32x4 fb[QW][2][3];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                f32x2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = frow[(size_t)i * LSC_ROW + pos[qi]];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint32_t p1, p2, p3;
                        ls_bf_split2(v[2 * i][c], v[2 * i + 1][c], p1, p2, p3);
                        fb[qi][c][0][i] = p1; fb[qi][c][1][i] = p2; fb[qi][c][2][i] = p3;
                    }
            }
            // every operand is in registers before the first MFMA; none of these registers may be written again until the
            // MFMAs that read them have left the matrix pipe (see below)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        f32x16 d = acc[qi][jt][c];
                        // small terms first
                        if (NPP >= 3) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][NPP >= 3 ? 2 : 0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][0]), d, 0, 0, 0);
                        if (NPP >= 2) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][NPP >= 2 ? 1 : 0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][1]), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][2]), d, 0, 0, 0);
                        if (NPP >= 2) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][NPP >= 2 ? 1 : 0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][0]), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][1]), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[jt][0], __builtin_bit_cast(ls_bf16x8, fb[qi][c][0]), d, 0, 0, 0);
                        acc[qi][jt][c] = d;
                    }
            // No drain is needed here.  History: the FIRST version of this kernel (round 3; operand loads interleaved with the MFMA
            // chains, never committed) produced 1-5 wrong items per 4000 with 8 waves per workgroup, and a "drain" - a dependent read
            // of every accumulator, then a use of every operand register - went in together with the restructuring above (every
            // operand in registers before the first MFMA) on the theory that LDS reads issued right behind the last MFMA landed in
            // its A / B source registers before the MFMA had read them.  Round 4 tested that theory on the hardware
            // (tools/mfma_war_probe.hip, profiles/r04_mfma_war_probe.txt): chains of 1 / 6 / 12 MFMAs followed after 0 ... 64 wait
            // states by ds_read_b128 OR v_mov_b32 into their A and / or B registers, the SIMD's other wave idle or issuing MFMAs back
            // to back - 36 variants x 1.3e8 accumulator values, not one wrong.  gfx950 interlocks an in-flight MFMA's sources against
            // both writers; that hazard does not exist, so the drain was not what cured the first version - the restructuring was
            // (with operands fetched between the MFMAs of a chain a slow wave could still be reading the spectra image / P slot of
            // chunk t after a fast wave had passed the next barrier and started to overwrite them: the reads must be retired before
            // the wave's own arrival at that barrier, which "all operands first" guarantees by construction).  The kernel as it stands is
            // bit-for-bit reproducible without the drain: 72 configurations x 12 runs, every item compared
            // (profiles/r04_ls_generic_stress_nodrain.txt; tests/stress_ls_generic.py, bounded form in tests/test_gpu_*.py).
            // ls_debug 64 puts the drain back for A/B runs.
            // End of round 4 (profiles/r04_ls_ringb_variants.txt): what IS seen, rarely and on some boxes only, with the
            // two-workgroups-per-CU instantiation <1, 4, 1, NPP, 2> (no longer selected: ls_ringb_min) is not a race at all - every bad
            // item is the result of ONE v_pk_add_f32 with op_sel (pk_add_mi / pk_add_pi of a transform stage) wrong in lanes 48-63,
            // the first launch after another kernel, where a wave of the CU's OTHER workgroup runs these MFMAs on the same SIMD.
            // Round 6 closed it: that instruction form loses its swapped-in operand under exactly that condition; the rotations are
            // single adds in every instantiation now.
            if (a.dbg & 64) {
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float tmp;
                            asm volatile("v_mov_b32 %0, %1" : "=v"(tmp) : "v"(acc[qi][jt][c][15]));
                        }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int k = 0; k < NPP; ++k) asm volatile("" ::"v"(pa[jt][k]));
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int k = 0; k < 3; ++k) asm volatile("" ::"v"(fb[qi][c][k]));
            }
        }
    }
    if (nitems > 0) store_item(blockIdx.x + (size_t)(nitems - 1) * gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Large-Nt variant (Nt > 64, where Nt spectra no longer fit the 160 KiB LDS): by linearity the
// despread is done FIRST, in the time domain, and only the despread rows are transformed:
//   Y[j][n] = sum_s P[j][s] x[s][64+n]          H[j][q] = FFT(Y[j])[f(q)] / (Nt ltf[q])
// One workgroup = one (packet, rx, chunk of 32 tx antennas).  The symbols stream through LDS in
// chunks of 32 (B operand of v_mfma_f32_32x32x2_f32, A operand = the 32x32 block of P), the 32
// despread rows stay in the accumulators, are then scattered (digit-reversed) into the same LDS
// buffer, transformed by ls_fft256_wave and written out.  The input of a (packet, rx) is read by
// Nt/32 workgroups (L2 / Infinity Cache absorb the re-reads); LDS = 64 KiB + tables.
constexpr int LSD_ROWS = 32;

__global__ __launch_bounds__(LS_THREADS) void ls_despread_first_kernel(const LsArgs a, int n_jc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tw_re = smem;
    float* tw_im = smem + LS_FFT;
    float* X = smem + 2 * LS_FFT;             // [32][2][LS_PLANE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = a.nt;
    const size_t blk = blockIdx.x / n_jc;
    const int jc = blockIdx.x % n_jc;

    tw_re[tid] = a.tw[tid];
    tw_im[tid] = a.tw[LS_FFT + tid];

    // accumulators: wave w owns sample tiles {w, w+4} of both planes
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][p][e] = 0.f;

    const int ja = jc * LSD_ROWS + l31;                   // A-operand row (tx antenna) of this lane
    const float* gre = a.ltf_re + blk * a.len_ltf + LS_CP + 4 * lane;
    const float* gim = a.ltf_im + blk * a.len_ltf + LS_CP + 4 * lane;
    for (int s0 = 0; s0 < nt; s0 += LSD_ROWS) {
        __syncthreads();                                   // previous chunk fully consumed
        // load 32 symbols (natural sample order), wave w takes rows w, w+4, ...
#pragma unroll
        for (int u = 0; u < LSD_ROWS / 4; ++u) {
            const int r = wave + 4 * u;
            const int s = s0 + r;
            f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
            if (s < nt) {
                vr = *reinterpret_cast<const f32x4*>(gre + (size_t)s * LS_SYM);
                vi = *reinterpret_cast<const f32x4*>(gim + (size_t)s * LS_SYM);
            }
            float* xr = X + (size_t)r * 2 * LS_PLANE;
            float* xi = xr + LS_PLANE;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p = ls_phys(4 * lane + c);
                xr[p] = vr[c];
                xi[p] = vi[c];
            }
        }
        __syncthreads();
        for (int ks = 0; ks < LSD_ROWS / 2; ++ks) {
            const int r = 2 * ks + hi;
            const int s = s0 + r;
            const float pv = (ja < nt && s < nt) ? a.P[ja * nt + s] : 0.f;
            const float* xr = X + (size_t)r * 2 * LS_PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = ls_phys((wave + 4 * i) * 32 + l31);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, xr[p], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, xr[LS_PLANE + p], acc[i][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // scatter Y[j][n] (C/D layout: col = lane&31 <-> n, row <-> j) to digit-reversed positions
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = (wave + 4 * i) * 32 + l31;
        const int rev = ((n & 3) << 6) | (((n >> 2) & 3) << 4) | (((n >> 4) & 3) << 2) | (n >> 6);
        const int p = ls_phys(rev);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float* yr = X + (size_t)j * 2 * LS_PLANE;
            yr[p] = acc[i][0][r];
            yr[LS_PLANE + p] = acc[i][1][r];
        }
    }
    __syncthreads();
    ls_fft_rows(X, wave, LSD_ROWS, tw_re, tw_im, lane);
    __syncthreads();
    // pick the 234 data bins, scale, store coalesced
    for (int idx = tid; idx < LSD_ROWS * LS_NDATA; idx += LS_THREADS) {
        const int j = idx / LS_NDATA;
        const int q = idx - j * LS_NDATA;
        const int jt = jc * LSD_ROWS + j;
        if (jt < nt) {
            const int p = ls_phys(a.bin_pos[q]);
            const float den = a.denom[q];
            const float* fr = X + (size_t)j * 2 * LS_PLANE;
            const size_t o = (blk * nt + jt) * LS_NDATA + q;
            a.h_re[o] = fr[p] / den;
            a.h_im[o] = fr[LS_PLANE + p] / den;
        }
    }
}

// i.i.d. CN(0,1) samples from a counter-based generator: element index -> splitmix64 ->
// two uniforms -> Box-Muller.  re/im each have variance 1/2.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void synth_white_kernel(uint64_t seed, uint64_t first_elem, size_t n, float* __restrict__ re,
                                   float* __restrict__ im) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = splitmix64(seed ^ splitmix64(first_elem + i));
        const float u1 = ((float)(uint32_t)(h >> 32) + 0.5f) * (1.0f / 4294967296.0f);
        const float u2 = ((float)(uint32_t)h + 0.5f) * (1.0f / 4294967296.0f);
        const float rad = sqrtf(-logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        re[i] = rad * cs;
        im[i] = rad * sn;
    }
}

}  // namespace csi
