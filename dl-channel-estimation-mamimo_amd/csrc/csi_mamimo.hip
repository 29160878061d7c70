// csi_mamimo.hip - C-ABI (include/csi_mamimo.h) and host-side orchestration of the MI355X
// channel-estimation hot path.  gfx950 only; built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC csi_mamimo.hip -o libcsi_mamimo.so
// (host code for the baseline x86-64 ISA; the AVX2 staging loops of csi_hostpipe.hpp carry their own target attribute), after
// band_kernel_gen.py -> clang -x assembler -mcpu=gfx950 -> ld.lld -> band8_hsaco.inc (_lib.build_library does all of it)
//
// What runs where (reference call sites in include/csi_mamimo.h):
//   csi_predict*        layer 0 once per (packet, rx)  -> gemm_f32_kernel<EPI_RAW> (optional split-K)
//                       (+ splitk_reduce_kernel)          then per pair (packet, rx, tx):
//                       hidden 1.. and regressor       -> pair_gemm_f32_kernel / gemm_f32_kernel
//   csi_predict_samples literal un-shared network      -> gemm_f32_kernel
//   csi_ls_estimate*    FFT + despread                 -> ls_estimate_kernel
#include "csi_context.hpp"
#include "csi_dnn_hs.hpp"
#include "csi_dnn_f32.hpp"
#include "csi_dnn_small.hpp"
#include "csi_dnn_bf16.hpp"
#include "csi_train.hpp"
#include "csi_hostpipe.hpp"
#include "csi_comm.hpp"

namespace {

// Which LS kernel serves this context.  With the Sylvester Hadamard pilot matrix the Walsh-Hadamard kernel on the
// LDS-DMA ring.  Any other P: FFT-first (all Nt spectra in LDS) up to ls_fft_first_max antennas, the ring kernels with
// the matrix-core despread up to Nt = 128 - bf16-split (ls_estimate_ringb_kernel) except for a pilot matrix of arbitrary
// floats at Nt <= 32, where the fp32 despread of ls_estimate_ring_kernel is as fast (profiles/r03_ls_probe_generic.txt) -
// the despread-first kernel beyond.  The older chunked
// kernel stays selectable through the "ls_kernel" option (tests, A/B runs).
enum LsMode { LS_AUTO = 0, LS_FFT_FIRST = 1, LS_CHUNKED = 2, LS_DESPREAD_FIRST = 3, LS_FWHT = 4, LS_FWHT2 = 5, LS_RING = 6, LS_RINGB = 7 };
struct LsPlan {
    int mode;
    const void* fn;
    size_t lds;
    int threads;
    int per_cu;           // resident workgroups per CU (persistent grids)
};
// the bf16-split ring kernel (ls_estimate_ringb_kernel) for this Nt / pilot: its shape and LDS need; fn == nullptr when Nt is outside 16 ... 128
struct LsRingB { const void* fn; size_t lds; int nw, per_cu; };
LsRingB ls_ringb_shape(const csi_ctx* c) {
    const int nt = c->cfg.nt, jt = (nt + 31) / 32, npp = std::min(3, std::max(1, c->p_pieces));
    LsRingB r{nullptr, 0, 8, 1};
    if (nt < 16 || nt > 128) return r;
    int nstg = 1, nf = 1;
#define LS_RB(J, W, NS, MB, DB)                                                                                    \
    {                                                                                                              \
        r.fn = npp == 1 ? (const void*)ls_estimate_ringb_kernel<J, W, NS, 1, MB, DB>                               \
               : (npp == 2 ? (const void*)ls_estimate_ringb_kernel<J, W, NS, 2, MB, DB> : (const void*)ls_estimate_ringb_kernel<J, W, NS, 3, MB, DB>); \
        r.nw = W; nstg = NS; r.per_cu = MB; nf = DB ? 2 : 1;                                                       \
    }
    // shapes as measured (profiles/r03_ls_probe_generic.txt); "ls_v2" = 1 selects the runner-up for A/B runs
    // one antenna tile (Nt <= 32).  Round 5: the product build runs this shape ONE workgroup per CU.  Its two-workgroups-per-CU form
    // (round 3's choice, 4 % faster) showed rare wrong first items on some parts of the pool (profiles/r04_ls_ringb_variants.txt: lanes
    // 48-63 of one packed add beside ANOTHER workgroup's bf16 MFMAs on the same SIMD; never with one workgroup per CU, whose barriers
    // keep the two waves of a SIMD in the same phase) and was never root-caused, so the shipped library cannot select it by any option:
    // it is compiled only into the hunt build (CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS).  The shape is padded to > 80 KiB of LDS below so
    // that the dispatcher cannot co-locate two of these workgroups either.
#ifdef CSI_LS_RACE_VARIANTS
    if (jt == 1) { if (c->ls_v2 == 1) LS_RB(1, 4, 2, 2, false) else LS_RB(1, 4, 1, 2, false) }
#else
    if (jt == 1) { if (c->ls_v2 == 1) LS_RB(1, 4, 2, 1, false) else LS_RB(1, 4, 1, 1, false) }
#endif
    else if (jt == 2) { if (c->ls_v2 == 1) LS_RB(2, 8, 1, 1, false) else LS_RB(2, 8, 1, 1, true) }
    else if (jt == 3) { if (c->ls_v2 == 1) LS_RB(3, 8, 2, 1, false) else LS_RB(3, 8, 1, 1, true) }
    else { if (c->ls_v2 == 1) LS_RB(4, 8, 1, 1, false) else LS_RB(4, 8, 1, 1, true) }
    // race hunt (tools/ls_race_fast.py): ls_debug bits 0x200 ... 0x8000 select the VAR 1 ... 64 forms (and a few sums) of the
    // two-workgroups-per-CU instantiation (ls_estimate.hip.h, lsc_stage0_write); pilots of one or two pieces only
#ifdef CSI_LS_RACE_VARIANTS       // build with CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS (python -c "import __graft_entry__ as g; g.build()")
    if (jt == 1 && c->ls_v2 != 1 && npp <= 2 && (c->ls_debug & 0xfe00)) {
        const int var = (c->ls_debug >> 9) & 127;
#define LS_RBV(V) if (var == V) r.fn = npp == 1 ? (const void*)ls_estimate_ringb_kernel<1, 4, 1, 1, 2, false, V> : (const void*)ls_estimate_ringb_kernel<1, 4, 1, 2, 2, false, V>;
        LS_RBV(1) LS_RBV(2) LS_RBV(4) LS_RBV(8) LS_RBV(5) LS_RBV(6) LS_RBV(12) LS_RBV(16) LS_RBV(20) LS_RBV(32) LS_RBV(36) LS_RBV(64) LS_RBV(68) LS_RBV(96) LS_RBV(48)
#undef LS_RBV
    }
#endif
#undef LS_RB
    r.lds = (size_t)(2 * LSC_NTW + nf * 16 * 2 * LSC_ROW + nstg * 16 * 2 * LS_FFT) * sizeof(float) + (size_t)(nstg + 1) * npp * jt * LSB_BLOCK * 2;
    if (r.lds > 160 * 1024) r.fn = nullptr;
#ifndef CSI_LS_RACE_VARIANTS
    if (jt == 1) r.lds = std::max(r.lds, (size_t)(81 * 1024));      // one workgroup per CU by construction (see above)
#endif
    r.per_cu = std::max(1, std::min(r.per_cu, (int)((160 * 1024) / r.lds)));
    return r;
}
LsPlan ls_plan(const csi_ctx* c) {
    const int nt = c->cfg.nt;
    int mode = c->ls_kernel;
    // Walsh-Hadamard despread: the Sylvester matrix itself, or (round 4) any signed row / column permutation of it - the kernel
    // then fetches the symbols and stores the antennas through the tables csi_set_pilot derived (PERM form)
    const bool perm = c->p_fast_ok && !c->p_fast_identity;
    const bool fwht_ok = (c->p_sylvester || (c->p_fast_ok && (c->p_fast_identity || c->ls_fast_perm))) && (nt == 16 || nt == 32 || nt == 64 || nt == 128);
    if ((mode == LS_FWHT || mode == LS_FWHT2) && !fwht_ok) mode = LS_AUTO;
    if (mode == LS_FWHT && perm) mode = LS_FWHT2;            // the round-1 kernel knows the Sylvester order only
    const LsRingB rb = ls_ringb_shape(c);
    if (mode == LS_RINGB && !rb.fn) mode = LS_AUTO;
    if (mode == LS_AUTO)
        mode = fwht_ok ? LS_FWHT2 : (nt <= c->ls_fft_first_max ? LS_FFT_FIRST : (nt <= 128 ? (rb.fn && nt >= c->ls_ringb_min && (c->p_pieces < 3 || nt > 32) ? LS_RINGB : LS_RING) : LS_DESPREAD_FIRST));
    if (mode == LS_FFT_FIRST && nt > 64) mode = LS_CHUNKED;
    if ((mode == LS_CHUNKED || mode == LS_RING) && (nt < 16 || nt > 128)) mode = nt < 16 ? LS_FFT_FIRST : LS_DESPREAD_FIRST;
    LsPlan p{};
    p.mode = mode;
    if (mode == LS_RINGB) {
        p.fn = rb.fn; p.lds = rb.lds; p.threads = 64 * rb.nw; p.per_cu = rb.per_cu;
        return p;
    }
    if (mode == LS_FWHT2) {
        // shape per Nt as measured (profiles/r02_ls_probe.txt); "ls_v2" = 1 selects the runner-up for A/B runs
        const int v = c->ls_v2;
        int split = 1, ch = 16, nstg = 1, nf = 1, maxcu = 2;
#define LS_V2(NTV, SP, CHV, NS, DB) { p.fn = (const void*)ls_estimate_fwht2_kernel<NTV, SP, CHV, NS, DB>; split = SP; ch = CHV; nstg = NS; nf = DB ? 2 : 1; }
        if (nt == 16) {
            if (v == 1) LS_V2(16, 1, 16, 1, false)
            else if (v == 3) { p.fn = (const void*)ls_estimate_fwht2_kernel<16, 1, 8, 1, false, 4>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }      // A/B: vector-address stores (round 3)
            else { p.fn = (const void*)ls_estimate_fwht2_kernel<16, 1, 8, 1, false, 4, false, true>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }
        }
        else if (nt == 32) {       // 8-symbol chunks, one slot: 38 KiB of LDS and 122 VGPRs - four workgroups per CU (0.379 ms; two with 16-symbol chunks: 0.402)
            if (v == 1) LS_V2(32, 1, 16, 1, false)
            else if (v == 3) { p.fn = (const void*)ls_estimate_fwht2_kernel<32, 1, 8, 1, false, 4>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }   // A/B: vector-address stores (round 3)
            // round 4: stores with the row base in scalar registers (SST): -3 ... -5 % at Nt = 32 / 64 (profiles/r04_ls_probe.txt)
            else { p.fn = (const void*)ls_estimate_fwht2_kernel<32, 1, 8, 1, false, 4, false, true>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }
        }
        else if (nt == 64) {
            if (v == 1) LS_V2(64, 1, 16, 1, false)
            else if (v == 3) LS_V2(64, 1, 8, 3, false)
            else { p.fn = (const void*)ls_estimate_fwht2_kernel<64, 1, 8, 3, false, 2, false, true>; split = 1; ch = 8; nstg = 3; nf = 1; }
        }
        else {
            if (v == 1) LS_V2(128, 2, 16, 3, false)
            else if (v == 3) LS_V2(128, 2, 16, 2, true)       // two spectra images: -6 %; vector-address stores
            else { p.fn = (const void*)ls_estimate_fwht2_kernel<128, 2, 16, 2, true, 1, false, true>; split = 2; ch = 16; nstg = 2; nf = 2; }    // + scalar-base stores: -1 ... -1.9 %
        }
#undef LS_V2
        if (perm) {        // same shapes as the defaults above, table-driven symbol fetch / antenna store
            if (nt == 16) { p.fn = (const void*)ls_estimate_fwht2_kernel<16, 1, 8, 1, false, 4, true>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }
            else if (nt == 32) { p.fn = (const void*)ls_estimate_fwht2_kernel<32, 1, 8, 1, false, 4, true>; split = 1; ch = 8; nstg = 1; nf = 1; maxcu = 4; }
            else if (nt == 64) { p.fn = (const void*)ls_estimate_fwht2_kernel<64, 1, 8, 3, false, 2, true>; split = 1; ch = 8; nstg = 3; nf = 1; maxcu = 2; }
            else { p.fn = (const void*)ls_estimate_fwht2_kernel<128, 2, 16, 2, true, 1, true>; split = 2; ch = 16; nstg = 2; nf = 2; maxcu = 2; }
        }
        p.lds = (size_t)(2 * LSC_NTW + nf * ch * 2 * LSC_ROW + nstg * ch * 2 * LS_FFT) * sizeof(float);
        p.threads = 256 * split;
        p.lds += (size_t)c->debug_ls_lds_pad;              // CSI_DEBUG_HOOKS=1 CSI_LS_LDS_PAD=<bytes>: unused LDS, i.e. fewer workgroups per CU (A/B runs)
        p.per_cu = std::max(1, std::min(split == 1 ? maxcu : 1, (int)((160 * 1024) / p.lds)));
    } else if (mode == LS_RING) {
        const int jt = (nt + 31) / 32, ldp = jt * 32;
        int nw = 4, nstg = 1, chs = 16, ringcu = 2;
#define LS_RING_K(J, W, C, NS) { p.fn = (const void*)ls_estimate_ring_kernel<J, W, C, NS>; nw = W; nstg = NS; chs = C; }
        // 8-symbol chunks where they waste fewer padded symbols (Nt = 24, 40, ...) and for two antenna tiles, where they let
        // two workgroups share a CU (measured: profiles/r02_ls_probe.txt); "ls_v2" = 1 flips the choice for A/B runs
        bool ch8 = jt == 2 || (jt == 1 && (nt + 7) / 8 * 8 < (nt + 15) / 16 * 16);
        if (c->ls_v2 == 1) ch8 = !ch8;
        // one antenna tile: 8-symbol chunks and one slot leave room for three workgroups per CU (Nt = 32: 0.438 against 0.470 ms)
        if (jt == 1 && c->ls_v2 == 0) { p.fn = (const void*)ls_estimate_ring_kernel<1, 4, 8, 1, 3>; nw = 4; nstg = 1; chs = 8; ringcu = 3; }
        else if (jt == 1) { if (c->ls_v2 == 2) LS_RING_K(1, 4, 8, 3) else LS_RING_K(1, 4, 16, 1) }
        else if (jt == 2) { if (ch8) LS_RING_K(2, 4, 8, 2) else LS_RING_K(2, 8, 16, 3) }
        else if (jt == 3) LS_RING_K(3, 8, 16, 2)
        else LS_RING_K(4, 8, 16, 1)
#undef LS_RING_K
        p.lds = (size_t)(2 * LSC_NTW + chs * 2 * LSC_ROW + nstg * chs * 2 * LS_FFT + 32 * jt * (ldp + 1)) * sizeof(float);
        p.threads = 64 * nw;
        p.per_cu = std::max(1, std::min(nw == 4 ? ringcu : 1, (int)((160 * 1024) / p.lds)));
    } else if (mode == LS_FWHT) {
        p.fn = nt == 16 ? (const void*)ls_estimate_fwht_kernel<16> : nt == 32 ? (const void*)ls_estimate_fwht_kernel<32>
               : nt == 64 ? (const void*)ls_estimate_fwht_kernel<64> : (const void*)ls_estimate_fwht_kernel<128, 2>;
        p.lds = (size_t)(16 * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
        p.threads = nt == 128 ? 512 : 256;
        p.per_cu = nt == 16 ? 3 : (nt == 128 ? 1 : 2);
    } else if (mode == LS_FFT_FIRST) {
        p.fn = nt <= 32 ? (const void*)ls_estimate_kernel<8> : (const void*)ls_estimate_kernel<16>;
        p.lds = (size_t)(nt * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
        p.threads = LS_THREADS;
        p.per_cu = std::max(1, std::min(8, (int)((160 * 1024) / p.lds)));
    } else if (mode == LS_CHUNKED) {
        const int jt = (nt + 31) / 32;
        p.fn = jt == 1 ? (const void*)ls_estimate_chunked_kernel<1, 4, 16>
               : jt == 2 ? (const void*)ls_estimate_chunked_kernel<2, 4, 16>
                       : (jt == 3 ? (const void*)ls_estimate_chunked_kernel<3, 8, 32> : (const void*)ls_estimate_chunked_kernel<4, 8, 32>);
        p.lds = (size_t)((jt <= 2 ? 16 : 32) * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
        p.threads = jt <= 2 ? 256 : 512;
        p.per_cu = jt <= 2 ? 2 : 1;                       // register-limited: 2 waves per SIMD
    } else {
        p.fn = (const void*)ls_despread_first_kernel;
        p.lds = (size_t)(LSD_ROWS * 2 * LS_PLANE + 2 * LS_FFT) * sizeof(float);
        p.threads = LS_THREADS;
        p.per_cu = 2;
    }
    return p;
}
bool ls_default_fwht2(const csi_ctx* c, size_t* lds_bytes) {
    if (c->ls_v2 != 0 || c->ls_debug != 0 || c->ls_kernel != LS_AUTO || (c->p_fast_ok && !c->p_fast_identity)) return false;
    const LsPlan p = ls_plan(c);
    *lds_bytes = p.lds;
    return p.mode == LS_FWHT2 && p.threads == 256;
}
LsArgs ls_args(const csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, float* d_h_re, float* d_h_im) {
    const csi_config& cf = c->cfg;
    LsArgs a{};
    a.P = c->P; a.Ppad = c->Ppad; a.Pbf = reinterpret_cast<const uint16_t*>(c->Pbf); a.ldp = (cf.nt + 31) / 32 * 32; a.dbg = c->ls_debug;
    a.tw = c->tw; a.bin_pos = c->bin_pos; a.denom = c->denom;
    a.nt = cf.nt; a.len_ltf = cf.len_ltf;
    a.perm = c->p_tables;
    a.ltf_re = d_ltf_re; a.ltf_im = d_ltf_im; a.h_re = d_h_re; a.h_im = d_h_im;
    return a;
}
int ls_prepare(csi_ctx* c) {
    if (c->cfg.nt == 0) return CSI_OK;
    const LsPlan p = ls_plan(c);
    HIP_TRY(c, hipFuncSetAttribute(p.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    return CSI_OK;
}

// Is P a signed row / column permutation of the Sylvester Hadamard matrix H[a][u] = (-1)^popcount(a & u)?  (helperGetP of the
// reference's toolbox is un-vendored, helperMIMOChannelEstimate.m:13; the 802.11 VHT mapping matrix [1 -1 1 1; 1 1 -1 1; 1 1 1 -1;
// -1 1 1 1] doubled up recursively is of this kind without being in the Sylvester order.)  Normalise: cs[s] = P[0][s] makes the first
// row +1, then rs[j] = P[j][0] cs[0] the first column; the normalised matrix N = diag(rs) P diag(cs) of such a P is H with rows and
// columns PERMUTED only: N[j][s] = H[sigma(j)][tau(s)] (rows of H multiply like XOR of their indices).  Label log2(nt) independent
// rows of N with the unit vectors (any basis does: sigma' = A sigma, tau' = A^-T tau leave the inner product alone), read tau(s) off
// their signs in column s and sigma(j) off row j's signs in the columns whose tau is a unit vector, then VERIFY every entry.
// On success perm[0][u] = tau^-1(u) | (cs < 0 ? 256 : 0), perm[1][r] = sigma^-1(r) | (rs < 0 ? 256 : 0).
bool pilot_decompose(const float* P, int nt, int perm[2][CSI_WIRE_MAX_NT], bool* identity) {
    if (nt < 2 || nt > CSI_WIRE_MAX_NT || (nt & (nt - 1))) return false;
    for (size_t i = 0; i < (size_t)nt * nt; ++i)
        if (P[i] != 1.0f && P[i] != -1.0f) return false;
    int n = 0;
    while ((1 << n) < nt) ++n;
    std::vector<int> cs(nt), rs(nt);
    for (int s = 0; s < nt; ++s) cs[s] = P[s] < 0 ? -1 : 1;
    for (int j = 0; j < nt; ++j) rs[j] = (P[(size_t)j * nt] < 0 ? -1 : 1) * cs[0];
    typedef std::pair<uint64_t, uint64_t> bits;              // a row of N as the set of its -1 columns
    std::vector<bits> row(nt);
    for (int j = 0; j < nt; ++j) {
        bits b{0, 0};
        for (int s = 0; s < nt; ++s)
            if (P[(size_t)j * nt + s] * (float)(rs[j] * cs[s]) < 0) (s < 64 ? b.first : b.second) |= (uint64_t)1 << (s & 63);
        row[j] = b;
    }
    // greedy basis: a row outside the group generated so far extends it
    std::vector<bits> span{bits{0, 0}};
    std::vector<int> basis;
    for (int j = 0; j < nt && (int)basis.size() < n; ++j) {
        if (std::find(span.begin(), span.end(), row[j]) != span.end()) continue;
        basis.push_back(j);
        const size_t m = span.size();
        for (size_t k = 0; k < m; ++k) span.push_back(bits{span[k].first ^ row[j].first, span[k].second ^ row[j].second});
    }
    if ((int)basis.size() != n) return false;
    std::vector<int> tau(nt), sigma(nt), tau_inv(nt, -1), sigma_inv(nt, -1);
    for (int s = 0; s < nt; ++s) {
        int t = 0;
        for (int i = 0; i < n; ++i)
            if (((s < 64 ? row[basis[i]].first : row[basis[i]].second) >> (s & 63)) & 1) t |= 1 << i;
        tau[s] = t;
        if (tau_inv[t] >= 0) return false;
        tau_inv[t] = s;
    }
    for (int j = 0; j < nt; ++j) {
        int g = 0;
        for (int i = 0; i < n; ++i) {
            const int col = tau_inv[1 << i];
            if (((col < 64 ? row[j].first : row[j].second) >> (col & 63)) & 1) g |= 1 << i;
        }
        sigma[j] = g;
        if (sigma_inv[g] >= 0) return false;
        sigma_inv[g] = j;
    }
    for (int j = 0; j < nt; ++j)
        for (int s = 0; s < nt; ++s) {
            const float want = (float)(rs[j] * cs[s]) * ((__builtin_popcount(sigma[j] & tau[s]) & 1) ? -1.0f : 1.0f);
            if (P[(size_t)j * nt + s] != want) return false;
        }
    bool ident = true;
    for (int u = 0; u < nt; ++u) {
        perm[0][u] = tau_inv[u] | (cs[tau_inv[u]] < 0 ? 256 : 0);
        perm[1][u] = sigma_inv[u] | (rs[sigma_inv[u]] < 0 ? 256 : 0);
        ident = ident && perm[0][u] == u && perm[1][u] == u;
    }
    *identity = ident;
    return true;
}

// device tables of the PERM Walsh-Hadamard kernel from c->p_perm: [4][nt] = source symbol, its sign, byte offset of the output antenna's row, its sign
int pilot_fast_tables(csi_ctx* c) {
    const int nt = c->cfg.nt;
    if (c->p_tables) { hipFree(c->p_tables); c->p_tables = nullptr; }
    if (!c->p_fast_ok || nt <= 0 || nt > CSI_WIRE_MAX_NT) return CSI_OK;
    c->p_fast_identity = true;
    std::vector<int> t((size_t)4 * nt);
    const float one = 1.0f, minus = -1.0f;
    for (int k = 0; k < 2; ++k)
        for (int u = 0; u < nt; ++u) {
            const int v = c->p_perm[k][u];
            if ((v & 255) >= nt) return fail(c, CSI_ERR_INVALID_ARG, "pilot permutation table entry %d out of range", v);
            t[(size_t)(2 * k) * nt + u] = k == 0 ? (v & 255) : (v & 255) * LS_NDATA * (int)sizeof(float);      // source symbol; BYTE offset of the output antenna's row inside an item
            std::memcpy(&t[(size_t)(2 * k + 1) * nt + u], (v & 256) ? &minus : &one, 4);
            c->p_fast_identity = c->p_fast_identity && v == u;
        }
    // Row 1 as the Walsh-Hadamard kernels consume it (ls_estimate_fwht2_kernel, PERM): the input signs S of a chunk of CH symbols are
    // multiplied out into butterfly coefficients, so that the kernel spends no instruction on them.  Per chunk: [S_r S_{r+8}, r < 8:
    // the fold of the two-threads-per-bin kernel (Nt = 128, CH = 16)], S_0, then for the levels of stride hh = 1, 2, 4 of the
    // 8-point transform one coefficient S_{i0} S_{i0+hh} per group i0 = 0, 2 hh, ..
    if (nt >= 16) {
        const int CH = nt == 128 ? 16 : 8, CHH = 8, L0 = CH - CHH;
        std::vector<float> sg(nt), cf(nt);
        std::memcpy(sg.data(), &t[(size_t)nt], sizeof(float) * nt);
        for (int ch = 0; ch < nt / CH; ++ch) {
            const float* s = sg.data() + ch * CH;
            float* o = cf.data() + ch * CH;
            float pend[8];
            for (int r = 0; r < CHH; ++r) { pend[r] = s[r]; if (L0) o[r] = s[r] * s[r + CHH]; }
            o[L0] = pend[0];
            int idx = L0 + 1;
            for (int hh = 1; hh < CHH; hh <<= 1) {
                for (int grp = 0; grp < CHH / (2 * hh); ++grp) o[idx + grp] = pend[grp * 2 * hh] * pend[grp * 2 * hh + hh];
                idx += CHH / (2 * hh);
            }
        }
        std::memcpy(&t[(size_t)nt], cf.data(), sizeof(float) * nt);
    }
    if (hipMalloc((void**)&c->p_tables, t.size() * sizeof(int)) != hipSuccess)
        return fail(c, CSI_ERR_NOMEM, "device allocation of %zu bytes failed", t.size() * sizeof(int));
    HIP_TRY(c, hipMemcpy(c->p_tables, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
    return CSI_OK;
}

int check_ready(csi_ctx* c, bool need_models, int model = -1) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "single-input context (nt=0): only csi_predict_samples is available");
    if (!c->pilot_ok) return fail(c, CSI_ERR_NOT_READY, "csi_set_pilot has not been called");
    if (need_models) {
        for (int d = 0; d < 2; ++d) {
            if (model >= 0 && d != model) continue;
            if (!c->model[d].loaded)
                return fail(c, CSI_ERR_NOT_READY, "weights of the %s model are not loaded", d ? "imag" : "real");
            if (!c->model[d].table_ok) {
                int rc = build_pilot_table(c, c->model[d]);
                if (rc) return rc;
            }
        }
    }
    return CSI_OK;
}

// hipGraph replay of a device-pointer call ("use_graph"): the 1st call with a key runs eagerly (it sizes every
// buffer), the 2nd is captured - the whole launch sequence of the call: range-guard memsets, magnitude sample,
// layer 0 (+ slab sum), per-pair layers, regressor, for both component models and every packet chunk, and the LS
// kernel for csi_estimate_device - later calls replay it.  Any reallocation / weight / pilot / option change
// drops the cache.  Calls that fork a second stream (small_call_overlap) or profile per kernel run eagerly.
template <typename Run>
int graph_or_run(csi_ctx* c, const GraphEntry& key, Run&& run) {
    if (!c->use_graph || c->prof_on) return run();
    auto same = [&](const GraphEntry& g) {
        return g.in_re == key.in_re && g.in_im == key.in_im && g.out_re == key.out_re && g.out_im == key.out_im && g.h_re == key.h_re &&
               g.h_im == key.h_im && g.npkt == key.npkt;
    };
    GraphEntry* ge = nullptr;
    for (auto& g : c->graphs)
        if (same(g)) ge = &g;
    if (!ge) {
        if (c->graphs.size() >= 16) drop_graphs(c);
        c->graphs.push_back(key);
        ge = &c->graphs.back();
    }
    if (ge->exec) {
        ++c->graph_replays;
        c->hs_launches += ge->hs_launches;         // the replayed kernels feed the range guard like eager ones
        HIP_TRY(c, hipGraphLaunch(ge->exec, c->stream));
        return CSI_OK;
    }
    if (ge->seen++ == 0) return run();
    const int64_t launches = c->hs_launches;
    HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = run();
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e_end != hipSuccess) return fail(c, CSI_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e_end));
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e_inst != hipSuccess) return fail(c, CSI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e_inst));
    for (auto& g : c->graphs)
        if (same(g)) { g.exec = exec; g.hs_launches = c->hs_launches - launches; }
    HIP_TRY(c, hipGraphLaunch(exec, c->stream));
    return CSI_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

int csi_abi_version(void) { return CSI_ABI_VERSION; }

const char* csi_last_error(const csi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int csi_create(const csi_config* cfg, csi_ctx** out) {
    if (!cfg || !out) return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: null argument");
    *out = nullptr;
    if (cfg->nt < 0 || cfg->nr < 1 || cfg->n_out < 1 || cfg->n_hidden < 1 || cfg->n_hidden > CSI_MAX_HIDDEN)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: bad shape nt=%d nr=%d n_hidden=%d n_out=%d",
                    cfg->nt, cfg->nr, cfg->n_hidden, cfg->n_out);
    // nt == 0: single-input model without pilot input (DNN.py:180,234); csi_predict_samples only
    if (cfg->nt > 0 && cfg->len_ltf != LS_SYM * cfg->nt)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: len_ltf (%d) must be 320*nt (%d)", cfg->len_ltf,
                    LS_SYM * cfg->nt);
    if (cfg->nt == 0 && (cfg->len_ltf < 4 || cfg->len_ltf % 4))
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: single-input width %d must be a positive multiple of 4", cfg->len_ltf);
    if (cfg->nt % 4)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: nt must be a multiple of 4 (16-byte rows), got %d", cfg->nt);
    for (int i = 0; i < cfg->n_hidden; ++i)
        if (cfg->hidden[i] < 4 || cfg->hidden[i] % 4)
            return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: hidden[%d]=%d must be a positive multiple of 4", i,
                        cfg->hidden[i]);
    if (cfg->dtype != CSI_DTYPE_F32 && cfg->dtype != CSI_DTYPE_BF16)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: unknown dtype %d", cfg->dtype);
    if (cfg->dtype == CSI_DTYPE_BF16) {
        for (int i = 0; i < cfg->n_hidden; ++i)
            if (cfg->hidden[i] % 8)
                return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: bf16 needs hidden widths that are multiples of 8 (hidden[%d]=%d)", i, cfg->hidden[i]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CSI_ERR_NO_DEVICE, "csi_create: no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_create: device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, CSI_ERR_HIP, "csi_create: hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CSI_ERR_NO_DEVICE, "csi_create: device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);

    csi_ctx* c = new csi_ctx();
    if (const char* h = std::getenv("CSI_DEBUG_HOOKS")) if (h[0] == '1') if (const char* d = std::getenv("CSI_SMALL_TILE16")) c->debug_small_tile16 = d[0] == '1';   // (once: not in the call path)
    if (const char* h = std::getenv("CSI_DEBUG_HOOKS")) if (h[0] == '1') if (const char* d = std::getenv("CSI_LS_LDS_PAD")) c->debug_ls_lds_pad = std::atoi(d);
    if (const char* h = std::getenv("CSI_DEBUG_HOOKS")) if (h[0] == '1') if (const char* d = std::getenv("CSI_BF16_FORK_LATE")) c->debug_bf16_fork_late = d[0] == '1';   // A/B: bf16 contexts fork the second stream behind the LS kernel
    c->cfg = *cfg;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (c->cfg.bn_eps <= 0.f) c->cfg.bn_eps = 1e-3f;
    c->d_in = cfg->len_ltf + cfg->nt;
    if (const char* e = std::getenv("CSI_FORCE_PAIR_TILE")) c->force_pair_tile = std::atoi(e);
    if (const char* e = std::getenv("CSI_LS_FFT_FIRST_MAX")) c->ls_fft_first_max = std::min(64, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("CSI_LS_DEBUG")) c->ls_debug = std::atoi(e);
    if (const char* e = std::getenv("CSI_LS_KERNEL")) c->ls_kernel = std::min(7, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("CSI_LS_V2")) c->ls_v2 = std::atoi(e);
    if (const char* e = std::getenv("CSI_HS_VM")) c->hs_vm_cast = c->hs_vm_pair = std::min(3, std::max(0, std::atoi(e)));
    auto bail = [&](int code) {
        g_create_error = c->err;
        csi_destroy(c);
        return code;
    };
    if (hipSetDevice(cfg->device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(CSI_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->err = "hipStreamCreate failed";
        return bail(CSI_ERR_HIP);
    }
    {
        int widest = 1;
        for (int i = 0; i < cfg->n_hidden; ++i) widest = std::max(widest, cfg->hidden[i]);
        const size_t zb = ((size_t)widest + 64) * sizeof(float);
        if (hipMalloc((void**)&c->hs_zero, zb) != hipSuccess || hipMemset(c->hs_zero, 0, zb) != hipSuccess) {
            c->err = "device allocation failed";
            return bail(CSI_ERR_NOMEM);
        }
    }
    if (hipMalloc((void**)&c->hs_peak, 256) != hipSuccess || hipMemset(c->hs_peak, 0, 256) != hipSuccess) {
        c->err = "device allocation failed";
        return bail(CSI_ERR_NOMEM);
    }
    // LS constants.  Twiddles in double on the host so the table is correctly rounded.
    std::vector<float> tw(2 * LS_FFT);
    for (int u = 0; u < LS_FFT; ++u) {
        const double ang = -2.0 * M_PI * u / LS_FFT;
        tw[u] = (float)std::cos(ang);
        tw[LS_FFT + u] = (float)std::sin(ang);
    }
    // VHT-LTF literal of helperMIMOChannelEstimate.m:16-23 and the data-bin list of
    // generate_maMIMO_LTF.m:98-102, in fftshift-ed (1-based MATLAB) bin order.
    static const int ltfL[26] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const int ltfR[26] = {1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const int midA[11] = {-1, -1, -1, 1, 1, -1, 1, -1, 1, 1, -1};
    static const int midB[9] = {1, -1, 1, -1, 0, 1, -1, -1, 1};
    std::vector<int> ltf;
    auto push = [&](const int* v, int n) { ltf.insert(ltf.end(), v, v + n); };
    auto seg = [&]() { push(ltfL, 26); ltf.push_back(1); push(ltfR, 26); };
    ltf.assign(7, 0);
    seg(); push(midA, 11); seg(); push(midB, 9); seg(); push(midA, 11); seg();
    ltf.insert(ltf.end(), 6, 0);
    if ((int)ltf.size() != LS_FFT) { c->err = "internal: LTF literal length"; return bail(CSI_ERR_INVALID_ARG); }
    static const int pilots[8] = {26, 54, 90, 118, 140, 168, 204, 232};
    std::vector<int> bin_pos;
    std::vector<float> denom;
    for (int k1 = 1; k1 <= LS_FFT; ++k1) {
        bool skip = (k1 <= 7) || (k1 == 129) || (k1 >= 251);
        for (int pk : pilots) skip = skip || (k1 == pk);
        if (skip) continue;
        bin_pos.push_back((k1 - 1 + LS_FFT / 2) % LS_FFT);      // undo fftshift: shifted index -> FFT bin
        denom.push_back((float)cfg->nt * (float)ltf[k1 - 1]);
    }
    if ((int)bin_pos.size() != LS_NDATA) { c->err = "internal: data-bin count"; return bail(CSI_ERR_INVALID_ARG); }
    if (upload(c, &c->tw, tw.data(), tw.size())) return bail(CSI_ERR_HIP);
    if (upload(c, &c->denom, denom.data(), denom.size())) return bail(CSI_ERR_HIP);
    if (hipMalloc((void**)&c->bin_pos, LS_NDATA * sizeof(int)) != hipSuccess ||
        hipMemcpy(c->bin_pos, bin_pos.data(), LS_NDATA * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        c->err = "bin table upload failed";
        return bail(CSI_ERR_HIP);
    }
    if (ls_prepare(c) != CSI_OK) return bail(CSI_ERR_HIP);
    *out = c;
    return CSI_OK;
}

void csi_destroy(csi_ctx* c) {
    if (!c) return;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& sp : c->spans) { hipEventDestroy(sp.beg); hipEventDestroy(sp.end); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    drop_graphs(c);
    free_model(c->model[0]);
    free_model(c->model[1]);
    for (int d = 0; d < 2; ++d) tr_free(c->trainer[d]);
    delete c->hostpipe;
    comm_free(c);
    if (c->band_mod) hipModuleUnload(c->band_mod);
    if (c->P) hipFree(c->P);
    if (c->hs_peak) hipFree(c->hs_peak);
    if (c->hs_zero) hipFree(c->hs_zero);
    if (c->fuse_ws) hipFree(c->fuse_ws);
    if (c->small_ws) hipFree(c->small_ws);
    if (c->Ppad) hipFree(c->Ppad);
    if (c->Pbf) hipFree(c->Pbf);
    if (c->p_tables) hipFree(c->p_tables);
    if (c->tw) hipFree(c->tw);
    if (c->bin_pos) hipFree(c->bin_pos);
    if (c->denom) hipFree(c->denom);
    if (c->ws) hipFree(c->ws);
    if (c->stage) hipFree(c->stage);
    if (c->aux_ws) hipFree(c->aux_ws);
    if (c->aux_l0skinny) hipFree(c->aux_l0skinny);
    if (c->aux_skbuf) hipFree(c->aux_skbuf);
    if (c->aux_fuse_ws) hipFree(c->aux_fuse_ws);
    if (c->aux_fork) hipEventDestroy(c->aux_fork);
    if (c->aux_join) hipEventDestroy(c->aux_join);
    if (c->aux_stream) hipStreamDestroy(c->aux_stream);
    if (c->ls_fork) hipEventDestroy(c->ls_fork);
    if (c->ls_join) hipEventDestroy(c->ls_join);
    if (c->ls_stream) hipStreamDestroy(c->ls_stream);
    if (c->skbuf) hipFree(c->skbuf);
    if (c->l0skinny) hipFree(c->l0skinny);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int csi_load_weights(csi_ctx* c, int model, const csi_tensor* tensors, int n) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (model < 0 || model > 1 || !tensors || n <= 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const csi_config& cf = c->cfg;
    Model& m = c->model[model];
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    free_model(m);
    m.layers.resize(cf.n_hidden + 1);
    int fan_in = c->d_in;
    std::vector<float> prev_shift;          // BN shift of the previous layer (split engine: folded into this layer's bias)
    std::vector<float> prev_scale;          // BN scale of the previous layer (split engine, layer 1 only: folded into the split weights)
    for (int li = 0; li <= cf.n_hidden; ++li) {
        const bool reg = li == cf.n_hidden;
        const std::string base = reg ? std::string("fc_regressor") : "fc_dense" + std::to_string(li);
        const int out = reg ? cf.n_out : cf.hidden[li];
        const csi_tensor* k = find_tensor(tensors, n, base + ".kernel");
        const csi_tensor* b = find_tensor(tensors, n, base + ".bias");
        if (!k || !b || !k->data || !b->data) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: missing %s.kernel/.bias", base.c_str());
        if (k->rows != fan_in || k->cols != out || b->rows * b->cols != out)
            return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: %s.kernel is [%lld,%lld], expected [%d,%d]", base.c_str(),
                        (long long)k->rows, (long long)k->cols, fan_in, out);
        Layer& L = m.layers[li];
        L.in = fan_in;
        L.out = out;
        // transpose [in][out] -> [out][ldw] (K-major, K zero-padded to a multiple of the k-tile) on
        // the host, blocked for cache friendliness
        L.ldw = (fan_in + G_BK - 1) / G_BK * G_BK;
        std::vector<float> wt((size_t)out * L.ldw, 0.f);
        const int TB = 32;
        for (int i0 = 0; i0 < fan_in; i0 += TB)
            for (int o0 = 0; o0 < out; o0 += TB)
                for (int i = i0; i < std::min(fan_in, i0 + TB); ++i)
                    for (int o = o0; o < std::min(out, o0 + TB); ++o) wt[(size_t)o * L.ldw + i] = k->data[(size_t)i * out + o];
        int rc = CSI_OK;
        const bool bf16 = cf.dtype == CSI_DTYPE_BF16;
        auto rne = [](float f) {                       // fp32 -> bf16, round to nearest even
            uint32_t u;
            std::memcpy(&u, &f, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        if (bf16) {
            L.ldwb = (fan_in + B_BK - 1) / B_BK * B_BK;
            std::vector<uint16_t> wb((size_t)out * L.ldwb, 0);
            // bf16 mode: the BatchNormalization of the previous layer lives in THIS layer's operands,
            //     (relu(z) sc + sh) W + b  =  relu(z) (diag(sc) W) + (b + sh W),
            // scaled rows rounded once, the shift product kept in fp32 in the bias below - so every hidden activation is
            // bf16(relu(z)) and the kernels that build or read it carry no per-column affine (the fused first per-pair
            // layer generated its A operand slower than the matrix cores consumed it)
            for (int o = 0; o < out; ++o)
                for (int i = 0; i < fan_in; ++i) {
                    const float w = wt[(size_t)o * L.ldw + i];
                    wb[(size_t)o * L.ldwb + i] = rne(prev_scale.empty() ? w : (float)((double)w * (double)prev_scale[i]));
                }
            const size_t bytes = wb.size() * 2 + 256;
            if (hipMalloc((void**)&L.Wb, bytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
            HIP_TRY(c, hipMemset(L.Wb, 0, bytes));
            HIP_TRY(c, hipMemcpy(L.Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
            if (reg && li == 2 && cf.nt > 0 && out <= 256) {
                // the regressor behind ONE per-pair layer: a copy for the fused band kernel - 256 rows (rows >= out zero), k
                // permuted inside every group of 16 so that a lane's stage-1 accumulators are its stage-2 operand
                std::vector<uint16_t> wp((size_t)256 * L.ldwb, 0);
                for (int o = 0; o < out; ++o)
                    for (int i = 0; i < L.ldwb; ++i) wp[(size_t)o * L.ldwb + i] = wb[(size_t)o * L.ldwb + (i & ~15) + hs_band_kperm(i & 15)];
                const size_t pbytes = wp.size() * 2 + 256;
                if (hipMalloc((void**)&L.Wb_p, pbytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
                HIP_TRY(c, hipMemset(L.Wb_p, 0, pbytes));
                HIP_TRY(c, hipMemcpy(L.Wb_p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
            }
        } else {
            rc = upload(c, &L.Wt, wt.data(), wt.size());
            if (rc) return rc;
            // split-f16 copy for gemm_hs.hip.h: W * 2^wshift (largest magnitude in [2^12, 2^13)) as
            // hi + lo halves in groups of 16 k-columns; layer 0 keeps its LTF columns only (the pilot
            // rows live in the table T)
            const int kh = (li == 0 && cf.nt > 0) ? cf.len_ltf : fan_in;
            // layer 1 (the first per-pair layer; the regressor when there is one hidden layer) on the shared-layer-0
            // path: the pair kernel generates A = relu(L0 + T) without bn0, so bn0's scale multiplies the rows of
            // this copy - (relu(z) sc) W = relu(z) (diag(sc) W) - and bn0's shift sits in bias_hs below
            const bool fold_scale = li == 1 && cf.nt > 0 && !prev_scale.empty();
            auto wv = [&](int o, int i) {
                const float w = wt[(size_t)o * L.ldw + i];
                return fold_scale ? (float)((double)w * (double)prev_scale[i]) : w;
            };
            float wmax = 0.f;
            for (int o = 0; o < out; ++o)
                for (int i = 0; i < kh; ++i) wmax = std::max(wmax, std::fabs(wv(o, i)));
            int e = 0;
            if (wmax > 0.f && std::isfinite(wmax)) std::frexp(wmax, &e);
            L.wshift = std::max(-40, std::min(40, 13 - e));
            const float ws = std::ldexp(1.f, L.wshift);
            L.ldwh = 2 * ((kh + HS_G - 1) / HS_G * HS_G);
            std::vector<uint16_t> wh((size_t)out * L.ldwh, 0);
            // representation error of this split copy: worst OUTPUT ROW, ||x_o - (hi + lo)_o|| / ||x_o|| (a whole-matrix norm would be
            // carried by the very outlier that causes the damage: the scale follows max |w|, the rows far below it lose their lo halves)
            double e_worst = 0.0;
            for (int o = 0; o < out; ++o) {
                double e_num = 0.0, e_den = 0.0;
                for (int i = 0; i < kh; ++i) {
                    const float x = wv(o, i) * ws;
                    const _Float16 hi = (_Float16)x;
                    const _Float16 lo = (_Float16)(x - (float)hi);
                    uint16_t* d = &wh[(size_t)o * L.ldwh + (i >> 4) * 32 + (i & 15)];
                    std::memcpy(d, &hi, 2);
                    std::memcpy(d + 16, &lo, 2);
                    const double r = (double)x - ((double)(float)hi + (double)(float)lo);
                    e_num += r * r;
                    e_den += (double)x * (double)x;
                }
                if (e_den > 0.0 && std::isfinite(e_den)) e_worst = std::max(e_worst, std::sqrt(e_num / e_den));
                else if (!std::isfinite(e_den)) e_worst = 1.0;
            }
            m.hs_repr_err = std::max(m.hs_repr_err, e_worst);
            const size_t hbytes = wh.size() * 2 + 4096;
            if (hipMalloc((void**)&L.Wh, hbytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
            HIP_TRY(c, hipMemset(L.Wh, 0, hbytes));
            HIP_TRY(c, hipMemcpy(L.Wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
            if (reg && li == 2 && cf.nt > 0 && !prev_scale.empty()) {
                // the regressor behind ONE per-pair layer (the shipped 2-hidden-layer network): a second copy with that
                // layer's BN scale folded into the rows, for the kernel that runs the regressor behind the pair layer
                // without h2 leaving the CU (gemm_hs.hip.h: hs_fused_regressor); the unfused path keeps Wh
                float fmax = 0.f;
                for (int o = 0; o < out; ++o)
                    for (int i = 0; i < kh; ++i) fmax = std::max(fmax, std::fabs((float)((double)wt[(size_t)o * L.ldw + i] * (double)prev_scale[i])));
                int ef = 0;
                if (fmax > 0.f && std::isfinite(fmax)) std::frexp(fmax, &ef);
                L.wshift_f = std::max(-40, std::min(40, 13 - ef));
                const float wsf = std::ldexp(1.f, L.wshift_f);
                std::fill(wh.begin(), wh.end(), 0);
                double f_worst = 0.0;
                for (int o = 0; o < out; ++o) {
                    double f_num = 0.0, f_den = 0.0;
                    for (int i = 0; i < kh; ++i) {
                        const float x = (float)((double)wt[(size_t)o * L.ldw + i] * (double)prev_scale[i]) * wsf;
                        const _Float16 hi = (_Float16)x;
                        const _Float16 lo = (_Float16)(x - (float)hi);
                        uint16_t* d = &wh[(size_t)o * L.ldwh + (i >> 4) * 32 + (i & 15)];
                        std::memcpy(d, &hi, 2);
                        std::memcpy(d + 16, &lo, 2);
                        const double r = (double)x - ((double)(float)hi + (double)(float)lo);
                        f_num += r * r;
                        f_den += (double)x * (double)x;
                    }
                    if (f_den > 0.0 && std::isfinite(f_den)) f_worst = std::max(f_worst, std::sqrt(f_num / f_den));
                    else if (!std::isfinite(f_den)) f_worst = 1.0;
                }
                m.hs_repr_err = std::max(m.hs_repr_err, f_worst);
                if (hipMalloc((void**)&L.Wh_f, hbytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
                HIP_TRY(c, hipMemset(L.Wh_f, 0, hbytes));
                HIP_TRY(c, hipMemcpy(L.Wh_f, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
                if (out <= 256) {
                    // ... and the form the fused band kernel reads: 256 rows, k permuted inside every group of 16 so that a
                    // lane's stage-1 accumulator registers are its stage-2 operand (gemm_hs_band.hip.h)
                    std::vector<uint16_t> wp((size_t)256 * L.ldwh, 0);
                    for (int o = 0; o < out; ++o)
                        for (int g16 = 0; g16 < L.ldwh / 32; ++g16)
                            for (int pos = 0; pos < 16; ++pos) {
                                const size_t src = (size_t)o * L.ldwh + (size_t)g16 * 32 + hs_band_kperm(pos);
                                const size_t dst = (size_t)o * L.ldwh + (size_t)g16 * 32 + pos;
                                wp[dst] = wh[src];
                                wp[dst + 16] = wh[src + 16];
                            }
                    const size_t pbytes = wp.size() * 2 + 4096;
                    if (hipMalloc((void**)&L.Wh_p, pbytes) != hipSuccess) return fail(c, CSI_ERR_NOMEM, "weight allocation failed");
                    HIP_TRY(c, hipMemset(L.Wh_p, 0, pbytes));
                    HIP_TRY(c, hipMemcpy(L.Wh_p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
                }
            }
        }
        if (bf16 && li >= 1 && !prev_shift.empty()) {
            std::vector<float> bf(out);
            for (int o = 0; o < out; ++o) {
                double acc = b->data[o];
                for (int i = 0; i < fan_in; ++i) acc += (double)prev_shift[i] * (double)k->data[(size_t)i * out + o];
                bf[o] = (float)acc;
            }
            rc = upload(c, &L.bias, bf.data(), out);
        } else {
            rc = upload(c, &L.bias, b->data, out);
        }
        if (rc) return rc;
        if (!bf16 && li >= 1) {
            // split engine: the previous layer's BatchNormalization shift moves into this layer's bias,
            //     (relu(z) sc + sh) W + b  =  (relu(z) sc) W + (b + sh W),
            // so that the A operand keeps the exact zeros of the relu (about half of it) - the f16 matrix
            // pipe is power-bound and runs ~5 % faster on such operands
            std::vector<float> bf(out);
            for (int o = 0; o < out; ++o) {
                double acc = b->data[o];
                if (!prev_shift.empty())
                    for (int i = 0; i < fan_in; ++i) acc += (double)prev_shift[i] * (double)k->data[(size_t)i * out + o];
                bf[o] = (float)acc;
            }
            rc = upload(c, &L.bias_hs, bf.data(), out);
            if (rc) return rc;
        }
        std::vector<float> sc(out, 1.f), sh(out, 0.f);
        if (!reg && cf.use_bn) {
            const std::string bn = "bn" + std::to_string(li);
            const csi_tensor* ga = find_tensor(tensors, n, bn + ".gamma");
            const csi_tensor* be = find_tensor(tensors, n, bn + ".beta");
            const csi_tensor* mu = find_tensor(tensors, n, bn + ".moving_mean");
            const csi_tensor* va = find_tensor(tensors, n, bn + ".moving_variance");
            if (!ga || !be || !mu || !va) return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: missing %s.* (use_bn=1)", bn.c_str());
            for (const csi_tensor* t : {ga, be, mu, va})
                if (!t->data || t->rows * t->cols != out)
                    return fail(c, CSI_ERR_INVALID_ARG, "csi_load_weights: %s.* must have %d elements", bn.c_str(), out);
            for (int o = 0; o < out; ++o) {
                // keras non-fused inference: inv = rsqrt(var + eps) * gamma; y = x*inv + (beta - mean*inv)
                const float inv = (1.0f / std::sqrt(va->data[o] + cf.bn_eps)) * ga->data[o];
                sc[o] = inv;
                sh[o] = be->data[o] - mu->data[o] * inv;
            }
            // split-f16 engine: scale of the activations this layer hands on.  Where the next kernel reads relu * scale
            // (BatchNormalization output minus its shift; Layer::ashift): beta + gamma * (standardised relu) is bounded
            // by |beta| + 6 |gamma| at six moving standard deviations.  Where it reads the bare relu because the scale
            // sits in the next layer's weights (layer 0 on the shared path, the pair layer in front of the fused
            // regressor; Layer::ashift_pre): moving_mean + 6 moving standard deviations.  The bound is put at
            // 2^10..2^11 - 32x of head room for outliers before the range guard takes over, and values down to
            // 2^-13 of the bound keep a normal lo half
            float amax = 0.f, amax_pre = 0.f;
            for (int o = 0; o < out; ++o) {
                amax = std::max(amax, std::fabs(be->data[o]) + 6.f * std::fabs(ga->data[o]));
                amax_pre = std::max(amax_pre, std::fabs(mu->data[o]) + 6.f * std::sqrt(std::max(va->data[o], 0.f) + cf.bn_eps));
            }
            auto shift_for = [](float bound, int dflt) {
                int ea = 0;
                if (!(bound > 0.f) || !std::isfinite(bound)) return dflt;
                std::frexp(bound, &ea);
                return std::max(-8, std::min(14, 11 - ea));
            };
            L.ashift = shift_for(amax, L.ashift);
            L.ashift_pre = shift_for(amax_pre, L.ashift_pre);
        }
        if (!reg) {
            // bf16 mode: identity here, the real vectors went into the next layer (above)
            const std::vector<float> one(out, 1.f), zero(out, 0.f);
            rc = upload(c, &L.scale, bf16 ? one.data() : sc.data(), out);
            if (rc) return rc;
            rc = upload(c, &L.shift, bf16 ? zero.data() : sh.data(), out);
            if (rc) return rc;
        }
        if (li == 0 && cf.nt > 0) {
            // pilot rows of fc_dense0.kernel, [nt][h1] row-major as stored (bf16 mode: rounded like
            // every other weight, the table itself is evaluated in fp32)
            std::vector<float> w0p(k->data + (size_t)cf.len_ltf * out, k->data + (size_t)(cf.len_ltf + cf.nt) * out);
            if (bf16)
                for (float& v : w0p) { const uint32_t u = (uint32_t)rne(v) << 16; std::memcpy(&v, &u, 4); }
            rc = upload(c, &m.W0p, w0p.data(), w0p.size());
            if (rc) return rc;
            if (!bf16) {
                rc = upload(c, &m.W0rm, k->data, (size_t)cf.len_ltf * out);
                if (rc) return rc;
            }
        }
        prev_shift = sh;
        prev_scale = sc;
        fan_in = out;
    }
    m.loaded = true;
    m.table_ok = false;
    if (cf.dtype == CSI_DTYPE_F32) {
        m.hs_repr_ok = m.hs_repr_err <= 0x1p-20;
        // not an error: the fp32 MFMA kernels serve such a model.  It is reported through the counters "hs_weight_pins" /
        // "hs_weight_err_e12" (include/csi_mamimo.h), never through csi_last_error: a successful call leaves no error text behind
        // 
        if (!m.hs_repr_ok) ++c->hs_weight_pins;
    }
    return build_pilot_table(c, m);
}

// host-only: CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) - the per-tensor checksum of TensorFlow's TensorBundle files
// (keras_files.py); continuing value in, value out (both un-inverted outside: crc = 0 starts a new checksum)
__attribute__((target("sse4.2"))) static uint32_t crc32c_sse42(const unsigned char* p, size_t n, uint32_t c) {
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = __builtin_ia32_crc32qi(c, *p++); --n; }
    uint64_t c64 = c;
    for (; n >= 8; n -= 8, p += 8) { uint64_t v; std::memcpy(&v, p, 8); c64 = __builtin_ia32_crc32di(c64, v); }
    c = (uint32_t)c64;
    while (n--) c = __builtin_ia32_crc32qi(c, *p++);
    return c;
}
uint32_t csi_crc32c(const void* data, int64_t bytes, uint32_t crc) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    if (!p || bytes <= 0) return crc;
    if (__builtin_cpu_supports("sse4.2")) return crc32c_sse42(p, (size_t)bytes, c) ^ 0xFFFFFFFFu;
    for (int64_t i = 0; i < bytes; ++i) {
        c ^= p[i];
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    }
    return c ^ 0xFFFFFFFFu;
}

// host-only: which LS despread a pilot matrix gets (no context, no device)
int csi_pilot_classify(const float* P, int nt, int32_t* sym_src, int32_t* out_row) {
    if (!P || nt <= 0) return CSI_ERR_INVALID_ARG;
    int perm[2][CSI_WIRE_MAX_NT];
    bool ident = true;
    if (!pilot_decompose(P, nt, perm, &ident)) return 0;
    for (int u = 0; u < nt; ++u) {
        if (sym_src) sym_src[u] = perm[0][u];
        if (out_row) out_row[u] = perm[1][u];
    }
    return ident ? 1 : 2;
}

int csi_set_pilot(csi_ctx* c, const float* P) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!P || c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_set_pilot: null P or single-input context");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    int rc = upload(c, &c->P, P, (size_t)c->cfg.nt * c->cfg.nt);
    if (rc) return rc;
    {   // Sylvester Hadamard?  P[j][s] == (-1)^popcount(j & s) exactly -> the Walsh-Hadamard despread applies
        const int nt = c->cfg.nt;
        bool syl = nt >= 2 && (nt & (nt - 1)) == 0;
        for (int j = 0; syl && j < nt; ++j)
            for (int q = 0; q < nt; ++q)
                if (P[(size_t)j * nt + q] != ((__builtin_popcount(j & q) & 1) ? -1.0f : 1.0f)) { syl = false; break; }
        c->p_sylvester = syl;
        bool ident = true;
        c->p_fast_ok = pilot_decompose(P, nt, c->p_perm, &ident);
        c->p_fast_identity = ident;
        rc = pilot_fast_tables(c);
        if (rc) return rc;
        // how many bf16 pieces (8 significand bits each, truncation) the entries need: the bf16-split LS despread keeps that many
        int pieces = 1;
        for (size_t i = 0; i < (size_t)nt * nt && pieces < 3; ++i) {
            uint32_t u; std::memcpy(&u, &P[i], 4);
            uint32_t t = u & 0xffff0000u; float f1; std::memcpy(&f1, &t, 4);
            const float r1 = P[i] - f1;
            if (r1 != 0.f) {
                std::memcpy(&u, &r1, 4); t = u & 0xffff0000u; float f2; std::memcpy(&f2, &t, 4);
                pieces = std::max(pieces, r1 - f2 != 0.f ? 3 : 2);
            }
        }
        c->p_pieces = pieces;
        // the pieces in the operand order of v_mfma_f32_32x32x16_bf16: block (chunk of 16 symbols, piece, antenna tile) =
        // [k half][row 32][8 symbols], what lane (row + 32 half) of a wave reads as one 16-byte LDS word
        const int jt = (nt + 31) / 32, nch = (nt + 15) / 16;
        std::vector<uint16_t> pb((size_t)nch * 3 * jt * LSB_BLOCK, 0);
        for (int j = 0; j < nt; ++j)
            for (int s = 0; s < nt; ++s) {
                float x = P[(size_t)j * nt + s];
                for (int k = 0; k < 3; ++k) {
                    uint32_t u; std::memcpy(&u, &x, 4);
                    const uint32_t t = u & 0xffff0000u; float f; std::memcpy(&f, &t, 4);
                    pb[(((size_t)(s >> 4) * 3 + k) * jt + (j >> 5)) * LSB_BLOCK + (((s >> 3) & 1) * 32 + (j & 31)) * 8 + (s & 7)] = (uint16_t)(t >> 16);
                    x -= f;
                }
            }
        std::vector<float> pbf((pb.size() + 1) / 2);
        std::memcpy(pbf.data(), pb.data(), pb.size() * 2);
        rc = upload(c, &c->Pbf, pbf.data(), pbf.size());
        if (rc) return rc;
    }
    {   // zero-padded copy for the chunked LS kernel (rows / columns up to the next multiple of 32)
        const int nt = c->cfg.nt, ldp = (nt + 31) / 32 * 32;
        std::vector<float> pad((size_t)ldp * ldp, 0.f);
        for (int j = 0; j < nt; ++j) std::memcpy(&pad[(size_t)j * ldp], P + (size_t)j * nt, sizeof(float) * nt);
        rc = upload(c, &c->Ppad, pad.data(), pad.size());
        if (rc) return rc;
    }
    rc = ls_prepare(c);
    if (rc) return rc;
    c->pilot_ok = true;
    for (int d = 0; d < 2; ++d) {
        c->model[d].table_ok = false;
        rc = build_pilot_table(c, c->model[d]);
        if (rc) return rc;
    }
    return CSI_OK;
}

namespace {
// does csi_predict_device run the two component models of this call on two streams?
bool two_stream_call(csi_ctx* c, int64_t npkt) {
    const bool bf16 = c->cfg.dtype == CSI_DTYPE_BF16;
    return c->small_call_overlap && !c->prof_on && !c->use_graph && !c->in_graph_call && !c->in_host_pipeline &&
           (npkt * c->cfg.nr * std::max(c->cfg.nt, 1) <= (bf16 ? 262144 : 327680) || c->small_call_overlap == 2);
    // (fp32 contexts, round 6: 131 072 -> 327 680 pair rows.  With the register-blocked band kernels two streams win up to 2560 packets of the shipped shape -
    // 1025 packets 2396 -> 2239 us per call, 1408: 2997 -> 2893, 2048: 4117 -> 4047, 2560: 5293 -> 5252 - and lose from 3000 on (+1.8 %, 4000: +3 %);
    // bf16 contexts: even at 1280 packets, worse beyond.  tools/regime_probe.py small_call_overlap=1 small_call_overlap=2, profiles/r06_band_probe.txt (H))
}
int aux_stream_ensure(csi_ctx* c) {
    if (!c->aux_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
    }
    return CSI_OK;
}
}  // namespace

int csi_predict_device(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_out_re,
                       float* d_out_im) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!d_ltf_re || !d_ltf_im || !d_out_re || !d_out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_device: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto run = [&]() -> int {
        const bool bf16 = c->cfg.dtype == CSI_DTYPE_BF16;
        auto plane = [&](Model& m, const float* in, float* out) { return bf16 ? predict_plane_bf16(c, m, in, npkt, out) : predict_plane(c, m, in, npkt, out); };
        // the one-packet regime: both models in 1 + n_hidden launches on this stream (csi_dnn_small.hpp)
        if (small_call_ok(c, npkt)) return predict_small(c, d_ltf_re, d_ltf_im, npkt, d_out_re, d_out_im);
        // small call: the two component models are independent and each is a chain of short,
        // launch-latency-bound kernels - run the imag model on a second stream with its own scratch
        // Round 5: up to 98 304 pair rows (768 bands of the fused kernel per model - three rounds of the chip), not 64 preambles: a
        // component model's kernels of a mid-size call fill a fraction of the 256 CUs (64 packets = 64 bands), and the other model's
        // fill the rest - 24 ... 128 packets 1.25-1.55x, 384 packets +16 %, 500 packets +1.4 % (profiles/r05_regime_probe.txt).  Beyond
        // that the gain is below 1 % and not worth the second workspace; 2 = any size (A/B runs).  (With the fork in front of the LS kernel:
        // 131 072 rows - 1000 packets +2.8 %, 2000 equal, 4000 packets 2.5 % slower.)  bf16 contexts (sequential until the
        // end of round 5): up to 262 144 pair rows - Nt = 64: one packet 129 -> 95 us, 64 packets 271 -> 198, 500: 1478 -> 1196, 1000: 2380 -> 2096,
        // 2000 packets equal (profiles/r05_band_split_probe.txt).
        const bool overlap = two_stream_call(c, npkt);
        if (!overlap) {
            int r = plane(c->model[0], d_ltf_re, d_out_re);
            if (r) return r;
            return plane(c->model[1], d_ltf_im, d_out_im);
        }
        if (int ra = aux_stream_ensure(c)) return ra;
        if (!c->aux_preforked) {           // (csi_estimate_device forks in front of its LS kernel)
            HIP_TRY(c, hipEventRecord(c->aux_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
        }
        c->aux_preforked = false;
        auto swap_scratch = [&]() {
            std::swap(c->stream, c->aux_stream);
            std::swap(c->ws, c->aux_ws); std::swap(c->ws_bytes, c->aux_ws_bytes);
            std::swap(c->l0skinny, c->aux_l0skinny); std::swap(c->l0skinny_bytes, c->aux_l0skinny_bytes);
            std::swap(c->skbuf, c->aux_skbuf); std::swap(c->skbuf_bytes, c->aux_skbuf_bytes);
            std::swap(c->fuse_ws, c->aux_fuse_ws); std::swap(c->fuse_ws_bytes, c->aux_fuse_ws_bytes);
        };
        swap_scratch();                                   // imag model: aux stream, aux scratch
        c->models_in_flight = 2;
        int r = plane(c->model[1], d_ltf_im, d_out_im);
        hipError_t e = r ? hipSuccess : hipEventRecord(c->aux_join, c->stream);
        swap_scratch();
        if (r) { c->models_in_flight = 1; return r; }
        if (e == hipSuccess) r = plane(c->model[0], d_ltf_re, d_out_re);
        c->models_in_flight = 1;
        HIP_TRY(c, e);
        if (r) return r;
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->aux_join, 0));
        return CSI_OK;
    };
    return graph_or_run(c, GraphEntry{d_ltf_re, d_ltf_im, d_out_re, d_out_im, nullptr, nullptr, npkt, 0, nullptr}, run);
}

int csi_estimate_device(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_out_re, float* d_out_im,
                        float* d_h_re, float* d_h_im) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!d_ltf_re || !d_ltf_im || !d_out_re || !d_out_im || !d_h_re || !d_h_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_estimate_device: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto run = [&]() -> int {
        const bool g = c->use_graph;
        c->use_graph = false;                      // the two calls below are the graph's content, not graphs of their own
        c->in_graph_call = g;
        int r = CSI_OK;
        if (c->ls_overlap_cus > 0) {
            // LS beside the per-pair kernels: parked here, fired by the DNN path behind its first layer-0 kernel on a CU-masked side
            // stream (ls_deferred_fire); joined below.  A path that never reaches the hook fires it here, behind the DNN kernels.
            c->ls_deferred.active = true;
            c->ls_deferred.forked = false;
            c->ls_deferred.re = d_ltf_re; c->ls_deferred.im = d_ltf_im; c->ls_deferred.npkt = npkt; c->ls_deferred.h_re = d_h_re; c->ls_deferred.h_im = d_h_im;
            r = csi_predict_device(c, d_ltf_re, d_ltf_im, npkt, d_out_re, d_out_im);
            const int r2 = ls_deferred_fire(c);            // no-op when the hook has fired
            c->ls_deferred.active = false;
            if (c->ls_deferred.forked) {                   // also on an error: a forked stream must be joined (graph capture)
                const hipError_t e = hipStreamWaitEvent(c->stream, c->ls_join, 0);
                if (e != hipSuccess && !r) r = fail(c, CSI_ERR_HIP, "csi_estimate_device: joining the LS stream failed: %s", hipGetErrorString(e));
            }
            if (!r) r = r2;
        } else {
            // (one-packet calls, round 5: the LS kernel on a second stream beside the three DNN launches was built and measured -
            // 72.5 us per call against 64.2 in this order; the fork / join events cost more than the 8.7 us kernel hides, and it
            // runs 16.8 us beside the weight stream.  profiles/r05_small_call_trace.txt)
            // The second stream of a two-stream call is forked HERE, in front of the LS kernel: the imag model's chain needs the
            // preambles, not the LS result, and the cross-queue wait (6-7 us before its first kernel starts) passes under the LS kernel
            // (Round 6, first: bf16 contexts forked BEHIND the LS kernel, because 19 of 20 calls of 500 ... 1000 packets at Nt = 64 came back with wrong LS items when
            // the imag model's layer 0 - gemm_bf16_kernel, 64 KiB of LDS: its MFMA waves fit on an LS workgroup's CU - ran beside the LS kernel.  Root cause found
            // later in the round (tools/pk_opsel_probe.hip, profiles/r06_pk_opsel_probe.txt): a packed-fp32 instruction whose SECOND source takes its
            // low half from the high register (the transform's +-i rotations were v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]) loses that operand in lanes 48-63 while
            // another wave of the SIMD issues MFMAs.  The rotations are single adds now (ls_estimate.hip.h, CSI_LS_VAR_DEFAULT) and every context forks here again;
            // CSI_DEBUG_HOOKS=1 CSI_BF16_FORK_LATE=1 brings the late fork back for A/B runs.)
            if (c->aux_fork_early && !(c->cfg.dtype == CSI_DTYPE_BF16 && c->debug_bf16_fork_late) && !small_call_ok(c, npkt) && two_stream_call(c, npkt) && aux_stream_ensure(c) == CSI_OK) {
                hipError_t e = hipEventRecord(c->aux_fork, c->stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0);
                if (e != hipSuccess) r = fail(c, CSI_ERR_HIP, "csi_estimate_device: forking the second stream failed: %s", hipGetErrorString(e));
                else c->aux_preforked = true;
            }
            // one-packet calls (round 6): the LS estimate rides in the layer-0 launch of the DNN (small_l0_ls_kernel); predict_small takes it from
            // the context - a call that does not reach that kernel after all runs the LS kernel behind the DNN
            const bool ls_inside = !r && small_ls_fusable(c, npkt);
            if (ls_inside) { c->small_ls_h_re = d_h_re; c->small_ls_h_im = d_h_im; }
            else if (!r) r = csi_ls_estimate_device(c, d_ltf_re, d_ltf_im, npkt, d_h_re, d_h_im);
            if (!r) r = csi_predict_device(c, d_ltf_re, d_ltf_im, npkt, d_out_re, d_out_im);
            if (ls_inside && c->small_ls_h_re) {
                c->small_ls_h_re = c->small_ls_h_im = nullptr;
                if (!r) r = csi_ls_estimate_device(c, d_ltf_re, d_ltf_im, npkt, d_h_re, d_h_im);
            }
            c->aux_preforked = false;
        }
        c->use_graph = g;
        c->in_graph_call = false;
        return r;
    };
    return graph_or_run(c, GraphEntry{d_ltf_re, d_ltf_im, d_out_re, d_out_im, d_h_re, d_h_im, npkt, 0, nullptr}, run);
}

int csi_ls_estimate_device(csi_ctx* c, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt, float* d_h_re,
                           float* d_h_im) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!d_ltf_re || !d_ltf_im || !d_h_re || !d_h_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_ls_estimate_device: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    const LsPlan plan = ls_plan(c);
    const int n_jc = (cf.nt + LSD_ROWS - 1) / LSD_ROWS;
    HIP_TRY(c, hipSetDevice(cf.device));
    const int64_t nblk = npkt * cf.nr;
    LsArgs a = ls_args(c, d_ltf_re, d_ltf_im, d_h_re, d_h_im);
    const int64_t max_grid = ((int64_t)1 << 30) / n_jc;      // also keeps nb inside an int
    for (int64_t b0 = 0; b0 < nblk; b0 += max_grid) {
        const int64_t nb = std::min(max_grid, nblk - b0);
        a.ltf_re = d_ltf_re + (size_t)b0 * cf.len_ltf;
        a.ltf_im = d_ltf_im + (size_t)b0 * cf.len_ltf;
        a.h_re = d_h_re + (size_t)b0 * cf.nt * LS_NDATA;
        a.h_im = d_h_im + (size_t)b0 * cf.nt * LS_NDATA;
        const double pairs = (double)nb * cf.nt;
        int nb32 = (int)nb;
        ProfScope ps(c, K_LS_ESTIMATE, pairs * (10240.0 + 8.0 * LS_NDATA * cf.nt), pairs * (2560.0 + 1872.0));
        if (plan.mode == LS_DESPREAD_FIRST) {
            hipLaunchKernelGGL(ls_despread_first_kernel, dim3((unsigned)(nb * n_jc)), dim3(LS_THREADS), plan.lds, c->stream, a, n_jc);
        } else {
            // persistent grid: as many workgroups as can reside (x256 CUs)
            const unsigned grid = (unsigned)std::min<int64_t>(nb, (int64_t)(c->ls_grid_cus > 0 ? c->ls_grid_cus : 256) * ((c->ls_debug & 128) ? 1 : plan.per_cu));      // ls_debug 128: one workgroup per CU (race hunt)
            void* kargs[] = {(void*)&a, (void*)&nb32};
            HIP_TRY(c, hipLaunchKernel(plan.fn, dim3(grid), dim3(plan.threads), kargs, plan.lds, c->stream));
        }
        HIP_TRY(c, hipGetLastError());
    }
    return CSI_OK;
}

int csi_lmmse_estimate_device(csi_ctx* c, const float* d_h_re, const float* d_h_im, int64_t npkt, const float* d_hvec, int L,
                              const float* d_snr_db, float* d_out_re, float* d_out_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (c->cfg.nt == 0) return fail(c, CSI_ERR_INVALID_ARG, "single-input context (nt=0): no LMMSE estimate");
    if (npkt < 0 || L < 1 || (npkt > 0 && (!d_h_re || !d_h_im || !d_hvec || !d_snr_db || !d_out_re || !d_out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_lmmse_estimate_device: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    const int n_jc = (cf.nt + LM_RHS - 1) / LM_RHS;
    const int64_t nblk = npkt * cf.nr;
    const int64_t max_items = ((int64_t)1 << 30) / n_jc / cf.nr * cf.nr;      // whole packets per launch
    for (int64_t b0 = 0; b0 < nblk; b0 += max_items) {
        const int64_t nb = std::min(max_items, nblk - b0);
        LmmseArgs a{};
        a.h_re = d_h_re + (size_t)b0 * cf.nt * LM_N;
        a.h_im = d_h_im + (size_t)b0 * cf.nt * LM_N;
        a.hvec = d_hvec + (size_t)(b0 / cf.nr) * L;
        a.snr_db = d_snr_db + b0;
        a.o_re = d_out_re + (size_t)b0 * cf.nt * LM_N;
        a.o_im = d_out_im + (size_t)b0 * cf.nt * LM_N;
        a.nt = cf.nt; a.nr = cf.nr; a.L = L;
        // per right-hand side: 2 * n^2 complex multiply-adds = 16 n^2 flop (fp64)
        ProfScope ps(c, K_LMMSE, (double)nb * (cf.nt + n_jc) * 16.0 * LM_N * LM_N, (double)nb * cf.nt * LM_N * 16.0);
        hipLaunchKernelGGL(lmmse_levinson_kernel, dim3((unsigned)(nb * n_jc)), dim3(LM_THREADS), 0, c->stream, a, n_jc);
        HIP_TRY(c, hipGetLastError());
    }
    return CSI_OK;
}

int csi_lmmse_estimate(csi_ctx* c, const float* h_re, const float* h_im, int64_t npkt, const float* hvec, int L,
                       const float* snr_db, float* out_re, float* out_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (npkt < 0 || L < 1 || (npkt > 0 && (!h_re || !h_im || !hvec || !snr_db || !out_re || !out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_lmmse_estimate: bad argument");
    if (npkt == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    const size_t pkt_f = (size_t)cf.nr * cf.nt * LM_N;                     // floats per packet and plane
    int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / (int64_t)(4 * pkt_f * sizeof(float)));
    chunk = std::min(chunk, npkt);
    const size_t need = (4 * pkt_f + (size_t)L + cf.nr) * sizeof(float) * (size_t)chunk;
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, need);
    if (rc) return rc;
    float* d_re = reinterpret_cast<float*>(c->stage);
    float* d_im = d_re + pkt_f * chunk;
    float* d_ore = d_im + pkt_f * chunk;
    float* d_oim = d_ore + pkt_f * chunk;
    float* d_hv = d_oim + pkt_f * chunk;
    float* d_snr = d_hv + (size_t)L * chunk;
    for (int64_t p0 = 0; p0 < npkt; p0 += chunk) {
        const int64_t np = std::min(chunk, npkt - p0);
        HIP_TRY(c, hipMemcpyAsync(d_re, h_re + p0 * pkt_f, pkt_f * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_im, h_im + p0 * pkt_f, pkt_f * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_hv, hvec + p0 * L, (size_t)L * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_snr, snr_db + p0 * cf.nr, (size_t)cf.nr * np * sizeof(float), hipMemcpyHostToDevice, c->stream));
        rc = csi_lmmse_estimate_device(c, d_re, d_im, np, d_hv, L, d_snr, d_ore, d_oim);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(out_re + p0 * pkt_f, d_ore, pkt_f * np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(out_im + p0 * pkt_f, d_oim, pkt_f * np * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

// ---------------------------------------------------------------- accuracy metric
int csi_nmse_device(csi_ctx* c, const float* d_ref_re, const float* d_ref_im, const float* d_est_re, const float* d_est_im,
                    int64_t nlinks, int n_bins, float* d_per_link, double* mean_out) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (nlinks <= 0 || n_bins <= 0 || !d_ref_re || !d_ref_im || !d_est_re || !d_est_im || !mean_out)
        return fail(c, CSI_ERR_INVALID_ARG, "csi_nmse_device: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // scratch: the per-link ratios (unless the caller wants them) followed by the fp64 sum
    const size_t need = (d_per_link ? 0 : (size_t)nlinks * sizeof(float)) + 64;
    int rc = ensure_bytes(c, &c->skbuf, &c->skbuf_bytes, need + 64);
    if (rc) return rc;
    float* ratio = d_per_link ? d_per_link : reinterpret_cast<float*>(c->skbuf);
    double* d_sum = reinterpret_cast<double*>(c->skbuf + ((need - 64 + 63) / 64) * 64);
    HIP_TRY(c, hipMemsetAsync(d_sum, 0, sizeof(double), c->stream));
    {
        ProfScope ps(c, K_NMSE, 8.0 * (double)nlinks * n_bins, 16.0 * (double)nlinks * n_bins);
        const unsigned blocks = (unsigned)std::min<int64_t>((nlinks + 3) / 4, 8192);
        hipLaunchKernelGGL(nmse_links_kernel, dim3(blocks), dim3(256), 0, c->stream, d_ref_re, d_ref_im, d_est_re, d_est_im, nlinks, n_bins, ratio);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(nmse_sum_kernel, dim3(1), dim3(1024), 0, c->stream, ratio, nlinks, d_sum);
        HIP_TRY(c, hipGetLastError());
    }
    double sum = 0.0;
    HIP_TRY(c, hipMemcpyAsync(&sum, d_sum, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mean_out = sum / (double)nlinks;
    return CSI_OK;
}

int csi_nmse(csi_ctx* c, const float* ref_re, const float* ref_im, const float* est_re, const float* est_im, int64_t nlinks,
             int n_bins, double* mean_out) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (nlinks <= 0 || n_bins <= 0 || !ref_re || !ref_im || !est_re || !est_im || !mean_out)
        return fail(c, CSI_ERR_INVALID_ARG, "csi_nmse: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t row = (size_t)n_bins * sizeof(float);
    const int64_t chunk = std::min<int64_t>(nlinks, std::max<int64_t>(1, ((int64_t)256 << 20) / (int64_t)(4 * row)));
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, 4 * row * (size_t)chunk + 256);
    if (rc) return rc;
    float* d[4];
    for (int i = 0; i < 4; ++i) d[i] = reinterpret_cast<float*>(c->stage) + (size_t)i * chunk * n_bins;
    const float* h[4] = {ref_re, ref_im, est_re, est_im};
    double total = 0.0;
    for (int64_t l0 = 0; l0 < nlinks; l0 += chunk) {
        const int64_t nl = std::min(chunk, nlinks - l0);
        for (int i = 0; i < 4; ++i)
            HIP_TRY(c, hipMemcpyAsync(d[i], h[i] + (size_t)l0 * n_bins, row * (size_t)nl, hipMemcpyHostToDevice, c->stream));
        double mean = 0.0;
        rc = csi_nmse_device(c, d[0], d[1], d[2], d[3], nl, n_bins, nullptr, &mean);
        if (rc) return rc;
        total += mean * (double)nl;
    }
    *mean_out = total / (double)nlinks;
    return CSI_OK;
}

int csi_get_option(csi_ctx* c, const char* name, int64_t* value) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!name || !value) return fail(c, CSI_ERR_INVALID_ARG, "csi_get_option: null argument");
    const std::string n(name);
    if (n == "use_graph") *value = c->use_graph;
    else if (n == "force_tile") *value = c->force_pair_tile;
    else if (n == "xcd_order") *value = c->xcd_order;
    else if (n == "ls_fft_first_max") *value = c->ls_fft_first_max;
    else if (n == "small_call_overlap") *value = c->small_call_overlap;
    else if (n == "small_fused") *value = c->small_fused;
    else if (n == "small_ls_fused") *value = c->small_ls_fused;
    else if (n == "small_ls_launches") *value = c->small_ls_launches;
    else if (n == "small_calls") *value = c->small_calls;
    else if (n == "small_rows") *value = c->small_rows;
    else if (n == "small_rows_band") *value = c->small_rows_band;
    else if (n == "f32_engine") *value = c->f32_engine;
    else if (n == "hs_band") *value = c->hs_band;
    else if (n == "band4") *value = c->band4;
    else if (n == "band4_available") *value = (c->cfg.dtype == CSI_DTYPE_BF16 ? c->band_fn4_bf16 : c->band_fn4) != nullptr;
    else if (n == "band_launches") *value = c->band_launches;
    else if (n == "band_split") *value = c->band_split;
    else if (n == "aux_fork_early") *value = c->aux_fork_early;
    else if (n == "l0_stream") *value = c->l0_stream;
    else if (n == "l0_stream_ks") *value = c->l0_stream_ks;
    else if (n == "l0_stream_prepass_rows") *value = c->l0_stream_prepass_rows;
    else if (n == "l0_stream_max_rows") *value = c->l0_stream_max_rows;
    else if (n == "l0_stream_launches") *value = c->l0_stream_launches;
    else if (n == "band_split_launches") *value = c->band_split_launches;
    else if (n == "comm_bytes") *value = c->comm ? c->comm->bytes_broadcast : 0;
    else if (n == "comm_blobs") *value = c->comm ? c->comm->blobs_broadcast : 0;
    else if (n == "comm_world") *value = c->comm ? c->comm->world : 0;
    else if (n == "comm_rank") *value = c->comm ? c->comm->rank : -1;
    else if (n == "hs_act_shift") *value = c->hs_act_shift;
    else if (n == "hs_min_blocks") *value = c->hs_min_blocks;
    else if (n == "hs_fuse_regressor") *value = c->hs_fuse_regressor;
    else if (n == "hs_blocked") *value = c->hs_blocked;
    else if (n == "hs_in_shift") *value = c->hs_in_shift;
    else if (n == "bf16_fused_h1") *value = c->bf16_fused_h1;
    else if (n == "band_tail_split") *value = c->band_tail_split;
    else if (n == "band_tail_launches") *value = c->band_tail_launches;
    else if (n == "bf16_l0_fused_split") *value = c->bf16_l0_fused_split;
    else if (n == "bf16_l0_fused_split_launches") *value = c->bf16_l0_fused_split_launches;
    else if (n == "host_threads") *value = c->host_threads;
    else if (n == "ls_kernel") *value = c->ls_kernel;
    else if (n == "ls_v2") *value = c->ls_v2;
    else if (n == "ls_ringb_min") *value = c->ls_ringb_min;
    else if (n == "ls_pilot_pieces") *value = c->p_pieces;
    else if (n == "ls_mode") *value = ls_plan(c).mode;
    else if (n == "ls_per_cu") *value = ls_plan(c).per_cu;                    // read-only: resident workgroups per CU of the kernel the next LS call runs
    else if (n == "hp_stage_us") *value = c->hostpipe ? (int64_t)c->hostpipe->us_stage : 0;           // read-only: where the last csi_estimate_c128 spent its time
    else if (n == "hp_wait_stage_us") *value = c->hostpipe ? c->hostpipe->us_wait_stage : 0;
    else if (n == "hp_wait_out_us") *value = c->hostpipe ? c->hostpipe->us_wait_out : 0;
    else if (n == "hp_weave_us") *value = c->hostpipe ? c->hostpipe->us_weave : 0;
    else if (n == "hp_total_us") *value = c->hostpipe ? c->hostpipe->us_total : 0;
    else if (n == "hp_chunk_packets") *value = c->hp_chunk_packets;
    else if (n == "hp_side_threads") *value = c->hp_side_threads;
    else if (n == "hp_device_weave") *value = c->hp_device_weave;
    else if (n == "hp_direct_out_calls") *value = c->hp_direct_out_calls;
    else if (n == "ls_fast_perm") *value = c->ls_fast_perm;
    else if (n == "ls_overlap_cus") *value = c->ls_overlap_cus;
    else if (n == "ls_overlap_stride") *value = c->ls_overlap_stride;
    else if (n == "ls_pilot_fast") *value = !c->p_fast_ok ? 0 : (c->p_fast_identity ? 1 : 2);     // read-only: 0 generic P, 1 Sylvester Hadamard, 2 a signed permutation of it
    else if (n == "hs_vm_cast") *value = c->hs_vm_cast;
    else if (n == "hs_vm_pair") *value = c->hs_vm_pair;
    else if (n == "graph_replays") *value = c->graph_replays;                // read-only counters
    else if (n == "hs_launches") *value = c->hs_launches;
    else if (n == "hs_range_fallbacks") *value = c->hs_range_fallbacks;
    else if (n == "hs_weight_pins") *value = c->hs_weight_pins;                  // models pinned to the fp32 MFMA kernels by the load-time check of their split copies
    else if (n == "hs_weight_err_e12") *value = (int64_t)(1e12 * std::max(c->model[0].hs_repr_err, c->model[1].hs_repr_err));    // worst relative representation error x 1e12
    else if (n == "band_available") {                                            // 1: the fused band kernel's code object is embedded in this build and loads
#ifdef CSI_HAVE_BAND8
        hipFunction_t f = nullptr;
        band8_function(c, &f, false, true);
        *value = f != nullptr;
#else
        *value = 0;
#endif
    }
    else return fail(c, CSI_ERR_INVALID_ARG, "csi_get_option: unknown option '%s'", name);
    return CSI_OK;
}

int csi_set_option(csi_ctx* c, const char* name, int64_t value) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!name) return fail(c, CSI_ERR_INVALID_ARG, "csi_set_option: null name");
    const std::string n(name);
    if (n == "use_graph") {
        if (!value) drop_graphs(c);
        c->use_graph = value != 0;
    } else if (n == "force_tile") {
        if (value != 0 && value != 128 && value != 256) return fail(c, CSI_ERR_INVALID_ARG, "force_tile must be 0, 128 or 256");
        drop_graphs(c);
        c->force_pair_tile = (int)value;
    } else if (n == "xcd_order") {
        if (value < -1 || value > 1) return fail(c, CSI_ERR_INVALID_ARG, "xcd_order must be -1 (auto), 0 or 1");
        drop_graphs(c);
        c->xcd_order = (int)value;
    } else if (n == "ls_fft_first_max") {
        if (value < 0 || value > 64) return fail(c, CSI_ERR_INVALID_ARG, "ls_fft_first_max must be 0..64");
        c->ls_fft_first_max = (int)value;
        return ls_prepare(c);
    } else if (n == "ls_ringb_min") {
        if (value < 0 || value > 1024) return fail(c, CSI_ERR_INVALID_ARG, "ls_ringb_min must be 0..1024");
        c->ls_ringb_min = (int)value;
        return ls_prepare(c);
    } else if (n == "small_call_overlap") {
        c->small_call_overlap = value == 2 ? 2 : (value != 0);
    } else if (n == "small_ls_fused") {
        drop_graphs(c);
        c->small_ls_fused = value != 0;
    } else if (n == "small_fused") {
        drop_graphs(c);
        c->small_fused = value != 0;
    } else if (n == "small_rows") {
        if (value < 0 || value > 65536) return fail(c, CSI_ERR_INVALID_ARG, "small_rows must be 0..65536");
        drop_graphs(c);
        c->small_rows = (int)value;
    } else if (n == "small_rows_band") {
        if (value < 0 || value > 65536) return fail(c, CSI_ERR_INVALID_ARG, "small_rows_band must be 0..65536");
        drop_graphs(c);
        c->small_rows_band = (int)value;
    } else if (n == "f32_engine") {
        if (value < -1 || value > 1) return fail(c, CSI_ERR_INVALID_ARG, "f32_engine must be -1 (automatic), 0 (fp32 MFMA) or 1 (split f16)");
        drop_graphs(c);
        c->f32_engine = (int)value;
    } else if (n == "hs_blocked") {
        drop_graphs(c);
        c->hs_blocked = value != 0;
    } else if (n == "hs_fuse_regressor") {
        drop_graphs(c);
        c->hs_fuse_regressor = value != 0;
    } else if (n == "hs_band") {
        if (value < 0 || value > 3)
            return fail(c, CSI_ERR_INVALID_ARG, "hs_band must be 0 (separate kernels), 1 (band kernel where its LDS-staged form applies, every fp32 shape it serves), "
                                                "2 (bf16 contexts: also the form with per-lane loads, any nt) or 3 (fp32 contexts: only that form; A/B runs)");
        drop_graphs(c);
        c->hs_band = (int)value;
    } else if (n == "band4") {
        drop_graphs(c);
        c->band4 = value != 0;
    } else if (n == "l0_stream") {
        drop_graphs(c);
        c->l0_stream = value != 0;
    } else if (n == "l0_stream_max_rows") {
        if (value < 8 || value > 4096) return fail(c, CSI_ERR_INVALID_ARG, "l0_stream_max_rows must be 8..4096");
        drop_graphs(c);
        c->l0_stream_max_rows = (int)value;
    } else if (n == "l0_stream_prepass_rows") {
        if (value < 0 || value > 256) return fail(c, CSI_ERR_INVALID_ARG, "l0_stream_prepass_rows must be 0..256");
        drop_graphs(c);
        c->l0_stream_prepass_rows = (int)value;
    } else if (n == "l0_stream_ks") {
        if (value < 0 || value > 256) return fail(c, CSI_ERR_INVALID_ARG, "l0_stream_ks must be 0 (automatic) .. 256");
        drop_graphs(c);
        c->l0_stream_ks = (int)value;
    } else if (n == "aux_fork_early") {
        drop_graphs(c);
        c->aux_fork_early = value != 0;
    } else if (n == "band_split") {
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4)
            return fail(c, CSI_ERR_INVALID_ARG, "band_split must be -1 (automatic), 0 / 1 (never) or 2 / 4 (column splits of every band)");
        drop_graphs(c);
        c->band_split = (int)value;
    } else if (n == "hs_min_blocks") {
        if (value < 1 || value > 65536) return fail(c, CSI_ERR_INVALID_ARG, "hs_min_blocks must be 1..65536");
        drop_graphs(c);
        c->hs_min_blocks = (int)value;
    } else if (n == "hs_act_shift" || n == "hs_in_shift") {
        if (value != HS_SHIFT_AUTO && (value < -8 || value > 14)) return fail(c, CSI_ERR_INVALID_ARG, "%s must be -8..14 or 99 (automatic)", name);
        drop_graphs(c);
        (n == "hs_act_shift" ? c->hs_act_shift : c->hs_in_shift) = (int)value;
    } else if (n == "band_tail_split") {
        drop_graphs(c);
        c->band_tail_split = value != 0;
    } else if (n == "bf16_l0_fused_split") {
        drop_graphs(c);
        c->bf16_l0_fused_split = value != 0;
    } else if (n == "bf16_fused_h1") {
        drop_graphs(c);
        c->bf16_fused_h1 = value != 0;
    } else if (n == "host_threads") {
        if (value < 0 || value > 256) return fail(c, CSI_ERR_INVALID_ARG, "host_threads must be 0 (automatic) .. 256");
        if (c->hostpipe) { delete c->hostpipe; c->hostpipe = nullptr; }
        c->host_threads = (int)value;
    } else if (n == "hs_vm_cast" || n == "hs_vm_pair") {
        if (value < 0 || value > 3) return fail(c, CSI_ERR_INVALID_ARG, "%s must be 0 ... 3", name);
        drop_graphs(c);
        (n == "hs_vm_cast" ? c->hs_vm_cast : c->hs_vm_pair) = (int)value;
    } else if (n == "ls_overlap_cus" || n == "ls_overlap_stride") {
        if (value < 0 || value > 255) return fail(c, CSI_ERR_INVALID_ARG, "%s must be 0 (LS in front of the DNN kernels on one stream) .. 255", name);
#ifndef CSI_LS_RACE_VARIANTS
        // measured slower than the serial order and it puts LS workgroups beside other kernels' MFMA waves - the one
        // condition under which a packed add of the LS transform was ever seen wrong.  Not part of the shipped library.
        if (n == "ls_overlap_cus" && value > 0)
            return fail(c, CSI_ERR_INVALID_ARG, "ls_overlap_cus > 0 (the LS kernel on a CU-masked side stream beside the matrix kernels) is an experiment that "
                                                "measured slower than the serial order and is not part of the product build; "
                                                "rebuild with CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS to run it");
#endif
        drop_graphs(c);
        (n == "ls_overlap_cus" ? c->ls_overlap_cus : c->ls_overlap_stride) = (int)value;
    } else if (n == "hp_side_threads") {
        if (c->hostpipe) { delete c->hostpipe; c->hostpipe = nullptr; }
        c->hp_side_threads = value != 0;
    } else if (n == "hp_device_weave") {
        c->hp_device_weave = value != 0;
    } else if (n == "hp_chunk_packets") {
        if (value < 0 || value > (1 << 20)) return fail(c, CSI_ERR_INVALID_ARG, "hp_chunk_packets must be 0 (automatic) .. 2^20");
        c->hp_chunk_packets = (int)value;
    } else if (n == "ls_fast_perm") {
        c->ls_fast_perm = value != 0;
        drop_graphs(c);
        return ls_prepare(c);
    } else if (n == "ls_v2") {
        c->ls_v2 = (int)value;
        return ls_prepare(c);
    } else if (n == "ls_debug") {
        c->ls_debug = (int)value;          // timing experiments: results are wrong when non-zero (bits 1 / 2 / 4)
        return ls_prepare(c);              // bits 0x200 ... 0x1000 select another instantiation (ls_ringb_shape)
    } else if (n == "ls_kernel") {
        if (value < 0 || value > 7)
            return fail(c, CSI_ERR_INVALID_ARG, "ls_kernel must be 0 (auto), 1 (FFT first), 2 (chunked), 3 (despread first), 4 (Walsh-Hadamard), "
                                                "5 (Walsh-Hadamard, LDS-DMA fed), 6 (generic P, LDS-DMA fed, fp32 despread) or 7 (generic P, bf16-split despread)");
        c->ls_kernel = (int)value;
        return ls_prepare(c);
    } else {
        return fail(c, CSI_ERR_INVALID_ARG, "csi_set_option: unknown option '%s'", name);
    }
    return CSI_OK;
}


// ---------------------------------------------------------------- on-box fine-tuning
static int trainer_of(csi_ctx* c, int model, const char* fn, csi_trainer** t) {
    if (model < 0 || model > 1) return fail(c, CSI_ERR_INVALID_ARG, "%s: model must be 0 or 1", fn);
    if (!c->trainer[model]) return fail(c, CSI_ERR_NOT_READY, "%s: csi_train_begin was not called for model %d", fn, model);
    *t = c->trainer[model];
    return CSI_OK;
}

int csi_train_begin(csi_ctx* c, int model, const csi_train_config* tc, const csi_tensor* tensors, int n) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (model < 0 || model > 1 || !tc || n < 0 || (n > 0 && !tensors)) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_begin: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int rc = tr_begin(c, model, tc, tensors, n);
    if (rc && c->trainer[model]) { tr_free(c->trainer[model]); c->trainer[model] = nullptr; }
    return rc;
}

int csi_train_step(csi_ctx* c, int model, const float* x, const float* y, int64_t B, float noise_std, float* loss) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_step", &t)) return rc0;
    if (!x || !y || B < 2 || B > (1 << 20) || noise_std < 0.f) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_step: bad argument (2 <= B <= 2^20)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int rc = tr_backward(c, t, x, y, nullptr, (int)B, noise_std, nullptr);
    if (rc) return rc;
    const int rc2 = tr_apply(c, t);
    if (rc2) return rc2;
    if (loss) {
        HIP_TRY(c, hipMemcpyAsync(loss, t->loss, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

int csi_train_backward(csi_ctx* c, int model, const float* x, const float* y, int64_t B, float noise_std, float* loss) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_backward", &t)) return rc0;
    if (!x || !y || B < 2 || B > (1 << 20) || noise_std < 0.f) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_backward: bad argument (2 <= B <= 2^20)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return tr_backward(c, t, x, y, nullptr, (int)B, noise_std, loss);
}

int csi_train_grads(csi_ctx* c, int model, float** d_grads, int64_t* count) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_grads", &t)) return rc0;
    if (!d_grads || !count) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_grads: bad argument");
    *d_grads = t->gflat;
    *count = t->gcount;
    return CSI_OK;
}

int csi_train_apply(csi_ctx* c, int model) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_apply", &t)) return rc0;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return tr_apply(c, t);
}

int csi_train_eval(csi_ctx* c, int model, const float* x, const float* y, int64_t B, float* loss) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_eval", &t)) return rc0;
    if (!x || !y || !loss || B < 1 || B > (1 << 20)) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_eval: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return tr_eval(c, t, x, y, nullptr, (int)B, loss);
}

int csi_train_set_dataset(csi_ctx* c, int model, const float* ltf_table, int64_t n_rows, const int32_t* ltf_row, const int32_t* itx,
                          const float* y, int64_t N) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_set_dataset", &t)) return rc0;
    if (!ltf_table || !ltf_row || !y || n_rows <= 0 || N <= 0 || N > 0x7fffffff || (c->cfg.nt > 0 && !itx))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_train_set_dataset: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return tr_set_dataset(c, t, ltf_table, n_rows, ltf_row, itx, y, N);
}

// mode 0: step (backward + Adam), 1: backward only, 2: inference-mode loss
int csi_train_indexed(csi_ctx* c, int model, int mode, const int32_t* ids, int64_t B, float noise_std, float* loss) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_indexed", &t)) return rc0;
    if (!ids || mode < 0 || mode > 2 || B < (mode == 2 ? 1 : 2) || B > (1 << 20) || noise_std < 0.f || (mode == 2 && !loss))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_train_indexed: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (mode == 2) return tr_eval(c, t, nullptr, nullptr, ids, (int)B, loss);
    int rc = tr_backward(c, t, nullptr, nullptr, ids, (int)B, noise_std, mode == 1 ? loss : nullptr);
    if (rc || mode == 1) return rc;
    rc = tr_apply(c, t);
    if (rc) return rc;
    if (loss) {
        HIP_TRY(c, hipMemcpyAsync(loss, t->loss, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

int csi_train_set_lr(csi_ctx* c, int model, float lr) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_set_lr", &t)) return rc0;
    if (!(lr > 0.f)) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_set_lr: lr must be positive");
    t->tc.lr = lr;
    return CSI_OK;
}

int csi_train_get(csi_ctx* c, int model, const char* name, float* out, int64_t count) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_get", &t)) return rc0;
    if (!name || !out || count <= 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_train_get: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return tr_get(c, t, name, out, count);
}

int csi_train_end(csi_ctx* c, int model, int commit) {
    if (!c) return CSI_ERR_INVALID_ARG;
    csi_trainer* t = nullptr;
    if (int rc0 = trainer_of(c, model, "csi_train_end", &t)) return rc0;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = CSI_OK;
    if (commit) {
        // hand the trained tensors to the inference model exactly as a caller of csi_load_weights would
        const csi_config& cf = c->cfg;
        std::vector<std::string> names;
        std::vector<std::vector<float>> data;
        std::vector<std::pair<int64_t, int64_t>> shape;
        auto add = [&](const std::string& nm, int64_t r, int64_t q) {
            names.push_back(nm);
            shape.push_back({r, q});
            data.emplace_back((size_t)(r * q));
        };
        for (int li = 0; li <= cf.n_hidden; ++li) {
            const auto& l = t->layers[li];
            const std::string base = li == cf.n_hidden ? std::string("fc_regressor") : "fc_dense" + std::to_string(li);
            add(base + ".kernel", l.in, l.out);
            add(base + ".bias", 1, l.out);
            if (li < cf.n_hidden && cf.use_bn)
                for (const char* sfx : {".gamma", ".beta", ".moving_mean", ".moving_variance"}) add("bn" + std::to_string(li) + sfx, 1, l.out);
        }
        std::vector<csi_tensor> ts(names.size());
        for (size_t i = 0; i < names.size() && !rc; ++i) {
            rc = tr_get(c, t, names[i].c_str(), data[i].data(), (int64_t)data[i].size());
            ts[i] = csi_tensor{names[i].c_str(), data[i].data(), shape[i].first, shape[i].second};
        }
        if (!rc) rc = csi_load_weights(c, model, ts.data(), (int)ts.size());
    }
    hipStreamSynchronize(c->stream);
    tr_free(t);
    c->trainer[model] = nullptr;
    return rc;
}

// ---- multi-GPU: RCCL weight broadcast inside the C-ABI (csi_comm.hpp)
int csi_get_unique_id(char id[CSI_UNIQUE_ID_BYTES]) {
    if (!id) return fail(nullptr, CSI_ERR_INVALID_ARG, "csi_get_unique_id: null buffer");
    RcclApi& r = rccl();
    if (!r.lib) return fail(nullptr, CSI_ERR_HIP, "%s", r.why.c_str());
    nccl_uid u;
    NCCL_TRY(nullptr, r.GetUniqueId(&u));
    std::memcpy(id, u.internal, CSI_UNIQUE_ID_BYTES);
    return CSI_OK;
}

int csi_comm_init(csi_ctx* c, int rank, int world, const char id[CSI_UNIQUE_ID_BYTES]) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, CSI_ERR_INVALID_ARG, "csi_comm_init: bad rank %d / world %d", rank, world);
    RcclApi& r = rccl();
    if (!r.lib) return fail(c, CSI_ERR_HIP, "%s", r.why.c_str());
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    comm_free(c);
    c->comm = new csi_comm;
    c->comm->rank = rank;
    c->comm->world = world;
    nccl_uid u;
    std::memcpy(u.internal, id, CSI_UNIQUE_ID_BYTES);
    NCCL_TRY(c, r.CommInitRank(&c->comm->comm, world, u, rank));
    HIP_TRY(c, hipMalloc(&c->comm->wire, WIRE_STATUS_OFF + 64));
    return CSI_OK;
}

int csi_comm_destroy(csi_ctx* c) {
    if (!c) return CSI_ERR_INVALID_ARG;
    hipSetDevice(c->cfg.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    comm_free(c);
    return CSI_OK;
}

int csi_broadcast_weights(csi_ctx* c, int root) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!c->comm || !c->comm->comm) return fail(c, CSI_ERR_NOT_READY, "csi_broadcast_weights: csi_comm_init has not been called");
    csi_comm& cm = *c->comm;
    if (root < 0 || root >= cm.world) return fail(c, CSI_ERR_INVALID_ARG, "csi_broadcast_weights: root %d of %d ranks", root, cm.world);
    RcclApi& r = rccl();
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    const bool is_root = cm.rank == root;
    // 1. the host-side scalars that belong to the device blobs: root -> everybody
    WireMeta w;
    std::memset(&w, 0, sizeof w);
    if (is_root) {
        wire_fill(c, w);
        HIP_TRY(c, hipMemcpyAsync(cm.wire, &w, sizeof w, hipMemcpyHostToDevice, c->stream));
    }
    NCCL_TRY(c, r.Broadcast(cm.wire, cm.wire, sizeof w, NCCL_CHAR, root, cm.comm, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&w, cm.wire, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // 2. receivers: the record against the own csi_config, the old model out, buffers of the root's sizes
    std::vector<WBlob> blobs;
    int rc_local = CSI_OK;
    if (is_root) wire_blobs(c, w, blobs);
    else rc_local = wire_receive(c, w, blobs, "csi_broadcast_weights");
    // 3. every rank learns whether EVERY rank can take the blobs (one min all-reduce of a status word): a refusing rank - other
    //    csi_config, allocation failure - must not leave the others inside the grouped broadcast, and it must not enter it itself
    int32_t* d_status = reinterpret_cast<int32_t*>(static_cast<char*>(cm.wire) + WIRE_STATUS_OFF);
    int32_t ok = rc_local == CSI_OK ? 1 : 0;
    const std::string why_local = c->err;
    HIP_TRY(c, hipMemcpyAsync(d_status, &ok, sizeof ok, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(c, r.AllReduce(d_status, d_status, 1, NCCL_INT32, NCCL_MIN, cm.comm, c->stream));
    int32_t all_ok = 0;
    HIP_TRY(c, hipMemcpyAsync(&all_ok, d_status, sizeof all_ok, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    cm.bytes_broadcast = 0;
    cm.blobs_broadcast = 0;
    if (rc_local != CSI_OK) { c->err = why_local; return rc_local; }
    if (!all_ok) {
        if (!is_root) wire_drop_receiver(c);
        return fail(c, CSI_ERR_INVALID_ARG, "csi_broadcast_weights: another rank refused the root's record (its csi_config differs or its allocation failed; "
                                             "see csi_last_error there) - nothing was broadcast");
    }
    // 4. the blobs themselves, device to device, one group.  The group is always closed, also on an error inside it
    cm.blobs_broadcast = (int64_t)blobs.size();
    int rc_g = r.GroupStart();
    if (rc_g == 0) {
        int rc_b = 0;
        for (WBlob& b : blobs) {
            rc_b = r.Broadcast(*b.p, *b.p, b.bytes, NCCL_CHAR, root, cm.comm, c->stream);
            if (rc_b) break;
            cm.bytes_broadcast += (int64_t)b.bytes;
        }
        const int rc_e = r.GroupEnd();
        rc_g = rc_b ? rc_b : rc_e;
    }
    if (rc_g) {
        if (!is_root) wire_drop_receiver(c);
        return fail(c, CSI_ERR_HIP, "csi_broadcast_weights: ncclBroadcast group failed: %s", r.GetErrorString ? r.GetErrorString(rc_g) : "?");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // 5. what is derived locally
    if (!is_root) {
        const int rc = wire_finish(c, w);
        if (rc) { const std::string why = c->err; wire_drop_receiver(c); c->err = why; return rc; }
    }
    return CSI_OK;
}

// The same protocol inside one process: dst takes what src holds, device to device.  Runs every line of the receiver side of
// csi_broadcast_weights (wire_receive / wire_blobs / wire_finish) - which a single-GPU box cannot reach through RCCL, two ranks
// on one GPU being refused - and is useful by itself: a second context (another stream, another packet range, another GPU of the
// same process) without a second pass through csi_load_weights and the host copies of the tensors.
int csi_clone_weights(csi_ctx* dst, const csi_ctx* src_c) {
    if (!dst) return CSI_ERR_INVALID_ARG;
    csi_ctx* src = const_cast<csi_ctx*>(src_c);        // only pointer VALUES are read from it
    if (!src || src == dst) return fail(dst, CSI_ERR_INVALID_ARG, "csi_clone_weights: source context is null or the destination itself");
    HIP_TRY(dst, hipSetDevice(src->cfg.device));
    HIP_TRY(dst, hipStreamSynchronize(src->stream));   // whatever still writes the source's buffers (csi_train_end, a pilot table build)
    HIP_TRY(dst, hipSetDevice(dst->cfg.device));
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    drop_graphs(dst);
    WireMeta w;
    wire_fill(src, w);
    std::vector<WBlob> to, from;
    int rc = wire_receive(dst, w, to, "csi_clone_weights");
    if (rc) return rc;
    wire_blobs(src, w, from);
    if (to.size() != from.size()) { wire_drop_receiver(dst); return fail(dst, CSI_ERR_INVALID_ARG, "csi_clone_weights: internal - %zu buffers here, %zu on the source", to.size(), from.size()); }
    for (size_t i = 0; i < to.size(); ++i) {
        if (to[i].bytes != from[i].bytes || !*from[i].p) { wire_drop_receiver(dst); return fail(dst, CSI_ERR_INVALID_ARG, "csi_clone_weights: internal - buffer %zu differs", i); }
        const hipError_t e = src->cfg.device == dst->cfg.device
                                 ? hipMemcpyAsync(*to[i].p, *from[i].p, to[i].bytes, hipMemcpyDeviceToDevice, dst->stream)
                                 : hipMemcpyPeerAsync(*to[i].p, dst->cfg.device, *from[i].p, src->cfg.device, to[i].bytes, dst->stream);
        if (e != hipSuccess) { wire_drop_receiver(dst); return fail(dst, CSI_ERR_HIP, "csi_clone_weights: device copy failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    rc = wire_finish(dst, w);
    if (rc) { const std::string why = dst->err; wire_drop_receiver(dst); dst->err = why; return rc; }
    return CSI_OK;
}

int csi_synchronize(csi_ctx* c) {
    if (!c) return CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float hit = 0.f;
    bool low = false;
    int rc = hs_range_check(c, &hit, &low);
    if (rc) return rc;
    if (hit > 0.f)
        return fail(c, CSI_ERR_RANGE, "split-f16 engine: an operand reached %.4g after scaling (f16 limit 65504) in a device-pointer call since the "
                    "last check - the outputs of those calls are not valid; lower hs_in_shift / hs_act_shift or set f32_engine to 0 and run them again", (double)hit);
    if (low)
        return fail(c, CSI_ERR_RANGE, "split-f16 engine: a row of operands stayed below %.3g after scaling in a device-pointer call since the last check - "
                    "its results may miss the 1e-5 contract; raise hs_in_shift / hs_act_shift or set f32_engine to 0 and run those calls again", (double)HS_LOW_REPORT);
    return CSI_OK;
}

// Range guard of the split-f16 engine behind a host-buffer entry point (the stream is synchronised, the caller's buffers
// are still here): if an operand left the f16 range, or a row sat in its denormals, run `again` on the fp32 MFMA kernels.
// The retry must not replay a hipGraph captured with the split-engine kernels (the host pipeline calls the device
// entry points with recurring buffers and chunk sizes, i.e. recurring graph keys): graphs are dropped and graph use is
// off for its duration; and the guard is read once more behind it - nothing of the split engine may have run.
static int range_guard_retry(csi_ctx* c, const std::function<int()>& again) {
    float hit = 0.f;
    bool low = false;
    int rc = hs_range_check(c, &hit, &low);
    if (rc || (hit == 0.f && !low)) return rc;
    ++c->hs_range_fallbacks;
    const int engine = c->f32_engine;
    const bool graph = c->use_graph;
    drop_graphs(c);
    c->f32_engine = 0;
    c->use_graph = false;
    const int64_t launches = c->hs_launches;
    rc = again();
    c->f32_engine = engine;
    c->use_graph = graph;
    drop_graphs(c);
    if (rc) return rc;
    if (c->hs_launches != launches)
        return fail(c, CSI_ERR_RANGE, "range-guard retry: the split-f16 engine ran again although f32_engine was 0");
    return CSI_OK;
}

// ---- host-buffer entry points: two-slot pipeline of csi_hostpipe.hpp
static int host_packets(csi_ctx* c, const float* re, const float* im, int64_t npkt, float* o_re, float* o_im,
                        int n_out, bool ls) {
    return hp_packets(c, re, im, npkt, o_re, o_im, n_out,
                      [c, ls](const float* d_re, const float* d_im, int64_t np, float* d_ore, float* d_oim) {
                          return ls ? csi_ls_estimate_device(c, d_re, d_im, np, d_ore, d_oim) : csi_predict_device(c, d_re, d_im, np, d_ore, d_oim);
                      });
}

int csi_predict(csi_ctx* c, const float* ltf_re, const float* ltf_im, int64_t npkt, float* out_re, float* out_im) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_re || !ltf_im || !out_re || !out_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_predict: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    rc = host_packets(c, ltf_re, ltf_im, npkt, out_re, out_im, c->cfg.n_out, false);
    if (rc) return rc;
    // range guard of the split-f16 engine: an operand beyond the f16 range, or a row of operands deep in
    // its denormal range -> the same call again on the
    // fp32 MFMA kernels (the caller's buffers are still here), so that this entry point never returns inf
    return range_guard_retry(c, [&] { return host_packets(c, ltf_re, ltf_im, npkt, out_re, out_im, c->cfg.n_out, false); });
}

int csi_estimate_c128(csi_ctx* c, const double* ltf_c128, int64_t npkt, float* dnn_c64, float* ls_c64) {
    int rc = check_ready(c, dnn_c64 != nullptr);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_c128 || (!dnn_c64 && !ls_c64))))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_estimate_c128: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    rc = hp_estimate_c128(c, ltf_c128, false, npkt, dnn_c64, ls_c64);
    if (rc || !dnn_c64) return rc;
    // range guard of the split-f16 engine, as in csi_predict: repeat the DNN on the fp32 MFMA kernels
    return range_guard_retry(c, [&] { return hp_estimate_c128(c, ltf_c128, false, npkt, dnn_c64, nullptr); });
}

int csi_estimate_c64(csi_ctx* c, const float* ltf_c64, int64_t npkt, float* dnn_c64, float* ls_c64) {
    int rc = check_ready(c, dnn_c64 != nullptr);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_c64 || (!dnn_c64 && !ls_c64))))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_estimate_c64: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    rc = hp_estimate_c128(c, ltf_c64, true, npkt, dnn_c64, ls_c64);
    if (rc || !dnn_c64) return rc;
    return range_guard_retry(c, [&] { return hp_estimate_c128(c, ltf_c64, true, npkt, dnn_c64, nullptr); });
}

int csi_ls_estimate(csi_ctx* c, const float* ltf_re, const float* ltf_im, int64_t npkt, float* h_re, float* h_im) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (npkt < 0 || (npkt > 0 && (!ltf_re || !ltf_im || !h_re || !h_im)))
        return fail(c, CSI_ERR_INVALID_ARG, "csi_ls_estimate: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return host_packets(c, ltf_re, ltf_im, npkt, h_re, h_im, LS_NDATA, true);
}

int csi_predict_samples(csi_ctx* c, int model, const float* x, int64_t B, float* y) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (model < 0 || model > 1) return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_samples: model must be 0 or 1");
    Model& m = c->model[model];
    if (!m.loaded) return fail(c, CSI_ERR_NOT_READY, "weights of the %s model are not loaded", model ? "imag" : "real");
    if (B < 0 || (B > 0 && (!x || !y))) return fail(c, CSI_ERR_INVALID_ARG, "csi_predict_samples: bad argument");
    if (B == 0) return CSI_OK;
    const csi_config& cf = c->cfg;
    HIP_TRY(c, hipSetDevice(cf.device));
    int maxh = 0;
    for (int i = 0; i < cf.n_hidden; ++i) maxh = std::max(maxh, cf.hidden[i]);
    if (cf.dtype == CSI_DTYPE_BF16) {
        const int ldx = (c->d_in + B_BK - 1) / B_BK * B_BK;
        const size_t per_row_b = (size_t)c->d_in * 4 + (size_t)ldx * 2 + 2 * (size_t)maxh * 2 + (size_t)cf.n_out * 4;
        int64_t chunk_b = std::min<int64_t>(B, std::max<int64_t>(1, ((int64_t)512 << 20) / (int64_t)per_row_b));
        int rcb = ensure_bytes(c, &c->stage, &c->stage_bytes, per_row_b * (size_t)chunk_b + 1024);
        if (rcb) return rcb;
        char* base = c->stage;
        float* d_xf = reinterpret_cast<float*>(base);      base += (size_t)chunk_b * c->d_in * 4;
        float* d_yf = reinterpret_cast<float*>(base);      base += (size_t)chunk_b * cf.n_out * 4;
        bf16_t* d_xb = reinterpret_cast<bf16_t*>(base);    base += (size_t)chunk_b * ldx * 2;
        bf16_t* hb0 = reinterpret_cast<bf16_t*>(base);     base += (size_t)chunk_b * maxh * 2;
        bf16_t* hb1 = reinterpret_cast<bf16_t*>(base);
        for (int64_t r0 = 0; r0 < B; r0 += chunk_b) {
            const int nb = (int)std::min(chunk_b, B - r0);
            HIP_TRY(c, hipMemcpyAsync(d_xf, x + (size_t)r0 * c->d_in, (size_t)nb * c->d_in * 4, hipMemcpyHostToDevice, c->stream));
            {
                ProfScope ps(c, K_CAST_BF16, 0.0, 6.0 * nb * c->d_in);
                const unsigned blocks = (unsigned)std::min<size_t>(((size_t)nb * ldx + 255) / 256, 8192);
                hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, d_xf, d_xb, nb, c->d_in, ldx);
                HIP_TRY(c, hipGetLastError());
            }
            const Layer& l0 = m.layers[0];
            GemmBf16Args q{};
            q.A = d_xb; q.lda = ldx;
            q.Bt = l0.Wb; q.ldb = l0.ldwb;
            q.C = hb0; q.ldc = l0.out;
            q.M = nb; q.N = l0.out; q.K = c->d_in;
            q.bias = l0.bias; q.scale = l0.scale; q.shift = l0.shift;
            q.k_per_split = l0.ldwb;
            rcb = launch_gemm_bf16<EPI_BIAS_RELU_AFFINE, true>(c, K_NAIVE_DENSE0, q, 1);
            if (rcb) return rcb;
            // hidden layers 1.. ping-pong starting from hb1, so layer 1 never overwrites its input
            rcb = bf16_tail(c, m, hb0, nb, hb1, hb0, d_yf, 1);
            if (rcb) return rcb;
            HIP_TRY(c, hipMemcpyAsync(y + (size_t)r0 * cf.n_out, d_yf, (size_t)nb * cf.n_out * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        return CSI_OK;
    }
    const size_t per_row = ((size_t)c->d_in + 2 * (size_t)maxh + cf.n_out) * sizeof(float);
    int64_t chunk = std::max<int64_t>(1, ((int64_t)512 << 20) / (int64_t)per_row);
    chunk = std::min(chunk, B);
    int rc = ensure_bytes(c, &c->stage, &c->stage_bytes, per_row * (size_t)chunk + G_SLACK_FLOATS * sizeof(float));
    if (rc) return rc;
    float* d_x = reinterpret_cast<float*>(c->stage);
    float* hb[2];
    // the staging buffer is shared with csi_nmse / csi_lmmse_estimate (raw caller data): keep G_SLACK_FLOATS of zeros
    // behind the input rows, the K tail of the last row (d_in is rarely a multiple of the k-tile) reads into them
    hb[0] = d_x + (size_t)chunk * c->d_in + G_SLACK_FLOATS;
    hb[1] = hb[0] + (size_t)chunk * maxh;
    float* d_y = hb[1] + (size_t)chunk * maxh;
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t nb = std::min(chunk, B - r0);
        HIP_TRY(c, hipMemcpyAsync(d_x, x + (size_t)r0 * c->d_in, (size_t)nb * c->d_in * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(d_x + (size_t)nb * c->d_in, 0, G_SLACK_FLOATS * sizeof(float), c->stream));
        const float* cur = d_x;
        int cur_ld = c->d_in, w = 0;
        for (int li = 0; li <= cf.n_hidden; ++li) {
            const Layer& l = m.layers[li];
            GemmArgs q{};
            q.A = cur; q.lda = cur_ld;
            q.Bt = l.Wt; q.ldb = l.ldw;
            q.M = (int)nb; q.N = l.out; q.K = l.in;
            q.bias = l.bias; q.scale = l.scale; q.shift = l.shift;
            q.k_per_split = ((l.in + G_BK - 1) / G_BK) * G_BK;
            if (li == cf.n_hidden) {
                q.C = d_y; q.ldc = cf.n_out;
                rc = launch_gemm<EPI_BIAS>(c, K_REGRESSOR, q, 1);
            } else {
                q.C = hb[w]; q.ldc = l.out;
                rc = launch_gemm<EPI_BIAS_RELU_AFFINE>(c, li == 0 ? K_NAIVE_DENSE0 : K_DENSE_HIDDEN, q, 1);
                cur = hb[w];
                cur_ld = l.out;
                w ^= 1;
            }
            if (rc) return rc;
        }
        HIP_TRY(c, hipMemcpyAsync(y + (size_t)r0 * cf.n_out, d_y, (size_t)nb * cf.n_out * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CSI_OK;
}

// ---- device memory plumbing
int csi_device_malloc(csi_ctx* c, void** dptr, int64_t bytes) {
    if (!c || !dptr || bytes < 0) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_device_malloc: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    *dptr = nullptr;
    if (bytes == 0) return CSI_OK;
    if (hipMalloc(dptr, (size_t)bytes) != hipSuccess) {
        *dptr = nullptr;
        return fail(c, CSI_ERR_NOMEM, "csi_device_malloc: %lld bytes failed", (long long)bytes);
    }
    return CSI_OK;
}

int csi_device_free(csi_ctx* c, void* dptr) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!dptr) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(dptr));
    return CSI_OK;
}

int csi_host_malloc(csi_ctx* c, void** ptr, int64_t bytes) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (!ptr || bytes <= 0) return fail(c, CSI_ERR_INVALID_ARG, "csi_host_malloc: bad argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, CSI_ERR_NOMEM, "csi_host_malloc: %lld bytes of pinned host memory are not available", (long long)bytes);
    }
    return CSI_OK;
}

int csi_host_free(csi_ctx* c, void* ptr) {
    if (!c) {          // a buffer may outlive the context that allocated it (python finalizers run in any order)
        if (ptr && hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); return CSI_ERR_HIP; }
        return CSI_OK;
    }
    if (ptr) HIP_TRY(c, hipHostFree(ptr));
    return CSI_OK;
}

int csi_memcpy_h2d(csi_ctx* c, void* dst_dev, const void* src_host, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst_dev || !src_host))) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_memcpy_h2d: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

int csi_memcpy_d2h(csi_ctx* c, void* dst_host, const void* src_dev, int64_t bytes) {
    if (!c || bytes < 0 || (bytes > 0 && (!dst_host || !src_dev))) return c ? fail(c, CSI_ERR_INVALID_ARG, "csi_memcpy_d2h: bad argument") : CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSI_OK;
}

int csi_synth_white(csi_ctx* c, uint64_t seed, int64_t first_pkt, int64_t npkt, float* d_re, float* d_im) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (npkt < 0 || first_pkt < 0 || (npkt > 0 && (!d_re || !d_im))) return fail(c, CSI_ERR_INVALID_ARG, "csi_synth_white: bad argument");
    if (npkt == 0) return CSI_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t per_pkt = (size_t)c->cfg.nr * c->cfg.len_ltf;
    const size_t n = per_pkt * (size_t)npkt;
    ProfScope ps(c, K_SYNTH_WHITE, 0.0, 8.0 * n);
    hipLaunchKernelGGL(synth_white_kernel, dim3(2048), dim3(256), 0, c->stream, seed, (uint64_t)first_pkt * per_pkt, n, d_re, d_im);
    HIP_TRY(c, hipGetLastError());
    return CSI_OK;
}

// ---- profiling
int csi_profile_enable(csi_ctx* c, int on) {
    if (!c) return CSI_ERR_INVALID_ARG;
    int rc = prof_collect(c);
    c->prof_on = on != 0;
    return rc;
}

int csi_profile_reset(csi_ctx* c) {
    if (!c) return CSI_ERR_INVALID_ARG;
    int rc = prof_collect(c);
    for (int i = 0; i < K_COUNT; ++i) {
        c->prof_ms[i] = 0;
        c->prof_launches[i] = 0;
        c->prof_flops[i] = 0;
        c->prof_bytes[i] = 0;
    }
    return rc;
}

int csi_profile_band_skeleton(csi_ctx* c, int64_t rows, int iters, double* ms_per_launch, double* executed_flops) {
    if (!c) return CSI_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return band_skeleton_time(c, rows, iters, ms_per_launch, executed_flops);
}

int csi_profile_pcie(csi_ctx* c, int64_t h2d_bytes, int64_t d2h_bytes, double* ms_h2d, double* ms_d2h, double* ms_both) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (h2d_bytes < 0 || d2h_bytes < 0 || (h2d_bytes == 0 && d2h_bytes == 0)) return fail(c, CSI_ERR_INVALID_ARG, "csi_profile_pcie: bad byte counts");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return hp_pcie_probe(c, h2d_bytes, d2h_bytes, ms_h2d, ms_d2h, ms_both);
}

int csi_profile_num_kernels(void) { return K_COUNT; }

const char* csi_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }

int csi_profile_query(csi_ctx* c, int id, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    if (!c) return CSI_ERR_INVALID_ARG;
    if (id < 0 || id >= K_COUNT) return fail(c, CSI_ERR_INVALID_ARG, "csi_profile_query: kernel id %d out of range", id);
    int rc = prof_collect(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof_ms[id];
    if (launches) *launches = c->prof_launches[id];
    if (flops) *flops = c->prof_flops[id];
    if (bytes) *bytes = c->prof_bytes[id];
    return CSI_OK;
}

#if (CSI_LS_VAR_DEFAULT) & 512
// race-hunt build only (tools/ls_opsel_hunt.sh): the log of packed +-i rotations that differed from the scalar form
int csi_debug_opsel_log(unsigned* host, int n_dwords, int reset) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && host) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(csi::g_opsel_log), (size_t)n_dwords * 4);
    if (e == hipSuccess && reset) { const unsigned z = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(csi::g_opsel_log), &z, 4); }
    return e == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
