// csi_context.hpp - the csi_ctx object behind include/csi_mamimo.h and the host-side plumbing every
// entry point shares: error reporting, per-kernel HIP-event profiling, device buffers, weight storage.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/csi_mamimo.h"
#include "gemm_f32.hip.h"
#include "gemm_bf16.hip.h"
#include "gemm_hs.hip.h"
#include "gemm_hs_band.hip.h"
#include "l0_hs_stream.hip.h"
#include "ls_estimate.hip.h"
#include "lmmse.hip.h"
#include "metrics.hip.h"

using namespace csi;

namespace {

enum KernelId {
    K_LAYER0_LTF = 0,    // layer 0, LTF part, once per (packet, rx)
    K_SPLITK_REDUCE,     // deterministic split-K combine of layer 0
    K_PAIR_DENSE,        // first per-pair layer, h1 generated in the prologue  (dominant)
    K_DENSE_HIDDEN,      // further hidden layers
    K_REGRESSOR,         // fc_regressor
    K_LS_ESTIMATE,       // FFT + despread
    K_NAIVE_DENSE0,      // un-shared layer 0 of csi_predict_samples
    K_SYNTH_WHITE,
    K_PILOT_TABLE,
    K_CAST_BF16,         // fp32 -> bf16 of the preambles (bf16 mode)
    K_PAIR_H1_BF16,      // materialise h1 in bf16 (bf16 mode)
    K_LMMSE,             // Levinson solve of the LMMSE smoother
    K_TRAIN_GEMM,        // forward / dgrad / wgrad products of csi_train_step
    K_TRAIN_ELEMWISE,    // BatchNormalization, dropout, loss, Adam of csi_train_step
    K_NMSE,              // per-link NMSE metric
    K_PAIR_DENSE_TAIL,   // the last, partly filled round of band workgroups launched in column splits ("band_tail_split", round 6)
    K_COUNT
};
const char* const kKernelNames[K_COUNT] = {
    "layer0_ltf_gemm", "splitk_reduce", "pair_dense_gemm", "dense_hidden_gemm", "regressor_gemm",
    "ls_estimate", "naive_dense0_gemm", "synth_white", "pilot_table", "cast_bf16", "pair_h1_bf16", "lmmse_levinson",
    "train_gemm", "train_elementwise", "nmse_links", "pair_dense_tail"};

thread_local std::string g_create_error;

constexpr int CSI_WIRE_MAX_NT = 128;      // Hadamard-equivalent pilots are recognised up to this many antennas (the Walsh-Hadamard LS kernels' range)
constexpr int HS_SHIFT_AUTO = 99;         // "hs_in_shift" / "hs_act_shift" options: scale chosen from the data / from the model

struct Layer {
    float* Wt = nullptr;      // [out][ldw] (K-major, ldw = in rounded up to 32, zero padded)   fp32 mode
    int ldw = 0;
    bf16_t* Wb = nullptr;     // [out][ldwb] bf16 (K-major, ldwb = in rounded up to 64)          bf16 mode
    int ldwb = 0;
    bf16_t* Wb_p = nullptr;   // regressor behind one per-pair layer, bf16 mode: 256 rows, k permuted inside every group of 16 (band kernel)
    uint16_t* Wh_f = nullptr; // regressor behind ONE per-pair layer: Wh with that layer's BN scale folded into its rows (fused kernel)
    int wshift_f = 0;
    uint16_t* Wh_p = nullptr; // the same folded regressor weights, 256 rows (rows >= out zero), k permuted inside every group of 16
                              // (hs_band_kperm): the fused band kernel (gemm_hs_band.hip.h / band_kernel_gen.py)
    int ashift_pre = 4;       // like ashift for this layer's relu output BEFORE BatchNormalization (its scale lives in the next weights)
    uint16_t* Wh = nullptr;   // [out][ldwh] split-f16 ("hs", gemm_hs.hip.h) copy of Wt * 2^wshift; layer 0: LTF columns only
    int ldwh = 0;             // halves per row = 2 * (in rounded up to 16)
    int wshift = 0;
    int ashift = 4;           // split engine: this layer's OUTPUT activations are carried times 2^ashift (from its BN vectors at load);
                              // layer 0: its relu output BEFORE BatchNormalization (the scale lives in layer 1's split weights)
    float* bias = nullptr;    // [out]
    float* bias_hs = nullptr; // [out]  split engine (layers >= 1): bias + (BN shift of the previous layer) . W
    float* scale = nullptr;   // [out]  BN: gamma * rsqrt(var + eps)   (1 without BN)
    float* shift = nullptr;   // [out]  BN: beta - mean * scale        (0 without BN)
    int in = 0, out = 0;
};

struct Model {
    std::vector<Layer> layers;   // n_hidden dense layers + regressor (last)
    float* W0p = nullptr;        // [nt][H1] pilot rows of fc_dense0.kernel, row-major
    float* W0rm = nullptr;       // [lenLTF][H1] LTF rows of fc_dense0.kernel as stored (skinny layer-0 kernel)
    float* T = nullptr;          // [nt][H1] pilot table incl. bias
    float* T_hs = nullptr;       // [nt][H1] 2^T_hs_shift * T: what the split-f16 pair kernel adds (gemm_hs.hip.h)
    int T_hs_shift = HS_SHIFT_AUTO;   // HS_SHIFT_AUTO = not built
    float* T_sw = nullptr;       // T (bf16 contexts) / T_hs (split engine) in the slab order of the staged band kernels (band_tsw_kernel); rebuilt with the table
    bool T_sw_ok = false;
    uint16_t* Wt1 = nullptr;     // bf16 contexts: the pair layer's and the regressor's weights pre-tiled for the register-blocked band kernel
    uint16_t* Wt2 = nullptr;     // (band4_tile_kernel, built on first use from layers[1].Wb / layers[2].Wb_p)
    bool tiled_ok = false;
    float* l0_rowmax = nullptr;  // [4096] row maxima of a mid-size call's preambles (l0_row_max_kernel -> l0_hs_stream_kernel)
    bool loaded = false;
    bool table_ok = false;
    // csi_load_weights measures how well the split-f16 copies represent the fp32 matrices: the weight scale comes from max |w| of a
    // whole matrix, so a matrix whose bulk sits ~2^20 below its largest entry would lose its lo halves to the f16 denormals -
    // silently (the range guard watches activations).  Worst OUTPUT ROW ||w_o s - (hi + lo)_o|| / ||w_o s|| over the split matrices of
    // this model (per row: a whole-matrix norm is carried by the outlier itself); above 2^-20 the model is pinned to the fp32 MFMA
    // kernels (hs_static_ok) and "hs_weight_pins" counts it.
    double hs_repr_err = 0.0;
    bool hs_repr_ok = true;
};

struct GraphEntry {           // one captured csi_predict_device / csi_estimate_device call
    const void* in_re; const void* in_im; void* out_re; void* out_im;
    void* h_re; void* h_im;   // LS outputs (csi_estimate_device), null for csi_predict_device
    int64_t npkt;
    int seen;                 // eager runs with this key so far (capture happens on the 2nd call)
    hipGraphExec_t exec;
    int64_t hs_launches = 0;  // split-engine GEMMs inside the graph (a replay owes them to the range-guard bookkeeping)
};

struct ProfSpan {
    int id;
    hipEvent_t beg, end;
};

}  // namespace

struct csi_trainer;          // csi_train.hpp
struct csi_hostpipe;         // csi_hostpipe.hpp
struct csi_comm;             // csi_comm.hpp

struct csi_ctx {
    csi_config cfg;
    int d_in = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Model model[2];
    csi_trainer* trainer[2] = {nullptr, nullptr};   // on-box fine-tuning state per component model
    csi_hostpipe* hostpipe = nullptr;               // streams / pinned slots / host threads of the host-buffer entry points
    csi_comm* comm = nullptr;                       // RCCL communicator of csi_comm_init (weight broadcast)
    int host_threads = 0;                           // "host_threads" option: threads of the user <-> pinned copies (0 = automatic)
    int hp_side_threads = 1;                        // "hp_side_threads": 1 = input staging and result staging on their own threads beside the caller's enqueue loop, 0 = inline, in turn
    int hp_chunk_packets = 0;                       // "hp_chunk_packets": packets per pipeline slot of the host-buffer entry points (0 = automatic)
    int hp_device_weave = 1;                        // "hp_device_weave": csi_estimate_c128 with PINNED result arrays assembles the complex64 values on the device and downloads into the arrays themselves (0 = host threads weave, as for pageable arrays)
    int64_t hp_direct_out_calls = 0;                // "hp_direct_out_calls": calls that took that path
    float* P = nullptr;          // device [nt][nt]
    float* Pbf = nullptr;        // device: bf16 pieces of P in MFMA operand order (ls_pilot layout of ls_estimate_ringb_kernel), 2 per float
    float* Ppad = nullptr;       // device [ceil32(nt)][ceil32(nt)], zero padded (chunked LS kernel)
    bool pilot_ok = false;
    // LS constants
    float* tw = nullptr;         // [2][256]
    int* bin_pos = nullptr;      // [234]
    float* denom = nullptr;      // [234]
    // activation workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // layer-0 slabs + sum of the one-packet (skinny) path
    char* l0skinny = nullptr;
    size_t l0skinny_bytes = 0;
    // split-K slabs of the small-batch path
    char* skbuf = nullptr;
    size_t skbuf_bytes = 0;
    // second stream + scratch set so that the real and the imag model of a SMALL call run side by side
    // (a one-packet call is launch-latency bound: 12 short kernels in a row; see csi_predict_device)
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    char *aux_ws = nullptr, *aux_l0skinny = nullptr, *aux_skbuf = nullptr, *aux_fuse_ws = nullptr;
    size_t aux_ws_bytes = 0, aux_l0skinny_bytes = 0, aux_skbuf_bytes = 0, aux_fuse_ws_bytes = 0;
    int small_call_overlap = 1;  // "small_call_overlap" option
    // the one-packet regime (csi_dnn_small.hpp): both models of a call of <= 8 rx preambles in 1 + n_hidden launches
    int small_ls_fused = 1;      // "small_ls_fused": a one-packet csi_estimate_device call runs its LS estimate inside the layer-0 launch (small_l0_ls_kernel)
    float* small_ls_h_re = nullptr;   // set by csi_estimate_device around its predict call, consumed by predict_small
    float* small_ls_h_im = nullptr;
    int64_t small_ls_launches = 0;
    int debug_ls_lds_pad = 0;        // CSI_DEBUG_HOOKS=1 CSI_LS_LDS_PAD=<bytes>: the Walsh-Hadamard LS kernel asks for that much more LDS than it uses
    int debug_bf16_fork_late = 0;    // CSI_DEBUG_HOOKS=1 CSI_BF16_FORK_LATE=1: bf16 contexts fork the second stream of a two-stream call behind the LS kernel (A/B runs)
    int debug_small_tile16 = 0;  // CSI_DEBUG_HOOKS=1 CSI_SMALL_TILE16=1, read once at csi_create (A/B runs)
    int small_fused = 1;         // "small_fused" option: 0 = the general kernels (six launches per model on two streams)
    int small_rows = 1024;       // "small_rows": pair rows up to which a call takes it (and at most 64 preambles).  Measured (profiles/r05_regime_probe.txt):
                                 // 4 packets 117 us against 143 on the general kernels, 8 packets 162 / 170, 12 packets 251 / 246, 16 packets 283 / 252
    int l0_stream = 1;           // "l0_stream": layer 0 of a call of 9 ... 256 rx preambles on the weight-streaming split-f16 kernel (l0_hs_stream.hip.h)
    int l0_stream_ks = 0;        // "l0_stream_ks": its k ranges (0 = automatic: ~256 workgroups per component model up to 64 preambles, ~128 beyond)
    int64_t l0_stream_launches = 0;
    int l0_stream_max_rows = 1280;     // "l0_stream_max_rows": the largest call (rx preambles) the kernel takes (<= 4096), beyond 256 in row blocks (gridDim.z).
                                       // Measured (profiles/r05_band_split_probe.txt): 128 packets 400 -> 323 us, 256: 693 -> 595, 320: 862 -> 790; 500 packets 1108 -> 1134 (not taken)
    int l0_stream_prepass_rows = 64;   // "l0_stream_prepass_rows": beyond this many preambles the row scales come from l0_row_max_kernel
    int small_rows_band = 256;   // "small_rows_band": the same limit where the column-split band kernel serves the model (csi_dnn_hs.hpp): the general
                                 // path with it and the weight-streaming layer 0 takes 102 us at 3 ... 5 packets against 117-121 here (2 packets: 111 / 61)
    bool in_host_pipeline = false;   // a chunk of a host-buffer entry point is being enqueued: no second-stream fork inside (measured: the
                                 // two-stream arrangement costs the PCIe-bound pipeline 6 % - profiles/r05_regime_probe.txt)
    int64_t small_calls = 0;     // "small_calls": calls that took it
    char* small_ws = nullptr;    // its scratch: L0 of both models + ping-pong activations
    size_t small_ws_bytes = 0;
    // "ls_overlap_cus" = n > 0: inside csi_estimate_device the LS kernel runs on its own stream, restricted to n compute units
    // (hipExtStreamCreateWithCUMask), BESIDE the DNN kernels of the same packets instead of in front of them: it is HBM-bound and
    // draws little power, the matrix kernels are bound by the power budget and by one workgroup per CU - a few CUs lent to it cost
    // them less than the 0.4 ms it occupies the whole chip for (both orders give bit-identical results)
    int ls_overlap_cus = 0;
    int ls_overlap_stride = 0;   // "ls_overlap_stride": CU i of the n is mask bit i * stride (0 = spread evenly over 256)
    hipStream_t ls_stream = nullptr;
    int ls_stream_cus = 0, ls_stream_stride = 0;
    hipEvent_t ls_fork = nullptr, ls_join = nullptr;
    int ls_grid_cus = 0;         // set while the LS kernel is launched for the masked stream: persistent grid = this many CUs
    // csi_estimate_device parks the LS launch here; the DNN path fires it behind its FIRST layer-0 kernel (ls_deferred_fire), so that
    // the LS kernel runs beside the per-pair kernels (many rounds of workgroups per CU: a few CUs less cost them nothing measurable)
    // and not beside layer 0 (ONE round of 252 tiles on 256 CUs at config 2: any CU less costs it a whole second round)
    struct { bool active = false, forked = false; const float *re = nullptr, *im = nullptr; int64_t npkt = 0; float *h_re = nullptr, *h_im = nullptr; } ls_deferred;
    // staging for host-buffer entry points
    char* stage = nullptr;
    size_t stage_bytes = 0;
    int xcd_order = -1;          // option "xcd_order": -1 auto, 0 linear tile order, 1 XCD super-tile order
    bool use_graph = false;
    bool in_graph_call = false;  // csi_estimate_device is running the content of a (future) graph: no second-stream fork inside
    int64_t graph_replays = 0;   // hipGraphLaunch count ("graph_replays", read-only)
    std::vector<GraphEntry> graphs;
    int f32_engine = -1;         // "f32_engine" option: fp32 contexts, 0 = native fp32 MFMA kernels, 1 = split-f16 kernels (gemm_hs.hip.h)
                                 // wherever the shapes allow, -1 = split-f16 once a GEMM fills the chip (default)
    int hs_act_shift = HS_SHIFT_AUTO;        // split-f16: hidden activations are carried times 2^hs_act_shift (|h| < 65504 / 2^shift)
    float* hs_zero = nullptr;    // zeros, widest hidden layer: the BN shift the split-engine kernels see (it lives in the next layer's bias)
    unsigned* hs_peak = nullptr; // device word: range guard of the split engine (gemm_hs.hip.h), 0 = no operand came near the f16 limit
    size_t hs_lds_attr[20] = {};   // dynamic-LDS limit already raised on this context's device: layer 0 / pair (hs out) / pair (fp32 out) / pair + fused regressor
    int64_t hs_launches = 0;     // split-engine GEMMs launched so far / at the last range check
    int64_t hs_checked = 0;
    int64_t hs_range_fallbacks = 0;
    int64_t hs_weight_pins = 0;  // models csi_load_weights pinned to the fp32 MFMA kernels: their split-f16 weight copies were not fp32-grade (read-only option)
    int hs_blocked = 1;          // "hs_blocked": hs activation buffers between split-engine layers in the blocked layout (gemm_hs.hip.h)
    int hs_fuse_regressor = 0;   // "hs_fuse_regressor": two hidden layers -> the regressor runs inside the pair kernel (h2 stays on the CU).
                                 // Off by default: measured SLOWER (2.45 vs 1.88 ms per 262144 rows, profiles/r02_hs_probe.txt) - the
                                 // partial sums the four column tiles exchange cost what the h2 round trip cost
    char* fuse_ws = nullptr;     // its partial-sum slabs + row-tile flags
    size_t fuse_ws_bytes = 0;
    int hs_band = 1;             // "hs_band": two hidden layers -> first per-pair layer + regressor as ONE kernel, the assembly "band8"
                                 // kernel of band_kernel_gen.py (h2 stays in registers; measured -4 .. -7 % against the two kernels,
                                 // profiles/r03_band_probe.txt); 0 = the separate pair and regressor kernels
    hipModule_t band_mod = nullptr;          // its code object (embedded in the library, loaded on first use)
    hipFunction_t band_fn = nullptr;
    hipFunction_t band_fn_bf16 = nullptr;
    hipFunction_t band_fn_bf16_ns = nullptr;   // bf16 form without the staged T / L0 streams (nt outside 32 .. 64)
    hipFunction_t band_fn_ns = nullptr;        // split-f16 form without them (nt outside 16 .. 128)
    hipFunction_t band_fn_bf16_cs = nullptr;   // ... of the bf16 form
    hipFunction_t band_fn_cs = nullptr;        // column-split form: grid (bands, splits), split y computes N1 / splits of the hidden features
    hipFunction_t band_fn4_cs = nullptr;       // column-split launches of the register-blocked forms (csi_band4_cs / csi_band4_bf16_cs)
    hipFunction_t band_fn4_bf16_cs = nullptr;
    hipFunction_t band_fn4_p = nullptr, band_fn4_bf16_p = nullptr;      // persistent forms (one workgroup per CU walks the bands)
    bool band_hs_persist = false, band_bf16_persist = false;            // hooked A/B runs: CSI_BAND8_NAME / CSI_BAND8_BF16_NAME name a persistent form
    int n_cu = 256;                                                      // compute units of the device (csi_create)
    hipFunction_t band_fn4 = nullptr;          // register-blocked split-f16 form (band4_kernel_gen.py "csi_band4"), staged, 16 <= nt <= 128
    int band_hs_threads = 512;                 // workgroup size of band_fn (a CSI_BAND8_NAME variant named csi_band4* has 256)
    hipFunction_t band_fn4_bf16 = nullptr;     // register-blocked bf16 form (band4_kernel_gen.py: 4 waves x 512 registers, 256 threads), staged, 32 <= nt <= 64
    int band_bf16_threads = 512;               // workgroup size of band_fn_bf16 (a CSI_BAND8_BF16_NAME variant named csi_band4* has 256)
    int band4 = 1;                             // "band4": bf16 contexts take the register-blocked form where it applies (0 = csi_band8_bf16; A/B)
    bool band_failed = false;                // the code object could not be loaded: separate kernels from then on
    int64_t band_launches = 0;
    int band_split = -1;         // "band_split": calls with fewer bands than CUs split every band's hidden features over 2 / 4 workgroups (the
                                 // regressor sums of the splits are added in split order by band_split_sum_kernel); -1 = automatic (as many
                                 // splits as keep the workgroups of the models in flight within the 256 CUs), 0 / 1 = never, 2 / 4 = always
    int64_t band_split_launches = 0;
    int aux_fork_early = 1;      // "aux_fork_early": csi_estimate_device forks the second stream of a two-stream call in front of its LS kernel
    bool aux_preforked = false;
    int models_in_flight = 1;    // 2 while csi_predict_device runs the component models on two streams
    int hs_min_blocks = 48;      // automatic mode: the per-pair layers go to the split engine from this many 256x256 workgroups on
                                 // ("hs_min_blocks"; measured crossover at Nt=32, 1024x1024 with the two component models on two
                                 // streams: 24 packets - profiles/r05_regime_probe.txt; 80 = 40 packets before round 5); layer 0 from max(this, 128)
    int hs_in_shift = HS_SHIFT_AUTO;         // split-f16: the preamble samples times 2^hs_in_shift
    long long bf16_l0_fused_split_launches = 0;   // layer-0 products that took the fused kernel with K ranges because of "bf16_l0_fused_split" (get only)
    int band_tail_split = 1;     // "band_tail_split": a call of more bands than CUs whose last round of band workgroups would leave >= half of the CUs idle launches
                                 // that round in 2 or 4 column splits (one-stream calls; round 6); 0 = one launch
    long long band_tail_launches = 0;
    int bf16_l0_fused_split = 1; // "bf16_l0_fused_split": bf16 layer 0 of calls between the streaming kernel's range and 256 tiles on the fused kernel with K ranges (round 6); 0 = cast pass + 128 x 128 kernel
    int bf16_fused_h1 = 1;       // "bf16_fused_h1" option: 0 = materialise h1 (pair_h1_bf16_kernel) instead of generating it in the GEMM
    int p_pieces = 3;            // bf16 pieces (8 significand bits each) the entries of P need: 1 for +-1 pilots, 3 for arbitrary floats
    int ls_ringb_min = 33;       // "ls_ringb_min": from this Nt on a non-Hadamard pilot takes the bf16-split despread (ls_estimate_ringb_kernel).
                                 // Round 4: 33, not 16 - the one-antenna-tile form of that kernel (Nt <= 32, TWO workgroups per CU) produced, on
                                 // ONE box of the pool, 1-3 wrong items in ~1 % of the first launches of a fresh context (always the first item of
                                 // a CU's second workgroup, lane groups of an FFT stage; 7 events in 570 cycles there, none in 1170 on two other
                                 // boxes, none ever in the other LS kernels).  Not understood, so not selected: Nt <= 32 takes
                                 // the fp32 ring kernel (3-5 % slower); tools/ls_race_repro.py reproduces, "ls_kernel" 7 still forces it
    bool p_sylvester = false;    // csi_set_pilot saw the Sylvester Hadamard matrix (Walsh-Hadamard LS despread applies)
    // csi_set_pilot saw P = D1 Pi1 H Pi2 D2 (H Sylvester, Pi permutations, D signs; e.g. the 802.11 VHT 4x4 base doubled up): the
    // Walsh-Hadamard kernel applies with permuted / signed symbol loads and output rows.  p_perm[0][u] = source symbol of transform
    // input u (| 256 when it enters negated), p_perm[1][r] = output antenna of transform row r (| 256 when negated)
    bool p_fast_ok = false;
    bool p_fast_identity = true; // ... and both are the identity without signs (the Sylvester matrix itself): the kernel without tables
    int p_perm[2][CSI_WIRE_MAX_NT] = {};
    int* p_tables = nullptr;     // device: [4][nt] source symbol, its sign (float bits), output row, its sign
    int ls_fast_perm = 1;        // "ls_fast_perm": 0 = only the exact Sylvester matrix takes the Walsh-Hadamard kernel (A/B runs)
    int ls_debug = 0;            // CSI_LS_DEBUG / "ls_debug": skip phases of the chunked LS kernel (timing experiments only)
    int ls_kernel = 0;           // "ls_kernel" option / CSI_LS_KERNEL: 0 auto, 1 FFT-first, 2 chunked, 3 despread-first, 4 / 5 Walsh-Hadamard (register prefetch / LDS-DMA ring), 6 generic P on the ring (fp32 despread), 7 generic P, bf16-split despread (tests, A/B runs)
    int hs_vm_cast = 2, hs_vm_pair = 3;   // "hs_vm_cast" / "hs_vm_pair": vector-memory schedule of the layer-0 / pair kernel (gemm_hs.hip.h): 0 builtin LDS-DMA + one drain per sub-tile, 1 hand-counted, 2 + one more sub-tile of look-ahead, 3 + one load and one 24-MFMA segment per sub-tile
    int ls_v2 = 0;               // CSI_LS_V2: shape variant of the LDS-DMA fed Walsh-Hadamard kernel (experiments)
    int ls_fft_first_max = 15;   // FFT-first LS kernel (all Nt spectra in LDS) up to this Nt; from 16 on the ring kernel is faster (Nt = 16: 0.47 vs 0.72 ms); debug knob CSI_LS_FFT_FIRST_MAX
    int force_pair_tile = 0;     // debug knob CSI_FORCE_PAIR_TILE=128|256: forces the row-tile height of every GEMM (tests)
    // profiling
    bool prof_on = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[K_COUNT] = {0};
    int64_t prof_launches[K_COUNT] = {0};
    double prof_flops[K_COUNT] = {0};
    double prof_bytes[K_COUNT] = {0};
};

namespace {

int fail(csi_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, CSI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

struct ProfScope {
    csi_ctx* c;
    bool on;
    ProfSpan sp;
    ProfScope(csi_ctx* ctx, int id, double flops, double bytes) : c(ctx), on(ctx->prof_on) {
        if (!on) return;
        sp.id = id;
        sp.beg = take();
        sp.end = take();
        c->prof_launches[id] += 1;
        c->prof_flops[id] += flops;
        c->prof_bytes[id] += bytes;
        hipEventRecord(sp.beg, c->stream);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(sp.end, c->stream);
        c->spans.push_back(sp);
    }
    hipEvent_t take() {
        if (!c->ev_pool.empty()) {
            hipEvent_t e = c->ev_pool.back();
            c->ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
};

int prof_collect(csi_ctx* c) {
    if (c->spans.empty()) return CSI_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& sp : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.beg, sp.end) == hipSuccess) c->prof_ms[sp.id] += ms;
        c->ev_pool.push_back(sp.beg);
        c->ev_pool.push_back(sp.end);
    }
    c->spans.clear();
    return CSI_OK;
}

void drop_graphs(csi_ctx* c) {
    for (auto& g : c->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

// the CU-masked side stream of "ls_overlap_cus"
int ls_stream_ensure(csi_ctx* c) {
    const int n = std::max(1, std::min(255, c->ls_overlap_cus));
    const int stride = c->ls_overlap_stride > 0 ? c->ls_overlap_stride : std::max(1, 256 / n);
    if (c->ls_stream && c->ls_stream_cus == n && c->ls_stream_stride == stride) return CSI_OK;
    drop_graphs(c);
    if (c->ls_stream) { hipStreamSynchronize(c->ls_stream); hipStreamDestroy(c->ls_stream); c->ls_stream = nullptr; }
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const int bit = (i * stride) % 256;
        mask[bit >> 5] |= 1u << (bit & 31);
    }
    HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->ls_stream, 8, mask));
    if (!c->ls_fork) HIP_TRY(c, hipEventCreateWithFlags(&c->ls_fork, hipEventDisableTiming));
    if (!c->ls_join) HIP_TRY(c, hipEventCreateWithFlags(&c->ls_join, hipEventDisableTiming));
    c->ls_stream_cus = n;
    c->ls_stream_stride = stride;
    return CSI_OK;
}

// the LS launch csi_estimate_device parked: fork the side stream behind what the main stream holds so far, launch the LS kernel
// there, record the join event (csi_estimate_device waits for it before it returns)
int ls_deferred_fire(csi_ctx* c) {
    if (!c->ls_deferred.active) return CSI_OK;
    c->ls_deferred.active = false;
    const auto d = c->ls_deferred;
    int r = ls_stream_ensure(c);
    if (r) return r;
    HIP_TRY(c, hipEventRecord(c->ls_fork, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->ls_stream, c->ls_fork, 0));
    c->ls_deferred.forked = true;
    std::swap(c->stream, c->ls_stream);
    c->ls_grid_cus = c->ls_overlap_cus;
    r = csi_ls_estimate_device(c, d.re, d.im, d.npkt, d.h_re, d.h_im);
    c->ls_grid_cus = 0;
    const hipError_t e = hipEventRecord(c->ls_join, c->stream);
    std::swap(c->stream, c->ls_stream);
    if (r) return r;
    HIP_TRY(c, e);
    return CSI_OK;
}

int ensure_bytes(csi_ctx* c, char** buf, size_t* have, size_t need) {
    if (*have >= need) return CSI_OK;
    drop_graphs(c);           // captured launches point into the old buffer
    if (*buf) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    const size_t bytes = need + G_SLACK_FLOATS * sizeof(float);
    if (hipMalloc((void**)buf, bytes) != hipSuccess) {
        *buf = nullptr;
        return fail(c, CSI_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
    }
    HIP_TRY(c, hipMemsetAsync(*buf, 0, bytes, c->stream));     // never-written parts must be finite
    *have = need;
    return CSI_OK;
}

// Every device array a GEMM may read as its A side (or as a per-column vector) is followed by
// G_SLACK_FLOATS zeroed floats: the K tail of the last tile over-reads into finite memory.
int upload(csi_ctx* c, float** dst, const float* src, size_t n) {
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    const size_t bytes = (n + G_SLACK_FLOATS) * sizeof(float);
    if (hipMalloc((void**)dst, bytes) != hipSuccess)
        return fail(c, CSI_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
    HIP_TRY(c, hipMemset(*dst, 0, bytes));
    HIP_TRY(c, hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return CSI_OK;
}

void free_layer(Layer& l) {
    if (l.Wt) hipFree(l.Wt);
    if (l.Wb) hipFree(l.Wb);
    if (l.Wb_p) hipFree(l.Wb_p);
    if (l.Wh) hipFree(l.Wh);
    if (l.Wh_f) hipFree(l.Wh_f);
    if (l.Wh_p) hipFree(l.Wh_p);
    if (l.bias) hipFree(l.bias);
    if (l.bias_hs) hipFree(l.bias_hs);
    if (l.scale) hipFree(l.scale);
    if (l.shift) hipFree(l.shift);
    l = Layer();
}

void free_model(Model& m) {
    for (auto& l : m.layers) free_layer(l);
    m.layers.clear();
    if (m.W0p) hipFree(m.W0p);
    if (m.W0rm) hipFree(m.W0rm);
    m.W0rm = nullptr;
    if (m.T) hipFree(m.T);
    if (m.T_hs) hipFree(m.T_hs);
    if (m.T_sw) hipFree(m.T_sw);
    if (m.Wt1) hipFree(m.Wt1);
    if (m.Wt2) hipFree(m.Wt2);
    m.Wt1 = m.Wt2 = nullptr;
    m.tiled_ok = false;
    if (m.l0_rowmax) hipFree(m.l0_rowmax);
    m.W0p = m.T = m.T_hs = m.T_sw = m.l0_rowmax = nullptr;
    m.T_sw_ok = false;
    m.T_hs_shift = HS_SHIFT_AUTO;
    m.loaded = m.table_ok = false;
    m.hs_repr_err = 0.0;
    m.hs_repr_ok = true;
}

const csi_tensor* find_tensor(const csi_tensor* t, int n, const std::string& name) {
    for (int i = 0; i < n; ++i)
        if (t[i].name && name == t[i].name) return &t[i];
    return nullptr;
}

}  // namespace
