/* csi_mamimo.h - C-ABI of the MI355X (gfx950) massive-MIMO channel-estimation hot path.
 *
 * The reference (mauro-belgiovine/DL-channel-estimation-MaMIMO) has no FFI: the path sits
 * behind two Python call surfaces, which the entry points below replace one for one.
 * Citations are file:line in the reference repository.
 *
 *   reference interface                                             replaced by
 *   --------------------------------------------------------------  ---------------------------
 *   model construction, massiveMIMO_CSI_prediction_DNN.py:176-234   csi_create
 *   Model.load_weights(<d>_weights-improvement.hdf5)   DNN.py:334   csi_load_weights
 *   keras.models.load_model(<d>_keras_model)     inference.py:15-16  csi_load_weights
 *   dataset['P'] fed as seq_p     massiveMIMO_dataGenerator.py:311  csi_set_pilot
 *   Model.predict(generator) over packets, batch = nTX*nRX
 *        DNN.py:339-346 + sample assembly dataGenerator.py:299-316  csi_predict[_device]
 *   CSIPredictor.inference: X.real / X.imag -> predict x2 ->
 *        real + 1j*imag on complex128 batches    inference.py:24-32  csi_estimate_c128 (complex64 batches: csi_estimate_c64)
 *   Model.predict(x, batch_size=bs)              inference.py:29-30
 *        / DNN.py:434,470  (arbitrary rows [B, lenLTF+Nt])          csi_predict_samples
 *   ofdmdemod + helperMIMOChannelEstimate
 *        generate_maMIMO_LTF.m:336-342, helperMIMOChannelEstimate.m:24-36
 *                                                                   csi_ls_estimate[_device]
 *   LMMSE_ce per link   helperMIMOChannelEstimate.m:37-39, LMMSE_ce.m  csi_lmmse_estimate[_device]
 *   NMSE_subk           BER_test_maMIMO_LTF.m:675-686                csi_nmse[_device]
 *   --execTime profiler loop                       DNN.py:441-475   csi_profile_*
 *   Model.fit step (noise, BN, dropout, Adam)       DNN.py:272-316   csi_train_*
 *
 * Conventions: every function returns 0 on success or a negative csi_status; it never calls
 * exit().  All buffers are caller-owned, row-major, contiguous float32.  A context is bound
 * to one GPU and one HIP stream and is not thread-safe; use one context per GPU.  Host-buffer
 * entry points are synchronous; internally they pipeline upload, kernels and download over packet
 * chunks (pinned staging slots, a few host threads) and DMA directly from / to buffers the caller
 * has pinned (hipHostMalloc / hipHostRegister).  *_device entry points take device pointers, enqueue on the
 * context's stream and return without waiting; call csi_synchronize before reading results.
 *
 * Sample / output order everywhere: s = p*Nr*Nt + iRx*Nt + iTx
 * (create_massiveMIMO_CSIest_dnn_dataset.py:62), i.e. outputs are [Npkt][Nr][Nt][n_out],
 * which is MATLAB CSI(:, iTx, iRx) of packet p (BER_test_maMIMO_LTF.m:191-195).
 */
#ifndef CSI_MAMIMO_H
#define CSI_MAMIMO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSI_MAX_HIDDEN 8
#define CSI_ABI_VERSION 1

typedef enum {
    CSI_OK = 0,
    CSI_ERR_INVALID_ARG = -1,   /* null pointer, bad shape, unsupported size              */
    CSI_ERR_NOT_READY = -2,     /* predict before weights / pilot were loaded             */
    CSI_ERR_HIP = -3,           /* a HIP runtime call failed; text in csi_last_error       */
    CSI_ERR_NO_DEVICE = -4,     /* no gfx950 device visible                                */
    CSI_ERR_NOMEM = -5,         /* device allocation failed                                */
    CSI_ERR_RANGE = -6          /* split-f16 engine: an operand left the f16 range (csi_synchronize) */
} csi_status;

typedef enum {
    CSI_DTYPE_F32 = 0,          /* fp32 in / accumulate / out.  GEMMs that fill the chip run on the split-f16 engine (every fp32
                                 * operand as hi + lo f16 halves, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; same
                                 * 1e-5 contract, both f16 range ends guarded with an automatic repeat); smaller calls and
                                 * "f32_engine" = 0 run v_mfma_f32_32x32x2_f32 (exact fp32 products)                    */
    CSI_DTYPE_BF16 = 1          /* bf16 operands, fp32 accumulate MFMA, fp32 out            */
} csi_dtype;

/* Model + problem shape.  n_out = 234 and hidden = {1024, 1024} in the shipped pipeline
 * (full_pipeline_maMIMO_DNNEst.sh:40,47). */
typedef struct {
    int32_t nt;                      /* tx antennas = LTF symbols = pilot-row length        */
    int32_t nr;                      /* rx antennas per packet                              */
    int32_t len_ltf;                 /* samples per rx preamble, 320*nt                     */
    int32_t n_hidden;                /* number of Dense+relu(+BN) layers, 1..CSI_MAX_HIDDEN */
    int32_t hidden[CSI_MAX_HIDDEN];  /* their widths (--nn)                                 */
    int32_t n_out;                   /* fc_regressor width (nSubCarr)                       */
    int32_t use_bn;                  /* --useBN                                             */
    float   bn_eps;                  /* keras BatchNormalization epsilon, 1e-3              */
    int32_t dtype;                   /* csi_dtype                                           */
    int32_t device;                  /* HIP device ordinal                                  */
    int64_t workspace_bytes;         /* cap for activation workspace; 0 = default (4 GiB fp32, 8 GiB bf16) */
} csi_config;

/* One named weight tensor on the host.  Names are the keras ones:
 *   fc_dense<i>.kernel [in,out]   fc_dense<i>.bias [out]
 *   bn<i>.gamma / .beta / .moving_mean / .moving_variance [out]      (iff use_bn)
 *   fc_regressor.kernel [in,n_out]   fc_regressor.bias [n_out]
 * Rows 0..len_ltf-1 of fc_dense0.kernel multiply the LTF samples, rows len_ltf..len_ltf+nt-1
 * the pilot row (Concatenate([flatten, seq_p]), DNN.py:207-208). */
typedef struct {
    const char*  name;
    const float* data;
    int64_t      rows;               /* 1 for vectors                                       */
    int64_t      cols;
} csi_tensor;

typedef struct csi_ctx csi_ctx;

int  csi_abi_version(void);
int  csi_create(const csi_config* cfg, csi_ctx** out);
void csi_destroy(csi_ctx* ctx);
const char* csi_last_error(const csi_ctx* ctx);      /* ctx may be NULL: last create error  */

/* model: 0 = real, 1 = imag.  Tensors are copied (and re-laid-out) to the device. */
int  csi_load_weights(csi_ctx* ctx, int model, const csi_tensor* tensors, int n);
/* P [nt][nt], row j = pilot sequence of tx antenna j = MATLAB P(j,:) = dataset['P'][:, j]. */
int  csi_set_pilot(csi_ctx* ctx, const float* P);

/* Which LS despread csi_set_pilot will choose for P (host only; no context or device needed).  Returns 0: generic real P
 * (matrix-core despread), 1: the Sylvester Hadamard matrix, 2: a signed row / column permutation of it - e.g. the 802.11 VHT
 * 4x4 mapping matrix doubled up, what helperGetP (helperMIMOChannelEstimate.m:13, un-vendored toolbox code) yields - both served
 * by the Walsh-Hadamard kernel; negative csi_status on a bad argument.  For 1 / 2 the optional tables receive the decomposition
 * P[j][s] = rs[j] H[sigma(j)][tau(s)] cs[s]:  sym_src[u] = tau^-1(u) | (cs < 0 ? 256 : 0),  out_row[r] = sigma^-1(r) | (rs < 0 ? 256 : 0)
 * (nt entries each, nt <= 128). */
int  csi_pilot_classify(const float* P, int nt, int32_t* sym_src, int32_t* out_row);

/* CRC-32C (Castagnoli) of a host buffer - the per-tensor checksum of the SavedModel variable files the reference writes
 * (DNN.py:411) and keras_files.py verifies; host only (SSE4.2 when the CPU has it).  crc = 0 starts, a returned value continues. */
uint32_t csi_crc32c(const void* data, int64_t bytes, uint32_t crc);

/* DNN estimate of npkt packets.  ltf_re / ltf_im [npkt][nr][len_ltf]; out_re / out_im
 * [npkt][nr][nt][n_out].  Layer 0 is evaluated once per (packet, rx) and shared by the nt
 * pairs (the reference stores each rx preamble once for the same reason, mk.py:50-63). */
int  csi_predict(csi_ctx* ctx, const float* ltf_re, const float* ltf_im, int64_t npkt,
                 float* out_re, float* out_im);
int  csi_predict_device(csi_ctx* ctx, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt,
                        float* d_out_re, float* d_out_im);

/* Literal Model.predict: x [B][len_ltf+nt] -> y [B][n_out] through the un-shared network. */
int  csi_predict_samples(csi_ctx* ctx, int model, const float* x, int64_t B, float* y);

/* LS estimate: h_re / h_im [npkt][nr][nt][234]. */
int  csi_ls_estimate(csi_ctx* ctx, const float* ltf_re, const float* ltf_im, int64_t npkt,
                     float* h_re, float* h_im);
int  csi_ls_estimate_device(csi_ctx* ctx, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt,
                            float* d_h_re, float* d_h_im);
/* LS + DNN of device-resident packets as ONE unit (the per-packet work of generate_maMIMO_LTF.m:336-342 plus
 * DNN.py:346): with "use_graph" the whole launch sequence - LS kernel, range-guard memsets, magnitude sample,
 * layer 0, per-pair layers, regressor, both component models, every packet chunk - is captured into one hipGraph
 * on the 2nd identical call and replayed afterwards. */
int  csi_estimate_device(csi_ctx* ctx, const float* d_ltf_re, const float* d_ltf_im, int64_t npkt,
                         float* d_out_re, float* d_out_im, float* d_h_re, float* d_h_im);

/* Both estimators on the arrays the reference's deployment wrapper handles (inference.py:24-32: a numpy
 * complex128 batch in, ``output_real + 1j*output_imag`` = complex64 out): ltf_c128 [npkt][nr][len_ltf] as
 * interleaved (re, im) doubles; dnn_c64 [npkt][nr][nt][n_out] and ls_c64 [npkt][nr][nt][234] as interleaved
 * (re, im) floats, either may be NULL.  One upload of the preambles serves both; the complex128 -> 2 x float32
 * split and the complex64 interleave run in the staging copies of the host pipeline, chunk by chunk beside the
 * transfers and kernels, instead of as whole-array passes in the caller (X.real / X.imag, inference.py:29-31).
 * Result arrays in pinned host memory (csi_host_malloc) skip the result-side host pass: the complex64 values are
 * assembled on the device and downloaded into the arrays themselves (option "hp_device_weave", default 1; counter
 * "hp_direct_out_calls"); same bits either way. */
int  csi_estimate_c128(csi_ctx* ctx, const double* ltf_c128, int64_t npkt, float* dnn_c64, float* ls_c64);

/* The same call for a batch the caller already holds as complex64 (an addition: inference.py:39-43 insists on complex128, and
 * CSIPredictor.inference keeps that contract; a pipeline that produces its preambles in single precision need not widen them
 * first).  ltf_c64 [npkt][nr][len_ltf] as interleaved (re, im) floats; results as csi_estimate_c128.  Half the bytes to read on
 * the host and no conversion pass there: the interleaved chunk is uploaded as it is - straight from the caller's array when
 * that is pinned host memory (csi_host_malloc) - and split into the two float32 planes on the device (csrc/weave.hip.h).
 * Bit-identical with csi_estimate_c128 on a batch whose values are representable in single precision. */
int  csi_estimate_c64(csi_ctx* ctx, const float* ltf_c64, int64_t npkt, float* dnn_c64, float* ls_c64);

/* LMMSE smoothing of an LS estimate (the 'hDmmse' output of helperMIMOChannelEstimate.m:37-39,
 * LMMSE_ce.m:23-39 with Nfft = Np = 234, Nps = 1).  h_re / h_im: LS estimate [npkt][nr][nt][234];
 * hvec [npkt][L]: the vector the reference passes as LMMSE_ce's 'h' (generate_maMIMO_LTF.m:342
 * hands it the scatterer delays h_tau); snr_db [npkt][nr]: SNR(i) in dB; out like h. */
int  csi_lmmse_estimate(csi_ctx* ctx, const float* h_re, const float* h_im, int64_t npkt, const float* hvec, int L,
                        const float* snr_db, float* out_re, float* out_im);
int  csi_lmmse_estimate_device(csi_ctx* ctx, const float* d_h_re, const float* d_h_im, int64_t npkt, const float* d_hvec,
                               int L, const float* d_snr_db, float* d_out_re, float* d_out_im);

/* Accuracy metric of the reference's evaluation, NMSE_subk (BER_test_maMIMO_LTF.m:675-686): per link
 * ||ref - est||^2 / ||ref||^2 over the n_bins bins, mean over the nlinks links ([link][n_bins] planes, e.g.
 * the [npkt][nr][nt][234] outputs of csi_predict / csi_ls_estimate with nlinks = npkt*nr*nt).  Synchronous;
 * d_per_link (may be NULL) receives the per-link ratios, e.g. for per-packet averages
 * (snr_loop_testing.m:44,51,58). */
int  csi_nmse(csi_ctx* ctx, const float* ref_re, const float* ref_im, const float* est_re, const float* est_im,
              int64_t nlinks, int n_bins, double* mean_out);
int  csi_nmse_device(csi_ctx* ctx, const float* d_ref_re, const float* d_ref_im, const float* d_est_re, const float* d_est_im,
                     int64_t nlinks, int n_bins, float* d_per_link, double* mean_out);

/* ---- On-box fine-tuning: one optimiser step of the reference's fit(),
 * massiveMIMO_CSI_prediction_DNN.py:272-316, for one component model (fp32 contexts only).
 *   GaussianNoise on the LTF columns of x only (DNN.py:191-193; stddev per batch from the caller,
 *   changeNoisePower DNN.py:92-100) -> Dense(relu) -> BatchNormalization(batch statistics) ->
 *   Dropout after every hidden layer but the last (DNN.py:211-226) -> Dense(linear), loss 'mse',
 *   Adam (DNN.py:272-273).  Keras defaults: beta1 0.9, beta2 0.999, eps 1e-7, BN momentum 0.99. */
typedef struct {
    float    lr;            /* --lr (1e-4)                                                  */
    float    beta1, beta2, eps;
    float    bn_momentum;
    float    dropout;       /* --dropout (0.15)                                             */
    uint64_t seed;          /* noise / dropout / initialisation streams                     */
} csi_train_config;

/* Creates the trainer of model 0 (real) / 1 (imag) from the named tensors of csi_load_weights
 * (n > 0), or with Glorot-uniform kernels, zero biases and identity BatchNormalization (n == 0,
 * DNN.py:213,227).  The inference model of the context is untouched until csi_train_end(commit). */
int  csi_train_begin(csi_ctx* ctx, int model, const csi_train_config* cfg, const csi_tensor* tensors, int n);
/* x [B][len_ltf+nt] rows as Model.fit receives them (dataGenerator.py:299-316), y [B][n_out];
 * host buffers, synchronous when loss != NULL.  noise_std = 0 disables the AWGN layer. */
int  csi_train_step(csi_ctx* ctx, int model, const float* x, const float* y, int64_t B, float noise_std, float* loss);
/* Data-parallel form of a step: csi_train_backward computes the loss and every gradient of the
 * rank's batch without touching the parameters; csi_train_grads exposes all gradients as ONE
 * flat device buffer (regressor first, layer 0 last) for a single sum all-reduce (RCCL) that the
 * caller scales by 1/world; csi_train_apply runs Adam on whatever the buffer then holds.
 * csi_train_step == backward + apply.  BatchNormalization statistics stay per rank (keras
 * MirroredStrategy default). */
int  csi_train_backward(csi_ctx* ctx, int model, const float* x, const float* y, int64_t B, float noise_std, float* loss);
int  csi_train_grads(csi_ctx* ctx, int model, float** d_grads, int64_t* count);
int  csi_train_apply(csi_ctx* ctx, int model);
/* Resident training set (288 GB of HBM: the data stays on the device for the whole fit).  The
 * dataset is handed over once in the reference's own de-duplicated form
 * (create_massiveMIMO_CSIest_dnn_dataset.py:50-63): ltf_table [n_rows][len_ltf] = every rx preamble
 * of this component once, and per sample s its table row ltf_row[s], its tx index itx[s] (the
 * pilot columns are row itx[s] of csi_set_pilot's P) and its labels y[s][n_out].  A batch is then
 * B sample indices: csi_train_indexed(mode 0 = step, 1 = backward only, 2 = inference-mode loss)
 * gathers the rows on the device (AWGN applied on the fly) - no per-step host assembly or upload. */
int  csi_train_set_dataset(csi_ctx* ctx, int model, const float* ltf_table, int64_t n_rows, const int32_t* ltf_row,
                           const int32_t* itx, const float* y, int64_t N);
int  csi_train_indexed(csi_ctx* ctx, int model, int mode, const int32_t* ids, int64_t B, float noise_std, float* loss);
/* mse of the current parameters in inference mode (running statistics, no noise, no dropout):
 * the val_loss that EarlyStopping / ReduceLROnPlateau monitor (DNN.py:285-286). */
int  csi_train_eval(csi_ctx* ctx, int model, const float* x, const float* y, int64_t B, float* loss);
int  csi_train_set_lr(csi_ctx* ctx, int model, float lr);
/* Reads one tensor by its keras name (kernels as [in][out]); "grad:<name>" reads the gradient of
 * the last step (tests). */
int  csi_train_get(csi_ctx* ctx, int model, const char* name, float* out, int64_t count);
/* commit != 0: loads the trained tensors into the inference model (as csi_load_weights would,
 * pilot table included); then releases the trainer. */
int  csi_train_end(csi_ctx* ctx, int model, int commit);

int  csi_synchronize(csi_ctx* ctx);

/* Tuning / debug knobs (no reference counterpart).  name:
 *   "use_graph"        1: csi_predict_device replays a captured hipGraph when it is called again
 *                         with the same pointers and packet count (the reference's per-packet loop,
 *                         DNN.py:346, repeats one launch sequence); 0 (default): eager launches
 *   "force_tile"       128 | 256: row-tile height of every GEMM (0 = chosen by grid size)
 *   "xcd_order"        1 | 0: force the XCD super-tile / the linear tile order of the plain GEMMs
 *                         (-1 = automatic)
 *   "ls_fft_first_max" largest Nt served by the FFT-first LS kernel (default 15, max 64)
 *   "small_call_overlap" 1 (default): calls of at most 327 680 pair rows (2560 packets at Nt = 32, Nr = 4; 131 072 until round 6) run the real and the imag
 *                         model on two streams side by side - a mid-size call's kernels fill a fraction of the chip (24 ... 128 packets:
 *                         1.25-1.55x, 500 packets +1.4 %; device-pointer calls only: inside the host-buffer entry points' pipeline the
 *                         chunks stay on one stream, where the fork measured 6 % slower); 0: one after the other; 2: any size (A/B runs).
 *                         bf16 contexts: up to 262 144 pair rows.  "aux_fork_early" (default 1): csi_estimate_device forks the second
 *                         stream in front of its LS kernel - the imag model's chain needs the preambles, not the LS result (3 ... 32
 *                         packets 6 % faster); 0: behind it (A/B runs)
 *   "small_fused"      1 (default): a call of at most "small_rows" pair rows (default 1024 = 8 packets at Nt = 32, Nr = 4; and at most 64
 *                         rx preambles) - the reference's literal one-packet predict, DNN.py:339-346, and its small multiples - runs BOTH
 *                         component models in 1 + n_hidden launches: layer 0 as one weight-streaming kernel (up to 8 preambles) or on
 *                         fp32-MFMA tiles, every layer behind it as 16 x 16 / 32 x 32 fp32-MFMA tiles over the whole K, no split-K slabs
 *                         (csrc/small_call.hip.h); 0: the general kernels (A/B runs).  Read-only: "small_calls" (calls that took it).
 *                         "small_rows_band" (default 256 = 2 packets of that shape; calls of at most 8 preambles are not subject to it): the limit where the column-split band kernel serves
 *                         the model ("band_split": two hidden layers, 16 <= Nt <= 128, hidden[1] a multiple of 512) - from there
 *                         on the general path (weight-streaming layer 0 + that kernel) is the faster one
 *   "small_ls_fused"   1 (default): a csi_estimate_device call of at most 8 rx preambles runs its LS estimate INSIDE the layer-0 launch of the
 *                         one-packet path (csrc/small_call.hip.h: small_l0_ls_kernel - LS workgroups beside the weight-streaming ones;
 *                         Sylvester-ordered pilot, Nt = 16 / 32 / 64): one launch and 8 us less per call, same bits; 0: the LS kernel
 *                         in front (A/B runs).  Read-only: "small_ls_launches".
 *   "l0_stream"        1 (default): layer 0 of a call of 9 ... "l0_stream_max_rows" (default 1280) rx preambles runs on the weight-streaming split-f16 kernel
 *                         (csrc/l0_hs_stream.hip.h: every preamble row scaled by its own power of two, no range guard needed);
 *                         0: the general kernels.  "l0_stream_ks": its k ranges (0 = automatic); "l0_stream_prepass_rows": beyond
 *                         this many preambles (default 64) the row maxima come from their own small kernel.
 *                         Read-only: "l0_stream_launches"
 *   "f32_engine"       fp32 contexts: -1 (default) large GEMMs - at least half a round of 256x256 tiles - run on
 *                         the f16 matrix cores with split operands (x = hi + lo halves, three MFMA per
 *                         product, fp32 accumulation: the same 1e-5 contract at ~2.6x the fp32 MFMA rate,
 *                         gemm_hs.hip.h), small ones on the fp32 MFMA kernels - except that calls of 9 ... 1280 rx preambles run
 *                         layer 0 on the engine's weight-streaming kernel ("l0_stream") and models the column-split band kernel
 *                         serves ("band_split") run their per-pair layers on it at every size; 0: fp32 MFMA kernels only;
 *                         1: split engine wherever the layer shapes allow (hidden widths multiples of 16)
 *   "hs_blocked"       split engine: 1 (default) keeps the hidden activations between its layers in a blocked layout
 *                         ([16 rows][k-group of 16] = the 1 KiB one LDS-DMA piece of the next GEMM fetches, contiguous),
 *                         0 row-major (A-B)
 *   "hs_fuse_regressor" split engine, two hidden layers, n_out <= 256: 1 runs the regressor inside the first per-pair
 *                         layer's kernel (its 256 x 256 tile of activations becomes the A operand of a second product
 *                         on the CU; only partial sums reach memory).  0 (default): measured slower than two kernels
 *   "hs_min_blocks"    automatic mode: the per-pair layers take the split engine from this many 256x256 workgroups
 *                         on (default 48 = 24 packets of the shipped shape; 80 before the two-stream arrangement of round 5),
 *                         layer 0 from max(this, 128).  (Models the column-split band kernel serves take the split engine's
 *                         per-pair layers at every size - "band_split" - and are not subject to this threshold.)
 *   "hs_in_shift"      split engine: the preamble samples are carried times 2^shift.  99 (default): chosen per
 *                         launch on the device from a sampled maximum of the data, so that it lands at
 *                         2^13..2^14 (any input scaling is served); -8..14: fixed
 *   "hs_act_shift"     split engine: hidden activations are carried times 2^shift.  99 (default): per layer from
 *                         its BatchNormalization vectors at load (|beta| + 6 |gamma| lands at 2^10..2^11; 2^4
 *                         without BN); -8..14: fixed.  Operands that leave the f16 range at either end are
 *                         detected on the device: csi_predict repeats the call on the fp32 MFMA kernels by
 *                         itself, after device-pointer calls csi_synchronize returns CSI_ERR_RANGE
 *   "band_tail_split"  fp32 contexts, 1 (default, round 6): a one-stream call of more bands (128 pair rows) than compute units whose LAST round of band workgroups would
 *                      fill at most half (a quarter) of them launches that round in 2 (4) column splits; 0 = one launch.  "band_tail_launches" (get only)
 *                      counts the calls that did.
 *   "bf16_l0_fused_split" bf16 mode: 1 (default, round 6) runs layer 0 of calls between the weight-streaming kernel's range and 256 tiles
 *                      of the fused 256 x 256 kernel (321 ... 4095 packets at Nt = 64, Nr = 4) on that kernel with its K cut into ranges;
 *                      0 = a cast pass plus the 128 x 128 kernel, as before.
 *                      "bf16_l0_fused_split_launches" (get only) counts the layer-0 products that took that form.
 *   "bf16_fused_h1"    bf16 mode: 1 (default) generates the first per-pair activations inside the GEMM,
 *                         0 materialises them in HBM first (tests / A-B)
 *   "host_threads"     threads that copy between the caller's (pageable) buffers and the pinned
 *                         slots of the host-buffer entry points (0 = automatic: cores / 8, between 2 and 24)
 *   "ls_kernel"        0: automatic, 1: FFT-first (all Nt spectra in LDS, Nt <= 64), 2: chunked
 *                         FFT-first (16 <= Nt <= 128), 3: despread-first (any Nt), 4: Walsh-Hadamard
 *                         despread (Nt = 16 / 32 / 64 / 128 and P the Sylvester Hadamard matrix), 5: the same
 *                         fed by an LDS-DMA ring (chosen automatically for that P), 6: generic P on
 *                         the LDS-DMA ring, fp32 matrix-core despread, 7: generic P, despread on the bf16 matrix
 *                         cores with every fp32 value cut exactly into three bf16 pieces (one of the two is chosen
 *                         automatically for any other P, 16 <= Nt <= 128); a choice the kernel cannot serve falls back
 *   "ls_ringb_min"     the smallest Nt at which a non-Hadamard P takes kernel 7 (default 33; below, kernel 6 serves - DESIGN.md 4.2).
 *                         Kernel 7 runs ONE workgroup per CU at every Nt in the shipped library: its two-workgroups-per-CU form at
 *                         Nt <= 32 (rare wrong first items on some parts, never root-caused) exists only in the hunt build
 *                         (CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS) and no option of the product build selects it
 *   get only: "ls_mode" (the kernel the next LS call runs), "ls_per_cu" (its resident workgroups per CU),
 *                         "ls_pilot_pieces" (bf16 pieces the entries of P need: 1 - 3)
 *   "ls_v2"            1: the runner-up shape (chunk length / ring depth) of kernels 5 and 6, for A/B runs
 *   "hs_band"          1 (default): two hidden layers -> first per-pair layer + regressor as ONE kernel, h2 in registers (generated
 *                         gfx950 assembly, csrc/band_kernel_gen.py): the split engine of fp32 contexts (the form that streams the
 *                         L0 / pilot-table values through LDS at 16 <= nt <= 128, per-lane loads elsewhere) and bf16 contexts at
 *                         32 <= nt <= 64 (streamed form); 0: the separate kernels (A/B runs); 2: bf16 contexts take the per-lane form
 *                         at any other nt as well (measured slower than the separate kernels); 3: only the per-lane forms (A/B runs).
 *                         Read-only: "band_launches".
 *   "band4"            1 (default): bf16 contexts take the REGISTER-BLOCKED form of that kernel where its streamed form applies (32 <= nt <= 64;
 *                         csrc/band4_kernel_gen.py "csi_band4_bf16": 4 waves x 512 registers, every weight fragment against two row
 *                         groups, weights pre-tiled at first use); 0: the 8-wave form "csi_band8_bf16" (A/B runs; same bf16 roundings,
 *                         fp32 sums in another order).  Read-only: "band4_available".
 *   "band_split"       -1 (default): a call with fewer bands of 128 pair rows than the part has CUs (24 ... 64 packets of the shipped
 *                         shape) splits every band's hidden features over 2 or 4 workgroups ("csi_band8_cs") and adds their regressor
 *                         sums in split order - same arithmetic, the final fp32 sums associate differently (1 ulp class);
 *                         0 / 1: never; 2 / 4: always (A/B runs, tests).  Read-only: "band_split_launches".
 *   "hs_vm_cast", "hs_vm_pair"  vector-memory schedule of the split-f16 layer-0 / first per-pair kernel: 0 builtin LDS-DMA
 *                         with one drain per sub-tile, 1 hand-counted waits, 2 + one more sub-tile of look-ahead (default for
 *                         layer 0), 3 + one load and one 24-MFMA segment per sub-tile (default for the pair layer); same
 *                         results bit for bit, for A/B runs (tools/vm_ab.sh)
 *   "ls_fast_perm"     1 (default): a pilot matrix that is a signed row / column permutation of the Sylvester Hadamard matrix
 *                         takes the Walsh-Hadamard LS kernel through permutation tables; 0: the generic kernels (A/B runs)
 *   "ls_overlap_cus", "ls_overlap_stride"  csi_estimate_device: run the LS kernel on a side stream masked to this many CUs beside
 *                         the DNN kernels.  An experiment that measured slower than the serial order: the product
 *                         build REFUSES "ls_overlap_cus" > 0 (CSI_ERR_INVALID_ARG with text); the hunt build
 *                         (CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS) accepts it
 *   "hp_side_threads"  host-buffer entry points: 1 (default) input staging and result staging on their own threads beside the
 *                         caller's enqueue loop; 0: inline on the calling thread, in turn (A/B runs)
 *   "hp_chunk_packets" packets per pipeline slot of csi_estimate_c128 (0 = automatic)
 *   "hp_device_weave"  1 (default): csi_estimate_c128 with pinned result arrays assembles the complex64 values on the device
 *   "ls_debug"         development switches of the LS kernels (tools/ls_race_*.py); write-only, 0 in production */
int  csi_set_option(csi_ctx* ctx, const char* name, int64_t value);
/* Current value of an option, or of the read-only values: "hs_launches" (split-engine GEMMs launched), "hs_range_fallbacks"
 * (csi_predict calls repeated on the fp32 MFMA kernels), "hs_weight_pins" / "hs_weight_err_e12" (layers pinned to the fp32 kernels
 * at load because their split copies were not fp32-grade; worst relative error x 1e12), "band_available" (the assembly band kernel
 * is embedded in this build), "graph_replays", "ls_pilot_fast" (0 generic / 1 Sylvester / 2 permuted pilot), "comm_world",
 * "comm_rank", "comm_blobs", "comm_bytes" (communicator and last broadcast), "hp_direct_out_calls", and where the last pipelined
 * host-buffer call spent its time in microseconds: "hp_total_us", "hp_stage_us", "hp_wait_stage_us", "hp_wait_out_us", "hp_weave_us". */
int  csi_get_option(csi_ctx* ctx, const char* name, int64_t* value);

/* Device-memory plumbing so that a host program needs no other GPU runtime. */
int  csi_device_malloc(csi_ctx* ctx, void** dptr, int64_t bytes);
int  csi_device_free(csi_ctx* ctx, void* dptr);
/* Pinned (page-locked) host memory: buffers from here are DMA'd directly by the host-buffer entry
 * points instead of being staged through the library's own pinned slots. */
int  csi_host_malloc(csi_ctx* ctx, void** ptr, int64_t bytes);
int  csi_host_free(csi_ctx* ctx, void* ptr);          /* ctx may be NULL (a buffer that outlived its context) */
int  csi_memcpy_h2d(csi_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int  csi_memcpy_d2h(csi_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
/* i.i.d. CN(0,1) preambles generated on the device by a counter-based RNG (element index
 * -> value), for workloads too large to stage through the host (SURVEY.md 8d 'white'). */
int  csi_synth_white(csi_ctx* ctx, uint64_t seed, int64_t first_pkt, int64_t npkt,
                     float* d_ltf_re, float* d_ltf_im);

/* ---- multi-GPU: packets shard over the GPUs of a node (one process and one context per GPU), the weights are shared and
 * read-only, outputs stay sharded.  The single collective of the path is the load-time broadcast of the
 * weights, here inside the library on RCCL (ncclBroadcast over xGMI), device to device on the context's stream.  The
 * reference has no multi-device code; this replaces what `Model.load_weights` (DNN.py:334) would otherwise do once per GPU.
 * RCCL (librccl.so.1) is dlopen'ed by csi_comm_init; nothing else in the library needs it.
 *
 *   rank 0:      csi_get_unique_id(id);  -> hand the 128 bytes to the other ranks (file, environment, socket, MPI ...)
 *   every rank:  csi_create(...);  csi_comm_init(ctx, rank, world, id);
 *   rank root:   csi_load_weights(ctx, 0, ...);  csi_load_weights(ctx, 1, ...);  csi_set_pilot(ctx, P);
 *   every rank:  csi_broadcast_weights(ctx, root);      -> every context is loaded; no host copy of the weights elsewhere
 *   every rank:  csi_predict[_device] / csi_ls_estimate[_device] on its own packet range  */
#define CSI_UNIQUE_ID_BYTES 128
int  csi_get_unique_id(char id[CSI_UNIQUE_ID_BYTES]);            /* ncclGetUniqueId; errors: csi_last_error(NULL) */
int  csi_comm_init(csi_ctx* ctx, int rank, int world, const char id[CSI_UNIQUE_ID_BYTES]);   /* ncclCommInitRank on the context's device (collective) */
int  csi_comm_destroy(csi_ctx* ctx);
/* Both component models and the pilot matrix as csi_load_weights / csi_set_pilot left them on `root`: the re-laid-out fp32
 * matrices, their split-f16 / bf16 forms, bias and BatchNormalization vectors, P - ncclBroadcast of the device buffers
 * themselves, then the pilot tables are rebuilt locally.  Collective; synchronous at return.  Contexts must share one csi_config
 * (shape and dtype): every rank checks the root's record against its own, the ranks agree on the outcome (one ncclAllReduce of a
 * status word) BEFORE the buffers move, and if any rank refuses, every rank returns an error (the refusing one with the reason)
 * instead of waiting inside the broadcast; a receiver that failed holds no model and no pilot afterwards.
 * "comm_bytes" / "comm_blobs" (csi_get_option) report what the last call moved. */
int  csi_broadcast_weights(csi_ctx* ctx, int root);
/* The same transfer inside ONE process: `dst` takes both component models and the pilot matrix as they sit in `src` (device to
 * device on dst's stream, hipMemcpyPeer when the contexts live on different GPUs), then rebuilds its pilot tables - a second
 * context (another stream, packet range or GPU of the process) without a second csi_load_weights.  It walks exactly the receiver
 * side of csi_broadcast_weights (same record, same buffer list, same rebuild), which is how a one-GPU box tests that code.
 * The contexts must agree in nt, len_ltf, hidden widths, n_out, use_bn and dtype (nr, device and workspace may differ); a
 * mismatch is refused with text and leaves `dst` empty (nothing loaded, no pilot).  Synchronous at return. */
int  csi_clone_weights(csi_ctx* dst, const csi_ctx* src);

/* Per-kernel HIP-event timing on the context's stream (the reference's --execTime). */
int  csi_profile_enable(csi_ctx* ctx, int on);
int  csi_profile_reset(csi_ctx* ctx);
/* The practical ceiling of the dominant kernel, measured: the fused per-pair kernel's MFMA + barrier skeleton (no operand
 * conversion, no operand streams, no LDS traffic; operand registers filled once with the loaded model's own split weights) over
 * `rows` pair rows, `iters` launches timed with HIP events.  executed_flops = f16 flop per launch (3 MFMA products per multiply,
 * padded regressor tile included).  The part clocks to its power budget (~1.4 kW, 1.8-2.0 GHz under these
 * kernels), so the table peak of 2.5 PFLOP/s at 2.4 GHz is not reachable on real data; this is what is.  fp32 contexts with the
 * two-hidden-layer model loaded; the outputs it writes are garbage and go to the context's workspace. */
int  csi_profile_band_skeleton(csi_ctx* ctx, int64_t rows, int iters, double* ms_per_launch, double* executed_flops);
/* The host link, measured in this process: h2d_bytes up and d2h_bytes down between pinned host memory and device memory on the
 * host pipeline's two copy streams - each direction alone and both at once (ms).  ms_both is the floor of a host-buffer call that
 * moves these byte counts; bench.py reports the host-buffer entry points as a fraction of it. */
int  csi_profile_pcie(csi_ctx* ctx, int64_t h2d_bytes, int64_t d2h_bytes, double* ms_h2d, double* ms_d2h, double* ms_both);
int  csi_profile_num_kernels(void);
const char* csi_profile_kernel_name(int kernel_id);
/* total_ms / launches / flops / bytes accumulated since the last reset. */
int  csi_profile_query(csi_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches,
                       double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* CSI_MAMIMO_H */
