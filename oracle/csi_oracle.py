"""CPU oracle for the massive-MIMO channel-estimation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (reference repo
mauro-belgiovine/DL-channel-estimation-MaMIMO, mounted read-only at /root/reference
in the build container; paths below are relative to it).  Nothing here is shipped
in the product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

PARITY PIN STATUS (see DESIGN.md "Oracle"):
  * sample assembly / ordering (a-3) and the CSIPredictor pre/post-processing and
    real/imag recombination (a-8, a-9) are PINNED: ``tests/golden/make_golden.py``
    imported the reference's own pure-numpy code (``massiveMIMO_dataGenerator.py``,
    ``inference.py``) in the build container and the outputs are committed under
    ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them.
  * the OFDM demodulation convention (a-1: symbol split, CP window 64..319, un-scaled 256-FFT,
    DC in the middle) is PINNED against the reference's own numpy statement of it,
    ``massiveMIMO_dataGenerator.py:425-453`` (method 'reshape'), executed by
    ``tests/golden/make_golden.py``; its per-symbol spectra are committed as
    ``tests/golden/ref_ofdm_reshape_nt4.npz`` and ``ofdm_demod`` reproduces them
    (``test_ofdm_demod_matches_reference_reshape_method``).  The MATLAB ``ofdmdemod`` call
    itself (generate_maMIMO_LTF.m:336-338) cannot run here.
  * the Dense / BatchNormalization / Dropout arithmetic (a-4..a-7) lives in TensorFlow
    2.3 (README.md:27-28) and the LS despread (a-2) in MATLAB R2020b + toolboxes
    (README.md:29-30); neither is vendored nor installable here and the reference has
    no tests, golden vectors or weights.  For those rows: PARITY UNPINNED.  They are
    restated from the call sites and the published layer definitions, and checked by
    exact identities (LS known-answer round trip, torch.nn.functional cross-check).

Shapes follow the reference: a "sample" is one (packet p, rx antenna r, tx antenna t)
link with index ``s = p*Nr*Nt + r*Nt + t`` (create_massiveMIMO_CSIest_dnn_dataset.py:62).
"""
import numpy as np

# ---------------------------------------------------------------------------
# OFDM constants  (packet_generation/phased_arr/generate_maMIMO_LTF.m:96-102)
# ---------------------------------------------------------------------------
FFT_LEN = 256            # prm.FFTLength                      generate_maMIMO_LTF.m:96
CP_LEN = 64              # prm.CyclicPrefixLength             generate_maMIMO_LTF.m:97
SYM_LEN = FFT_LEN + CP_LEN
N_DATA = 234             # prm.numCarriers                    generate_maMIMO_LTF.m:98
BN_EPS = 1e-3            # keras BatchNormalization() default epsilon (DNN.py:217)


def null_carrier_indices():
    """1-based guard + DC bins, generate_maMIMO_LTF.m:99  ``[1:7 129 256-5:256]'``."""
    return np.array(list(range(1, 8)) + [129] + list(range(251, 257)), dtype=np.int64)


def pilot_carrier_indices():
    """1-based pilot bins, generate_maMIMO_LTF.m:100."""
    return np.array([26, 54, 90, 118, 140, 168, 204, 232], dtype=np.int64)


def data_carrier_indices():
    """1-based data bins ``prm.CarriersLocations`` = setdiff(1:256, nulls U pilots),
    generate_maMIMO_LTF.m:101-102.  Returned sorted ascending, length 234."""
    non_data = set(null_carrier_indices().tolist()) | set(pilot_carrier_indices().tolist())
    idx = np.array([k for k in range(1, FFT_LEN + 1) if k not in non_data], dtype=np.int64)
    assert idx.size == N_DATA
    return idx


def vht_ltf_256():
    """The 256-bin VHT-LTF frequency sequence literal of helperMIMOChannelEstimate.m:16-23
    (index 0 here = MATLAB index 1 = most negative frequency after fftshift)."""
    ltf_left = [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1,
                1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1]
    ltf_right = [1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1,
                 -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]
    mid_a = [-1, -1, -1, 1, 1, -1, 1, -1, 1, 1, -1]
    mid_b = [1, -1, 1, -1, 0, 1, -1, -1, 1]
    seq = ([0] * 7 + ltf_left + [1] + ltf_right + mid_a
           + ltf_left + [1] + ltf_right + mid_b
           + ltf_left + [1] + ltf_right + mid_a
           + ltf_left + [1] + ltf_right + [0] * 6)
    seq = np.array(seq, dtype=np.float64)
    assert seq.size == FFT_LEN
    return seq


# ---------------------------------------------------------------------------
# a-1  OFDM demodulation (MATLAB ofdmdemod call sites generate_maMIMO_LTF.m:336-338,
#      BER_test_maMIMO_LTF.m:324-326; python convention massiveMIMO_dataGenerator.py:437-453)
# ---------------------------------------------------------------------------
def ofdm_demod(ltf, nt):
    """ltf: complex [..., lenLTF] time-domain preamble of ONE rx antenna, lenLTF = 320*nt.
    Returns rxsym complex [..., 234, nt] (data bins x LTF symbol), i.e. MATLAB
    ``rxOFDM(:, 1:numTx, iRx)``.

    Convention: column-major split into nt symbols of 320 samples
    (massiveMIMO_dataGenerator.py:437-439); symbol offset = CP length so the FFT window
    is samples 64..319 of each symbol (:442-443); unscaled 256-point FFT (:452); fftshift
    ALONG THE FREQUENCY AXIS ONLY so DC lands on 1-based bin 129 (generate_maMIMO_LTF.m:99;
    :453 of the python file shifts both axes, which is a bug in dead code); keep the 234
    data bins (generate_maMIMO_LTF.m:98-102)."""
    ltf = np.asarray(ltf)
    if ltf.dtype != np.complex128:
        # the oracle computes in double, as MATLAB does: numpy >= 2 would transform a complex64 preamble in SINGLE precision
        # (~1e-7 relative; found by the plain-C statement of this file, oracle/csi_oracle_c.c, which disagreed at that level)
        ltf = ltf.astype(np.complex128)
    assert ltf.shape[-1] == SYM_LEN * nt
    sym = ltf.reshape(ltf.shape[:-1] + (nt, SYM_LEN))          # [..., s, n]
    win = sym[..., CP_LEN:CP_LEN + FFT_LEN]
    spec = np.fft.fft(win, n=FFT_LEN, axis=-1)
    spec = np.fft.fftshift(spec, axes=-1)                      # index i <-> MATLAB bin i+1
    data = spec[..., data_carrier_indices() - 1]               # [..., s, 234]
    return np.swapaxes(data, -1, -2)                           # [..., 234, s]


# ---------------------------------------------------------------------------
# a-2  LS channel estimate  (helperMIMOChannelEstimate.m:8-41)
# ---------------------------------------------------------------------------
def ls_from_rxsym(rxsym, P):
    """rxsym complex [..., 234, nt] (bins x symbol);  P [nt, nt] with ROW j = pilot mapping
    sequence of tx antenna j over the nt LTF symbols (MATLAB ``P(j,:)``).
    hD(:,j,i) = rxsym * Puse(:,j) ./ denom,  Puse = P'  (conjugate transpose, :24),
    denom = nltf .* ltf(CarriersLocations) (:26-27,:33-36).
    Returns H complex [..., 234, nt] (bins x tx)."""
    P = np.asarray(P)
    nt = P.shape[0]
    puse = P.conj().T                                           # [s, j]
    denom = nt * vht_ltf_256()[data_carrier_indices() - 1]      # [234]
    return (rxsym @ puse) / denom[:, None]


def ls_estimate(ltf, P):
    """Full LS path of one or many rx-antenna preambles.
    ltf complex [..., lenLTF]  ->  H complex [..., nt, 234]   (tx-major rows, matching the
    DNN output layout ``H[pkt, iRx, iTx, k]``, BER_test_maMIMO_LTF.m:191-195)."""
    nt = np.asarray(P).shape[0]
    h = ls_from_rxsym(ofdm_demod(ltf, nt), P)                   # [..., 234, nt]
    return np.swapaxes(h, -1, -2)


# ---------------------------------------------------------------------------
# a-3  sample assembly  (massiveMIMO_dataGenerator.py:299-316)
# ---------------------------------------------------------------------------
def assemble_batch(dataset, d, list_ids, len_ltf=None):
    """Mirror of DataGenerator.__data_generation for datasource 'matlab_maMimo',
    method 'default'.  ``dataset`` is the pickle dict written by
    create_massiveMIMO_CSIest_dnn_dataset.py:125 ({X:[N,2] (key,iTx), y:{real,imag},
    LTF:{key:{real,imag}}, P, simParams}).  Returns (Xsig [B,lenLTF,1], Xp [B,nTX], y [B,234])
    as float64 (np.empty default dtype, :303-305)."""
    n = len(list_ids)
    first_key = dataset['X'][list_ids[0], 0]
    if len_ltf is None:
        len_ltf = dataset['LTF'][first_key][d].shape[0]          # loadDataset :27
    nt = dataset['simParams']['nTX']
    xsig = np.empty((n, len_ltf, 1))
    xp = np.empty((n, nt))
    y = np.empty((n, dataset['y'][d].shape[1]))
    for i, six in enumerate(list_ids):
        xsig[i] = dataset['LTF'][dataset['X'][six, 0]][d][0:len_ltf][:, np.newaxis]   # :307-308
        xp[i] = dataset['P'][:, dataset['X'][six, 1]]                                  # :311
        y[i] = dataset['y'][d][six, :]                                                 # :314
    return xsig, xp, y


def pilot_rows_from_dataset_P(P_py):
    """dataset['P'] is the h5py read of the MATLAB matrix, i.e. its transpose
    (create_massiveMIMO_CSIest_dnn_dataset.py:37); the DNN sees ``P_py[:, iTx]`` =
    MATLAB ``P(iTx, :)``.  Returns the [nt, nt] array whose ROW t is that vector."""
    return np.ascontiguousarray(np.asarray(P_py).T)


def samples_from_packets(ltf, P, d):
    """Packed-array twin of assemble_batch.  ltf complex [Npkt, Nr, lenLTF]; P [nt,nt]
    with row t = pilot sequence of tx t.  Returns X [Npkt*Nr*Nt, lenLTF+Nt] = the
    Flatten+Concatenate input of the FC model (DNN.py:207-208) in dataset sample order
    s = p*Nr*Nt + r*Nt + t (create_massiveMIMO_CSIest_dnn_dataset.py:62)."""
    ltf = np.asarray(ltf)
    npkt, nr, len_ltf = ltf.shape
    nt = P.shape[0]
    part = ltf.real if d == 'real' else ltf.imag
    x = np.empty((npkt, nr, nt, len_ltf + nt), dtype=part.dtype)
    x[..., :len_ltf] = part[:, :, None, :]
    x[..., len_ltf:] = np.asarray(P, dtype=part.dtype)[None, None, :, :]
    return x.reshape(npkt * nr * nt, len_ltf + nt)


# ---------------------------------------------------------------------------
# a-4..a-7  the FC regressor  (massiveMIMO_CSI_prediction_DNN.py:176-234)
# ---------------------------------------------------------------------------
def bn_inference(x, gamma, beta, mean, var, eps=BN_EPS):
    """keras BatchNormalization at inference on a 2-D input (non-fused path):
    inv = gamma * rsqrt(var + eps);  y = x*inv + (beta - mean*inv)."""
    dt = x.dtype
    inv = (gamma / np.sqrt(var + dt.type(eps))).astype(dt)
    return x * inv + (beta - mean * inv).astype(dt)


def fc_forward(x, w, dtype=np.float64):
    """Forward of ONE model (real or imag).  x [B, lenLTF+Nt] already flattened and
    concatenated ([flatten(seq_in), seq_p], DNN.py:207-208).  ``w`` is a dict with keras
    names: fc_dense{i}.kernel [in,out], fc_dense{i}.bias, optional bn{i}.gamma/beta/
    moving_mean/moving_variance, fc_regressor.kernel/bias, and 'bn_eps'.
    Dense relu (DNN.py:211-214) -> BatchNormalization (:215-219, applied AFTER the relu)
    -> Dropout (:222, identity at inference) ... -> Dense linear (:227)."""
    h = np.asarray(x, dtype=dtype)
    eps = float(w.get('bn_eps', BN_EPS))
    cv = lambda a: np.asarray(a, dtype=dtype)          # no copy when already of that dtype
    i = 0
    while f'fc_dense{i}.kernel' in w:
        h = h @ cv(w[f'fc_dense{i}.kernel']) + cv(w[f'fc_dense{i}.bias'])
        h = np.maximum(h, dtype(0))
        if f'bn{i}.gamma' in w:
            h = bn_inference(h, cv(w[f'bn{i}.gamma']), cv(w[f'bn{i}.beta']), cv(w[f'bn{i}.moving_mean']),
                             cv(w[f'bn{i}.moving_variance']), eps)
        i += 1
    return h @ cv(w['fc_regressor.kernel']) + cv(w['fc_regressor.bias'])


# ---------------------------------------------------------------------------
# a-8  predict over packets in dataset order; real/imag recombination
#      (DNN.py:339-346; inference.py:29-31; output layout BER_test_maMIMO_LTF.m:191-195)
# ---------------------------------------------------------------------------
def predict_packets(ltf, P, w_real, w_imag, dtype=np.float64, pkt_batch=None):
    """Literal restatement of the reference test loop: one batch of Nt*Nr rows per packet
    (set_batchsize(nTX*nRX), DNN.py:339) through the NAIVE network (no layer-1 sharing),
    for the real then the imag model.  Returns (out_real, out_imag) float [Npkt,Nr,Nt,234]."""
    ltf = np.asarray(ltf)
    npkt, nr, _ = ltf.shape
    nt = P.shape[0]
    outs = []
    for d, w in (('real', w_real), ('imag', w_imag)):
        n_out = w['fc_regressor.bias'].shape[0]
        out = np.empty((npkt, nr, nt, n_out), dtype=dtype)
        step = pkt_batch or 1
        for p0 in range(0, npkt, step):
            x = samples_from_packets(ltf[p0:p0 + step], np.asarray(P), d)
            y = fc_forward(x, w, dtype)
            out[p0:p0 + step] = y.reshape(-1, nr, nt, n_out)
        outs.append(out)
    return outs[0], outs[1]


# ---------------------------------------------------------------------------
# bf16-operand emulation (BASELINE.json config 3).  Not a reference behaviour: the reference
# runs fp32.  It defines what the CSI_DTYPE_BF16 kernels are expected to compute - operands
# (inputs, weights, hidden activations) rounded to bfloat16 round-to-nearest-even, exact
# products, wide accumulation, fp32 bias / relu epilogue, BatchNormalization folded into the next layer's
# operands before they are rounded - so that their error against the
# fp64 oracle can be split into 'format' and 'implementation' parts.
# ---------------------------------------------------------------------------
def bf16_round(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def fc_forward_bf16(x, w):
    """Literal network with bf16 operands.  x [B, lenLTF+Nt] float.

    The BatchNormalization behind a hidden layer is carried by the NEXT layer's operands, as csi_load_weights does
    in bf16 mode:  (relu(z) sc + sh) W + b = relu(z) (diag(sc) W) + (b + sh W)  - the scaled kernel rows are rounded
    to bf16 once, the shift product stays wide - so every hidden activation is bf16(relu(z))."""
    eps = float(w.get('bn_eps', BN_EPS))
    h = bf16_round(x).astype(np.float64)
    sc = sh = None
    i = 0
    while True:
        name = f'fc_dense{i}' if f'fc_dense{i}.kernel' in w else 'fc_regressor'
        k = w[name + '.kernel'].astype(np.float64)
        b = w[name + '.bias'].astype(np.float64)
        if sc is not None:
            b = b + sh @ k
            k = k * sc[:, None]
        z = h @ bf16_round(k.astype(np.float32)).astype(np.float64) + b.astype(np.float32).astype(np.float64)
        if name == 'fc_regressor':
            return z.astype(np.float32)
        z = np.maximum(z.astype(np.float32), np.float32(0))
        sc = sh = None
        if f'bn{i}.gamma' in w:
            inv = (np.float32(1) / np.sqrt(w[f'bn{i}.moving_variance'].astype(np.float32) + np.float32(eps))) * w[f'bn{i}.gamma'].astype(np.float32)
            sc = inv.astype(np.float64)
            sh = (w[f'bn{i}.beta'].astype(np.float32) - w[f'bn{i}.moving_mean'].astype(np.float32) * inv).astype(np.float64)
        h = bf16_round(z).astype(np.float64)
        i += 1


def predict_packets_bf16(ltf, P, w_real, w_imag):
    ltf = np.asarray(ltf)
    npkt, nr, _ = ltf.shape
    nt = P.shape[0]
    outs = []
    for d, w in (('real', w_real), ('imag', w_imag)):
        x = samples_from_packets(ltf, np.asarray(P, dtype=np.float32), d)
        outs.append(fc_forward_bf16(x, w).reshape(npkt, nr, nt, -1))
    return outs[0], outs[1]


# ---------------------------------------------------------------------------
# Emulation of the split-f16 GEMM engine of fp32 contexts (csrc/gemm_hs.hip.h): every operand is
# carried as hi + lo float16 halves of 2^shift * x and a product is a_hi*b_hi + a_hi*b_lo + a_lo*b_hi.
# Accumulation here is fp64, so this isolates the REPRESENTATION error of the scheme (the device
# accumulates in fp32 like the fp32 MFMA path does).
# ---------------------------------------------------------------------------
def split_f16(x, shift):
    """(hi, lo) float64 arrays with hi + lo ~= 2^shift * x, both exactly float16 numbers."""
    xs = np.asarray(x, dtype=np.float32) * np.float32(2.0 ** shift)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def weight_shift(w):
    """Power of two that puts the largest |w| into [2^12, 2^13) (csi_load_weights)."""
    m = float(np.max(np.abs(w)))
    return 13 - int(np.frexp(m)[1]) if m > 0 else 0


def dense_split_f16(a, w, a_shift):
    """a [M,K] @ w [K,N] with split operands and the dropped lo*lo term, scaled back."""
    sw = weight_shift(w)
    a_hi, a_lo = split_f16(a, a_shift)
    w_hi, w_lo = split_f16(w, sw)
    return (a_hi @ w_hi + a_hi @ w_lo + a_lo @ w_hi) * 2.0 ** -(a_shift + sw)


def fc_forward_split_f16(x, w, in_shift=4, act_shift=4):
    """Literal network, every dense product through dense_split_f16; bias / relu / BN in fp32."""
    eps = float(w.get('bn_eps', BN_EPS))
    h = np.asarray(x, dtype=np.float32)
    i = 0
    while f'fc_dense{i}.kernel' in w:
        z = dense_split_f16(h, w[f'fc_dense{i}.kernel'], in_shift if i == 0 else act_shift) + w[f'fc_dense{i}.bias'].astype(np.float64)
        z = np.maximum(z.astype(np.float32), np.float32(0))
        if f'bn{i}.gamma' in w:
            z = bn_inference(z, w[f'bn{i}.gamma'].astype(np.float32), w[f'bn{i}.beta'].astype(np.float32),
                             w[f'bn{i}.moving_mean'].astype(np.float32), w[f'bn{i}.moving_variance'].astype(np.float32), eps)
        h = z
        i += 1
    return (dense_split_f16(h, w['fc_regressor.kernel'], in_shift if i == 0 else act_shift) + w['fc_regressor.bias'].astype(np.float64)).astype(np.float32)


def predict_packets_shared(ltf, P, w_real, w_imag, dtype=np.float64):
    """Same function as predict_packets evaluated with layer 0 shared across the Nt pairs of an rx
    antenna (z0 = LTF.W0[:lenLTF] + P_t.W0[lenLTF:] + b0) - algebraically identical, Nt times
    cheaper; used where the literal form is too slow for the checker (Nt = 128 shapes)."""
    ltf = np.asarray(ltf)
    npkt, nr, len_ltf = ltf.shape
    nt = P.shape[0]
    outs = []
    for d, w in (('real', w_real), ('imag', w_imag)):
        eps = float(w.get('bn_eps', BN_EPS))
        part = (ltf.real if d == 'real' else ltf.imag).astype(dtype).reshape(npkt * nr, len_ltf)
        k0 = w['fc_dense0.kernel'].astype(dtype)
        l0 = part @ k0[:len_ltf]
        t = np.asarray(P, dtype=dtype) @ k0[len_ltf:] + w['fc_dense0.bias'].astype(dtype)
        h = np.maximum(l0[:, None, :] + t[None, :, :], 0).reshape(npkt * nr * nt, -1)
        i = 0
        while True:
            if f'bn{i}.gamma' in w:
                h = bn_inference(h, w[f'bn{i}.gamma'].astype(dtype), w[f'bn{i}.beta'].astype(dtype),
                                 w[f'bn{i}.moving_mean'].astype(dtype), w[f'bn{i}.moving_variance'].astype(dtype), eps)
            i += 1
            if f'fc_dense{i}.kernel' not in w:
                break
            h = np.maximum(h @ w[f'fc_dense{i}.kernel'].astype(dtype) + w[f'fc_dense{i}.bias'].astype(dtype), 0)
        y = h @ w['fc_regressor.kernel'].astype(dtype) + w['fc_regressor.bias'].astype(dtype)
        outs.append(y.reshape(npkt, nr, nt, -1))
    return outs[0], outs[1]


def recombine(out_real, out_imag):
    """inference.py:31  ``output_real + 1j * output_imag``."""
    return out_real + 1j * out_imag


# ---------------------------------------------------------------------------
# a-9  CSIPredictor pre/post-processing  (inference.py:35-68)
# ---------------------------------------------------------------------------
def preprocess_rice_renew(input_batch):
    """inference.py:39-44: requires complex128, otherwise the reference prints an error
    and exits(-1); here a ValueError carries the same message."""
    if np.asarray(input_batch).dtype != np.complex128:
        raise ValueError('[CSIPredictor] ERROR: Input batch must be of type np.complex128')
    return input_batch


def postprocess_rice_renew(output_batch):
    """inference.py:52-66: 52 outputs -> 64 bins [0*6 | o[0:26] | 0 | o[26:52] | 0*5],
    then ifftshift along axis 1."""
    output_batch = np.asarray(output_batch)
    if output_batch.shape[1] != 52:
        raise ValueError('[CSIPredictor] ERROR: Output samples must have size 52 (assuming FFTLen = 64).')
    b = output_batch.shape[0]
    tmp = np.concatenate((np.zeros((b, 6)), output_batch[:, 0:26], np.zeros((b, 1)),
                          output_batch[:, 26:], np.zeros((b, 5))), axis=1)
    return np.fft.ifftshift(tmp, axes=1)


# ---------------------------------------------------------------------------
# SURVEY 8f-3  LMMSE smoothing of an LS estimate  (LMMSE_ce.m:23-39, called per link from
#              helperMIMOChannelEstimate.m:37-39 with Nfft = Np = 234, Nps = 1)
# ---------------------------------------------------------------------------
def lmmse_ce(h_tilde, nfft, np_, nps, h, snr_db):
    """Literal restatement of LMMSE_ce.m.  h_tilde complex [Np] (one link's LS estimate), h the
    vector the reference passes as 'channel impulse response' (generate_maMIMO_LTF.m:342 hands it
    the scatterer delays h_tau), snr_db scalar.  Returns complex [Nfft]."""
    h = np.asarray(h, dtype=np.complex128).reshape(-1)
    snr = 10.0 ** (snr_db * 0.1)                                       # :23
    k = np.arange(h.size)                                              # :27
    hh = np.vdot(h, h)                                                 # h*h'
    tmp = h * np.conj(h) * k                                           # :28
    r = np.sum(tmp) / hh
    r2 = (tmp @ k) / hh                                                # :29
    tau_rms = np.sqrt(r2 - r ** 2)                                     # :30
    df = 1.0 / nfft                                                    # :31
    j2pi_tau_df = 1j * 2 * np.pi * tau_rms * df                        # :32
    K1 = np.arange(nfft)[:, None]
    K2 = np.arange(np_)[None, :]
    rf = 1.0 / (1.0 + j2pi_tau_df * (K1 - K2 * nps))                   # :33-34
    K3 = np.arange(np_)[:, None]
    K4 = np.arange(np_)[None, :]
    rf2 = 1.0 / (1.0 + j2pi_tau_df * nps * (K3 - K4))                  # :35-36
    rpp = rf2 + np.eye(np_) / snr                                      # :38
    return rf @ np.linalg.inv(rpp) @ np.asarray(h_tilde, dtype=np.complex128)   # :39


def lmmse_estimate(h_ls, h, snr_db):
    """helperMIMOChannelEstimate.m:33-39 with isMMSE: every link (tx j, rx i) of every packet is
    smoothed on its own.  h_ls complex [npkt, nr, nt, 234]; h [npkt, L]; snr_db [npkt, nr]
    (SNR(i) per rx antenna).  Returns complex128 [npkt, nr, nt, 234]."""
    h_ls = np.asarray(h_ls)
    npkt, nr, nt, n = h_ls.shape
    out = np.empty(h_ls.shape, dtype=np.complex128)
    for p in range(npkt):
        for i in range(nr):
            for j in range(nt):
                out[p, i, j] = lmmse_ce(h_ls[p, i, j], n, n, 1, h[p], snr_db[p, i])
    return out


# ---------------------------------------------------------------------------
# a-12  per-link NMSE  (BER_test_maMIMO_LTF.m:675-686)
# ---------------------------------------------------------------------------
def nmse_subk(h_ref, h_est):
    """h_* complex [..., Nr, Nt, 234] (or any [..., link, 234]): per link
    ||ref-est||^2 / ||ref||^2 over the bins, mean over all links."""
    h_ref = np.asarray(h_ref, dtype=np.complex128)          # double, whatever the caller holds (complex64 sums would be single)
    diff = h_ref - np.asarray(h_est, dtype=np.complex128)
    num = np.sum(np.abs(diff) ** 2, axis=-1)
    den = np.sum(np.abs(h_ref) ** 2, axis=-1)
    return float(np.mean(num / den))


def row_rel_err(y, y_ref):
    """Norm-relative error per output row, the form the 1e-5 fp32 contract is stated in
    (SURVEY.md section 7 'hard parts'): max over rows of ||y-ref||2 / ||ref||2."""
    y = np.asarray(y, dtype=np.float64).reshape(-1, np.asarray(y).shape[-1])
    r = np.asarray(y_ref, dtype=np.float64).reshape(y.shape)
    return float(np.max(np.linalg.norm(y - r, axis=1) / np.linalg.norm(r, axis=1)))


# ---------------------------------------------------------------------------
# Synthetic data (the reference ships no weights and no datasets)
# ---------------------------------------------------------------------------
def hadamard(n):
    """Sylvester-Hadamard matrix; stand-in for the un-vendored helperGetP(numSTS)
    (helperMIMOChannelEstimate.m:13).  In the real pipeline P is read from the dataset."""
    assert n & (n - 1) == 0
    h = np.array([[1.0]])
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def make_weights(rng, d_in, hidden, n_out, use_bn=True, dtype=np.float32):
    """Keras-initialiser-like random weights: glorot-uniform kernels (DNN.py:213,227),
    small biases, non-trivial BN statistics so that folding errors would show."""
    w = {'bn_eps': BN_EPS}
    fan_in = d_in
    for i, h in enumerate(hidden):
        lim = np.sqrt(6.0 / (fan_in + h))
        w[f'fc_dense{i}.kernel'] = rng.uniform(-lim, lim, (fan_in, h)).astype(dtype)
        w[f'fc_dense{i}.bias'] = (0.01 * rng.standard_normal(h)).astype(dtype)
        if use_bn:
            w[f'bn{i}.gamma'] = rng.uniform(0.5, 1.5, h).astype(dtype)
            w[f'bn{i}.beta'] = (0.1 * rng.standard_normal(h)).astype(dtype)
            w[f'bn{i}.moving_mean'] = (0.1 * rng.standard_normal(h)).astype(dtype)
            w[f'bn{i}.moving_variance'] = rng.uniform(0.5, 1.5, h).astype(dtype)
        fan_in = h
    lim = np.sqrt(6.0 / (fan_in + n_out))
    w['fc_regressor.kernel'] = rng.uniform(-lim, lim, (fan_in, n_out)).astype(dtype)
    w['fc_regressor.bias'] = (0.01 * rng.standard_normal(n_out)).astype(dtype)
    return w


def make_structured_packets(rng, npkt, nr, P, snr_db=None, n_taps=8):
    """Packets with a KNOWN channel: H[k,j,i] = FFT of an n_taps complex Gaussian CIR per
    link; rx[k,s,i] = ltf[k] * sum_j H[k,j,i] P[j,s] on all non-null bins; OFDM-modulated
    (ifftshift -> ifft -> CP); the amplitude scaling of generate_maMIMO_LTF.m:303-304 is left
    out so that the known answer is H itself.  Returns
    (ltf complex128 [npkt,nr,320*nt], H complex128 [npkt,nr,nt,234]).  With snr_db=None the
    LS estimate of the returned preamble equals H to rounding when P P^H = nt I."""
    P = np.asarray(P, dtype=np.float64)
    nt = P.shape[0]
    cir = (rng.standard_normal((npkt, nr, nt, n_taps)) + 1j * rng.standard_normal((npkt, nr, nt, n_taps)))
    cir *= np.exp(-0.5 * np.arange(n_taps))[None, None, None, :] / np.sqrt(2.0)
    hfull = np.fft.fftshift(np.fft.fft(cir, n=FFT_LEN, axis=-1), axes=-1)     # [p,r,j,256]
    ltf_seq = vht_ltf_256()
    # frequency-domain LTF symbols: X[p,r,s,k] = ltf[k] * sum_j H[p,r,j,k] P[j,s]
    xf = np.einsum('prjk,js->prsk', hfull, P) * ltf_seq[None, None, None, :]
    xt = np.fft.ifft(np.fft.ifftshift(xf, axes=-1), axis=-1)                  # [p,r,s,256]
    sym = np.concatenate([xt[..., -CP_LEN:], xt], axis=-1)                    # CP + body
    ltf = sym.reshape(npkt, nr, nt * SYM_LEN)
    if snr_db is not None:
        sig_pow = np.mean(np.abs(ltf) ** 2)
        npow = sig_pow / (10.0 ** (snr_db / 10.0))
        noise = (rng.standard_normal(ltf.shape) + 1j * rng.standard_normal(ltf.shape)) * np.sqrt(npow / 2.0)
        ltf = ltf + noise
    h_true = hfull[..., data_carrier_indices() - 1]                           # [p,r,j,234]
    return ltf, h_true


# ------------------------------------------------------------------------------------------------
# Training step.  Restates, in fp64 numpy, what keras does for the reference's
# fit() (massiveMIMO_CSI_prediction_DNN.py:191-193 AWGN on the LTF input, :211-227 the layer stack,
# :272-273 Adam + mse).  Keras is not available here: BatchNormalization (training: biased batch
# variance, eps 1e-3, running statistics moved with momentum 0.99), Dropout (kept units scaled by
# 1/(1-rate)), mse (mean over all elements) and Adam (lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
# p -= lr_t*m/(sqrt(v)+1e-7)) follow the published keras definitions - parity unpinned.
def train_forward_backward(w, x, y, use_bn=True, bn_eps=1e-3, noise=None, masks=None, dropout=0.0):
    """One forward/backward pass.  w: dict of keras-named tensors; x [B, d_in], y [B, n_out];
    noise: array added to x (already scaled) or None; masks: list of keep masks (bool [B, width])
    per hidden layer or None.  Returns (loss, grads dict, batch statistics [(mean, var), ...])."""
    w = {k: np.asarray(v, np.float64) for k, v in w.items() if k != 'bn_eps'}
    n_hidden = sum(1 for k in w if k.startswith('fc_dense') and k.endswith('.kernel'))
    h = np.asarray(x, np.float64) + (0.0 if noise is None else np.asarray(noise, np.float64))
    B = h.shape[0]
    cache = []
    stats = []
    for i in range(n_hidden):
        inp = h
        a = np.maximum(inp @ w[f'fc_dense{i}.kernel'] + w[f'fc_dense{i}.bias'], 0.0)
        if use_bn:
            mu, var = a.mean(0), a.var(0)
            istd = 1.0 / np.sqrt(var + bn_eps)
            xhat = (a - mu) * istd
            o = xhat * w[f'bn{i}.gamma'] + w[f'bn{i}.beta']
            stats.append((mu, var))
        else:
            xhat, istd, o = None, None, a
        keep = None
        if masks is not None and i < n_hidden - 1 and dropout > 0.0:
            keep = np.asarray(masks[i], bool)
            o = np.where(keep, o / (1.0 - dropout), 0.0)
        cache.append((inp, a, xhat, istd, keep))
        h = o
    out = h @ w['fc_regressor.kernel'] + w['fc_regressor.bias']
    diff = out - np.asarray(y, np.float64)
    loss = float(np.mean(diff ** 2))
    g = {}
    dout = 2.0 * diff / diff.size
    g['fc_regressor.kernel'] = h.T @ dout
    g['fc_regressor.bias'] = dout.sum(0)
    dh = dout @ w['fc_regressor.kernel'].T
    for i in range(n_hidden - 1, -1, -1):
        inp, a, xhat, istd, keep = cache[i]
        if keep is not None:
            dh = np.where(keep, dh / (1.0 - dropout), 0.0)
        if use_bn:
            g[f'bn{i}.gamma'] = (dh * xhat).sum(0)
            g[f'bn{i}.beta'] = dh.sum(0)
            da = w[f'bn{i}.gamma'] * istd / B * (B * dh - g[f'bn{i}.beta'] - xhat * g[f'bn{i}.gamma'])
        else:
            da = dh
        dz = da * (a > 0)
        g[f'fc_dense{i}.kernel'] = inp.T @ dz
        g[f'fc_dense{i}.bias'] = dz.sum(0)
        if i > 0:
            dh = dz @ w[f'fc_dense{i}.kernel'].T
    return loss, g, stats


def adam_init(w):
    keys = [k for k in w if 'moving' not in k and k != 'bn_eps']
    return {'t': 0, 'm': {k: np.zeros_like(np.asarray(w[k], np.float64)) for k in keys},
            'v': {k: np.zeros_like(np.asarray(w[k], np.float64)) for k in keys}}


def train_step_reference(w, state, x, y, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, use_bn=True, bn_eps=1e-3,
                         bn_momentum=0.99, noise=None, masks=None, dropout=0.0):
    """One optimiser step in fp64; returns (loss, new weights, grads).  state from adam_init (updated in place)."""
    loss, g, stats = train_forward_backward(w, x, y, use_bn, bn_eps, noise, masks, dropout)
    state['t'] += 1
    t = state['t']
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    new = {k: np.asarray(v, np.float64).copy() for k, v in w.items() if k != 'bn_eps'}
    for k, gk in g.items():
        state['m'][k] = beta1 * state['m'][k] + (1.0 - beta1) * gk
        state['v'][k] = beta2 * state['v'][k] + (1.0 - beta2) * gk * gk
        new[k] = new[k] - lr_t * state['m'][k] / (np.sqrt(state['v'][k]) + eps)
    for i, (mu, var) in enumerate(stats):
        new[f'bn{i}.moving_mean'] = new[f'bn{i}.moving_mean'] * bn_momentum + mu * (1.0 - bn_momentum)
        new[f'bn{i}.moving_variance'] = new[f'bn{i}.moving_variance'] * bn_momentum + var * (1.0 - bn_momentum)
    return loss, new, g


def eval_loss_reference(w, x, y, use_bn=True, bn_eps=1e-3):
    """Inference-mode mse (val_loss)."""
    ww = dict(w)
    ww['bn_eps'] = bn_eps
    if not use_bn:
        ww = {k: v for k, v in ww.items() if not k.startswith('bn') or k == 'bn_eps'}
    out = fc_forward(np.asarray(x, np.float64), ww, np.float64)
    return float(np.mean((out - np.asarray(y, np.float64)) ** 2))
