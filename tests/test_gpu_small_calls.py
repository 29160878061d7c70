"""GPU tests (-m gpu; every call through the C-ABI of libcsi_mamimo.so, checked against the numpy oracle on identical seeded inputs at the
1e-5 norm-relative contract of BASELINE.json unless a test states its own): small and mid-size calls: the one-packet path (massiveMIMO_CSI_prediction_DNN.py:339-346), routing, fuzzed shapes, the 500-packet call of full_pipeline_maMIMO_DNNEst.sh:44-48."""
import os
import sys


import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


TOL = 1e-5


def _weights(oracle, seed, nt, hidden, use_bn=True, n_out=234):
    rng = np.random.default_rng(seed)
    d_in = 320 * nt + nt
    return (oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn),
            oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn))


SMALL_CASES = [
    # nt, nr, npkt, hidden, use_bn, n_out
    (32, 4, 1, (1024, 1024), True, 234),     # DNN.py:339-346: one packet of the shipped model = 128 rows
    (32, 4, 2, (1024, 1024), True, 234),     # 8 preambles: the largest call the path takes at Nr = 4
    (8, 2, 1, (64, 64), True, 234),          # 16 pair rows: one row tile, three of its four waves without rows
    (8, 2, 4, (64, 64), False, 234),         # --useBN off
    (4, 1, 7, (32, 48, 40), True, 234),      # three hidden layers (ping-pong buffers), K = 1280: the k loop's tail
    (8, 2, 3, (100, 36), True, 234),         # widths that are not multiples of 16: K tail of the tiles over zero-padded weights
    (8, 2, 2, (64,), True, 52),              # single hidden layer: the per-pair layer IS the regressor; 52 outputs (inference.py:58)
    (12, 2, 3, (40, 24), True, 234),         # Nt not a power of two: pair rows straddle (packet, rx) boundaries inside a tile
    (64, 2, 1, (96, 64), True, 234),         # Nt = 64
    (128, 1, 1, (64, 64), True, 234),        # Nt = 128: 128 pair rows from ONE preamble
    (12, 2, 3, (72, 1100), True, 234),       # a wide per-pair layer on the 32 x 32 tiles: ragged rows (72) and columns (1100), K = 72 (3 groups for 16 k-parts); regressor K = 1100
    (32, 4, 8, (1024, 1024), True, 234),     # 8 packets of the shipped model: 32 preambles - layer 0 on the tiles, its epilogue writes the per-pair input (EPI_H1)
    (32, 2, 16, (1024, 1024), True, 234),    # 16 packets of Nr = 2: 32 preambles, 1024 pair rows - the largest call the path takes by default
    (8, 2, 12, (64, 64), True, 234),         # 24 preambles, ragged row tiles
    (4, 1, 40, (32, 48, 40), False, 234),    # 40 preambles of Nt = 4, three hidden layers, no BN, K = 1280
    (12, 2, 9, (40, 24), True, 52),          # 18 preambles, Nt = 12, 52 outputs
    (8, 2, 3, (512, 320), True, 234),        # ("small_fused" = 0 leg: the split-K latency path of every general kernel, K >= 256)
    (8, 2, 1, (256,), False, 234),           # one packet, single hidden layer, no BN (general kernels: split-K regressor)
]


def _pilot(rng, nt, orthogonal=True):
    from oracle import csi_oracle as o
    if orthogonal:
        P = o.hadamard(nt)
        return (P[rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[:, None]).astype(np.float64)
    return rng.integers(-3, 4, (nt, nt)).astype(np.float64)


def _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn=True, n_out=234, **kw):
    e = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=n_out, use_bn=use_bn, **kw)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    return e


def test_config1_csipredictor_500_packets_0db(pkg, oracle, tmp_path):
    """BASELINE configs[0]: Nt=32, Nr=4, TEST_Npkt=500, SNR 0 dB through the reference's deployment surface -
    CSIPredictor(model_path, experiment='matlab_maMimo').inference (inference.py:24-32) on the shipped architecture, the
    model folders written the way the reference's test run leaves them (DNN.py:411), LS through estimate()."""
    nt, nr, npkt, hidden = 32, 4, 500, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = pkg.synth.hadamard(nt)
    e0 = pkg.CsiEngine(nt, nr, hidden=hidden)
    for d, w in (('real', w_re), ('imag', w_im)):
        pkg.CSIModel(e0, d).load_weights(w).save(str(tmp_path / f'{d}_keras_model'), pilot=P)
    e0.close()
    pred = pkg.CSIPredictor(str(tmp_path), experiment='matlab_maMimo')
    ltf = np.concatenate([blk for _, _, blk in pkg.synth.mixed_snr_batch(2024, nr, P, per_level=npkt, levels=(0.0,))]).astype(np.complex128)
    assert ltf.shape == (npkt, nr, 320 * nt)
    csi = pred.inference(ltf)
    assert csi.shape == (npkt, nr, nt, 234) and csi.dtype == np.complex64 and np.isfinite(csi.view(np.float32)).all()
    pick = [0, 1, 249, 498, 499]
    r_re, r_im = oracle.predict_packets_shared(ltf[pick].astype(np.complex64), P, w_re, w_im)
    ref = oracle.recombine(r_re, r_im)
    assert rel_rows(np.concatenate([csi[pick].real, csi[pick].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # literal Model.predict of one packet's 128 samples (DNN.py:339-346) agrees with the packet path
    x = oracle.samples_from_packets(ltf[249:250].astype(np.complex64), P.astype(np.float32), 'real')
    lit = pred.model_real.predict(x)
    assert rel_rows(lit.reshape(nr, nt, 234), r_re[2]) < TOL
    dnn, h_ls = pred.estimate(ltf)
    assert np.array_equal(dnn, csi)
    ref_ls = oracle.ls_estimate(ltf[pick].astype(np.complex64), P)
    assert rel_rows(np.concatenate([h_ls[pick].real, h_ls[pick].imag], -1), np.concatenate([ref_ls.real, ref_ls.imag], -1)) < TOL
    # the figure the pipeline reports for this configuration: NMSE_subk (BER_test_maMIMO_LTF.m:675-686) of the DNN vs the LS labels
    assert abs(oracle.nmse_subk(h_ls[pick], csi[pick]) - oracle.nmse_subk(ref_ls, ref)) < 1e-4 * oracle.nmse_subk(ref_ls, ref)
    with pytest.raises(SystemExit) as ex:
        pred.inference(ltf.astype(np.complex64))
    assert ex.value.code == -1


@pytest.mark.parametrize('nt,nr,npkt,hidden,use_bn,n_out', SMALL_CASES)
def test_small_call_path_matches_oracle_and_general_kernels(pkg, oracle, nt, nr, npkt, hidden, use_bn, n_out):
    """small_l0_gemv_kernel + small_tile_gemm_kernel (PAIR / plain, both epilogues) against the fp64 oracle on the same packets, against
    the general kernels ("small_fused" = 0) to rounding, and run-to-run bit-identical (no atomics, fixed summation order)."""
    rng = np.random.default_rng(5000 + nt * 10 + npkt)
    w_re, w_im = _weights(oracle, 4321 + nt, nt, hidden, use_bn, n_out)
    P = _pilot(rng, nt, orthogonal=False)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn, n_out)
    assert e.get_option('small_fused') == 1
    e.set_option('small_rows_band', 65536)     # (the shipped model beyond 512 pair rows goes to the column-split band kernel by default: test below)
    n0 = e.get_option('small_calls')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('small_calls') == n0 + 1, 'a call of %d preambles must take the one-packet path' % (npkt * nr)
    assert o_re.shape == (npkt, nr, nt, n_out) and o_re.dtype == np.float32
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), 'run-to-run identical'
    e.set_option('small_fused', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('small_calls') == n0 + 2
    assert rel_rows(g_re, r_re) < TOL and rel_rows(g_im, r_im) < TOL
    assert rel_rows(o_re, g_re) < 5e-6 and rel_rows(o_im, g_im) < 5e-6


def test_small_call_limits_and_the_literal_predict(pkg, oracle):
    """More than "small_rows" pair rows (or more than 64 preambles) take the general kernels; the small path equals the literal un-shared network
    (csi_predict_samples: Keras Model.predict semantics, DNN.py:346) on the same packet; device-resident calls and a replayed hipGraph of
    the one-packet call give the same bits as the host call."""
    nt, nr, hidden = 32, 4, (256, 128)
    rng = np.random.default_rng(77)
    w_re, w_im = _weights(oracle, 99, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, 3, nr, P, snr_db=0.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    n0 = e.get_option('small_calls')
    e.set_option('small_rows', 256)
    e.predict(ltf[:3])                                        # 384 pair rows > "small_rows"
    assert e.get_option('small_calls') == n0
    e.set_option('small_rows', 4096)
    big = np.concatenate([ltf] * 6)[:17]                      # 17 packets = 68 preambles: beyond the path's 64 whatever "small_rows" says
    e.predict(big)
    assert e.get_option('small_calls') == n0
    o_re, o_im = e.predict(ltf[:1])
    assert e.get_option('small_calls') == n0 + 1
    # literal network on the assembled samples [LTF ; P_t] of packet 0 (gen.py:299-316 order)
    y_re = e.predict_samples('real', oracle.samples_from_packets(ltf[:1], P, 'real').astype(np.float32))
    y_im = e.predict_samples('imag', oracle.samples_from_packets(ltf[:1], P, 'imag').astype(np.float32))
    assert rel_rows(o_re.reshape(-1, 234), y_re) < 5e-6 and rel_rows(o_im.reshape(-1, 234), y_im) < 5e-6
    # device-resident + graph replay
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf[:1].real)), e.to_device(np.ascontiguousarray(ltf[:1].imag))
    q_re, q_im = e.empty((1, nr, nt, 234)), e.empty((1, nr, nt, 234))
    e.set_option('use_graph', 1)
    g0 = e.get_option('graph_replays')
    for _ in range(4):
        e.predict_device(d_re, d_im, 1, q_re, q_im)
        e.synchronize()
        assert np.array_equal(q_re.download(), o_re) and np.array_equal(q_im.download(), o_im)
    assert e.get_option('graph_replays') >= g0 + 2
    e.set_option('use_graph', 0)


def test_500_packet_call_of_the_pipeline(pkg, oracle):
    """full_pipeline_maMIMO_DNNEst.sh:44-48 hands `--test` the 500 packets of ONE SNR level: that call (64 000 pair rows, the split-f16
    engine + band kernel) against the fp64 oracle on packets spread over the batch, at the lowest and the highest SNR of setenv.sh."""
    nt, nr, hidden, npkt = 32, 4, (1024, 1024), 500
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    for snr in (-25.0, 10.0):
        rng = np.random.default_rng(int(2000 + snr))
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=snr)[0].astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
        o_re, o_im = e.predict(ltf)
        assert e.get_option('hs_launches') > 0 and e.get_option('hs_range_fallbacks') == 0 and e.get_option('band_launches') > 0
        sel = [0, 123, 250, 377, 499]
        r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
        assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
        h = e.ls_estimate(ltf)
        r_ls = oracle.ls_estimate(ltf[sel], P)
        assert rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), np.concatenate([r_ls.real, r_ls.imag], -1)) < TOL
        e.close()


def test_default_routing_of_small_calls_of_the_shipped_model(pkg, oracle):
    """Nt = 32, Nr = 4, FC 1024 x 1024: up to 2 packets (256 pair rows, 8 preambles) the one-packet path, from 3 packets the general path
    with the weight-streaming layer 0 and the column-split band kernel (measured faster from there, profiles/r05_band_split_probe.txt) -
    each against the fp64 oracle."""
    nt, nr, hidden = 32, 4, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    rng = np.random.default_rng(12)
    ltf = oracle.make_structured_packets(rng, 8, nr, P, snr_db=0.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    for npkt, small in ((1, True), (2, True), (3, False), (8, False)):
        s0, b0 = e.get_option('small_calls'), e.get_option('band_split_launches')
        o_re, o_im = e.predict(ltf[:npkt])
        assert (e.get_option('small_calls') == s0 + 1) == small and (e.get_option('band_split_launches') == b0 + 2) == (not small), npkt
        r_re, r_im = oracle.predict_packets(ltf[:npkt], P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL, npkt
    e.close()


def test_fuzz_small_call_shapes(pkg, oracle):
    """Seeded, bounded fuzz of the one-packet path: random antenna counts, preamble counts (1 ... 64, at most 1024 pair rows), one to three
    hidden layers of random widths (multiples of 4, not of the 16 / 32 tiles), with and without BatchNormalization, random output widths -
    every case against the fp64 oracle and against the general kernels, and the path must have been taken."""
    rng = np.random.default_rng(20250930)
    done = 0
    while done < 14:
        nt = int(rng.choice([4, 8, 12, 16, 32, 64]))
        nr = int(rng.integers(1, 5))
        npkt = int(rng.integers(1, 17))
        if npkt * nr > 64 or npkt * nr * nt > 1024:
            continue
        nh = int(rng.integers(1, 4))
        hidden = tuple(int(4 * rng.integers(2, 76)) for _ in range(nh))
        use_bn = bool(rng.integers(0, 2))
        n_out = int(rng.choice([52, 234, 100]))
        w_re, w_im = _weights(oracle, int(rng.integers(1 << 30)), nt, hidden, use_bn, n_out)
        P = _pilot(rng, nt, orthogonal=False)
        ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn, n_out)
        o_re, o_im = e.predict(ltf)
        case = (nt, nr, npkt, hidden, use_bn, n_out)
        assert e.get_option('small_calls') == 1, case
        r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL, case
        e.set_option('small_fused', 0)
        g_re, g_im = e.predict(ltf)
        assert rel_rows(o_re, g_re) < 5e-6 and rel_rows(o_im, g_im) < 5e-6, case
        e.close()
        done += 1


def test_fuzz_mid_size_calls(pkg, oracle):
    """Seeded, bounded fuzz of the mid-size routing: random antenna counts, 9 ... 600 rx preambles, hidden widths that do and do not admit
    the band kernel / its column split - whatever combination of l0_hs_stream_kernel, csi_band8(_cs), the separate split-engine kernels
    and the fp32 MFMA kernels serves the call, the result meets the contract against the fp64 oracle and repeats bit for bit."""
    rng = np.random.default_rng(20251001)
    done, streamed, split = 0, 0, 0
    while done < 12:
        nt = int(rng.choice([16, 32, 64]))
        nr = int(rng.integers(1, 5))
        npkt = int(rng.integers(3, 160))
        if not 9 <= npkt * nr <= 600 or npkt * nr * nt > 40000:
            continue
        h1 = int(rng.choice([128, 192, 208, 256, 320]))
        h2 = int(rng.choice([256, 512, 1024, 96]))
        hidden = (h1, h2)
        w_re, w_im = _weights(oracle, int(rng.integers(1 << 30)), nt, hidden)
        P = oracle.hadamard(nt)
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=float(rng.uniform(-10, 20)))[0].astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
        o_re, o_im = e.predict(ltf)
        case = (nt, nr, npkt, hidden)
        sel = sorted(set(int(i) for i in rng.integers(0, npkt, 3)) | {0, npkt - 1})
        r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
        assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL, case
        p_re, p_im = e.predict(ltf)
        assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), case
        streamed += e.get_option('l0_stream_launches') > 0
        split += e.get_option('band_split_launches') > 0
        assert e.get_option('hs_range_fallbacks') == 0, case
        e.close()
        done += 1
    assert streamed >= 6 and split >= 2, (streamed, split)


def test_mid_size_call_under_a_small_workspace_and_under_graph_replay(pkg, oracle):
    """The mid-size routing in the two situations that change its launch plan: a workspace budget that cuts the call into several packet
    chunks (each chunk takes the streaming layer 0 with its own k ranges and the column-split band kernel) and a captured hipGraph of the
    device call (one stream, so four column splits instead of two) - both against the fp64 oracle and the one-chunk eager call."""
    nt, nr, hidden, npkt = 32, 2, (256, 512), 40
    rng = np.random.default_rng(31)
    w_re, w_im = _weights(oracle, 41, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=2.0)[0].astype(np.complex64)
    sel = [0, 17, 39]
    r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    o_re, o_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == 2 and e.get_option('band_split_launches') == 2
    assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
    # ~13 packets per chunk: layer-0 slabs (33 k ranges of 26 preambles x 256) dominate the per-packet need
    per_pkt = nr * 256 * 4 * 33 + nr * nt * 512 * 4
    small = _engine(pkg, nt, nr, hidden, w_re, w_im, P, workspace_bytes=13 * per_pkt)
    c_re, c_im = small.predict(ltf)
    assert small.get_option('l0_stream_launches') >= 6, 'several chunks, each on the streaming kernel'
    assert rel_rows(c_re[sel], r_re) < TOL and rel_rows(c_im[sel], r_im) < TOL
    assert rel_rows(c_re, o_re) < 2e-6 and rel_rows(c_im, o_im) < 2e-6
    small.close()
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real)), e.to_device(np.ascontiguousarray(ltf.imag))
    q = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    e.estimate_device(d_re, d_im, npkt, *q); e.synchronize()
    eager = [a.download() for a in q]
    assert rel_rows(eager[0][sel], r_re) < TOL
    e.set_option('use_graph', 1)
    g0 = e.get_option('graph_replays')
    for _ in range(4):
        e.estimate_device(d_re, d_im, npkt, *q); e.synchronize()
    assert e.get_option('graph_replays') >= g0 + 2
    graph = [a.download() for a in q]
    assert rel_rows(graph[0][sel], r_re) < TOL and rel_rows(graph[1][sel], r_im) < TOL
    assert np.array_equal(graph[2], eager[2]) and np.array_equal(graph[3], eager[3]), 'LS planes: same kernel, same bits'
    assert rel_rows(graph[0], eager[0]) < 2e-6 and rel_rows(graph[1], eager[1]) < 2e-6
    e.set_option('use_graph', 0)
    e.close()


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(32, 4, 1, (1024, 1024)), (32, 4, 2, (1024, 1024)), (64, 4, 1, (512, 512)), (64, 2, 3, (256, 512)),
                                               (16, 2, 1, (256, 256)), (16, 1, 7, (128,))])
def test_ls_inside_the_layer0_launch_of_a_one_packet_call(pkg, oracle, nt, nr, npkt, hidden):
    """`small_ls_fused` = 1 (default): LS + DNN of a call of at most 8 preambles in 1 + n_hidden launches.  Both bodies are the functions the
    separate kernels call: every output bit-identical with the four-launch form, on EVERY one of 300 calls (an LS wave beside foreign matrix
    instructions once produced rare wrong items - profiles/r04_ls_ringb_variants.txt; layer 0 of this path has none), and inside 1e-5 of the oracle."""
    rng = np.random.default_rng(nt + npkt)
    d_in = 320 * nt + nt
    w_re, w_im = oracle.make_weights(rng, d_in, list(hidden), 234), oracle.make_weights(rng, d_in, list(hidden), 234)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=3.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real, np.float32)), e.to_device(np.ascontiguousarray(ltf.imag, np.float32))
    o = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    e.set_option('small_ls_fused', 0)
    e.estimate_device(d_re, d_im, npkt, *o)
    e.synchronize()
    base = [x.download(0, npkt) for x in o]
    e.set_option('small_ls_fused', 1)
    n0 = e.get_option('small_ls_launches')
    for it in range(300):
        for x in o[2:]:
            x.upload(np.zeros((npkt, nr, nt, 234), np.float32)) if it % 50 == 0 else None
        e.estimate_device(d_re, d_im, npkt, *o)
        e.synchronize()
        got = [x.download(0, npkt) for x in o]
        assert all(np.array_equal(a, b) for a, b in zip(got, base)), 'call %d differs from the four-launch form' % it
    assert e.get_option('small_ls_launches') == n0 + 300 and e.get_option('small_calls') >= 300
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    r_ls = oracle.ls_estimate(ltf, P)
    assert rel_rows(base[0], r_re) < 1e-5 and rel_rows(base[1], r_im) < 1e-5
    assert rel_rows(base[2], r_ls.real) < 1e-5 and rel_rows(base[3], r_ls.imag) < 1e-5
    # a pilot matrix outside the Sylvester order (table-driven LS kernel) keeps the separate LS launch
    if nt >= 8:
        P2 = P[:, ::-1].copy()
        e.set_pilot(P2)
        n1 = e.get_option('small_ls_launches')
        e.estimate_device(d_re, d_im, npkt, *o)
        e.synchronize()
        assert e.get_option('small_ls_launches') == n1
    e.close()


@pytest.mark.parametrize('dtype,nt,sizes', [('bf16', 64, (256, 500, 600)), ('f32', 32, (24, 256, 700))])
def test_ls_planes_of_two_stream_calls_equal_the_one_stream_call(pkg, oracle, dtype, nt, sizes):
    """Round 6.  csi_estimate_device runs the two component models of a mid-size call on two streams; fp32 contexts fork the second one in FRONT of the
    LS kernel.  In bf16 contexts that put bf16-MFMA waves of the imag model's layer 0 beside the LS waves and the LS planes came back with wrong items
    (19 of 20 calls at 500 ... 1000 packets, profiles/r06_small_calls.txt): there the fork sits behind the LS kernel now.  Every call of every size: LS
    planes bit-identical with the one-stream call, DNN planes run-to-run identical, a sampled packet against the oracle."""
    nr, hidden = 4, (256, 256)
    rng = np.random.default_rng(5)
    w = [oracle.make_weights(rng, 320 * nt + nt, list(hidden), 234) for _ in range(2)]
    P = oracle.hadamard(nt)
    e = pkg.CsiEngine(nt, nr, hidden=hidden, dtype=dtype)
    e.load_weights('real', w[0])
    e.load_weights('imag', w[1])
    e.set_pilot(P)
    for n in sizes:
        d_re, d_im = e.empty((n, nr, e.len_ltf)), e.empty((n, nr, e.len_ltf))
        e.synth_white(11, 0, n, d_re, d_im)
        o = [e.empty((n, nr, nt, 234)) for _ in range(4)]
        e.set_option('small_call_overlap', 0)
        e.estimate_device(d_re, d_im, n, *o)
        e.synchronize()
        ref = [a.download() for a in o[2:]]
        ltf = d_re.download(n - 1, 1) + 1j * d_im.download(n - 1, 1)
        r_ls = oracle.ls_estimate(ltf, P)
        assert rel_rows(ref[0][-1:], r_ls.real) < 1e-5 and rel_rows(ref[1][-1:], r_ls.imag) < 1e-5
        e.set_option('small_call_overlap', 1)
        dnn0 = None
        for it in range(12):
            e.estimate_device(d_re, d_im, n, *o)
            e.synchronize()
            got = [a.download() for a in o]
            assert np.array_equal(got[2], ref[0]) and np.array_equal(got[3], ref[1]), '%s %d packets, call %d: LS planes differ from the one-stream call' % (dtype, n, it)
            dnn0 = dnn0 or got[:2]
            assert np.array_equal(got[0], dnn0[0]) and np.array_equal(got[1], dnn0[1])
        del d_re, d_im, o
    e.close()
