"""GPU tests of the small-call regimes (-m gpu, through the C-ABI).  Round 6: the LS estimate of a one-packet csi_estimate_device call inside
the layer-0 launch of the DNN (csrc/small_call.hip.h: small_l0_ls_kernel) - the step massiveMIMO_CSI_prediction_DNN.py:339-346 takes per packet,
with the LS estimate generate_maMIMO_LTF.m:336-349 derives from the same preamble."""
import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu


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
