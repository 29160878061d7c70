#!/usr/bin/env python3
"""band4_hs_check.py - the register-blocked split-f16 band kernel (csi_band4) against csi_band8 and the fp64 oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dl_channel_estimation_mamimo_amd as pkg
from oracle import csi_oracle as o


def rel_rows(a, b):
    return float(np.max(np.linalg.norm((a - b).reshape(-1, a.shape[-1]), axis=1) / np.maximum(np.linalg.norm(b.reshape(-1, b.shape[-1]), axis=1), 1e-30)))


def check(nt, nr, npkt, hidden=(1024, 1024), n_out=234):
    rng = np.random.default_rng(nt + npkt)
    w_re = o.make_weights(rng, 320 * nt + nt, list(hidden), n_out)
    w_im = o.make_weights(rng, 320 * nt + nt, list(hidden), n_out)
    P = o.hadamard(nt) if nt & (nt - 1) == 0 else rng.choice([-1.0, 1.0], (nt, nt))
    ltf = o.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=n_out)
    e.load_weights('real', w_re); e.load_weights('imag', w_im); e.set_pilot(P)
    e.set_option('f32_engine', 1); e.set_option('band_split', 0); e.set_option('small_fused', 0)
    e.set_option('band4', 0)
    a_re, a_im = e.predict(ltf)
    e.set_option('band4', 1)
    n0, h0 = e.get_option('band_launches'), e.get_option('hs_range_fallbacks')
    b_re, b_im = e.predict(ltf)
    print('nt %d nr %d npkt %d hidden %s n_out %d: band4_available %d band launches %d fallbacks %d' % (nt, nr, npkt, hidden, n_out, e.get_option('band4_available'),
          e.get_option('band_launches') - n0, e.get_option('hs_range_fallbacks') - h0), flush=True)
    print('   band4 vs band8: rel %.3e / %.3e finite %s identical %s' % (rel_rows(b_re, a_re), rel_rows(b_im, a_im), np.isfinite(b_re).all(), np.array_equal(b_re, a_re)), flush=True)
    r_re, r_im = o.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    print('   vs fp64 oracle: band4 %.3e / %.3e   band8 %.3e' % (rel_rows(b_re, r_re), rel_rows(b_im, r_im), rel_rows(a_re, r_re)), flush=True)
    c_re, _ = e.predict(ltf)
    print('   run-to-run identical', np.array_equal(c_re, b_re), flush=True)
    e.close()


if __name__ == '__main__':
    check(32, 4, 8)
    check(16, 2, 9, hidden=(128, 256))
    check(48, 2, 7, hidden=(512, 256), n_out=52)
    check(128, 2, 3, hidden=(256, 512))
    check(32, 4, 37)
