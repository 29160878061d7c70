#!/usr/bin/env python3
"""band4_check.py - the register-blocked bf16 band kernel (band4_kernel_gen.py) against csi_band8_bf16 and the oracle's bf16 emulation,
then both timed at BASELINE configs[2].  usage: python tools/band4_check.py [nt nr npkt]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dl_channel_estimation_mamimo_amd as pkg
from oracle import csi_oracle as o


def rel_rows(a, b):
    return float(np.max(np.linalg.norm((a - b).reshape(-1, a.shape[-1]), axis=1) / np.maximum(np.linalg.norm(b.reshape(-1, b.shape[-1]), axis=1), 1e-30)))


def check(nt, nr, npkt, hidden=(1024, 1024)):
    rng = np.random.default_rng(nt + npkt)
    w_re = o.make_weights(rng, 320 * nt + nt, list(hidden), 234)
    w_im = o.make_weights(rng, 320 * nt + nt, list(hidden), 234)
    P = o.hadamard(nt) if nt & (nt - 1) == 0 else rng.choice([-1.0, 1.0], (nt, nt))
    ltf = o.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
    e.load_weights('real', w_re); e.load_weights('imag', w_im); e.set_pilot(P)
    e.set_option('force_tile', 256)
    e.set_option('band4', 0)
    a_re, a_im = e.predict(ltf)
    e.set_option('band4', 1)
    n0 = e.get_option('band_launches')
    b_re, b_im = e.predict(ltf)
    print('nt %d nr %d npkt %d hidden %s: band4_available %d launches %d' % (nt, nr, npkt, hidden, e.get_option('band4_available'), e.get_option('band_launches') - n0), flush=True)
    print('   band4 vs band8: rel %.3e / %.3e   finite %s   identical %s' % (rel_rows(b_re, a_re), rel_rows(b_im, a_im), np.isfinite(b_re).all(), np.array_equal(b_re, a_re)), flush=True)
    if npkt <= 64:
        r_re, r_im = o.predict_packets_bf16(ltf.astype(np.complex64), P, w_re, w_im)
        print('   vs bf16 emulation: band4 %.3e  band8 %.3e' % (rel_rows(b_re, r_re), rel_rows(a_re, r_re)), flush=True)
    c_re, _ = e.predict(ltf)
    print('   run-to-run identical', np.array_equal(c_re, b_re), flush=True)
    bad = np.argwhere(~np.isclose(b_re, a_re, rtol=2e-2, atol=2e-3))
    if len(bad):
        print('   mismatches', len(bad), 'first', bad[:8].tolist())
    e.close()


if __name__ == '__main__':
    if len(sys.argv) > 3:
        check(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        check(64, 4, 8)
        check(32, 4, 5, hidden=(256, 512))
        check(48, 2, 7, hidden=(512, 256))
        check(64, 4, 37)
