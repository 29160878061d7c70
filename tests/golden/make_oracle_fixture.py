#!/usr/bin/env python3
"""Writes tests/golden/oracle_nt4.npz: a small, fully seeded problem (Nt=4, Nr=2, hidden 64x64,
3 structured packets at 5 dB) with the fp64 oracle's outputs - LS estimate, DNN real / imag outputs,
recombined CSI - so that (1) a later edit of oracle/ that changes its numbers is caught and (2) the
HIP path has a committed target that does not depend on running the oracle.  SURVEY.md 8c.
    python tests/golden/make_oracle_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import csi_oracle as o   # noqa: E402


def main():
    rng = np.random.default_rng(20240928)
    nt, nr, npkt, hidden = 4, 2, 3, (64, 64)
    P = o.hadamard(nt)
    w_re = o.make_weights(rng, 321 * nt, hidden, 234)
    w_im = o.make_weights(rng, 321 * nt, hidden, 234)
    ltf, H = o.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)
    ltf = ltf.astype(np.complex64)
    ls = o.ls_estimate(ltf, P)
    out_re, out_im = o.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    out = {'nt': nt, 'nr': nr, 'P': P, 'ltf': ltf, 'H_true': H, 'ls': ls, 'dnn_real': out_re, 'dnn_imag': out_im,
           'csi': o.recombine(out_re, out_im)}
    for tag, w in (('re', w_re), ('im', w_im)):
        for k, v in w.items():
            if isinstance(v, np.ndarray):
                out[f'w_{tag}.{k}'] = v
    path = os.path.join(HERE, 'oracle_nt4.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
