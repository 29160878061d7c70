#!/usr/bin/env python3
"""Eager launches vs one replayed hipGraph for the whole LS + DNN step (csi_estimate_device, "use_graph"), at BASELINE
config 5's shape (Nt=128, Nr=16, multi-chunk) and at config 2's, plus the small-call regime where launch overhead is
visible.  GPU box:  python tools/graph_probe.py  > profiles/rNN_graph_probe.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402


def run(nt, nr, npkt, reps):
    rng = np.random.default_rng(0)
    e = pkg.CsiEngine(nt, nr, hidden=(1024, 1024))
    w = pkg.synth.make_weights(rng, nt, (1024, 1024))
    e.load_weights('real', w)
    e.load_weights('imag', w)
    e.set_pilot(pkg.synth.hadamard(nt))
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(1, 0, npkt, d_re, d_im)
    outs = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    res = {}
    for mode in ('eager', 'graph'):
        e.set_option('use_graph', int(mode == 'graph'))
        for _ in range(3):
            e.estimate_device(d_re, d_im, npkt, *outs)
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.estimate_device(d_re, d_im, npkt, *outs)
        e.synchronize()
        res[mode] = (time.perf_counter() - t0) / reps * 1e3
    e.close()
    print(f'Nt={nt:3d} Nr={nr:2d} {npkt:5d} packets/step: eager {res["eager"]:9.3f} ms  graph {res["graph"]:9.3f} ms  '
          f'({npkt * nr * nt / res["graph"] / 1e3:7.2f} M pairs/s replayed, graph/eager = {res["graph"] / res["eager"]:.3f})')


if __name__ == '__main__':
    for nt, nr, npkt, reps in ((128, 16, 520, 5), (128, 16, 256, 8), (32, 4, 4000, 10), (32, 4, 64, 50), (32, 4, 8, 200), (32, 4, 1, 300)):
        run(nt, nr, npkt, reps)
