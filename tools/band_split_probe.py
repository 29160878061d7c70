#!/usr/bin/env python3
"""Column-split band kernel ("band_split"): mid-size calls (Nt=32 Nr=4, FC 1024x1024), device-resident csi_estimate_device -
per-call time of 20 queued calls with band_split 0 / automatic / 2 / 4, the largest difference of each setting's outputs from
band_split = 0 relative to the output norm, and the split launches taken.  usage: band_split_probe.py   (SIZES=8,16,... to override)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

nt, nr, hidden = 32, 4, (1024, 1024)
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden)
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
sizes = [int(x) for x in os.environ.get('SIZES', '8,12,16,24,32,48,64,96,128,256').split(',')]
nmax = max(sizes)
d_re, d_im = eng.empty((nmax, nr, eng.len_ltf)), eng.empty((nmax, nr, eng.len_ltf))
eng.synth_white(1, 0, nmax, d_re, d_im)
o = [eng.empty((nmax, nr, nt, 234)) for _ in range(4)]
settings = [0, -1, 2, 4]
if os.environ.get('ENGINE'):                       # ENGINE=1: the split-f16 engine at every size (the automatic mode starts it at 24 packets)
    eng.set_option('f32_engine', int(os.environ['ENGINE']))
print('%8s' % 'packets', *['%34s' % ('band_split=%d' % s) for s in settings])
for n in sizes:
    row, ref = [], None
    for sp in settings:
        eng.set_option('band_split', sp)
        before = eng.get_option('band_split_launches')
        for _ in range(5):
            eng.estimate_device(d_re, d_im, n, *o)
        eng.synchronize()
        taken = eng.get_option('band_split_launches') - before
        out = np.concatenate([o[0].download(0, n).ravel(), o[1].download(0, n).ravel()])
        if ref is None:
            ref = out
        err = float(np.linalg.norm(out - ref) / max(np.linalg.norm(ref), 1e-30))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                eng.estimate_device(d_re, d_im, n, *o)
            eng.synchronize()
            ts.append((time.perf_counter() - t0) / 20)
        row.append('%8.1f us %6.2f M/s d=%.1e %s' % (np.median(ts) * 1e6, n * nr * nt / np.median(ts) / 1e6, err, 'cs' if taken else '--'))
    print('%8d' % n, *['%34s' % r for r in row])
