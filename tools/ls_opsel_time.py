#!/usr/bin/env python3
"""ls_opsel_time.py - LS kernel time per launch for the library at CSI_LIBRARY_PATH (tools/ls_opsel_hunt.sh builds one per form of the transform's packed
operations): Hadamard pilot (Walsh-Hadamard kernel) and a generic +-1 pilot (ring kernels), the bench's sizes.  usage: ls_opsel_time.py"""
import os, sys, time
os.environ['CSI_DEBUG_HOOKS'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
out = []
for (nt, nr, n) in ((32, 4, 4000), (64, 4, 5000), (128, 16, 800), (16, 4, 8000)):
    for kind in ('hadamard', 'pm1'):
        eng = pkg.CsiEngine(nt, nr, hidden=(64, 64))
        rng = np.random.default_rng(3)
        if kind == 'hadamard': P = pkg.synth.hadamard(nt)
        else:
            while True:
                P = rng.choice([-1.0, 1.0], size=(nt, nt))
                if np.linalg.cond(P) < 1e5: break
        eng.set_pilot(P)
        d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
        eng.synth_white(12, 0, n, d_re, d_im)
        h = [eng.empty((n, nr, nt, 234)) for _ in range(2)]
        best = 1e9
        for rep in range(4):
            eng.ls_estimate_device(d_re, d_im, n, *h); eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): eng.ls_estimate_device(d_re, d_im, n, *h)
            eng.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        print('Nt=%d Nr=%d %d packets %s: %.1f us' % (nt, nr, n, kind, best * 1e6), flush=True)
        del d_re, d_im, h, eng
