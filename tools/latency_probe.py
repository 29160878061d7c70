#!/usr/bin/env python3
"""Latency of the device-resident path for small packet counts (the reference's literal
per-packet call is npkt = 1)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

nt, nr, hidden = 32, 4, (1024, 1024)
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden)
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
import itertools
for graph, npkt in itertools.product((0, 1), (1, 2, 8, 32, 128, 512)):
    eng.set_option('use_graph', graph)
    d_re, d_im = eng.empty((npkt, nr, eng.len_ltf)), eng.empty((npkt, nr, eng.len_ltf))
    eng.synth_white(1, 0, npkt, d_re, d_im)
    o_re, o_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    h_re, h_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    for _ in range(5):
        eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im); eng.predict_device(d_re, d_im, npkt, o_re, o_im)
    eng.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
        eng.predict_device(d_re, d_im, npkt, o_re, o_im)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    eng.profile_enable(True); eng.profile_reset()
    eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im); eng.predict_device(d_re, d_im, npkt, o_re, o_im); eng.synchronize()
    prof = {k: round(v['ms'] * 1e3, 1) for k, v in eng.profile().items() if v['launches']}
    eng.profile_enable(False)
    print('graph=%d npkt=%4d  median %.1f us  min %.1f us  -> %.0f pairs/s   kernels(us)=%s' % (
        graph, npkt, np.median(ts) * 1e6, min(ts) * 1e6, npkt * nr * nt / np.median(ts), prof))
