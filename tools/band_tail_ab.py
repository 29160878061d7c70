#!/usr/bin/env python3
"""band_tail_ab.py - "band_tail_split" 1 against 0 on one-stream calls whose last round of band workgroups is nearly empty: per-call time (DNN only, both models,
10 queued) and the outputs of the two forms against each other.  usage: band_tail_ab.py f32|bf16 [packets ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
nt, nr = (64, 4) if dtype == 'bf16' else (32, 4)
sizes = [int(x) for x in sys.argv[2:]] or ([5000, 2570, 1290, 2600, 3000] if dtype == 'bf16' else [2100, 2200, 2600, 3100, 4000, 4150])
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=(1024, 1024), dtype=dtype)
eng.load_weights('real', pkg.synth.make_weights(rng, nt, (1024, 1024))); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, (1024, 1024)))
eng.set_pilot(pkg.synth.hadamard(nt))
nmax = max(sizes)
d_re, d_im = eng.empty((nmax, nr, eng.len_ltf)), eng.empty((nmax, nr, eng.len_ltf))
eng.synth_white(1, 0, nmax, d_re, d_im)
o = [eng.empty((nmax, nr, nt, 234)) for _ in range(2)]
print('%s Nt=%d Nr=%d' % (dtype, nt, nr))
for n in sizes:
    t, outs, took = {}, {}, 0
    for rep in range(2):
        for v in (0, 1):
            eng.set_option('band_tail_split', v)
            n0 = eng.get_option('band_tail_launches')
            for _ in range(2): eng.predict_device(d_re, d_im, n, *o)
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): eng.predict_device(d_re, d_im, n, *o)
            eng.synchronize()
            t[v] = min(t.get(v, 1e9), (time.perf_counter() - t0) / 8)
            if v: took = (eng.get_option('band_tail_launches') - n0) // 10
            outs[v] = o[0].download(0, n)
    a, b = outs[0].reshape(-1, 234).astype(np.float64), outs[1].reshape(-1, 234).astype(np.float64)
    rel = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1e-30)
    bands = n * nr * nt // 128
    print('%6d packets = %6d bands = %.2f rounds: one launch %9.1f us | tail split %9.1f us (%d tail launches per call) ratio %.3f | rows that differ %d, max row-rel %.2e, finite %s' % (
        n, bands, bands / 256, t[0] * 1e6, t[1] * 1e6, took, t[1] / t[0], int((rel > 0).sum()), rel.max(), bool(np.isfinite(b).all())), flush=True)
