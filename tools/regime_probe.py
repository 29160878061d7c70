#!/usr/bin/env python3
"""Mid-size calls (8 ... 500 packets, Nt=32 Nr=4, FC 1024x1024), device-resident, LS + DNN as one csi_estimate_device call: per-call
time of 20 queued calls for option settings given as name=value[,name=value] groups.  usage: regime_probe.py "a=1" "a=2,b=0" ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

nt, nr, hidden = int(os.environ.get('NT', '32')), int(os.environ.get('NR', '4')), (1024, 1024)      # NT / NR / DTYPE=bf16 from the environment
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype=os.environ.get('DTYPE', 'f32'))
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
sizes = [int(x) for x in os.environ.get('SIZES', '8,16,32,64,96,128,192,256,500').split(',')]
nmax = max(sizes)
d_re, d_im = eng.empty((nmax, nr, eng.len_ltf)), eng.empty((nmax, nr, eng.len_ltf))
eng.synth_white(1, 0, nmax, d_re, d_im)
o = [eng.empty((nmax, nr, nt, 234)) for _ in range(4)]
groups = sys.argv[1:] or ['small_call_overlap=1']
print('%8s' % 'packets', *['%28s' % g for g in groups])
for n in sizes:
    row = []
    for g in groups:
        for kv in g.split(','):
            k, v = kv.split('=')
            eng.set_option(k, int(v))
        for _ in range(5):
            eng.estimate_device(d_re, d_im, n, *o)
        eng.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                eng.estimate_device(d_re, d_im, n, *o)
            eng.synchronize()
            ts.append((time.perf_counter() - t0) / 20)
        row.append('%10.1f us %8.2f M pairs/s' % (np.median(ts) * 1e6, n * nr * nt / np.median(ts) / 1e6))
    print('%8d' % n, *row)
