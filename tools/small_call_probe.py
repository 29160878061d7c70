#!/usr/bin/env python3
"""One-packet calls in a loop (run under `rocprofv3 --kernel-trace --output-format csv`): what the kernels of a call take on the
device and what lies between them.  usage: small_call_probe.py [n_calls] [small_fused 0|1] [estimate|separate]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else 'estimate'
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nt, nr, hidden = 32, 4, (1024, 1024)
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden)
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
eng.set_option('small_fused', fused)
eng.set_option('use_graph', graph)
d_re, d_im = eng.empty((1, nr, eng.len_ltf)), eng.empty((1, nr, eng.len_ltf))
eng.synth_white(1, 0, 1, d_re, d_im)
o = [eng.empty((1, nr, nt, 234)) for _ in range(4)]


def call():
    if mode == 'estimate':
        eng.estimate_device(d_re, d_im, 1, *o)
    else:
        eng.ls_estimate_device(d_re, d_im, 1, o[2], o[3])
        eng.predict_device(d_re, d_im, 1, o[0], o[1])


for _ in range(10):
    call()
eng.synchronize()
lat = []
for _ in range(n_calls):
    t0 = time.perf_counter()
    call()
    eng.synchronize()
    lat.append(time.perf_counter() - t0)
t0 = time.perf_counter()
for _ in range(n_calls):
    call()
eng.synchronize()
pip = (time.perf_counter() - t0) / n_calls
print('use_graph=%d ' % graph, end='')
print('small_fused=%d mode=%s: latency median %.1f us (min %.1f), pipelined %.1f us per call' % (fused, mode, np.median(lat) * 1e6, min(lat) * 1e6, pip * 1e6))
