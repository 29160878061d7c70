#!/usr/bin/env python3
"""ls_opsel_count.py - how many bf16 two-stream calls (second stream forked in FRONT of the LS kernel, the default again) come back with
LS planes that differ from the one-stream call, and which transform outputs the wrong bins trace back to.  The library under test comes from
CSI_LIBRARY_PATH (tools/ls_opsel_hunt.sh).  usage: ls_opsel_count.py [calls]"""
import os, sys, time
os.environ['CSI_DEBUG_HOOKS'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
from oracle import csi_oracle as orc          # the checker's bin map, to name FFT positions
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nt, nr, hidden = 64, 4, (1024, 1024)
NAT = (np.asarray(orc.data_carrier_indices()) - 1 + 128) % 256        # data bin -> natural-order FFT position
rng = np.random.default_rng(1)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
for n in (500, 1000):
    d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
    eng.synth_white(11, 0, n, d_re, d_im)
    o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
    eng.set_option('small_call_overlap', 0)
    eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
    ref = [a.download() for a in o]
    eng.set_option('small_call_overlap', 1)
    wrong, items, census = 0, 0, {}
    t0 = time.time()
    for it in range(calls):
        eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
        g2 = o[2].download()
        d = g2 != ref[2]
        if not d.any(): continue
        wrong += 1
        for (p, r) in sorted({(i[0], i[1]) for i in np.argwhere(d.any(axis=(2, 3))).tolist()}):
            items += 1
            pos = np.unique(NAT[np.flatnonzero(d[p, r].any(axis=0))])
            for mod in (256, 64, 16):
                res = np.unique(pos % mod)
                if res.max() - res.min() < 16 and (mod == 256 or len(res) <= 16): break
            key = 'positions %d..%d mod %d' % (res.min(), res.max(), mod)
            census[key] = census.get(key, 0) + 1
    print('packets %4d: %d of %d calls wrong, %d items; %s  (%.1f s)' % (n, wrong, calls, items, sorted(census.items(), key=lambda kv: -kv[1])[:6], time.time() - t0), flush=True)

# the LS kernel's time in this form: 4000-packet launches at Nt = 64 (this context), HIP-side queue of 20
n = 4000
d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
eng.synth_white(12, 0, n, d_re, d_im)
h = [eng.empty((n, nr, nt, 234)) for _ in range(2)]
for rep in range(3):
    eng.ls_estimate_device(d_re, d_im, n, *h); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): eng.ls_estimate_device(d_re, d_im, n, *h)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 20
print('LS kernel, Nt = 64, 4000 packets: %.1f us per launch (20 queued)' % (dt * 1e6))
