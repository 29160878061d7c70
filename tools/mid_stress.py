#!/usr/bin/env python3
"""Run-to-run stress of the mid-size routing (l0_hs_stream / l0_bf16_stream, csi_band8_cs / csi_band8_bf16_cs, two streams): every call of a
loop must reproduce the first call's outputs bit for bit.  usage: mid_stress.py [iterations]   (NT / NR / DTYPE from the environment)"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nt, nr, hidden = int(os.environ.get('NT', '32')), int(os.environ.get('NR', '4')), (1024, 1024)
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype=os.environ.get('DTYPE', 'f32'))
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
bad = 0
for n in (3, 8, 24, 64, 96, 144, 256):
    d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
    eng.synth_white(7, 0, n, d_re, d_im)
    o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
    ref = None
    t0 = time.time()
    for it in range(iters):
        # a burst of queued calls, then the check of the last one: the calls of a burst overlap on the two streams
        for _ in range(4):
            eng.estimate_device(d_re, d_im, n, *o)
        eng.synchronize()
        h = tuple(zlib.crc32(a.download().tobytes()) for a in o)
        if ref is None:
            ref = h
        elif h != ref:
            bad += 1
            print('MISMATCH packets %d iteration %d: %s != %s' % (n, it, h, ref))
    print('packets %4d: %d bursts of 4 calls, %d mismatches so far, %.1f s' % (n, iters, bad, time.time() - t0))
print('STRESS %s' % ('FAILED' if bad else 'ok'))
