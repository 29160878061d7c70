#!/usr/bin/env python3
"""band4_stress.py - run-to-run stress of the register-blocked band kernels at full size and of the one-packet call with the LS estimate inside the
layer-0 launch: every iteration must reproduce the first one bit for bit (checksums of the downloaded planes).
usage: band4_stress.py [iterations]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for dtype, nt, nr, sizes in (('f32', 32, 4, (1, 2, 700, 4000)), ('bf16', 64, 4, (600, 2500))):
    rng = np.random.default_rng(1)
    hidden = (1024, 1024)
    eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype=dtype)
    eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
    eng.set_pilot(pkg.synth.hadamard(nt))
    for n in sizes:
        d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
        eng.synth_white(11, 0, n, d_re, d_im)
        o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
        ref, t0 = None, time.time()
        n4, nl = eng.get_option('band_launches'), eng.get_option('small_ls_launches')
        reps = iters * (20 if n <= 2 else 1)
        for it in range(reps):
            for _ in range(3):
                eng.estimate_device(d_re, d_im, n, *o)
            eng.synchronize()
            h = tuple(zlib.crc32(a.download().tobytes()) for a in o)
            if ref is None:
                ref = h
            elif h != ref:
                bad += 1
                print('MISMATCH %s packets %d iteration %d' % (dtype, n, it))
        print('%s Nt=%d packets %5d: %d x 3 calls, band launches %d (band4 %d), LS-inside launches %d, mismatches so far %d, %.1f s' % (
            dtype, nt, n, reps, eng.get_option('band_launches') - n4, eng.get_option('band4'), eng.get_option('small_ls_launches') - nl, bad, time.time() - t0), flush=True)
        del d_re, d_im, o
    eng.close()
print('STRESS %s' % ('FAILED' if bad else 'ok'))
