#!/usr/bin/env python3
"""Throughput of the LMMSE smoother at the bench shape (Nt=32, Nr=4)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
import ctypes

nt, nr, npkt = 32, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = pkg.CsiEngine(nt, nr, hidden=(8,))
eng.set_pilot(pkg.synth.hadamard(nt))
d_re, d_im = eng.empty((npkt, nr, eng.len_ltf)), eng.empty((npkt, nr, eng.len_ltf))
eng.synth_white(3, 0, npkt, d_re, d_im)
h_re, h_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
o_re, o_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
rng = np.random.default_rng(0)
hv = eng.to_device(np.sort(np.abs(rng.standard_normal((npkt, 100)))).astype(np.float32))
snr = eng.to_device(rng.choice([-10.0, 0.0, 10.0], size=(npkt, nr)).astype(np.float32))
lib, ctx = eng._lib, eng._ctx
def run():
    eng._check(lib.csi_lmmse_estimate_device(ctx, h_re.ptr, h_im.ptr, npkt, hv.ptr, 100, snr.ptr, o_re.ptr, o_im.ptr))
run(); eng.synchronize()
eng.profile_enable(True); eng.profile_reset()
t0 = time.perf_counter()
for _ in range(3): run()
eng.synchronize()
dt = (time.perf_counter() - t0) / 3
p = eng.profile()['lmmse_levinson']
print('LMMSE npkt=%d: %.3f ms per call, %.1f us/packet, %.2f M links/s, %.1f TFLOP/s fp64 (Levinson flops)' % (
    npkt, dt * 1e3, dt / npkt * 1e6, npkt * nr * nt / dt / 1e6, p['flops'] / p['ms'] / 1e9))
