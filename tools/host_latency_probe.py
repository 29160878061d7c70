import sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
import dl_channel_estimation_mamimo_amd as pkg
nt, nr, hidden = 32, 4, (1024, 1024)
rng = np.random.default_rng(0)
eng = pkg.CsiEngine(nt, nr, hidden=hidden)
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
for npkt in (1, 8, 64):
    re = rng.standard_normal((npkt, nr, 320*nt), dtype=np.float32); im = re.copy()
    out = (np.zeros((npkt, nr, nt, 234), np.float32), np.zeros((npkt, nr, nt, 234), np.float32))
    for _ in range(5): eng.predict(re, im, out=out)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); eng.predict(re, im, out=out); ts.append(time.perf_counter() - t0)
    print('host predict npkt', npkt, 'median %.1f us' % (np.median(ts) * 1e6), 'min %.1f' % (min(ts) * 1e6))
