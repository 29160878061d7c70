import sys, time, numpy as np
sys.path.insert(0, '.')
import dl_channel_estimation_mamimo_amd as pkg
rng = np.random.default_rng(0)
nt, nr, npkt = 32, 4, 4000
e = pkg.CsiEngine(nt, nr, hidden=(1024, 1024))
w = pkg.synth.make_weights(rng, nt, (1024, 1024))
e.load_weights('real', w); e.load_weights('imag', w); e.set_pilot(pkg.synth.hadamard(nt))
d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
e.synth_white(1, 0, npkt, d_re, d_im)
outs = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
for rep in range(3):
    for mode in (1, 2):
        e.set_option('small_call_overlap', mode)
        for _ in range(2): e.estimate_device(d_re, d_im, npkt, *outs)
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): e.estimate_device(d_re, d_im, npkt, *outs)
        e.synchronize()
        print('overlap' if mode == 2 else 'serial ', (time.perf_counter() - t0) / 10 * 1e3, 'ms')
