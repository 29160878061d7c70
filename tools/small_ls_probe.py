#!/usr/bin/env python3
"""small_ls_probe.py - one-packet csi_estimate_device with the LS estimate inside the layer-0 launch ("small_ls_fused" = 1) against the
four-launch form (0): same bits, latency (call + csi_synchronize) and time per call of a queued loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import dl_channel_estimation_mamimo_amd as pkg


def run(nt, nr, npkts=(1, 2), hidden=(1024, 1024)):
    rng = np.random.default_rng(5)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); e.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
    e.set_pilot(pkg.synth.hadamard(nt))
    n = max(npkts)
    d_re, d_im = e.empty((n, nr, e.len_ltf)), e.empty((n, nr, e.len_ltf))
    e.synth_white(3, 0, n, d_re, d_im)
    for k in npkts:
        o = [e.empty((k, nr, nt, 234)) for _ in range(4)]
        res = {}
        for mode in (0, 1, 0, 1):
            e.set_option('small_ls_fused', mode)
            n0 = e.get_option('small_ls_launches')
            for _ in range(5):
                e.estimate_device(d_re, d_im, k, *o)
            e.synchronize()
            lat = []
            for _ in range(40):
                t0 = time.perf_counter(); e.estimate_device(d_re, d_im, k, *o); e.synchronize(); lat.append(time.perf_counter() - t0)
            q = []
            for _ in range(6):
                t0 = time.perf_counter()
                for _ in range(20):
                    e.estimate_device(d_re, d_im, k, *o)
                e.synchronize(); q.append((time.perf_counter() - t0) / 20)
            outs = [x.download(0, k) for x in o]
            res.setdefault(mode, []).append((np.median(lat) * 1e6, np.median(q) * 1e6, outs, e.get_option('small_ls_launches') - n0))
        a, b = res[0][-1], res[1][-1]
        same = all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
        print('nt %d nr %d packets %d: 4 launches latency %.1f / queued %.1f us | LS inside layer 0: %.1f / %.1f us (fused launches %d) | bit-identical %s, finite %s' % (
            nt, nr, k, a[0], a[1], b[0], b[1], b[3], same, all(np.isfinite(x).all() for x in b[2])), flush=True)
    e.close()


if __name__ == '__main__':
    run(32, 4, npkts=(1, 2, 3, 8, 24, 64))
    run(64, 4, npkts=(1, 2, 8, 33))
    run(16, 2, npkts=(1, 3, 20), hidden=(256, 256))
