#!/usr/bin/env python3
"""Device-resident DNN time per call over the packet count, fp32 MFMA kernels vs split-f16 engine
(chooses the crossover of the automatic mode, csi_dnn_hs.hpp HS_MIN_BLOCKS).
GPU box: python tools/engine_sweep.py [nt nr]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    hidden = (1024, 1024)
    rng = np.random.default_rng(1)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', pkg.synth.make_weights(rng, nt, hidden))
    e.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
    e.set_pilot(pkg.synth.hadamard(nt))
    nmax = 2048
    d_re, d_im = e.empty((nmax, nr, e.len_ltf)), e.empty((nmax, nr, e.len_ltf))
    e.synth_white(7, 0, nmax, d_re, d_im)
    o_re, o_im = e.empty((nmax, nr, nt, 234)), e.empty((nmax, nr, nt, 234))
    thr = [int(a) for a in sys.argv[3:]] or [128]
    print('packets  rows     fp32-MFMA ms   split ms   auto ms at hs_min_blocks =', thr)
    for npkt in (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 128, 192, 256, 512, 2048):
        res = []
        for engine, mb in [(0, 128), (1, 128)] + [(-1, t) for t in thr]:
            e.set_option('f32_engine', engine)
            e.set_option('hs_min_blocks', mb)
            ts = []
            for i in range(12):
                e.synchronize()
                t0 = time.perf_counter()
                e.predict_device(d_re, d_im, npkt, o_re, o_im)
                e.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append(float(np.median(ts[4:])))
        pairs = npkt * nr * nt
        print('%7d %6d   ' % (npkt, pairs) + '  '.join('%8.3f' % (r * 1e3) for r in res))


if __name__ == '__main__':
    main()
