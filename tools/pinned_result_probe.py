#!/usr/bin/env python3
"""The deployment surface (inference.py:24-32: complex128 batch in, complex64 estimates out; csi_estimate_c128) with pageable result
arrays (host threads weave out of the staging buffer) against PINNED result arrays (complex values assembled on the device, downloads
land in the caller's arrays: weave_c64_kernel, csi_hostpipe.hpp) - same input, alternating, in one process.  Seconds, not minutes:
    python tools/pinned_result_probe.py [--packets 4000] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--packets', type=int, default=4000)
    ap.add_argument('--nt', type=int, default=32)
    ap.add_argument('--nr', type=int, default=4)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--ls', action='store_true', help='DNN + LS (default: DNN only, what CSIPredictor.inference returns)')
    a = ap.parse_args()
    nt, nr, npkt = a.nt, a.nr, a.packets
    t00 = time.time()
    rng = np.random.default_rng(0)
    e = pkg.CsiEngine(nt, nr, hidden=(1024, 1024))
    w = pkg.synth.make_weights(rng, nt, (1024, 1024))
    e.load_weights('real', w)
    e.load_weights('imag', w)
    e.set_pilot(pkg.synth.hadamard(nt))
    x = np.empty((npkt, nr, 320 * nt), np.complex128)
    x.real = rng.standard_normal(x.shape[1:])
    x.imag = x.real[:, ::-1]
    shape = (npkt, nr, nt, 234)
    mk = {'pageable': lambda: np.zeros(shape, np.complex64), 'pinned  ': lambda: e.pinned_empty(shape, np.complex64)}
    outs = {k: (f(), f() if a.ls else None) for k, f in mk.items()}
    for k in outs:
        for o in outs[k]:
            if o is not None:
                o[...] = 0
    print('setup %.1f s' % (time.time() - t00), flush=True)
    ts = {k: [] for k in outs}
    where = {}
    for rep in range(a.reps + 1):
        for k, o in outs.items():
            t0 = time.perf_counter()
            e.estimate(x, ls=a.ls, out=o)
            if rep:
                ts[k].append((time.perf_counter() - t0) * 1e3)
                where[k] = 'last call: total %.1f ms, stager busy %.1f, caller waits: staged chunk %.1f, download %.1f, weave threads %.1f' % tuple(
                    e.get_option(n) / 1e3 for n in ('hp_total_us', 'hp_stage_us', 'hp_wait_stage_us', 'hp_wait_out_us', 'hp_weave_us'))
    same = all(np.array_equal(outs['pageable'][i], outs['pinned  '][i]) for i in range(2) if outs['pageable'][i] is not None)
    for k, v in ts.items():
        print('%s result arrays: %s ms  -> %.2f M pairs/s (best) | %s' % (k, ' '.join('%.2f' % t for t in v), npkt * nr * nt / min(v) / 1e3, where[k]))
    print('bit-identical: %s; hp_direct_out_calls %d; %s, Nt=%d Nr=%d %d packets' % (same, e.get_option('hp_direct_out_calls'), 'DNN + LS' if a.ls else 'DNN only', nt, nr, npkt))
    up, down = x.nbytes // 2, sum(o.nbytes for o in outs['pinned  '] if o is not None)
    a_ms, b_ms, ab_ms = e.pcie_probe(up, down)
    print('link, bare pinned copies of the same bytes: %.2f GB up %.2f ms, %.2f GB down %.2f ms, both at once %.2f ms' % (up / 1e9, a_ms, down / 1e9, b_ms, ab_ms))


if __name__ == '__main__':
    main()
