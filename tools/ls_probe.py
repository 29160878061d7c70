#!/usr/bin/env python3
"""LS kernel A/B on one MI355X: every Walsh-Hadamard / chunked variant at the three benchmark shapes, same box, same
inputs, HIP-event time per launch (engine profile) and the result checked against the first variant's output.

    python tools/ls_probe.py [--shapes 32x4x4000,64x4x5000,128x16x2000] [--reps 5] [--dbg 0,4,...]

Prints one line per (shape, kernel, variant, ls_debug mask): ms per launch, algorithmic TB/s (2560 B in + 1872 B
out per pair) and its fraction of 8 TB/s.  ls_debug masks skip phases (1 transforms, 2 despread,
4 stores, 8 first butterfly stage): timing experiments only, results are wrong by design."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_channel_estimation_mamimo_amd.engine import CsiEngine            # noqa: E402
from dl_channel_estimation_mamimo_amd.synth import hadamard              # noqa: E402


def _rows(d, first, cnt, nr, nt):
    """packets [first, first+cnt) of a [npkt, nr, nt, 234] result living at the start of the (larger) probe buffer"""
    import ctypes
    out = np.empty((cnt, nr, nt, 234), dtype=np.float32)
    e = d.engine
    e._check(e._lib.csi_memcpy_d2h(e._ctx, out.ctypes.data, d.ptr + first * nr * nt * 234 * 4, out.nbytes))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='32x4x4000,64x4x5000,128x16x2000')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--warm', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--variants', default='4:0,5:0,5:1,5:2,5:3,2:0')
    ap.add_argument('--dbg', default='0')
    ap.add_argument('--generic', action='store_true', help='non-Hadamard pilot matrix (only the generic kernels apply)')
    ap.add_argument('--pilot', default='qr', help="with --generic: 'qr' random orthogonal floats (3 bf16 pieces), 'q16' the same rounded "
                    "to 16 significand bits (2 pieces), 'perm' random +-1 entries (1 piece)")
    args = ap.parse_args()
    for shape in args.shapes.split(','):
        nt, nr, npkt = (int(v) for v in shape.split('x'))
        eng = CsiEngine(nt, nr, hidden=(32,), n_out=234)
        if not args.generic:
            P = hadamard(nt).astype(np.float32)
        else:
            rng = np.random.default_rng(1)
            P = np.linalg.qr(rng.standard_normal((nt, nt)))[0].astype(np.float32) * np.float32(np.sqrt(nt))
            if args.pilot == 'q16':
                P = (P.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)
            elif args.pilot == 'perm':
                P = rng.choice([-1.0, 1.0], (nt, nt)).astype(np.float32)
        eng.set_pilot(P)
        d_re, d_im = eng.empty((npkt, nr, 320 * nt)), eng.empty((npkt, nr, 320 * nt))
        eng.synth_white(7, 0, npkt, d_re, d_im)
        d_hr, d_hi = eng.empty((npkt, nr, nt, 256)), eng.empty((npkt, nr, nt, 256))       # 256: room for the pitch experiment (ls_debug 16)
        pairs = npkt * nr * nt
        ref = None
        cases = []
        for kv in args.variants.split(','):
            k, v = (int(x) for x in kv.split(':'))
            for dbg in (int(x) for x in args.dbg.split(',')):
                cases.append((k, v, dbg))
        times = {c: [] for c in cases}
        errs = {}
        for rnd in range(args.rounds):                     # variants interleaved: box drift hits all of them alike
            for (k, v, dbg) in cases:
                eng.set_option('ls_v2', v)
                eng.set_option('ls_kernel', k)
                eng.set_option('ls_debug', dbg)
                for _ in range(args.warm if rnd == 0 else 2):
                    eng.ls_estimate_device(d_re, d_im, npkt, d_hr, d_hi)
                eng.synchronize()
                eng.profile_enable(True)
                eng.profile_reset()
                for _ in range(args.reps):
                    eng.ls_estimate_device(d_re, d_im, npkt, d_hr, d_hi)
                eng.synchronize()
                pr = eng.profile()['ls_estimate']
                eng.profile_enable(False)
                times[(k, v, dbg)].append(pr['ms'] / pr['launches'])
                if dbg == 0 and rnd == 0:
                    n = min(npkt, 64)
                    h = _rows(d_hr, 0, n, nr, nt) + 1j * _rows(d_hi, 0, n, nr, nt)
                    tail = _rows(d_hr, npkt - 1, 1, nr, nt)
                    if ref is None:
                        ref = (h, tail)
                        errs[(k, v, dbg)] = 'reference'
                    else:
                        errs[(k, v, dbg)] = 'max rel diff %.2e' % max(float(np.max(np.abs(h - ref[0])) / np.max(np.abs(ref[0]))),
                                                                       float(np.max(np.abs(tail - ref[1])) / np.max(np.abs(ref[1]))))
        for (k, v, dbg) in cases:
            ts = sorted(times[(k, v, dbg)])
            ms = ts[len(ts) // 2]
            tbs = pairs * (2560 + 1872) / (ms * 1e-3) / 1e12
            print('Nt=%3d Nr=%2d pkts=%5d  kernel %d v%d dbg %2d : median %7.3f ms (min %7.3f)  %5.2f TB/s  %.3f of 8 TB/s   %s'
                  % (nt, nr, npkt, k, v, dbg, ms, ts[0], tbs, tbs / 8, errs.get((k, v, dbg), '')), flush=True)
        eng.close()


if __name__ == '__main__':
    main()
