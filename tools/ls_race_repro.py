#!/usr/bin/env python3
"""Repro loop for the rare bad FIRST run of the bf16-split LS kernel at Nt = 24 (tests/stress_ls_generic.py): the stress's own
sequence restricted to a few shapes, many times in one process, with knobs to bisect.
    python tools/ls_race_repro.py --shapes 16x4x2000,24x4x1500 --loops 30 [--side 0] [--device]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg            # noqa: E402
import stress_ls_generic as st                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='16x4x2000,24x4x1500')
    ap.add_argument('--loops', type=int, default=30)
    ap.add_argument('--side', type=int, default=1)
    ap.add_argument('--device', action='store_true', help='device-resident calls instead of the host-buffer pipeline')
    ap.add_argument('--kinds', default='pm1,q16,qr')
    ap.add_argument('--dbg', type=int, default=0, help='ls_debug for the kernel-7 calls: 128 = one workgroup per CU, 256 = LDS pre-filled with NaN')
    ap.add_argument('--first', type=int, default=6, help='kernel run before kernel 7 (6 = the stress; 0 = none)')
    ap.add_argument('--warm', type=int, default=0, help='packets of a small kernel-7 launch in front of the three full ones (device mode): is it the cold start of the code?')
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    events, firsts = 0, 0
    for loop in range(a.loops):
        for shape in a.shapes.split(','):
            nt, nr, npkt = (int(v) for v in shape.split('x'))
            ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
            for kind in a.kinds.split(','):
                e = pkg.CsiEngine(nt, nr, hidden=(8,))
                e.set_option('hp_side_threads', a.side)
                e.set_pilot(st.pilot(rng, nt, kind))
                if a.device:
                    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real, np.float32)), e.to_device(np.ascontiguousarray(ltf.imag, np.float32))
                    o_re, o_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))

                    def call():
                        e.ls_estimate_device(d_re, d_im, npkt, o_re, o_im)
                        e.synchronize()
                        return o_re.download() + 1j * o_im.download()
                else:
                    call = lambda: e.ls_estimate(ltf)
                e.set_option('ls_kernel', 6)
                h6 = call() if a.first else None
                e.set_option('ls_kernel', 7)
                e.set_option('ls_v2', 0)
                e.set_option('ls_debug', a.dbg)
                if a.warm and a.device:
                    e.ls_estimate_device(d_re, d_im, a.warm, o_re, o_im)
                    e.synchronize()
                hs = [call() for _ in range(3)]
                e.set_option('ls_debug', 0)
                if h6 is None:
                    e.set_option('ls_kernel', 6)
                    h6 = call()
                firsts += 1
                for k, h in enumerate(hs):
                    n_items = npkt * nr
                    d = np.abs(h - h6).reshape(n_items, -1).max(1) / np.abs(h6).reshape(n_items, -1).max(1)
                    if not (d <= 2e-6).all():
                        events += 1
                        st.describe('loop %d Nt=%d %s call %d' % (loop, nt, kind, k), h, h6, limit=4)
                e.close()
    print('events: %d in %d engine cycles (shapes %s, side %d, device %s, first %d, dbg %d, warm %d)' % (events, firsts, a.shapes, a.side, a.device, a.first, a.dbg, a.warm))


if __name__ == '__main__':
    main()
