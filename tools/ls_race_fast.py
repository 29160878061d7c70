#!/usr/bin/env python3
"""Fast form of tools/ls_race_repro.py for the rare bad FIRST launch of ls_estimate_ringb_kernel<1, 4, 1, NPP, 2> (two workgroups
per CU): the events only ever hit the first item of a CU's second workgroup, so a cycle needs no more than two items per workgroup -
256 packets x 4 rx = 1024 items instead of 8000 - and runs ~20 x faster.  Variants (ls_debug bits, csi_mamimo.hip ls_ringb_shape):
    0x200  s_waitcnt lgkmcnt(0) behind every transform stage's LDS writes
    0x400  op_sel operations never in place (early-clobber destinations)
    0x800  idle cycles between a stage's last VALU operation and its first ds_write
    0x1000 no LDS-DMA of the wave in flight across the "spectra complete" barrier
    0x2000 sources of the op_sel adds live until the stage's writes are out,  0x4000 two idle cycles behind every op_sel operation,
    0x8000 no op_sel / packed operation at all; sums combine (0x8800 = 0x800 + 0x8000: the instantiations csi_mamimo.hip lists)
    128    one workgroup per CU,  256  LDS pre-filled with NaN
--variants a,b,c interleaves them cycle by cycle (same box, same minutes).
    python tools/ls_race_fast.py --loops 2000 --variants 0,0x200,0x400"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg            # noqa: E402
import stress_ls_generic as st                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nt', type=int, default=16)
    ap.add_argument('--nr', type=int, default=4)
    ap.add_argument('--npkt', type=int, default=256)
    ap.add_argument('--loops', type=int, default=1000)
    ap.add_argument('--kinds', default='pm1,q16')
    ap.add_argument('--variants', default='0')
    ap.add_argument('--seconds', type=float, default=0, help='stop after this many seconds (0 = run all loops)')
    ap.add_argument('--nok6', action='store_true', help='with --reuse: the fp32 ring kernel runs ONCE per engine (reference), afterwards only the kernel under test is launched')
    ap.add_argument('--reuse', action='store_true', help='ONE engine per variant and pilot kind for the whole run (is the fresh context needed?)')
    a = ap.parse_args()
    variants = [int(v, 0) for v in a.variants.split(',')]
    kinds = a.kinds.split(',')
    rng = np.random.default_rng(5)
    nt, nr, npkt = a.nt, a.nr, a.npkt
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    l_re, l_im = np.ascontiguousarray(ltf.real, np.float32), np.ascontiguousarray(ltf.imag, np.float32)
    pilots = {k: st.pilot(rng, nt, k) for k in kinds}
    events = {v: 0 for v in variants}
    cycles = {v: 0 for v in variants}
    where = {v: [] for v in variants}
    kept, refs = {}, {}
    t0 = time.time()

    def engine(kind):
        e = pkg.CsiEngine(nt, nr, hidden=(8,))
        e.set_pilot(pilots[kind])
        d_re, d_im = e.to_device(l_re), e.to_device(l_im)
        o_re, o_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
        return e, d_re, d_im, o_re, o_im

    for loop in range(a.loops):
        if a.seconds and time.time() - t0 > a.seconds:
            break
        for kind in kinds:
            for v in variants:
                if a.reuse:
                    if (kind, v) not in kept:
                        kept[(kind, v)] = engine(kind)
                    e, d_re, d_im, o_re, o_im = kept[(kind, v)]
                else:
                    e, d_re, d_im, o_re, o_im = engine(kind)

                def call():
                    e.ls_estimate_device(d_re, d_im, npkt, o_re, o_im)
                    e.synchronize()
                    return o_re.download() + 1j * o_im.download()
                if a.nok6 and a.reuse and (kind, v) in refs:
                    h6 = refs[(kind, v)]
                else:
                    e.set_option('ls_debug', 0)
                    e.set_option('ls_kernel', 6)
                    h6 = refs[(kind, v)] = call()
                e.set_option('ls_kernel', 7)
                e.set_option('ls_v2', 0)
                e.set_option('ls_debug', v)
                hs = [call() for _ in range(2)]
                e.set_option('ls_debug', 0)
                cycles[v] += 1
                n_items = npkt * nr
                ref = np.abs(h6).reshape(n_items, -1).max(1)
                for k, h in enumerate(hs):
                    d = np.abs(h - h6).reshape(n_items, -1).max(1) / ref
                    if not (d <= 2e-6).all():
                        events[v] += 1
                        where[v].append((loop, kind, k, np.nonzero(~(d <= 2e-6))[0].tolist()[:6]))
                        st.describe('variant %#x loop %d Nt=%d %s call %d' % (v, loop, nt, kind, k), h, h6, limit=3)
                if not a.reuse:
                    for arr in (d_re, d_im, o_re, o_im):
                        arr.free()              # (a DeviceArray outlives its engine only as a leak)
                    e.close()
    dt = time.time() - t0
    for v in variants:
        print('variant %#6x: events %d in %d cycles  %s' % (v, events[v], cycles[v], where[v][:8]))
    print('%.1f s, %.1f cycles/s (Nt=%d Nr=%d %d packets, kinds %s, reuse %s)' % (dt, sum(cycles.values()) / dt, nt, nr, npkt, a.kinds, a.reuse))


if __name__ == '__main__':
    main()
