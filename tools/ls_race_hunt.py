#!/usr/bin/env python3
"""Hunts the rare wrong items of the generic-P LS kernels (round 4: tests/stress_ls_generic.py still sees ~1 bad launch in a few
hundred with the bf16-split despread, drain or no drain).  Launches one configuration many times on device-resident inputs, compares
every launch with the fp32-despread kernel's result and, for every bad item, prints where it sits (workgroup, position in its
workgroup's sequence) and what is wrong in it (antennas x bins pattern).

    python tools/ls_race_hunt.py [--nt 24] [--nr 4] [--npkt 1500] [--kernel 7] [--v2 0] [--launches 600] [--pilot pm1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nt', type=int, default=24)
    ap.add_argument('--nr', type=int, default=4)
    ap.add_argument('--npkt', type=int, default=1500)
    ap.add_argument('--kernel', type=int, default=7)
    ap.add_argument('--v2', type=int, default=0)
    ap.add_argument('--dbg', type=int, default=0)
    ap.add_argument('--launches', type=int, default=600)
    ap.add_argument('--pilot', default='pm1')
    ap.add_argument('--between', default='', help="'dnn': run a DNN predict between LS launches (other kernels in flight, clocks moving)")
    ap.add_argument('--fresh', type=int, default=0, help='N > 0: N cycles of [new engine, set_pilot, host-path LS with kernel 6, then kernel 7 twice] - '
                    'the sequence of tests/stress_ls_generic.py, where the bad launch was the first one of the freshly selected kernel')
    ap.add_argument('--side', type=int, default=1, help='hp_side_threads of the host pipeline (--fresh)')
    a = ap.parse_args()
    if a.fresh:
        return fresh(a)
    rng = np.random.default_rng(5)
    nt, nr, npkt = a.nt, a.nr, a.npkt
    if a.pilot == 'pm1':
        P = rng.choice([-1.0, 1.0], (nt, nt))
    else:
        P = np.linalg.qr(rng.standard_normal((nt, nt)))[0] * np.sqrt(nt)
    e = pkg.CsiEngine(nt, nr, hidden=(64, 64))
    e.set_pilot(P)
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real, np.float32)), e.to_device(np.ascontiguousarray(ltf.imag, np.float32))
    h_re, h_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.set_option('ls_kernel', 6)
    e.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
    e.synchronize()
    ref = h_re.download() + 1j * h_im.download()
    e.set_option('ls_kernel', a.kernel)
    e.set_option('ls_v2', a.v2)
    e.set_option('ls_debug', a.dbg)
    print('Nt=%d Nr=%d items=%d kernel %d (mode %d) v2=%d dbg=%d pilot=%s pieces=%d' % (nt, nr, npkt * nr, a.kernel, e.get_option('ls_mode'), a.v2, a.dbg,
                                                                                 a.pilot, e.get_option('ls_pilot_pieces')), flush=True)
    scale = np.abs(ref).reshape(npkt * nr, -1).max(1)
    bad_launches = 0
    for it in range(a.launches):
        h_re.upload(np.zeros((1, nr, nt, 234), np.float32))             # touch: nothing stale
        e.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
        e.synchronize()
        h = h_re.download() + 1j * h_im.download()
        d = np.abs(h - ref).reshape(npkt * nr, nt, 234)
        item_err = d.reshape(npkt * nr, -1).max(1) / scale
        bad = np.nonzero(item_err > 2e-6)[0]
        if len(bad):
            bad_launches += 1
            print('launch %d: %d bad items %s' % (it, len(bad), bad.tolist()[:20]), flush=True)
            for b in bad[:6]:
                w = d[b] > 2e-6 * scale[b]
                ants = np.nonzero(w.any(1))[0]
                bins = np.nonzero(w.any(0))[0]
                print('   item %d: %d wrong values; antennas %s; bins n=%d first %s  bins mod 4 %s mod 32 hist %s; max rel err %.3g' % (
                    b, int(w.sum()), ants.tolist()[:40], len(bins), bins.tolist()[:12], np.bincount(bins % 4, minlength=4).tolist(),
                    np.bincount(bins // 32, minlength=8).tolist(), item_err[b]), flush=True)
    print('bad launches: %d of %d' % (bad_launches, a.launches))


def describe(d, scale, bad, limit=6):
    for b in bad[:limit]:
        w = d[b] > 2e-6 * scale[b]
        ants = np.nonzero(w.any(1))[0]
        bins = np.nonzero(w.any(0))[0]
        print('   item %d: %d wrong values; antennas %s; bins n=%d first %s  bins mod 4 %s, by 32 %s; max abs err / item max %.3g' % (
            b, int(w.sum()), ants.tolist()[:40], len(bins), bins.tolist()[:12], np.bincount(bins % 4, minlength=4).tolist(),
            np.bincount(bins // 32, minlength=8).tolist(), float(d[b].max() / scale[b])), flush=True)


def fresh(a):
    rng = np.random.default_rng(5)
    nt, nr, npkt = a.nt, a.nr, a.npkt
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    bad_cycles = 0
    for cyc in range(a.fresh):
        P = rng.choice([-1.0, 1.0], (nt, nt)) if a.pilot == 'pm1' else np.linalg.qr(rng.standard_normal((nt, nt)))[0] * np.sqrt(nt)
        e = pkg.CsiEngine(nt, nr, hidden=(8,))
        e.set_option('hp_side_threads', a.side)
        e.set_pilot(P)
        e.set_option('ls_kernel', 6)
        h6 = e.ls_estimate(ltf)
        e.set_option('ls_kernel', a.kernel)
        e.set_option('ls_v2', a.v2)
        e.set_option('ls_debug', a.dbg)
        hs = [e.ls_estimate(ltf) for _ in range(3)]
        scale = np.abs(h6).reshape(npkt * nr, -1).max(1)
        for k, h in enumerate(hs):
            d = np.abs(h - h6).reshape(npkt * nr, nt, 234)
            bad = np.nonzero(d.reshape(npkt * nr, -1).max(1) / scale > 2e-6)[0]
            if len(bad):
                bad_cycles += 1
                print('cycle %d, call %d of the fresh kernel: %d bad items %s' % (cyc, k, len(bad), bad.tolist()[:20]), flush=True)
                describe(d, scale, bad)
        e.close()
    print('cycles with a bad call: %d of %d (Nt=%d kernel %d v2=%d dbg=%d side_threads=%d)' % (bad_cycles, a.fresh, nt, a.kernel, a.v2, a.dbg, a.side))


if __name__ == '__main__':
    main()
