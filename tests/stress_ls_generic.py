#!/usr/bin/env python3
"""Stress run of the generic-P LS kernels on one MI355X (not a pytest file; `python tests/stress_ls_generic.py [--reps 20]`).

The bf16-split despread kernel (ls_kernel 7) is run `reps` times per (Nt, pilot kind, ring depth) over many more items than
resident workgroups; EVERY item of every run is compared bit for bit with the first run and, to rounding, with the fp32
matrix-core despread (ls_kernel 6).  Written after a race was found in the first version of the kernel (LDS reads of the next
chunk issued while the last MFMAs of a chunk were still in the pipe: 1-5 wrong items per 4000, only with two waves per SIMD):
the drain that went in with the fix turned out not to be what closed it (round 4: tools/mfma_war_probe.hip shows the
suspected hardware hazard does not exist, and this script is clean without the drain); it is off by default now, `--drain`
(ls_debug 64) puts it back."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg            # noqa: E402


def pilot(rng, nt, kind):
    if kind == 'pm1':
        return rng.choice([-1.0, 1.0], (nt, nt))
    P = np.linalg.qr(rng.standard_normal((nt, nt)))[0].astype(np.float32) * np.float32(np.sqrt(nt))
    if kind == 'q16':
        P = (P.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)
    return P.astype(np.float64)


def describe(tag, h, ref, grid_hint=None, limit=8):
    """where the bad items of one run sit and what is wrong inside them (printed whenever a run differs)"""
    n_items = h.shape[0] * h.shape[1]
    nt = h.shape[2]
    d = np.abs(h - ref).reshape(n_items, nt, 234)
    scale = np.abs(ref).reshape(n_items, -1).max(1)
    bad = np.nonzero(~(d.reshape(n_items, -1).max(1) <= 2e-6 * scale))[0]          # NaN counts as bad
    print('   !! %s: %d bad items of %d: %s' % (tag, len(bad), n_items, bad.tolist()[:24]), flush=True)
    for b in bad[:limit]:
        w = ~(d[b] <= 2e-6 * scale[b])
        ants = np.nonzero(w.any(1))[0]
        bins = np.nonzero(w.any(0))[0]
        print('      item %d (packet %d rx %d): %d wrong values; antennas %s; %d bins, first %s, bins mod 4 %s, per 32 %s; worst abs err / item max %.3g; got max %.3g ref max %.3g'
              % (b, b // h.shape[1], b % h.shape[1], int(w.sum()), ants.tolist()[:32], len(bins), bins.tolist()[:10], np.bincount(bins % 4, minlength=4).tolist(),
                 np.bincount(bins // 32, minlength=8).tolist(), float(d[b].max() / scale[b]), float(np.abs(h.reshape(n_items, -1)[b]).max()), float(scale[b])), flush=True)


def run(reps=10, shapes='16x4x2000,24x4x1500,32x4x1500,48x4x800,64x4x800,96x4x400,128x4x300', no_drain=True, budget_s=None, quiet=False, runs=None, small_ringb=False):
    """Returns the number of bad items.  budget_s bounds the wall time (the loops stop between configurations once it is spent);
    `runs` is an alias of `reps` (pytest caller)."""
    import time
    reps = runs if runs is not None else reps
    t0 = time.time()
    spent = lambda: budget_s is not None and time.time() - t0 > budget_s
    say = (lambda *a, **k: None) if quiet else print       # (describe() always prints)
    rng = np.random.default_rng(5)
    total_bad = 0
    for shape in shapes.split(','):
        nt, nr, npkt = (int(v) for v in shape.split('x'))
        ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
        for kind in ('pm1', 'q16', 'qr'):
            if spent() or (nt <= 32 and not small_ringb):
                break          # (round 4: the one-antenna-tile form of kernel 7 is no longer selected by the library - see csi_context.hpp, ls_ringb_min)
            e = pkg.CsiEngine(nt, nr, hidden=(8,))
            e.set_pilot(pilot(rng, nt, kind))
            e.set_option('ls_kernel', 6)
            h6 = e.ls_estimate(ltf)
            e.set_option('ls_kernel', 7)
            for v in (0, 1):
                e.set_option('ls_v2', v)
                e.set_option('ls_debug', 0 if no_drain else 64)
                first, bad_runs, bad_items = None, 0, 0
                for _ in range(reps):
                    h = e.ls_estimate(ltf)
                    if first is None:
                        first = h
                        d = np.abs(h - h6).reshape(npkt * nr, -1).max(1) / np.abs(h6).reshape(npkt * nr, -1).max(1)
                        n6 = int((d > 2e-6).sum())
                        if n6:
                            describe('Nt=%d %s v%d first run against the fp32 despread' % (nt, kind, v), h, h6)
                    nb = int((h != first).reshape(npkt * nr, -1).any(1).sum())
                    if nb and not n6:
                        describe('Nt=%d %s v%d a later run against the first' % (nt, kind, v), h, first)
                    bad_runs += nb > 0
                    bad_items += nb
                total_bad += bad_items + n6
                say(f'Nt={nt:3d} items={npkt * nr:5d} pilot={kind:3s} pieces={e.get_option("ls_pilot_pieces")} shape v{v}: '
                    f'{reps} runs, {bad_runs} differ from the first ({bad_items} items), {n6} items off the fp32 despread', flush=True)
            e.close() if hasattr(e, 'close') else None
    # the round-2 kernels the same way: Walsh-Hadamard ring kernel (the headline LS kernel; round 4: also its table-driven form for a
    # signed permutation of the Sylvester matrix) and the fp32 matrix-core ring kernel
    for shape in shapes.split(','):
        nt, nr, npkt = (int(v) for v in shape.split('x'))
        ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
        pow2 = nt & (nt - 1) == 0
        Hs = pkg.synth.hadamard(nt) if pow2 else None
        Hp = (rng.choice([-1.0, 1.0], nt)[:, None] * Hs[rng.permutation(nt)][:, rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[None, :]) if pow2 else None
        for kernel, P in ((5, Hs), (5, Hp), (6, pilot(rng, nt, 'qr'))):
            if P is None or spent():
                continue
            e = pkg.CsiEngine(nt, nr, hidden=(8,))
            e.set_pilot(P)
            e.set_option('ls_kernel', kernel)
            for v in (0, 1):
                e.set_option('ls_v2', v)
                first, bad_runs, bad_items = None, 0, 0
                for _ in range(reps):
                    h = e.ls_estimate(ltf)
                    first = h if first is None else first
                    nb = int((h != first).reshape(npkt * nr, -1).any(1).sum())
                    if nb:
                        describe('Nt=%d kernel %d v%d a later run against the first' % (nt, kernel, v), h, first)
                    bad_runs += nb > 0
                    bad_items += nb
                total_bad += bad_items
                say(f'Nt={nt:3d} items={npkt * nr:5d} kernel {kernel} (mode {e.get_option("ls_mode")}, pilot class {e.get_option("ls_pilot_fast")}) shape v{v}: {reps} runs, {bad_runs} differ from the first ({bad_items} items)', flush=True)
    say('TOTAL bad items:', total_bad)
    return total_bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--no-drain', action='store_true', help='(the default since round 4: kept for old command lines)')
    ap.add_argument('--drain', action='store_true', help='run the bf16-split kernel WITH the round-3 MFMA drain (ls_debug 64)')
    ap.add_argument('--shapes', default='16x4x2000,24x4x1500,32x4x1500,48x4x800,64x4x800,96x4x400,128x4x300')
    ap.add_argument('--budget-s', type=float, default=None)
    ap.add_argument('--small-ringb', action='store_true', help='also force kernel 7 at Nt <= 32 (two workgroups per CU: the configuration with the open rare failure)')
    args = ap.parse_args()
    return 1 if run(args.reps, args.shapes, not args.drain, args.budget_s, small_ringb=args.small_ringb) else 0


if __name__ == '__main__':
    sys.exit(main())
