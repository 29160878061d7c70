#!/usr/bin/env python3
"""Round 4: the Walsh-Hadamard LS kernel on Hadamard-EQUIVALENT pilot matrices (signed row / column permutations of the Sylvester
matrix: the 802.11 VHT 4x4 base doubled up, a random signed permutation) next to the Sylvester matrix itself and next to the
generic-P kernels the same pilots took in round 3 (`ls_fast_perm` = 0).  Variants interleaved on one box, HIP-event time per
launch, median over rounds; every fast result is compared with the generic kernel's on the first packets.

    python tools/ls_pilot_probe.py [--shapes 16x4x8000,32x4x4000,64x4x5000,128x16x2000] [--rounds 5] [--reps 10]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_channel_estimation_mamimo_amd.engine import CsiEngine            # noqa: E402
from dl_channel_estimation_mamimo_amd.synth import hadamard              # noqa: E402

P_VHT4 = np.array([[1, -1, 1, 1], [1, 1, -1, 1], [1, 1, 1, -1], [-1, 1, 1, 1]], np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='16x4x8000,32x4x4000,64x4x5000,128x16x2000')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    for shape in args.shapes.split(','):
        nt, nr, npkt = (int(v) for v in shape.split('x'))
        rng = np.random.default_rng(nt)
        H = hadamard(nt).astype(np.float64)
        pilots = {'sylvester': H, 'vht_kron': np.kron(hadamard(nt // 4), P_VHT4),
                  'signed_perm': rng.choice([-1.0, 1.0], nt)[:, None] * H[rng.permutation(nt)][:, rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[None, :]}
        pilots['cols_only'] = H[:, rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[None, :]       # symbol fetch permuted, antenna store in order
        pilots['rows_only'] = rng.choice([-1.0, 1.0], nt)[:, None] * H[rng.permutation(nt)]        # the other way round
        eng = CsiEngine(nt, nr, hidden=(32,), n_out=234)
        d_re, d_im = eng.empty((npkt, nr, 320 * nt)), eng.empty((npkt, nr, 320 * nt))
        eng.synth_white(7, 0, npkt, d_re, d_im)
        d_hr, d_hi = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
        cases = [('sylvester', 1, 0), ('vht_kron', 1, 0), ('signed_perm', 1, 0), ('cols_only', 1, 0), ('rows_only', 1, 0), ('vht_kron', 0, 0), ('signed_perm', 0, 0)]
        cases.insert(1, ('sylvester', 1, 3))      # A/B of the store form on the table-free kernel: v3 = vector addresses (default: scalar base)
        times = {c: [] for c in cases}
        info = {}
        ref = {}
        for rnd in range(args.rounds):
            for c in cases:
                name, fast, v2 = c
                eng.set_pilot(pilots[name])
                eng.set_option('ls_fast_perm', fast)
                eng.set_option('ls_v2', v2)
                for _ in range(5 if rnd == 0 else 2):
                    eng.ls_estimate_device(d_re, d_im, npkt, d_hr, d_hi)
                eng.synchronize()
                eng.profile_enable(True)
                eng.profile_reset()
                for _ in range(args.reps):
                    eng.ls_estimate_device(d_re, d_im, npkt, d_hr, d_hi)
                eng.synchronize()
                pr = eng.profile()['ls_estimate']
                eng.profile_enable(False)
                times[c].append(pr['ms'] / pr['launches'])
                if rnd == 0:
                    n = min(npkt, 32)
                    h = d_hr.download(0, n) + 1j * d_hi.download(0, n)
                    info[c] = 'mode %d class %d' % (eng.get_option('ls_mode'), eng.get_option('ls_pilot_fast'))
                    if fast == 0 or name == 'sylvester':
                        ref.setdefault(name, h)
                    if name in ref and ref[name] is not h:
                        info[c] += '  max rel diff vs %s %.2e' % ('generic kernel' if name != 'sylvester' else 'default kernel', float(np.max(np.abs(h - ref[name])) / np.max(np.abs(ref[name]))))
                    elif name not in ref:
                        info[c] += '  (checked against the generic kernel below)'
                        ref['_pending_' + name] = h
                if rnd == 0 and fast == 0 and ('_pending_' + name) in ref:
                    hp = ref.pop('_pending_' + name)
                    info[(name, 1, 0)] = info[(name, 1, 0)].replace('(checked against the generic kernel below)',
                                                                      'max rel diff vs generic kernel %.2e' % float(np.max(np.abs(hp - ref[name])) / np.max(np.abs(ref[name]))))
        pairs = npkt * nr * nt
        base = None
        for c in cases:
            ts = sorted(times[c])
            ms = ts[len(ts) // 2]
            base = ms if base is None else base
            tbs = pairs * (2560 + 1872) / (ms * 1e-3) / 1e12
            print('Nt=%3d Nr=%2d pkts=%5d  %-12s fast_perm=%d v%d : median %7.3f ms (min %7.3f)  %.3f of 8 TB/s  %+5.1f %% vs sylvester   %s'
                  % (nt, nr, npkt, c[0], c[1], c[2], ms, ts[0], tbs / 8, 100.0 * (ms / base - 1.0), info.get(c, '')), flush=True)
        eng.close()


if __name__ == '__main__':
    main()
