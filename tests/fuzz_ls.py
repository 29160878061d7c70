#!/usr/bin/env python3
"""Random LS shapes against the oracle (GPU box):  python tests/fuzz_ls.py [cases] [seed]; a seeded, bounded run of the same
cases is part of the gpu test suite (tests/test_gpu_ls.py::test_fuzz_ls_cases).

Every case: antenna count, rx count and packet count at random (item counts below, at and far above the resident grid of
the persistent kernels), the Sylvester Hadamard pilot matrix or a generic one, the automatic kernel and every kernel
that can serve the shape ('ls_kernel' 1-6, 'ls_v2' 0 / 1); checked on the first, two middle and the last packets."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg            # noqa: E402
from oracle import csi_oracle as oracle                  # noqa: E402


def rel_rows(a, b):
    a = np.asarray(a, np.float64).reshape(-1, a.shape[-1])
    b = np.asarray(b, np.float64).reshape(-1, b.shape[-1])
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)))


def run_case(rng, c=0, log=print):
    """One random case; returns the number of (kernel, variant) combinations that failed."""
    fails = 0
    nt = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 96, 100, 128]))
    nr = int(rng.integers(1, 5))
    npkt = int(rng.choice([1, 2, 3, 7, 33, 64, 65, 129, 300, 517])) if nt <= 64 else int(rng.choice([1, 2, 5, 33, 130]))
    pow2 = nt & (nt - 1) == 0
    had = pow2 and rng.random() < 0.6
    P = oracle.hadamard(nt) if had else rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(P)
    sel = np.unique(np.r_[0, npkt // 3, npkt // 2, npkt - 1])
    ref = oracle.ls_estimate(np.asarray(ltf)[sel].astype(np.complex64), P)
    refc = np.concatenate([ref.real, ref.imag], -1)
    kernels = [0] + ([4, 5] if had and nt >= 16 else []) + ([1] if nt <= 64 else []) + ([2, 6] if 16 <= nt <= 128 else []) + [3]
    worst = 0.0
    for k in kernels:
        for v in ((0, 1) if k in (5, 6) else (0,)):
            e.set_option('ls_kernel', k)
            e.set_option('ls_v2', v)
            h = e.ls_estimate(ltf)
            err = rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), refc)
            fin = bool(np.isfinite(h.view(np.float32)).all()) and (npkt < 4 or np.abs(h).sum(axis=(1, 2, 3)).min() > 0)
            worst = max(worst, err)
            if not (err < 2e-6 and fin):
                fails += 1
                log('FAIL nt=%d nr=%d npkt=%d had=%d kernel=%d v=%d err=%.2e finite=%s' % (nt, nr, npkt, had, k, v, err, fin))
    e.close()
    log('%3d nt=%3d nr=%d npkt=%4d %s kernels=%s worst=%.2e' % (c, nt, nr, npkt, 'hadamard' if had else 'generic ', kernels, worst))
    return fails


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    fails = 0
    for c in range(cases):
        fails += run_case(rng, c, lambda m: print(m, flush=True))
    print('FAILURES:', fails)
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
