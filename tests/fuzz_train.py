#!/usr/bin/env python3
"""Randomised sweep of the training step (not collected by pytest; GPU box: python tests/fuzz_train.py [cases] [seed]):
random widths / depth / batch size / BatchNormalization on-off; loss and every gradient of one step
(noise and dropout off) against the fp64 oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402
from oracle import csi_oracle as o               # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(cases):
        nt = int(rng.choice([4, 8, 16]))
        nh = int(rng.integers(1, 4))
        hidden = tuple(int(8 * rng.integers(1, 25)) for _ in range(nh))
        n_out = int(rng.choice([16, 64, 234]))
        B = int(rng.integers(2, 300))
        use_bn = bool(rng.integers(0, 2))
        d_in = 321 * nt
        w = o.make_weights(rng, d_in, hidden, n_out, use_bn=use_bn)
        x = rng.standard_normal((B, d_in)).astype(np.float32)
        y = rng.standard_normal((B, n_out)).astype(np.float32)
        e = pkg.CsiEngine(nt, 1, hidden=hidden, n_out=n_out, use_bn=use_bn)
        e.train_begin('real', weights=w, lr=1e-3, dropout=0.0, seed=i)
        loss = e.train_step('real', x, y)
        ref = {k: np.asarray(v, np.float64) for k, v in w.items() if k != 'bn_eps'}
        rloss, _, g = o.train_step_reference(ref, o.adam_init(ref), x, y, lr=1e-3, use_bn=use_bn)
        worst = max(rel(e.train_get('real', 'grad:' + k), gk) for k, gk in g.items())
        ok = abs(loss - rloss) < 5e-5 * max(1.0, rloss) and worst < 5e-4
        bad += not ok
        print(f'{i:3d} nt={nt:2d} hidden={hidden} n_out={n_out} B={B:3d} bn={int(use_bn)} loss_err={abs(loss - rloss):.1e} grad={worst:.1e} {"ok" if ok else "FAIL"}')
        e.train_end('real', commit=False)
    print('FAILURES:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
