#!/usr/bin/env python3
"""Times csi_train_step at the shipped training shape (pipe.sh:40: Nt=32, --nn 1024 1024, --bs 256,
default_SNR) and prints the per-kernel split.  GPU box:  python tools/train_probe.py [--bs 256] [--nt 32]"""
import argparse
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nt', type=int, default=32)
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--hidden', type=int, nargs='+', default=[1024, 1024])
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    e = pkg.CsiEngine(a.nt, 4, hidden=a.hidden)
    d_in = 321 * a.nt
    x = rng.standard_normal((a.bs, d_in)).astype(np.float32)
    y = rng.standard_normal((a.bs, 234)).astype(np.float32)
    e.train_begin('real', lr=1e-4, dropout=0.15, seed=1)
    for _ in range(3):
        e.train_step('real', x, y, noise_std=0.1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = e.train_step('real', x, y, noise_std=0.1)
    dt = (time.perf_counter() - t0) / a.steps
    print(f'train_step: {dt * 1e3:.3f} ms/step incl. H2D of the batch ({a.bs / dt:.0f} samples/s), loss {loss:.4f}')
    e.profile_enable(True)
    for _ in range(5):
        e.train_step('real', x, y, noise_std=0.1)
    for k, v in e.profile().items():
        if v['launches']:
            print('  %-20s %5d launches  %9.3f ms/step  %8.1f TFLOP/s' % (k, v['launches'] // 5, v['ms'] / 5, v['flops'] / max(v['ms'], 1e-9) / 1e9))
    # resident dataset: 512 rx preambles x nt tx = the samples of 128 packets, batches addressed by index
    e.set_pilot(np.eye(a.nt) * 2.0 - 1.0)
    n_rows = 512
    table = rng.standard_normal((n_rows, 320 * a.nt)).astype(np.float32)
    N = n_rows * a.nt
    e.train_set_dataset('real', table, np.repeat(np.arange(n_rows), a.nt), np.tile(np.arange(a.nt), n_rows),
                        rng.standard_normal((N, 234)).astype(np.float32))
    ids = [rng.permutation(N)[:a.bs] for _ in range(a.steps)]
    e.profile_enable(False)
    for i in range(3):
        e.train_step_indexed('real', ids[i], noise_std=0.1)
    t0 = time.perf_counter()
    for i in range(a.steps):
        e.train_step_indexed('real', ids[i], noise_std=0.1)
    dt = (time.perf_counter() - t0) / a.steps
    print(f'train_step_indexed (resident dataset): {dt * 1e3:.3f} ms/step ({a.bs / dt:.0f} samples/s)')
    t0 = time.perf_counter()
    for _ in range(5):
        e.train_eval('real', x, y)
    print(f'train_eval: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per batch')
    e.train_end('real', commit=False)


if __name__ == '__main__':
    main()
