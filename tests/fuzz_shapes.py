#!/usr/bin/env python3
"""Randomised shape sweep (GPU box: python tests/fuzz_shapes.py [cases] [seed]; a seeded, bounded run of the same cases is part
of the gpu test suite, tests/test_gpu_dnn_f32.py::test_fuzz_shape_cases):
random Nt / Nr / packet counts / hidden widths / depth / BatchNormalization on-off / dtype / GEMM engine, the shared
layer-0 path, the literal path and the LS estimate against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402
from oracle import csi_oracle as o               # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1, a.shape[-1])
    b = np.asarray(b, np.float64).reshape(-1, b.shape[-1])
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)))


def run_case(rng, i=0, log=print):
    """One random case; True if every check of it passed."""
    nt = int(rng.choice([4, 8, 12, 16, 20, 32, 40, 64]))
    nr = int(rng.choice([1, 2, 3, 4, 8]))
    npkt = int(rng.integers(1, 30))
    if nt <= 16 and rng.random() < 0.3:
        npkt = int(rng.integers(60, 160))            # several 256-row tiles per GEMM
    nh = int(rng.integers(1, 4))
    hidden = tuple(int(8 * rng.integers(1, 33)) for _ in range(nh))
    use_bn = bool(rng.integers(0, 2))
    dtype = 'bf16' if rng.random() < 0.3 else 'f32'
    tile = int(rng.choice([0, 0, 128, 256]))
    engine = int(rng.choice([-1, 0, 1, 1]))            # fp32 contexts: automatic / fp32 MFMA kernels / split-f16 engine
    if engine == 1 and rng.random() < 0.8:
        hidden = tuple(int(16 * rng.integers(1, 17)) for _ in range(nh))     # widths the split engine serves
        if nh == 2 and rng.random() < 0.5:
            hidden = (hidden[0], int(256 * rng.integers(1, 4)))              # second width a multiple of 256: the fused regressor applies
    w_re = o.make_weights(rng, 321 * nt, hidden, 234, use_bn=use_bn)
    w_im = o.make_weights(rng, 321 * nt, hidden, 234, use_bn=use_bn)
    P = o.hadamard(nt) if (nt & (nt - 1)) == 0 and rng.random() < 0.5 else rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=hidden, use_bn=use_bn, dtype=dtype)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('force_tile', tile)
    fuse = blocked = 0
    if dtype == 'f32':
        e.set_option('f32_engine', engine)
        fuse, blocked = int(rng.integers(0, 2)), int(rng.integers(0, 2))       # split engine: fused regressor / activation layout
        e.set_option('hs_fuse_regressor', fuse)
        e.set_option('hs_blocked', blocked)
    o_re, o_im = e.predict(ltf)
    h = e.ls_estimate(ltf)
    k = min(npkt, 4)
    if dtype == 'f32':
        r_re, r_im = o.predict_packets(ltf[:k], P, w_re, w_im, np.float64, pkt_batch=k)
        tol = 1e-5
    else:
        r_re, r_im = o.predict_packets_bf16(ltf[:k], P, w_re, w_im)
        tol = 6e-3
    err = max(rel(o_re[:k], r_re), rel(o_im[:k], r_im))
    x = o.samples_from_packets(ltf[:1], P.astype(np.float32), 'real')
    lit = e.predict_samples('real', x)
    err_lit = rel(lit, o.fc_forward(x, w_re, np.float64) if dtype == 'f32' else o.fc_forward_bf16(x, w_re))
    ref = o.ls_estimate(ltf[:k], P)
    err_ls = rel(np.concatenate([h[:k].real, h[:k].imag], -1), np.concatenate([ref.real, ref.imag], -1))
    ok = err < tol and err_lit < tol and err_ls < 1e-5 and np.isfinite(o_re).all() and np.isfinite(h.view(np.float32)).all()
    log(f'{i:3d} nt={nt:3d} nr={nr} npkt={npkt:3d} hidden={hidden} bn={int(use_bn)} {dtype} tile={tile:3d} engine={engine:2d} fuse={fuse} blk={blocked} hs={e.get_option("hs_launches"):2d} '
          f'dnn={err:.2e} literal={err_lit:.2e} ls={err_ls:.2e} {"ok" if ok else "FAIL"}')
    return bool(ok)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(cases):
        bad += not run_case(rng, i)
    print('FAILURES:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
