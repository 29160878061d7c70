#!/usr/bin/env python3
"""bf16_l0_split_fuzz.py - random bf16 shapes in the range "bf16_l0_fused_split" serves (more preambles than the weight-streaming kernel takes, fewer than 256
tiles of the fused kernel): the fused kernel with K ranges against the form it replaces (row-wise norm-relative), run to run, and two packets against the oracle's
bf16-operand emulation.  usage: bf16_l0_split_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
from oracle import csi_oracle as o
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1, a.shape[-1]); b = np.asarray(b, np.float64).reshape(-1, b.shape[-1])
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)))


bad = 0
for i in range(cases):
    nt = int(rng.choice([8, 16, 32, 64])); nr = int(rng.choice([1, 2, 3, 4]))
    h1 = int(rng.choice([256, 512, 768, 1024, 1536])); h2 = int(rng.choice([64, 128, 256, 512]))
    nh = int(rng.choice([1, 2, 2, 2]))
    hidden = (h1, h2)[:nh]
    m1 = int(rng.integers(1300, 5000)); npkt = max(2, m1 // nr)
    if npkt * nr * nt * max(hidden) * 4 > 6e9: npkt = int(6e9 / (nr * nt * max(hidden) * 4))
    w = [o.make_weights(rng, 320 * nt + nt, list(hidden), 234) for _ in range(2)]
    P = o.hadamard(nt)
    ltf = o.make_structured_packets(rng, npkt, nr, P, snr_db=8.0)[0].astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
    e.load_weights('real', w[0]); e.load_weights('imag', w[1]); e.set_pilot(P)
    # device-resident call (the host-buffer entry points cut a call into chunks of their pinned slots: small chunks never reach this path)
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real)), e.to_device(np.ascontiguousarray(ltf.imag))
    outs = [e.empty((npkt, nr, nt, 234)) for _ in range(2)]

    def run():
        e.predict_device(d_re, d_im, npkt, *outs); e.synchronize()
        return [x.download() for x in outs]
    n0 = e.get_option('bf16_l0_fused_split_launches')
    a = run(); took = e.get_option('bf16_l0_fused_split_launches') - n0
    a2 = run()
    e.set_option('bf16_l0_fused_split', 0)
    b = run()
    sel = [0, npkt - 1]
    r = o.predict_packets_bf16(ltf[sel], P, w[0], w[1])
    d_ab = max(rel(a[0], b[0]), rel(a[1], b[1])); d_or = max(rel(a[0][sel], r[0]), rel(a[1][sel], r[1])); d_or0 = max(rel(b[0][sel], r[0]), rel(b[1][sel], r[1]))
    same = np.array_equal(a[0], a2[0]) and np.array_equal(a[1], a2[1])
    ok = same and d_or < 8e-3 and np.isfinite(a[0]).all() and (d_ab < 1.2e-2)
    bad += not ok
    print('case %2d nt %2d nr %d packets %4d hidden %s: K-range launches %d, vs other form %.2e, vs bf16 emulation %.2e (other form %.2e), run to run %s %s' % (
        i, nt, nr, npkt, hidden, took, d_ab, d_or, d_or0, same, '' if ok else '<-- FAILED'), flush=True)
    e.close()
print('RESULT', 'ok' if not bad else '%d FAILED' % bad)
