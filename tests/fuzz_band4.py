#!/usr/bin/env python3
"""fuzz_band4.py - random shapes the register-blocked band kernels serve (csrc/band4_kernel_gen.py: csi_band4 in fp32 contexts at 16 <= Nt <= 128, csi_band4_bf16 in
bf16 contexts at 32 <= Nt <= 64): against the 8-wave kernels on the same operands (other order of the fp32 sums: 2e-6) and against the oracle (fp64 at 1e-5 /
bf16 emulation at 4e-3).  Ragged last bands, odd output counts (the row-per-lane store path), one ... four column steps, K1 from the smallest served.
usage: fuzz_band4.py [cases] [seed]      (a bounded run is part of the gpu suite: tests/test_gpu_dnn_f32.py::test_fuzz_band4_cases)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def rel_rows(a, b):
    a = np.asarray(a, np.float64).reshape(-1, a.shape[-1]); b = np.asarray(b, np.float64).reshape(a.shape)
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)))


def run_case(rng, i, log=print):
    import dl_channel_estimation_mamimo_amd as pkg
    from oracle import csi_oracle as o
    bf16 = bool(rng.integers(0, 2))
    nt = int(rng.choice([32, 40, 48, 64] if bf16 else [16, 24, 32, 40, 48, 64, 96, 100, 128]))
    nr = int(rng.integers(1, 5))
    npkt = int(rng.integers(1, max(2, 6000 // (nt * nr) + 1)))
    h1 = int(rng.choice([256, 384, 512, 1024] if bf16 else [128, 192, 256, 512, 1024, 1088]))
    h2 = int(rng.choice([256, 512, 768, 1024]))
    n_out = int(rng.choice([18, 52, 233, 234, 256]))     # (a row of 2 outputs makes the norm-relative measure a cancellation lottery: 3e-5 for either kernel)
    use_bn = bool(rng.integers(0, 2))
    w = [o.make_weights(rng, 320 * nt + nt, [h1, h2], n_out, use_bn=use_bn) for _ in range(2)]
    P = o.hadamard(nt) if nt & (nt - 1) == 0 else rng.choice([-1.0, 1.0], (nt, nt))
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=(h1, h2), n_out=n_out, use_bn=use_bn, dtype='bf16' if bf16 else 'f32')
    e.load_weights('real', w[0]); e.load_weights('imag', w[1]); e.set_pilot(P)
    if bf16:
        e.set_option('force_tile', 256)
    else:
        e.set_option('f32_engine', 1); e.set_option('small_fused', 0)
    e.set_option('band_split', 0)
    n0 = e.get_option('band_launches')
    b_re, b_im = e.predict(ltf)
    served = e.get_option('band_launches') - n0
    e.set_option('band4', 0)
    a_re, a_im = e.predict(ltf)
    k = min(npkt, 3)
    if bf16:
        r_re, r_im = o.predict_packets_bf16(ltf[:k], P, w[0], w[1])
        tol = 4e-3
    else:
        r_re, r_im = o.predict_packets(ltf[:k], P, w[0], w[1], np.float64, pkt_batch=k)
        tol = 1e-5
    d48 = max(rel_rows(b_re, a_re), rel_rows(b_im, a_im))
    dor = max(rel_rows(b_re[:k], r_re), rel_rows(b_im[:k], r_im))
    fb = e.get_option('hs_range_fallbacks') if not bf16 else 0
    ok = served == 2 and np.isfinite(b_re).all() and d48 < 2e-6 and dor < tol and e.get_option('band4_available') == 1
    log('%3d %s nt=%3d nr=%d npkt=%3d rows=%5d hidden=(%d, %d) n_out=%3d bn=%d band launches %d fallbacks %d band4-vs-band8 %.2e oracle %.2e %s' % (
        i, 'bf16' if bf16 else 'f32 ', nt, nr, npkt, npkt * nr * nt, h1, h2, n_out, use_bn, served, fb, d48, dor, 'ok' if ok else 'FAILED'))
    e.close()
    return ok


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = sum(0 if run_case(rng, i) else 1 for i in range(cases))
    print('FAILURES: %d' % bad)
