#!/usr/bin/env python3
"""ls_opsel_log.py - with the library built by `tools/ls_opsel_hunt.sh build 512` (every +-i rotation of the LS transform computed as the packed op_sel
instruction AND in scalar operations, differences logged on the device): run the reproducible case and say what the wrong packed results ARE.
usage: CSI_DEBUG_HOOKS=1 CSI_LIBRARY_PATH=build_variants/libcsi_v512.so ls_opsel_log.py [calls]"""
import os, sys, ctypes
os.environ['CSI_DEBUG_HOOKS'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ls_debug = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
lib = pkg.load_library()
lib.csi_debug_opsel_log.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
nt, nr, hidden = 64, 4, (1024, 1024)
rng = np.random.default_rng(1)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
if ls_debug: eng.set_option('ls_debug', ls_debug); print('ls_debug', ls_debug)
n = 1000
d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
eng.synth_white(11, 0, n, d_re, d_im)
o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
buf = np.zeros(4 + 16 * 4096, np.uint32)


def read_log(reset=1):
    assert lib.csi_debug_opsel_log(buf.ctypes.data, buf.size, reset) == 0
    cnt = int(buf[0]); e = buf[4:4 + 16 * min(cnt, 4096)].reshape(-1, 16).copy()
    return cnt, e


eng.set_option('small_call_overlap', 0)
eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
cnt, _ = read_log()
print('one stream: %d packed results differ from the single operations' % cnt)
eng.set_option('small_call_overlap', 1)
alle = []
for it in range(calls):
    eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
    cnt, e = read_log()
    print('two streams, call %d: %d differ' % (it, cnt))
    alle.append(e)
e = np.concatenate(alle) if alle else np.zeros((0, 16), np.uint32)
if len(e):
    kind = e[:, 0] & 255; tid = e[:, 0] >> 8; lane = tid & 63; wave = tid >> 6
    f = e[:, 2:16].copy().view(np.float32)
    a0, a1, b0, b1, p0, p1, a0l, a1l, b0l, b1l, q0, q1, s0, s1 = (f[:, i] for i in range(14))
    U = lambda x: x.view(np.uint32)
    print('entries %d: add_mi %d / add_pi %d; lane histogram by 16: %s; waves %s; %d workgroups' % (
        len(e), int((kind == 0).sum()), int((kind == 1).sum()), np.bincount(lane // 16, minlength=4).tolist(), np.bincount(wave, minlength=4).tolist(), len(np.unique(e[:, 1]))))
    pw = (U(p0) != U(s0)) | (U(p1) != U(s1)); qw = (U(q0) != U(s0)) | (U(q1) != U(s1))
    moved = (U(a0) != U(a0l)) | (U(a1) != U(a1l)) | (U(b0) != U(b0l)) | (U(b1) != U(b1l))
    print('  first execution wrong: %d   second execution (8+ cycles later) wrong: %d   both: %d   operands read differently before / after: %d' % (
        int(pw.sum()), int(qw.sum()), int((pw & qw).sum()), int(moved.sum())))
    sgn = np.where(kind == 0, 1.0, -1.0).astype(np.float32)
    cands = {
        'the other rotation': (a0 - sgn * b1, a1 + sgn * b0), 'no negation': (a0 + b1, a1 + b0), 'both negated': (a0 - b1, a1 - b0),
        'no swap, neg on its half': (np.where(kind == 0, a0 + b0, a0 - b0), np.where(kind == 0, a1 - b1, a1 + b1)), 'no swap no neg': (a0 + b0, a1 + b1),
        'a': (a0, a1), 'b': (b0, b1), 'b swapped': (b1, b0), 'zero': (0 * a0, 0 * a0),
        'lo from both halves of b.lo: (a0 + b0, a1 - b0)': (a0 + sgn * b0, a1 - sgn * b0), 'hi: (a0 + b1, a1 - b1)': (a0 + sgn * b1, a1 - sgn * b1),
    }
    for nm, (r0, r1, w) in (('first', (p0, p1, pw)), ('second', (q0, q1, qw))):
        lo_w = (U(r0) != U(s0)) & w; hi_w = (U(r1) != U(s1)) & w
        print('  %s execution: lo half wrong %d, hi half wrong %d' % (nm, int(lo_w.sum()), int(hi_w.sum())))
        for name, c in cands.items():
            m0 = (U(r0) == U(c[0].astype(np.float32))) & lo_w; m1 = (U(r1) == U(c[1].astype(np.float32))) & hi_w
            if m0.any() or m1.any(): print('      = "%s": lo %d, hi %d' % (name, int(m0.sum()), int(m1.sum())))
    print('  first entries: kind tid | a | b | first result | second result | single-operation result')
    for i in range(min(10, len(e))):
        print('   ', int(kind[i]), int(tid[i]), '|', float(a0[i]), float(a1[i]), '|', float(b0[i]), float(b1[i]), '|', float(p0[i]), float(p1[i]), '|', float(q0[i]), float(q1[i]), '|', float(s0[i]), float(s1[i]))
