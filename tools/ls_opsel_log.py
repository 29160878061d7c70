#!/usr/bin/env python3
"""ls_opsel_log.py - with the library built by `tools/ls_opsel_hunt.sh build 512` (every +-i rotation of the LS transform computed as the packed op_sel
instruction AND in scalar operations, differences logged on the device): run the reproducible case and say what the wrong packed results ARE.
usage: CSI_DEBUG_HOOKS=1 CSI_LIBRARY_PATH=build_variants/libcsi_v512.so ls_opsel_log.py [calls]"""
import os, sys, ctypes
os.environ['CSI_DEBUG_HOOKS'] = '1'
os.environ['CSI_BF16_FORK_EARLY'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = pkg.load_library()
lib.csi_debug_opsel_log.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
nt, nr, hidden = 64, 4, (1024, 1024)
rng = np.random.default_rng(1)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
eng.set_pilot(pkg.synth.hadamard(nt))
n = 1000
d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
eng.synth_white(11, 0, n, d_re, d_im)
o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
buf = np.zeros(4 + 8 * 4096, np.uint32)


def read_log(reset=1):
    assert lib.csi_debug_opsel_log(buf.ctypes.data, buf.size, reset) == 0
    cnt = int(buf[0]); e = buf[4:4 + 8 * min(cnt, 4096)].reshape(-1, 8).copy()
    return cnt, e


eng.set_option('small_call_overlap', 0)
eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
cnt, _ = read_log()
print('one stream: %d packed results differ from the scalar form' % cnt)
eng.set_option('small_call_overlap', 1)
alle = []
for it in range(calls):
    eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
    cnt, e = read_log()
    print('two streams, call %d: %d differ' % (it, cnt))
    alle.append(e)
e = np.concatenate(alle) if alle else np.zeros((0, 8), np.uint32)
if len(e):
    kind = e[:, 0] & 255; tid = e[:, 0] >> 8; lane = tid & 63; wave = tid >> 6
    f = e[:, 1:7].copy().view(np.float32)
    ax, ay, bx, by, px, py = (f[:, i] for i in range(6))
    print('entries %d: kind add_mi %d / add_pi %d; lanes min %d max %d; lane histogram by 16: %s; waves %s; workgroups %d distinct' % (
        len(e), int((kind == 0).sum()), int((kind == 1).sum()), lane.min(), lane.max(), np.bincount(lane // 16, minlength=4).tolist(),
        np.bincount(wave, minlength=4).tolist(), len(np.unique(e[:, 7]))))
    sgn = np.where(kind == 0, 1.0, -1.0).astype(np.float32)
    good = (ax + sgn * by, ay - sgn * bx)
    cands = {
        'correct': good,
        'the OTHER rotation (neg on the other half)': (ax - sgn * by, ay + sgn * bx),
        'no negation': (ax + by, ay + bx),
        'both negated': (ax - by, ay - bx),
        'no swap, neg kept on its half': (ax + sgn * bx if False else np.where(kind == 0, ax + bx, ax - bx), np.where(kind == 0, ay - by, ay + by)),
        'no swap, no neg': (ax + bx, ay + by),
        'a': (ax, ay), 'b': (bx, by), 'swapped b': (by, bx),
    }
    for half, (p, gi) in (('lo', (px, 0)), ('hi', (py, 1))):
        wrong = p.view(np.uint32) != good[gi].astype(np.float32).view(np.uint32)
        print('  %s half wrong in %d entries' % (half, int(wrong.sum())))
        for name, c in cands.items():
            m = (p.view(np.uint32) == c[gi].astype(np.float32).view(np.uint32)) & wrong
            if m.any(): print('      = "%s" in %d' % (name, int(m.sum())))
    print('  first entries (kind, tid, a, b, packed result, scalar result):')
    for i in range(min(12, len(e))):
        print('   ', int(kind[i]), int(tid[i]), (float(ax[i]), float(ay[i])), (float(bx[i]), float(by[i])), (float(px[i]), float(py[i])), (float(good[0][i]), float(good[1][i])))
