#!/usr/bin/env python3
"""ls_wrong_block_probe.py - WHAT is wrong in the LS planes of a bf16 call whose second stream is forked in front of the LS kernel (the repro of
profiles/r06_small_calls.txt (4); the order every context uses again since the rotations are single adds; build the old form with tools/ls_opsel_hunt.sh build 0).  For every wrong (packet, rx) item of the
first bad calls: which antennas / bins / planes differ, and whether the difference is one chunk's contribution to one output block (missing, doubled,
sign), another item's values, or a stale previous value.  usage: ls_wrong_block_probe.py [packets] [calls]"""
import os, sys
os.environ['CSI_DEBUG_HOOKS'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nt, nr, hidden, CH = 64, 4, (1024, 1024), 8
rng = np.random.default_rng(1)
eng = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
eng.load_weights('real', pkg.synth.make_weights(rng, nt, hidden)); eng.load_weights('imag', pkg.synth.make_weights(rng, nt, hidden))
P = pkg.synth.hadamard(nt)
eng.set_pilot(P)
d_re, d_im = eng.empty((n, nr, eng.len_ltf)), eng.empty((n, nr, eng.len_ltf))
eng.synth_white(11, 0, n, d_re, d_im)
x = d_re.download().astype(np.float64) + 1j * d_im.download().astype(np.float64)      # [n][nr][len_ltf]
o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]
eng.set_option('small_call_overlap', 0)
eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
ref = [a.download() for a in o]
eng.set_option('small_call_overlap', 1)


from oracle import csi_oracle as orc      # the checker, used by this probe only to locate the data bins
BINS = np.asarray(orc.data_carrier_indices())
H = P.astype(np.float64)


def spectra(item):
    """[nt][256]: FFT of every LTF symbol of a (packet, rx) item (cyclic prefix dropped)."""
    sym = x[item].reshape(nt, 320)[:, 64:]
    return np.fft.fft(sym, axis=1)


seen = 0
for it in range(calls):
    # poison the LS planes so that a store that never happens shows (the previous call's values are the same numbers otherwise)
    for a in o[2:]: a.upload(np.full((n, nr, nt, 234), 7.0, np.float32))
    eng.estimate_device(d_re, d_im, n, *o); eng.synchronize()
    got = [a.download() for a in o]
    d = (got[2] != ref[2]) | (got[3] != ref[3])
    if not d.any(): continue
    items = sorted({(i[0], i[1]) for i in np.argwhere(d).tolist()})
    print('call %d: %d values differ in %d items' % (it, int(d.sum()), len(items)))
    for (p, r) in items[:6]:
        di = d[p, r]                                   # [nt][234]
        ants = np.flatnonzero(di.any(axis=1)); bins_bad = np.flatnonzero(di.any(axis=0))
        gre, gim, rre, rim = got[2][p, r], got[3][p, r], ref[2][p, r], ref[3][p, r]
        dz = (gre - rre).astype(np.float64) + 1j * (gim - rim).astype(np.float64)
        line = '  item (%d,%d): antennas %s  bins %d (%d..%d)  |diff| max %.3g  ref rms %.3g' % (
            p, r, ants.tolist(), len(bins_bad), bins_bad.min(), bins_bad.max(), np.abs(dz).max(), np.sqrt(np.mean(rre ** 2 + rim ** 2)))
        print(line)
        print('     poison left (7.0): re %d im %d values' % (int((gre == 7.0).sum()), int((gim == 7.0).sum())))
        # equals another item's values?
        blk = slice(ants.min(), ants.max() + 1)
        for (pp, rr) in [(p, (r + k) % nr) for k in range(1, nr)] + [((p + k) % n, r) for k in (-1, 1)]:
            if np.array_equal(gre[blk], ref[2][pp, rr][blk]): print('     = item (%d,%d) of the reference' % (pp, rr))
        # which SYMBOL carries the error: the antennas are a Hadamard transform of the symbols, so transform the difference back
        fft_bins = BINS[bins_bad] % 256
        print('     FFT bins of the wrong data bins: %s' % fft_bins.tolist())
        E = (H.T @ dz) if H.shape == (nt, nt) else None          # [symbol][bin]
        pw = np.abs(E).sum(axis=1); s0 = int(np.argmax(pw))
        print('     error energy per symbol: top %s (of total %.3g)' % ([(int(i), float(np.round(pw[i], 3))) for i in np.argsort(-pw)[:3]], pw.sum()))
        F = spectra((p, r))
        e0 = E[s0][bins_bad]                                       # error of symbol s0's spectrum at the wrong bins, in output scaling
        # output scaling: ref = c * H^T-ish combination; estimate c from the whole item:  R = H.T @ ref  ~ c * F[:, BINS]
        rz = rre.astype(np.float64) + 1j * rim.astype(np.float64)
        R = H.T @ rz
        for name, Fm in (('F', F[:, BINS % 256]), ('conj F', np.conj(F[:, BINS % 256]))):
            c = np.vdot(Fm, R) / np.vdot(Fm, Fm)
            print('     scale fit against %s: c = %.4g%+.4gj, residual %.3g' % (name, c.real, c.imag, np.abs(R - c * Fm).max()))
        Fm = F[:, BINS % 256]; c = np.vdot(Fm, R) / np.vdot(Fm, Fm)
        true0 = c * Fm[s0][bins_bad]
        print('     symbol %d at the wrong bins: |true| rms %.3g, |error| rms %.3g, error/true median %.3g' % (
            s0, np.sqrt(np.mean(np.abs(true0) ** 2)), np.sqrt(np.mean(np.abs(e0) ** 2)), np.median(np.abs(e0) / np.abs(true0))))
        # candidates for what was used instead of symbol s0's spectrum
        cands = {'zero': 0 * true0}
        for (pp, rr, tag) in [(p, (r + k) % nr, 'rx+%d' % k) for k in range(1, nr)]:
            cands['same symbol, ' + tag] = c * spectra((pp, rr))[s0][BINS % 256][bins_bad]
        for ds_ in (-8, -1, 1, 8, 16, 24):
            if 0 <= s0 + ds_ < nt: cands['symbol %+d, same item' % ds_] = c * Fm[s0 + ds_][bins_bad]
        for k in (1, 2, 3, 4, 8, 16, 32, 64, 128, 256, 512):
            for sg in (-1, 1):
                ib = p * nr + r + sg * k
                if 0 <= ib < n * nr: cands['same symbol, item %+d' % (sg * k)] = c * spectra((ib // nr, ib % nr))[s0][BINS % 256][bins_bad]
        res = sorted(((np.abs(true0 + e0 - v).max(), k) for k, v in cands.items()))
        print('     what stood in for it: best %s' % [(k, float(np.round(e, 4))) for e, k in res[:3]])
        # ratio pattern
        with np.errstate(all='ignore'):
            q = (gre[ants] / rre[ants]).ravel(); q = q[np.isfinite(q)]
        print('     got/ref (re) quantiles: %s' % np.round(np.quantile(q, [0.05, 0.25, 0.5, 0.75, 0.95]), 3).tolist())
    seen += 1
    if seen >= 3: break
print('RESULT', 'no bad call' if not seen else 'bad calls analysed: %d' % seen)
