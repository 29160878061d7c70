#!/usr/bin/env python3
"""rocprofv3 --memory-copy-trace --kernel-trace CSVs of tools/c128_trace.py -> how the host pipeline's copies overlap:
per direction: bytes, busy time (union of intervals), GB/s while busy; time both directions are busy at once; kernel busy time;
over the LAST call (the longest gap in the copy stream separates the calls)."""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def both(a, b):
    ev = [(s, 1, 0) for s, e in a] + [(e, -1, 0) for s, e in a] + [(s, 0, 1) for s, e in b] + [(e, 0, -1) for s, e in b]
    ev.sort()
    na = nb = 0
    last, tot = None, 0
    for t, da, db in ev:
        if last is not None and na > 0 and nb > 0:
            tot += t - last
        na += da
        nb += db
        last = t
    return tot


def main(d):
    import glob
    cp = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)[0]
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    copies = []
    with open(cp) as f:
        for r in csv.DictReader(f):
            copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'], r.get('Bytes') or r.get('Size') or '0'))
    copies.sort()
    # last call = after the longest gap among the big copies
    big = [c for c in copies if (c[1] - c[0]) > 50000]
    gaps = [(big[i + 1][0] - big[i][1], i) for i in range(len(big) - 1)]
    cut = big[max(gaps)[1] + 1][0] if gaps else 0
    last = [c for c in copies if c[0] >= cut]
    t0, t1 = min(c[0] for c in last), max(c[1] for c in last)
    h2d = [(s, e) for s, e, d_, _ in last if 'HOST_TO_DEVICE' in d_.upper() or 'H2D' in d_.upper()]
    d2h = [(s, e) for s, e, d_, _ in last if 'DEVICE_TO_HOST' in d_.upper() or 'D2H' in d_.upper()]
    nb = lambda sel: sum(int(b) for s, e, d_, b in last if (s, e) in set(sel))
    kern = []
    with open(kt) as f:
        for r in csv.DictReader(f):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if s >= t0 - 2000000 and e <= t1 + 2000000:
                kern.append((s, e))
    print('last call: span %.2f ms (first copy start -> last copy end)' % ((t1 - t0) / 1e6))
    for name, iv in (('H2D', h2d), ('D2H', d2h)):
        by, bz = nb(iv), union(iv)
        print('  %s: %3d copies, %.3f GB, busy %.2f ms, %.1f GB/s while busy' % (name, len(iv), by / 1e9, bz / 1e6, by / max(bz, 1)))
    print('  both directions busy at once: %.2f ms' % (both(h2d, d2h) / 1e6))
    print('  kernels: %d launches, busy %.2f ms; kernels beside a copy: %.2f ms' % (len(kern), union(kern) / 1e6, both(kern, h2d + d2h) / 1e6))
    print('  no copy in flight: %.2f ms of the span' % (((t1 - t0) - union(h2d + d2h)) / 1e6))


if __name__ == '__main__':
    main(sys.argv[1])
