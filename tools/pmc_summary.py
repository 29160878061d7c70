#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite file: per kernel, mean of every counter and mean duration."""
import collections
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    try:
        for name, cn, val, d in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            agg[name][cn].append(val)
            dur[name].append(d)
    except sqlite3.OperationalError:
        pass
    if not agg:
        for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            print('%-90s calls %6d  total %12.1f us  avg %10.2f us  %5.1f %%' % (r[0][:90], r[1], r[2] / 1e3 if r[2] > 1e7 else r[2], r[3] / 1e3 if r[3] > 1e6 else r[3], r[4]))
        return
    for k, v in agg.items():
        print(k[:110], ' n=%d' % (len(next(iter(v.values())))), ' avg_dur_us=%.1f' % (sum(dur[k]) / len(dur[k]) / 1e3))
        for cn, vals in sorted(v.items()):
            print('    %-28s %16.1f' % (cn, sum(vals) / len(vals)))


if __name__ == '__main__':
    main(sys.argv[1])
