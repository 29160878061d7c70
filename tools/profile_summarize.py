#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by tools/profile_config.sh (rounds 1-4: tools/profile_round.sh, its headline-only predecessor) into the committed summaries under
profiles/:  <tag>_kernel_stats.txt (from the kernel trace), <tag>_pmc.txt (per-kernel mean of every
counter, one pass per counter group) and <tag>_traffic.json (HBM bytes per launch of each kernel,
read by bench.py for roofline.traffic).

Launches are bucketed by (kernel name, total grid size): the same kernel serves calls of very different
sizes (a 4000-packet step, a 2-packet parity check), and a mean over such a mix says nothing about
either.  Every bucket is listed; the FULL-SIZE bucket of a kernel (its largest grid, ties -> most
launches) is the one that goes into <tag>_traffic.json and is marked '*'.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in
KiB and come from separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so the read side is doubled:  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import os
import sys


def short(name):
    return name.replace('void csi::', '').replace('csi::', '').split('(')[0]


def full_size_bucket(buckets):
    """buckets: {grid: n_launches} of one kernel -> the grid of its full-size launches."""
    return max(buckets, key=lambda g: (g, buckets[g]))


def read_trace(path):
    """kernel trace -> {kernel: {grid: [durations ns]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            out[short(r['Kernel_Name'])][grid].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return out


def read_pmc(path):
    """counter collection -> {kernel: {grid: {counter: [values]}}}, {kernel: {grid: [durations]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    dur = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg, dur
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k, grid = short(r['Kernel_Name']), int(r['Grid_Size'])
            agg[k][grid][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                dur[k][grid].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg, dur


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main(src, tag, dst='profiles', cmd='python bench.py --steps 20 --warmup 5 --no-cpu-baseline --check 0 --no-latency --host-path 0 --no-other-configs --no-next-rows'):
    os.makedirs(dst, exist_ok=True)
    trace = read_trace(os.path.join(src, 'kt', 'kt_kernel_trace.csv'))
    total = sum(sum(d) for k in trace for d in trace[k].values())
    lines = [f'# rocprofv3 --kernel-trace --stats -- {cmd}',
             '# one row per (kernel, grid size) bucket; * = the full-size bucket of that kernel',
             '# %-58s %10s %6s %14s %12s %7s %12s %12s' % ('kernel', 'grid', 'calls', 'total_us', 'avg_us', 'pct', 'min_us', 'max_us')]
    rows = []
    for k in trace:
        if k.startswith('__amd'):
            continue
        full = full_size_bucket({g: len(d) for g, d in trace[k].items()})
        for g, d in trace[k].items():
            rows.append((sum(d), k, g, d, g == full))
    for tot, k, g, d, is_full in sorted(rows, reverse=True):
        lines.append('%-60s %10d %6d %14.1f %12.2f %7.2f %12.2f %12.2f' % (
            ('*' if is_full else ' ') + k[:59], g, len(d), tot / 1e3, mean(d) / 1e3, 100.0 * tot / max(total, 1), min(d) / 1e3, max(d) / 1e3))
    open(os.path.join(dst, f'{tag}_kernel_stats.txt'), 'w').write('\n'.join(lines) + '\n')

    out = ['# per-(kernel, grid) MEAN counter value per launch; one rocprofv3 --pmc pass per group (never combined with traces)',
           '# * = the full-size bucket of that kernel (the one in %s_traffic.json)' % tag]
    fetch, write, launches = {}, {}, {}
    for grp in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_l2'):
        agg, dur = read_pmc(os.path.join(src, grp, 'pmc_counter_collection.csv'))
        out.append(f'\n## pass {grp}')
        for k in sorted(agg):
            if k.startswith('__amd'):
                continue
            full = full_size_bucket({g: len(dur[k][g]) for g in agg[k]})
            for g in sorted(agg[k], reverse=True):
                a, d = agg[k][g], dur[k][g]
                out.append('%s%-44s grid=%d launches=%d avg_dur_us=%.1f' % ('*' if g == full else ' ', k[:44], g, len(d), mean(d) / 1e3))
                for cn in sorted(a):
                    v = mean(a[cn])
                    out.append('    %-28s %18.1f' % (cn, v))
                    if g == full and cn == 'FETCH_SIZE':
                        fetch[k], launches[k] = v, (g, len(d))
                    if g == full and cn == 'WRITE_SIZE':
                        write[k], launches[k] = v, (g, len(d))
                if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_BUSY_CYCLES' in a:
                    mf, bz = mean(a['SQ_VALU_MFMA_BUSY_CYCLES']), mean(a['SQ_BUSY_CYCLES'])
                    # SQ_BUSY_CYCLES is summed over the 32 shader engines, MFMA busy over the 1024 SIMDs
                    out.append('    %-28s %18.3f' % ('=> MFMA pipe utilisation', (mf / 1024.0) / (bz / 32.0) if bz else 0.0))
                    out.append('    %-28s %18.3f' % ('=> clock GHz (profiled)', (bz / 32.0) / mean(d) if d else 0.0))
                if 'TCC_HIT_sum' in a and 'TCC_MISS_sum' in a:
                    h, m = sum(a['TCC_HIT_sum']), sum(a['TCC_MISS_sum'])
                    out.append('    %-28s %18.3f' % ('=> L2 hit rate', h / (h + m) if h + m else 0.0))
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        fb, wb = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        traffic[k] = dict(grid=launches[k][0], launches=launches[k][1], fetch_size_kib=fetch.get(k), write_size_kib=write.get(k),
                          hbm_bytes_per_launch=2.0 * fb + wb, correction='(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 read-side x2')
    open(os.path.join(dst, f'{tag}_pmc.txt'), 'w').write('\n'.join(out) + '\n')
    json.dump(traffic, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)
    print('\n'.join(lines))
    print(json.dumps(traffic, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:])
