#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by tools/profile_round.sh into the committed summaries under
profiles/:  <tag>_kernel_stats.txt (kernel-trace --stats), <tag>_pmc.txt (per-kernel mean of every
counter, one pass per counter group) and <tag>_traffic.json (HBM bytes per launch of each kernel,
read by bench.py for roofline.traffic).

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


def read_pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg, dur
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg, dur


def main(src, tag, dst='profiles'):
    os.makedirs(dst, exist_ok=True)
    lines = ['# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0',
             '# %-60s %6s %14s %12s %7s %12s %12s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct', 'min_us', 'max_us')]
    with open(os.path.join(src, 'kt', 'kt_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            lines.append('%-62s %6d %14.1f %12.2f %7.2f %12.2f %12.2f' % (
                short(r['Name'])[:62], int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3,
                float(r['Percentage']), float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    open(os.path.join(dst, f'{tag}_kernel_stats.txt'), 'w').write('\n'.join(lines) + '\n')

    out = ['# per-kernel MEAN counter value per launch; one rocprofv3 --pmc pass per group (never combined with traces)']
    traffic = {}
    fetch, write = {}, {}
    for grp in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_l2'):
        agg, dur = read_pmc(os.path.join(src, grp, 'pmc_counter_collection.csv'))
        out.append(f'\n## pass {grp}')
        for k in sorted(agg):
            if k.startswith('__amd'):
                continue
            n = len(next(iter(agg[k].values())))
            out.append('%-44s launches=%d avg_dur_us=%.1f' % (k[:44], n, sum(dur[k]) / len(dur[k]) / 1e3))
            for cn in sorted(agg[k]):
                v = sum(agg[k][cn]) / len(agg[k][cn])
                out.append('    %-28s %18.1f' % (cn, v))
                if cn == 'FETCH_SIZE':
                    fetch[k] = v
                if cn == 'WRITE_SIZE':
                    write[k] = v
            a = agg[k]
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_BUSY_CYCLES' in a:
                mf = sum(a['SQ_VALU_MFMA_BUSY_CYCLES']) / len(a['SQ_VALU_MFMA_BUSY_CYCLES'])
                bz = sum(a['SQ_BUSY_CYCLES']) / len(a['SQ_BUSY_CYCLES'])
                # SQ_BUSY_CYCLES is summed over the 32 shader engines, MFMA busy over the 1024 SIMDs
                out.append('    %-28s %18.3f' % ('=> MFMA pipe utilisation', (mf / 1024.0) / (bz / 32.0) if bz else 0.0))
                out.append('    %-28s %18.3f' % ('=> clock GHz (profiled)', (bz / 32.0) / (sum(dur[k]) / len(dur[k])) if dur[k] else 0.0))
            if 'TCC_HIT_sum' in a and 'TCC_MISS_sum' in a:
                h = sum(a['TCC_HIT_sum']); m = sum(a['TCC_MISS_sum'])
                out.append('    %-28s %18.3f' % ('=> L2 hit rate', h / (h + m) if h + m else 0.0))
    for k in sorted(set(fetch) | set(write)):
        fb, wb = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        traffic[k] = dict(fetch_size_kib=fetch.get(k), write_size_kib=write.get(k),
                          hbm_bytes_per_launch=2.0 * fb + wb, correction='(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 read-side x2')
    open(os.path.join(dst, f'{tag}_pmc.txt'), 'w').write('\n'.join(out) + '\n')
    json.dump(traffic, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)
    print('\n'.join(lines))
    print('\n'.join(out))
    print(json.dumps(traffic, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
