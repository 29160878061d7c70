#!/usr/bin/env python3
"""Socket power and shader clock while ONE kernel runs back to back (round-3 verdict, "pin down the ceiling you divide by").

For every target a child process loops the kernel for `--secs` seconds; after a warm-up this script takes `--samples` readings
0.4 s apart of the socket power and the shader clock - from sysfs (hwmon power1_average / power1_input, pp_dpm_sclk) and from
`rocm-smi --showpower --showclocks` as a cross-check - and prints them with the child's own ms / launch and TFLOP/s.

    python tools/power_probe.py [--secs 7] [--samples 10] [--targets band8,skeleton,skeleton_rnd,layer0,hs2hs,blas_band,blas_8192]

Needs tools/band_probe.bin, tools/hs_probe.bin (hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/<x>.hip -o tools/<x>.bin) and the
all-variants code object of the band kernel (tools/build_band8.sh -> tools/band8.hsaco)."""
import argparse
import glob
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

BLAS_LOOP = r'''
import sys, time, torch
m, n, k, secs = (int(v) for v in sys.argv[1:5])
a = torch.relu(torch.randn(m, k, device='cuda')).half()
b = (torch.randn(k, n, device='cuda') * 0.03).half()
c = a @ b
torch.cuda.synchronize()
print('LOOP START', flush=True)
t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        c = a @ b
    torch.cuda.synchronize(); it += 20
dt = time.perf_counter() - t0
print('loop blas f16 M=%d N=%d K=%d (relu operand): %d launches, %.3f ms each, %.0f TF f16 executed' % (m, n, k, it, dt / it * 1e3, 2.0 * m * n * k * it / dt / 1e12))
'''


def sysfs_sample():
    out = {}
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        for name in ('power1_average', 'power1_input'):
            for f in glob.glob(os.path.join(card, 'hwmon', 'hwmon*', name)):
                try:
                    out.setdefault('power_w', float(open(f).read()) / 1e6)
                except (OSError, ValueError):
                    pass
        try:
            for ln in open(os.path.join(card, 'pp_dpm_sclk')):
                if '*' in ln:
                    m = re.search(r'(\d+)\s*Mhz', ln, re.I)
                    if m:
                        out.setdefault('sclk_mhz', float(m.group(1)))
        except OSError:
            pass
        if out:
            break
    return out


def smi_sample():
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=20)
    except (OSError, subprocess.TimeoutExpired):
        return {}
    out = {}
    for ln in r.stdout.splitlines():
        if 'Power' in ln and 'W' in ln:
            m = re.search(r':\s*([0-9.]+)\s*$', ln.strip()) or re.search(r'([0-9.]+)\s*W?\s*$', ln.strip())
            if m:
                out.setdefault('smi_power_w', float(m.group(1)))
        if 'sclk' in ln:
            m = re.search(r'\((\d+)\s*Mhz\)', ln, re.I)
            if m:
                out.setdefault('smi_sclk_mhz', float(m.group(1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--secs', type=float, default=7.0)
    ap.add_argument('--samples', type=int, default=10)
    ap.add_argument('--targets', default='idle,band8,skeleton,skeleton_rnd,layer0,hs2hs,mfma,blas_band,blas_8192')
    ap.add_argument('--hsaco', default=os.path.join(HERE, 'band8.hsaco'))
    args = ap.parse_args()
    env = dict(os.environ, BAND8_HSACO=args.hsaco)
    secs = str(args.secs)
    bp, hp = os.path.join(HERE, 'band_probe.bin'), os.path.join(HERE, 'hs_probe.bin')
    cmds = {
        'idle': None,
        'band8': [bp, 'loop', '32', 'csi_band8', secs],
        'skeleton': [bp, 'loop', '32', 'csi_band8_skeleton', secs],
        'skeleton_rnd': [bp, 'loop', '32', 'csi_band8_skeleton_rnd', secs],
        'band8_noaside': [bp, 'loop', '32', 'csi_band8_noaside', secs],
        'layer0': [hp, 'loop', '1', 'relu', 'cast', secs],
        'hs2hs': [hp, 'loop', '1', 'relu', 'hs2hs', secs],
        'mfma': [hp, 'loop', '1', 'relu', 'mfma', secs],
        'pair': [hp, 'loop', '1', 'relu', 'pair', secs],
        'blas_band': [sys.executable, '-c', BLAS_LOOP, '262144', '1024', '1024', str(int(args.secs))],
        'blas_8192': [sys.executable, '-c', BLAS_LOOP, '8192', '8192', '8192', str(int(args.secs))],
    }
    print('target            power W (sysfs)         sclk MHz (sysfs)        power W (rocm-smi)      sclk MHz (rocm-smi)     child')
    for t in args.targets.split(','):
        cmd = cmds[t]
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, bufsize=1) if cmd else None
        head = []
        if child:                       # the child says when its set-up is over and the loop runs
            for ln in child.stdout:
                head.append(ln.rstrip())
                if 'LOOP START' in ln:
                    break
        time.sleep(1.5)                 # clocks and the power controller settled
        rows = []
        for _ in range(args.samples):
            s = sysfs_sample()
            s.update(smi_sample())
            rows.append(s)
            time.sleep(0.4)
            if child and child.poll() is not None:
                break
        tail = ''
        if child:
            out, _ = child.communicate()
            tail = ((out.strip().splitlines() or head) or [''])[-1]

        def stat(key):
            v = [r[key] for r in rows if key in r]
            return '%7.1f (%6.1f..%6.1f) n=%d' % (sum(v) / len(v), min(v), max(v), len(v)) if v else 'n/a'.ljust(22)
        print('%-16s  %-22s  %-22s  %-22s  %-22s  %s' % (t, stat('power_w'), stat('sclk_mhz'), stat('smi_power_w'), stat('smi_sclk_mhz'), tail), flush=True)


if __name__ == '__main__':
    main()
