#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (csi_ls_estimate + csi_predict) with
pre-allocated, pre-touched caller buffers: pageable numpy arrays, and pinned ones
(csi_host_malloc -> detected by the library, DMA'd directly).  GPU box:
    python tools/hostpath_probe.py [--packets 4000] [--threads 0]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl_channel_estimation_mamimo_amd as pkg   # noqa: E402


def fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--packets', type=int, default=4000)
    ap.add_argument('--nt', type=int, default=32)
    ap.add_argument('--nr', type=int, default=4)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    nt, nr, npkt = a.nt, a.nr, a.packets
    rng = np.random.default_rng(0)
    e = pkg.CsiEngine(nt, nr, hidden=(1024, 1024))
    w = pkg.synth.make_weights(rng, nt, (1024, 1024))
    e.load_weights('real', w)
    e.load_weights('imag', w)
    e.set_pilot(pkg.synth.hadamard(nt))
    if a.threads:
        e.set_option('host_threads', a.threads)
    lib, ctx = e._lib, e._ctx

    def buffers(pinned):
        shp_i, shp_o = (npkt, nr, 320 * nt), (npkt, nr, nt, 234)
        if pinned:
            mk = lambda s: e.pinned_empty(s)
        else:
            mk = lambda s: np.empty(s, np.float32)
        re, im, ore, oim = mk(shp_i), mk(shp_i), mk(shp_o), mk(shp_o)
        re[...] = rng.standard_normal(shp_i[1:], dtype=np.float32)
        im[...] = re
        ore[...] = 0
        oim[...] = 0
        return re, im, ore, oim

    for pinned in (False, True):
        re, im, ore, oim = buffers(pinned)
        gb = (re.nbytes + im.nbytes) * 2 + (ore.nbytes + oim.nbytes) * 2
        for name, fn in (('csi_predict', lib.csi_predict), ('csi_ls_estimate', lib.csi_ls_estimate)):
            fn(ctx, fp(re), fp(im), npkt, fp(ore), fp(oim))
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                rc = fn(ctx, fp(re), fp(im), npkt, fp(ore), fp(oim))
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            t = min(ts)
            moved = re.nbytes + im.nbytes + ore.nbytes + oim.nbytes
            print(f'{"pinned  " if pinned else "pageable"} {name:16s} {t * 1e3:8.2f} ms  {npkt * nr * nt / t / 1e6:7.2f} M pairs/s  '
                  f'{moved / t / 1e9:6.1f} GB/s over PCIe (both directions)')
    # the deployment surface: complex128 numpy batch in, complex64 numpy estimates out (csi_estimate_c128), by staging threads
    x = np.empty((npkt, nr, 320 * nt), np.complex128)
    x.real = rng.standard_normal(x.shape[1:])
    x.imag = x.real
    dnn = np.zeros((npkt, nr, nt, 234), np.complex64)
    ls = np.zeros((npkt, nr, nt, 234), np.complex64)
    up, down = x.nbytes // 2, dnn.nbytes
    a_ms, b_ms, ab_ms = e.pcie_probe(up, down)
    print(f'link: {up / 1e9:.2f} GB up alone {a_ms:.2f} ms ({up / a_ms / 1e6:.1f} GB/s), {down / 1e9:.2f} GB down alone {b_ms:.2f} ms ({down / b_ms / 1e6:.1f} GB/s), both at once {ab_ms:.2f} ms (os.cpu_count() = {os.cpu_count()})')
    try:
        print('cgroup cpu.max:', open('/sys/fs/cgroup/cpu.max').read().strip(), '| affinity:', len(os.sched_getaffinity(0)), 'cpus')
    except OSError as err:
        print('cgroup cpu.max: n/a', err)
    combos = [(a.threads, 0, 1)] if a.threads else [(0, 0, 1), (0, 0, 0), (16, 0, 1), (16, 0, 0), (32, 0, 1), (32, 0, 0), (0, 0, 1), (0, 0, 0), (32, 256, 1), (32, 256, 0), (32, 128, 1), (32, 128, 0)]
    for threads, chunk, side in combos:
        e.set_option('hp_side_threads', side)
        e.set_option('host_threads', threads)
        e.set_option('hp_chunk_packets', chunk)
        for what, kw in (('DNN + LS', dict(out=(dnn, ls))), ('DNN only', dict(ls=False, out=(dnn, None)))):
            e.estimate(x, **kw)
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                e.estimate(x, **kw)
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            moved = x.nbytes // 2 + dnn.nbytes + (ls.nbytes if 'LS' in what else 0)
            us = {k: e.get_option('hp_' + k + '_us') / 1e3 for k in ('total', 'stage', 'wait_stage', 'wait_out', 'weave')}
            print(f'c128 -> c64 {what:9s} host_threads={threads:2d} chunk={chunk:3d} side_threads={side} {t * 1e3:8.2f} ms  {npkt * nr * nt / t / 1e6:7.2f} M pairs/s  '
                  f'{moved / t / 1e9:6.1f} GB/s over PCIe (both directions) | last call: total {us["total"]:.1f} ms, stager busy {us["stage"]:.1f}, '
                  f'main waits: staged chunk {us["wait_stage"]:.1f}, download {us["wait_out"]:.1f}, weave {us["weave"]:.1f}')


if __name__ == '__main__':
    main()
