#!/usr/bin/env python3
"""bench.py - channel-estimates/s of the hot path (LS + DNN real + DNN imag -> complex CSI) on
MI355X.  One "step" = one pass of the whole path over one batch of synthetic packets that is
already resident in HBM.

Workload (N = 1): BASELINE.json configs[1] - Nt=32, Nr=4, 500 packets at each of the 8 SNR levels
{-25..10 dB} = 4000 packets = 512 000 pair-channels per step, shipped model FC 1024x1024 + BN
(full_pipeline_maMIMO_DNNEst.sh:40,47), fp32.  For N > 1 every rank runs the same per-GPU
workload on its own packet shard (weak scaling); the weights are broadcast once from rank 0
over RCCL before the timed region and there is no collective in the data path.

Data: the batch the pipeline feeds config 2 with - 500 structured sounding packets (8-tap channels,
reference amplitude scaling) at EACH of the 8 SNR levels, all 4000 in one launch (synth.mixed_snr_batch;
`--input white` = i.i.d. CN(0,1) generated on the device, the default for batches beyond 8000 packets) -
and random-initialised weights of the shipped architecture: the reference ships neither datasets nor weights.

Ranks: `--gpus N` under torchrun (WORLD_SIZE set) uses the ranks it is given; WITHOUT torchrun it starts
N ranks itself (one process per GPU, RCCL; CSI_DIST_BACKEND=gloo lets ranks share a GPU for a dry
run).  `n_gpus` in the line is the number of ranks that executed the step.  `--scaling weak` (default):
`--packets` per GPU; `--scaling strong`: `--packets` in total, sharded by contiguous packet ranges.

Arithmetic: fp32 in, fp32 accumulate, fp32 out.  The library's fp32 contexts run GEMMs that fill the
chip on the f16 matrix cores with split operands (x = hi + lo halves, 3 MFMA per product,
gemm_hs.hip.h): same 1e-5 contract (the line carries the measured error against the fp64 oracle),
~2.6x the fp32 MFMA rate.  `--engine native` times the fp32 MFMA kernels instead; the default run
also reports them as `native_fp32_engine` for comparison.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - dominant kernel (pair_dense_gemm) TFLOP/s from HIP events on the library's own
                 stream: algorithmic flops / time against the ceiling of the engine that ran -
                 2500 / 3 TFLOP/s for the split engine (f16 dense MFMA peak over the three products
                 per fp32 multiply), 157.3 TFLOP/s for the fp32 MFMA kernels
  cpu_baseline - the reference's per-packet naive fp32 loop restated on the host cores
                 (oracle/cpu_baseline.py), timed on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0
BF16_MATRIX_PEAK_TFLOPS = 2500.0    # dense, v_mfma_f32_32x32x16_bf16 (the same for v_mfma_f32_32x32x16_f16)
SPLIT_PRODUCTS = 3                  # f16 MFMAs per fp32-grade multiply on the split engine


def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start N ranks of this script (one process per GPU,
    LOCAL_RANK i -> device i) with a private rendezvous on 127.0.0.1, forward rank 0's line.  The port is found by
    bind-and-close, which another process can win before rank 0 binds it again: a launch whose rank 0 dies WITHOUT having
    printed a line and with an address-in-use / connection text on stderr is started once more on a fresh port, the
    first failure going to stderr (it is the launcher that retries a rendezvous, never a measurement)."""
    import socket
    import subprocess
    import uuid
    for attempt in range(3):
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        token = uuid.uuid4().hex                    # per-launch: dist.exchange_unique_id never takes a leftover of another launch
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', CSI_RCCL_ID_TOKEN=token)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None))
        out, err = procs[0].communicate()
        rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
        err = err.decode(errors='replace')
        sys.stderr.write(err)
        rendezvous_failed = any(rcs) and not out.strip() and any(t in err for t in ('EADDRINUSE', 'ddress already in use', 'Connection refused', 'connect() timed out', 'The server socket has failed'))
        if rendezvous_failed and attempt < 2:
            print('bench.py: rendezvous on 127.0.0.1:%d failed (rank exit codes %s); starting the ranks again on a fresh port' % (port, rcs), file=sys.stderr)
            continue
        break
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        sys.exit('bench.py: rank exit codes %s' % rcs)


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--nt', type=int, default=32)
    ap.add_argument('--nr', type=int, default=4)
    ap.add_argument('--packets', type=int, default=4000, help='packets per step: per GPU (--scaling weak) or in total (--scaling strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--input', default='auto', choices=['auto', 'mixed-snr', 'white'],
                    help='mixed-snr = structured packets, packets/8 at each of {-25..10} dB in one launch (host-generated); '
                         'white = CN(0,1) generated on the device; auto = mixed-snr up to 8000 packets per GPU')
    ap.add_argument('--no-latency', action='store_true', help='skip the one-packet latency loop (profiling runs)')
    ap.add_argument('--graph', action='store_true',
                    help='time replays of ONE hipGraph holding the whole step (csi_estimate_device + use_graph: LS, both DNNs, every chunk) - '
                         'BASELINE configs[4]; the per-kernel HIP-event breakdown then comes from one extra eager step outside the timed region')
    ap.add_argument('--hidden', type=int, nargs='+', default=[1024, 1024])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget-s', type=float, default=12.0)
    ap.add_argument('--no-ls', action='store_true', help='time the DNN only')
    ap.add_argument('--workspace-gb', type=float, default=0.0, help='activation workspace cap (0 = library default)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='f32 = the headline fp32 path; bf16 = BASELINE config 3 (use with --nt 64 --packets 5000)')
    ap.add_argument('--engine', default='auto', choices=['auto', 'native', 'split'],
                    help='fp32 contexts: auto = split-f16 engine for GEMMs that fill the chip (library default), '
                         'native = fp32 MFMA kernels only, split = split engine wherever the shapes allow')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=VALUE', help='csi_set_option before the timed region (A/B runs), repeatable')
    ap.add_argument('--check', type=int, default=8, help='packets checked against the oracle after timing (spread over the batch: with the mixed-SNR input one per SNR level)')
    ap.add_argument('--host-path', type=int, default=4000,
                    help='also time the host-buffer (PCIe-inclusive) entry points on this many packets (0 = skip; rank 0, N = 1)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='the default N = 1 run also measures BASELINE configs[2] and one GPU\'s share of configs[3] / configs[4] in fresh '
                         'processes after its own timed region and reports them under "other_configs" (never part of "value"); this skips them')
    ap.add_argument('--weights-via', default='rccl', choices=['rccl', 'torch'],
                    help='N > 1: rccl = rank 0 loads the weights into its context and the library broadcasts its device buffers '
                         '(csi_comm_init / csi_broadcast_weights: ncclBroadcast inside the C-ABI); torch = dist.broadcast_weights '
                         '(torch.distributed, host round trip).  CSI_DIST_BACKEND=gloo implies torch.')
    ap.add_argument('--no-next-rows', action='store_true', help='skip the "next_rows" legs (LMMSE smoother, one training step, LS on a non-Sylvester pilot) measured after the timed region')
    ap.add_argument('--legs-packets', default='', metavar='A,B',
                    help='N > 1 only: total packets of the configs[3] / configs[4] legs (default 50000,100000 = BASELINE.json; the CPU tests shrink them - and, given explicitly, the legs run beside any headline)')
    ap.add_argument('--no-regimes', action='store_true', help='skip the "regimes" leg (1 / 8 / 64 / 500-packet calls with their bounds) measured after the timed region')
    ap.add_argument('--full-line', action='store_true',
                    help='print the whole detail object as the line instead of the compact record (what the fresh-process legs of this script '
                         'ask of their children; by hand: bench_detail.json holds the same object)')
    ap.add_argument('--detail-file', default='', help='where the full detail object goes (default: bench_detail.json beside this script, and gpurun_out/ when that exists)')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='start the ranks, rendezvous, all-reduce a rank count and print it - no GPU work (checks the launch path on any host)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args.gpus)

    import dl_channel_estimation_mamimo_amd as pkg
    rank, world, local = pkg.dist.env_rank_world()
    backend = os.environ.get('CSI_DIST_BACKEND', 'nccl')
    if world > 1:
        # RCCL over xGMI; CSI_DIST_BACKEND=gloo lets several ranks share one GPU for a dry run
        pkg.dist.init_process_group(backend)
        assert pkg.dist.world_size() == world
    n_gpus = world                       # the ranks that execute the step - never the flag
    if args.rendezvous_only:
        lo, hi = pkg.dist.shard_range(args.packets, rank, world) if args.scaling == 'strong' else (rank * args.packets, (rank + 1) * args.packets)
        seen = pkg.dist.all_reduce_sum(1.0)
        pkts = pkg.dist.all_reduce_sum(float(hi - lo))
        infos = pkg.dist.gather_objects({'rank': rank, 'local_rank': local, 'pid': os.getpid(), 'packets': hi - lo, 'ms_per_step': None,
                                         'device': {'ordinal': None, 'name': 'none (rendezvous only)', 'pci': None, 'arch': None}})
        pkg.dist.barrier()
        if rank == 0:
            print(json.dumps({'rendezvous_only': True, 'n_gpus': n_gpus, 'ranks_seen': int(seen), 'requested': args.gpus,
                              'other_configs_scheduled': [dict(config=n_, flags=' '.join(f_), fits=fit_, per_rank_gb=round(gb_, 1))
                                                          for n_, _, f_, fit_, gb_ in multi_gpu_legs(args, world)] if world > 1 and (default_headline(args) or args.legs_packets) else [],
                              'packets_per_step': int(pkts), 'scaling': args.scaling, 'backend': backend if world > 1 else None,
                              'ranks_ms': [i['ms_per_step'] for i in infos], 'devices': [i['device'] for i in infos], 'ranks': infos}))
        return

    nt, nr, hidden = args.nt, args.nr, tuple(args.hidden)
    if args.scaling == 'strong':
        first, last = pkg.dist.shard_range(args.packets, rank, world)     # fixed total, contiguous packet ranges
    else:
        first, last = rank * args.packets, (rank + 1) * args.packets      # fixed work per GPU
    npkt = last - first
    total_pkts = args.packets if args.scaling == 'strong' else args.packets * world
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    if world > 1 and backend == 'nccl' and world > ndev:
        sys.exit('bench.py: %d ranks but %d GPUs visible (CSI_DIST_BACKEND=gloo shares a GPU for a dry run)' % (world, ndev))
    eng = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=234, use_bn=True, device=local % ndev, dtype=args.dtype,
                        workspace_bytes=int(args.workspace_gb * 2**30))

    # weights: created on rank 0, broadcast as one flat buffer (RCCL over xGMI when world > 1)
    wts = None
    if rank == 0:
        rng = np.random.default_rng(1234)
        wts = {'real': pkg.synth.make_weights(rng, nt, hidden), 'imag': pkg.synth.make_weights(rng, nt, hidden),
               'P': {'pilot': pkg.synth.hadamard(nt)}}
    via = 'single GPU'
    use_lib_rccl = world > 1 and args.weights_via == 'rccl' and backend == 'nccl'
    if use_lib_rccl:
        # can every rank load RCCL through the library at all?  (decided together: a rank that cannot must not leave the
        # others waiting inside ncclCommInitRank - then everybody takes the torch.distributed path)
        try:
            pkg.engine.get_unique_id()
            mine = 1.0
        except Exception as err:                       # noqa: BLE001
            print('bench.py: rank %d cannot use the library\'s RCCL path (%s)' % (rank, err), file=sys.stderr)
            mine = 0.0
        use_lib_rccl = pkg.dist.all_reduce_min(mine) > 0.5
    if use_lib_rccl:
        # the library's own communicator: rank 0 loads, every context receives the device buffers over RCCL.  An error on any rank
        # (reported by the library, not a hang) sends EVERY rank to the torch.distributed path below: the ranks agree afterwards.
        t_b = time.perf_counter()
        moved, ok = 0, 1.0
        try:
            eng.comm_init(rank, world, pkg.dist.exchange_unique_id(rank, world))
            if rank == 0:
                eng.load_weights('real', wts['real'])
                eng.load_weights('imag', wts['imag'])
                eng.set_pilot(wts['P']['pilot'])
            moved = eng.broadcast_weights(0)
        except Exception as err:                       # noqa: BLE001
            print('bench.py: rank %d: weight broadcast through the library failed (%s); torch.distributed path instead' % (rank, err), file=sys.stderr)
            ok = 0.0
        use_lib_rccl = pkg.dist.all_reduce_min(ok) > 0.5
        if not use_lib_rccl:
            try:
                eng.comm_destroy()
            except Exception:                          # noqa: BLE001
                pass
    if use_lib_rccl:
        via = 'csi_broadcast_weights: ncclBroadcast of %d device buffers, %.1f MB, %.0f ms incl. communicator setup and rank 0 load' % (
            eng.get_option('comm_blobs'), moved / 1e6, (time.perf_counter() - t_b) * 1e3)
        P_host = pkg.synth.hadamard(nt)             # the pilot matrix is a constant of the configuration (input synthesis below)
    else:
        if world > 1:
            wts = {k: pkg.dist.broadcast_weights(wts[k] if rank == 0 else None, src=0) for k in ('real', 'imag', 'P')}
            via = 'dist.broadcast_weights (torch.distributed, %s)' % backend
        eng.load_weights('real', wts['real'])
        eng.load_weights('imag', wts['imag'])
        eng.set_pilot(wts['P']['pilot'])
        P_host = wts['P']['pilot']

    # this rank's packet shard, resident in HBM before the timed region
    d_re, d_im = eng.empty((npkt, nr, eng.len_ltf)), eng.empty((npkt, nr, eng.len_ltf))
    mixed = args.input == 'mixed-snr' or (args.input == 'auto' and npkt <= 8000 and npkt % 8 == 0)
    if mixed:
        assert npkt % 8 == 0, 'mixed-snr input needs a multiple of 8 packets per rank'
        for p0, snr, blk in pkg.synth.mixed_snr_batch(2024 + 1 + 1000 * rank, nr, P_host, per_level=npkt // 8):
            d_re.upload(np.ascontiguousarray(blk.real), first=p0)
            d_im.upload(np.ascontiguousarray(blk.imag), first=p0)
    else:
        eng.synth_white(2024 + 1, first, npkt, d_re, d_im)
    d_ore, d_oim = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    d_hre, d_him = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    eng.synchronize()

    if args.dtype == 'f32':
        eng.set_option('f32_engine', {'auto': -1, 'native': 0, 'split': 1}[args.engine])
    for kv in args.option:
        name, value = kv.split('=')
        eng.set_option(name, int(value))

    # LS + DNN as one unit (csi_estimate_device) when the step is a hipGraph or when the LS kernel runs beside the DNN kernels
    one_unit = args.graph or (args.dtype == 'f32' and not args.no_ls and eng.get_option('ls_overlap_cus') > 0)

    def step():
        if one_unit:
            eng.estimate_device(d_re, d_im, npkt, d_ore, d_oim, d_hre, d_him)
            return
        if not args.no_ls:
            eng.ls_estimate_device(d_re, d_im, npkt, d_hre, d_him)
        eng.predict_device(d_re, d_im, npkt, d_ore, d_oim)

    if args.graph:
        assert not args.no_ls, '--graph times the whole step (LS + DNN)'
        eng.set_option('use_graph', 1)
        for _ in range(max(args.warmup, 4)):      # eager (allocates, which resets the graph cache), eager, capture, first replay
            step()
        eng.synchronize()
        assert eng.get_option('graph_replays') >= 1
    else:
        for _ in range(args.warmup):
            step()
        eng.synchronize()
        eng.profile_enable(True)                  # per-kernel HIP events on the library's stream (forces eager launches)
        eng.profile_reset()

    pkg.dist.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    dt_rank = time.perf_counter() - t0            # this rank's own steps (before waiting for the others)
    pkg.dist.barrier()
    dt = time.perf_counter() - t0
    dt = pkg.dist.all_reduce_max(dt)
    # per-rank evidence for the N > 1 line: who ran, on which device, how long its own steps took, and that its split
    # engine's range guard stayed silent
    infos = pkg.dist.gather_objects({'rank': rank, 'local_rank': local, 'pid': os.getpid(), 'packets': npkt,
                                     'ms_per_step': dt_rank / args.steps * 1e3, 'device': pkg.dist.device_identity(local % ndev),
                                     'hs_range_fallbacks': eng.get_option('hs_range_fallbacks') if args.dtype == 'f32' else None})
    if world > 1 and backend == 'nccl':
        ords = sorted(i['device']['ordinal'] for i in infos)
        assert pkg.dist.world_size() == world == len(infos), (pkg.dist.world_size(), world, len(infos))
        assert ords == list(range(world)), 'ranks did not land on %d distinct devices: %s' % (world, ords)

    graph_replays = eng.get_option('graph_replays') if args.graph else 0
    if args.graph:                                 # kernel breakdown: one eager step with events, outside the timed region
        eng.set_option('use_graph', 0)
        eng.profile_enable(True)
        eng.profile_reset()
        step()
        eng.synchronize()
    prof = eng.profile()
    eng.profile_enable(False)
    # the same steps once more WITHOUT the per-kernel events (round-3 verdict: the timed region carries them, so headline and
    # breakdown could not be told apart): what the event records between the kernels cost.  Rank-local, after the timed region.
    no_events = None
    if not args.graph:
        step(); step()
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        eng.synchronize()
        no_events = (time.perf_counter() - t1) / args.steps * 1e3
    pairs_per_step = total_pkts * nr * nt          # all ranks together
    value = pairs_per_step * args.steps / dt
    split_engine = args.dtype == 'f32' and eng.get_option('hs_launches') > 0
    band_any = eng.get_option('band_launches') > 0                                   # first per-pair layer + regressor as one kernel (hs_band)
    band_kernel = args.dtype == 'f32' and band_any
    # bf16 contexts: the register-blocked form (csrc/band4_kernel_gen.py) where it serves the call
    band4 = band_any and eng.get_option('band4') and eng.get_option('band4_available')
    band_name = ('csi_band4' if band4 else 'csi_band8') + ('' if band_kernel else '_bf16')
    # range guard of the split-f16 engine over the timed steps: a hit would have made eng.synchronize() raise
    # (CSI_ERR_RANGE) above; the counters go into the line
    guard = {'hs_launches': eng.get_option('hs_launches'), 'hs_range_fallbacks': eng.get_option('hs_range_fallbacks'),
             'csi_synchronize': 'clean'} if args.dtype == 'f32' else None

    # ---- host-buffer (PCIe-inclusive) entry points: measured first, before the CPU legs below start their BLAS / OpenMP thread
    # pools and fragment host memory (measured on one box: 50 ms here, 60 ms behind the oracle check)
    host_path = None
    if rank == 0 and world == 1 and args.host_path > 0:
        k = min(args.host_path, npkt)
        h_re, h_im = d_re.download(0, k), d_im.download(0, k)
        o_ls = (np.zeros((k, nr, nt, 234), np.float32), np.zeros((k, nr, nt, 234), np.float32))
        o_nn = (np.zeros((k, nr, nt, 234), np.float32), np.zeros((k, nr, nt, 234), np.float32))
        eng.predict(h_re, h_im, out=o_nn); eng.ls_estimate(h_re, h_im, out=o_ls)          # warm-up (staging slots)
        t0 = time.perf_counter()
        eng.ls_estimate(h_re, h_im, out=o_ls)
        eng.predict(h_re, h_im, out=o_nn)
        t1 = time.perf_counter() - t0
        host_path = {'packets': k, 'pairs_per_s': k * nr * nt / t1, 'ms': t1 * 1e3,
                     'note': 'csi_ls_estimate + csi_predict on pre-allocated pageable host buffers: staging + H2D + kernels + D2H, pipelined over packet chunks'}
        # the deployment surface (inference.py:24-32): complex128 batch in, complex64 estimates out, through the Python
        # wrapper - csi_estimate_c128: ONE upload for both estimators, split / interleave inside the staging copies
        x128 = np.empty(h_re.shape, np.complex128)
        x128.real, x128.imag = h_re, h_im
        bufs = (np.zeros((k, nr, nt, 234), np.complex64), np.zeros((k, nr, nt, 234), np.complex64))   # touched, like o_ls / o_nn above
        eng.estimate(x128, out=bufs)                                                       # warm-up (staging slots)
        t2s = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.estimate(x128, out=bufs)
            t2s.append(time.perf_counter() - t0)
        t2 = sorted(t2s)[1]
        host_path['python_c128_to_c64'] = {'pairs_per_s': k * nr * nt / t2, 'ms': t2 * 1e3, 'ms_all': [round(t * 1e3, 2) for t in t2s],
                                           'd2h_bytes': int(bufs[0].nbytes + bufs[1].nbytes),
                                           'note': 'CsiEngine.estimate (LS + DNN) on a complex128 numpy batch, complex64 numpy results; median of 3 calls. '
                                                   'Bound: the download of the two result arrays over one PCIe direction (~50 GB/s measured: '
                                                   'profiles/r03_c128_copy_trace.txt)'}
        # the reference's inference.py:24-32 returns the DNN estimate only: the same call without the LS array (half the download)
        t3s = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.estimate(x128, ls=False, out=(bufs[0], None))
            t3s.append(time.perf_counter() - t0)
        t3 = sorted(t3s)[1]
        host_path['python_c128_to_c64']['dnn_only'] = {'pairs_per_s': k * nr * nt / t3, 'ms': t3 * 1e3, 'ms_all': [round(t * 1e3, 2) for t in t3s],
                                                       'd2h_bytes': int(bufs[0].nbytes),
                                                       'note': 'CsiEngine.estimate(ls=False): what CSIPredictor.inference returns (model_real + 1j * model_imag).  Between this and the link\'s floor '
                                                               '(pcie_bound_ms) lie the two host passes a pageable complex128 / complex64 pair needs (split into float32 planes in pinned '
                                                               'staging, weave of the result planes): DRAM-bound loops on the CPUs the container grants (host_cpu_quota)'}
        # the plane entry point with buffers the CALLER has pinned (csi_host_malloc): no staging pass on the host at all
        try:
            pr, pi = eng.pinned_empty(h_re.shape), eng.pinned_empty(h_im.shape)
            po = (eng.pinned_empty(o_nn[0].shape), eng.pinned_empty(o_nn[1].shape))
            pr[...] = h_re; pi[...] = h_im
            eng.predict(pr, pi, out=po)
            tp = []
            for _ in range(3):
                t0 = time.perf_counter()
                eng.predict(pr, pi, out=po)
                tp.append(time.perf_counter() - t0)
            tpm = sorted(tp)[1]
            host_path['pinned_planes_dnn'] = {'ms': tpm * 1e3, 'pairs_per_s': k * nr * nt / tpm, 'ms_all': [round(t * 1e3, 2) for t in tp],
                                              'note': 'csi_predict on float32 planes in caller-pinned memory (csi_host_malloc): H2D / D2H straight from / into them'}
            del pr, pi, po
        except Exception as e:
            host_path['pinned_planes_error'] = repr(e)
        # the deployment surface once more with the RESULT array in pinned memory (a serving loop that reuses its arrays): the complex64
        # values are assembled on the device and the downloads land in the array itself - no host pass on the result side
        try:
            pc = eng.pinned_empty(bufs[0].shape, np.complex64)
            pc[...] = 0
            n_direct = eng.get_option('hp_direct_out_calls')
            eng.estimate(x128, ls=False, out=(pc, None))
            t4s = []
            for _ in range(3):
                t0 = time.perf_counter()
                eng.estimate(x128, ls=False, out=(pc, None))
                t4s.append(time.perf_counter() - t0)
            t4 = sorted(t4s)[1]
            host_path['python_c128_to_c64']['dnn_only_pinned_result'] = {
                'pairs_per_s': k * nr * nt / t4, 'ms': t4 * 1e3, 'ms_all': [round(t * 1e3, 2) for t in t4s],
                'bit_identical_with_pageable_result': bool(np.array_equal(pc, bufs[0])),
                'direct_downloads': eng.get_option('hp_direct_out_calls') - n_direct,
                'where_us': {n: eng.get_option(n) for n in ('hp_total_us', 'hp_stage_us', 'hp_wait_stage_us', 'hp_wait_out_us', 'hp_weave_us')},
                'note': 'CsiEngine.estimate(ls=False, out=(engine.pinned_empty(shape, complex64), None)): weave_c64_kernel + one download per chunk into the caller\'s array; '
                        'what is left is the input side (complex128 -> float32 planes on the host CPUs: where_us.hp_stage_us)'}
            del pc
        except Exception as e:                      # a side measurement never takes the line down
            host_path['python_c128_to_c64']['dnn_only_pinned_result_error'] = repr(e)
        # the same surface for a caller that holds its batch as complex64 in pinned memory (csi_estimate_c64, round-4 verdict next 7):
        # no host pass on either side - the chunk is uploaded as it is and split on the device, the result assembled there
        try:
            x64 = eng.pinned_empty(x128.shape, np.complex64)
            x64[...] = x128
            pc = eng.pinned_empty(bufs[0].shape, np.complex64)
            pc[...] = 0
            eng.estimate(x64, ls=False, out=(pc, None))
            t5s = []
            for _ in range(3):
                t0 = time.perf_counter()
                eng.estimate(x64, ls=False, out=(pc, None))
                t5s.append(time.perf_counter() - t0)
            t5 = sorted(t5s)[1]
            ref64 = np.empty(bufs[0].shape, np.complex64)
            eng.estimate(x64.astype(np.complex128), ls=False, out=(ref64, None))
            host_path['c64_pinned_in_and_out'] = {
                'dnn_only': {'pairs_per_s': k * nr * nt / t5, 'ms': t5 * 1e3, 'ms_all': [round(t * 1e3, 2) for t in t5s],
                             'bit_identical_with_c128_call_on_the_same_values': bool(np.array_equal(pc, ref64)),
                             'where_us': {n: eng.get_option(n) for n in ('hp_total_us', 'hp_stage_us', 'hp_wait_stage_us', 'hp_wait_out_us', 'hp_weave_us')}},
                'note': 'CsiEngine.estimate(complex64 array from pinned_empty, ls=False, out=(pinned complex64, None)) = csi_estimate_c64: the interleaved '
                        'chunk is DMA\'d from the caller\'s array, split_c64_kernel + kernels + weave_c64_kernel on the device, one download per chunk into the '
                        'caller\'s array.  Same bytes over the link as the complex128 call (8 B per sample after its host-side split), no host pass'}
            del x64, pc, ref64
        except Exception as e:                      # a side measurement never takes the line down
            host_path['c64_error'] = repr(e)
        try:
            host_path['host_cpu_quota'] = open('/sys/fs/cgroup/cpu.max').read().strip() + ' (cgroup cpu.max: quota / period us); os.cpu_count() = %d' % os.cpu_count()
        except OSError:
            host_path['host_cpu_quota'] = 'no cgroup v2 cpu.max; os.cpu_count() = %d' % os.cpu_count()
        # the link itself, same process, same byte counts: pinned host memory <-> device on the pipeline's two copy streams
        up = int(2 * h_re.nbytes)                      # two float32 planes (the complex128 batch is split on the host)
        try:
            for key, down in (('dnn_only', int(bufs[0].nbytes)), ('dnn_and_ls', int(bufs[0].nbytes + bufs[1].nbytes))):
                a, b, ab = eng.pcie_probe(up, down)
                tgt = host_path['python_c128_to_c64']['dnn_only'] if key == 'dnn_only' else host_path['python_c128_to_c64']
                tgt['pcie_bound_ms'] = round(ab, 3)
                tgt['pcie'] = {'h2d_bytes': up, 'd2h_bytes': down, 'h2d_alone_ms': round(a, 3), 'd2h_alone_ms': round(b, 3), 'both_ms': round(ab, 3),
                               'h2d_gbs': round(up / a / 1e6, 1), 'd2h_gbs': round(down / b / 1e6, 1),
                               'what': 'csi_profile_pcie: bare hipMemcpyAsync of these byte counts between pinned host memory and the device, 32 MiB pieces, two streams'}
                tgt['frac_of_pcie_bound'] = round(ab / tgt['ms'], 3)
                if key == 'dnn_only' and 'dnn_only_pinned_result' in host_path['python_c128_to_c64']:
                    pr_ = host_path['python_c128_to_c64']['dnn_only_pinned_result']
                    pr_['pcie_bound_ms'] = round(ab, 3)
                    pr_['frac_of_pcie_bound'] = round(ab / pr_['ms'], 3)
                if key == 'dnn_only' and 'c64_pinned_in_and_out' in host_path:
                    c64_ = host_path['c64_pinned_in_and_out']['dnn_only']
                    c64_['pcie_bound_ms'] = round(ab, 3)
                    c64_['frac_of_pcie_bound'] = round(ab / c64_['ms'], 3)
                if key == 'dnn_only' and 'pinned_planes_dnn' in host_path:
                    host_path['pinned_planes_dnn']['pcie_bound_ms'] = round(ab, 3)
                    host_path['pinned_planes_dnn']['frac_of_pcie_bound'] = round(ab / host_path['pinned_planes_dnn']['ms'], 3)
        except Exception as e:                      # a side measurement never takes the line down
            host_path['pcie_probe_error'] = repr(e)
        del x128, bufs

    # ---- cpu_baseline leg (rank 0, N = 1, after the timed region): the only place that touches oracle/.
    # It runs the CPU restatement of the reference on a bounded sample of the very same packets - timed
    # (--no-cpu-baseline skips the timing) and compared with what the GPU produced for them (--check 0
    # skips the comparison).  The oracle is the checker / the baseline here, never the thing measured.
    check, cpu_baseline = {}, None
    if rank == 0 and args.check > 0:               # N > 1: rank 0 checks packets of ITS shard
        from oracle import csi_oracle as o
        k = min(args.check, npkt)
        # spread over the batch, first and last included: with the mixed-SNR input (8 blocks of npkt / 8) k = 8 takes one packet
        # of every SNR level, the -25 dB and the +10 dB ends among them
        sel = sorted(set(int(round(j * (npkt - 1) / max(k - 1, 1))) for j in range(k))) if k > 1 else [0]
        take = lambda d: np.concatenate([d.download(p, 1) for p in sel])
        ltf = take(d_re) + 1j * take(d_im)
        r_re, r_im = o.predict_packets(ltf, wts['P']['pilot'], wts['real'], wts['imag'], np.float64, pkt_batch=k)
        g_re, g_im = take(d_ore), take(d_oim)
        check['dnn_rel_err'] = max(o.row_rel_err(g_re, r_re), o.row_rel_err(g_im, r_im))
        check['dnn_nmse_vs_fp64'] = o.nmse_subk(r_re + 1j * r_im, g_re + 1j * g_im)
        if args.dtype == 'bf16':
            b_re, b_im = o.predict_packets_bf16(ltf, wts['P']['pilot'], wts['real'], wts['imag'])
            check['dnn_rel_err_vs_bf16_emulation'] = max(o.row_rel_err(g_re, b_re), o.row_rel_err(g_im, b_im))
        if not args.no_ls:
            r_ls = o.ls_estimate(ltf, wts['P']['pilot'])
            check['ls_rel_err'] = max(o.row_rel_err(take(d_hre), r_ls.real), o.row_rel_err(take(d_him), r_ls.imag))
        check['packets'] = sel
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline as cb
        k = min(npkt, 256)
        ltf = (d_re.download(0, k) + 1j * d_im.download(0, k)).astype(np.complex64)
        r = cb.time_reference_loop(ltf, wts['P']['pilot'], wts['real'], wts['imag'], budget_s=args.cpu_budget_s)
        granted, how = cpus_granted()
        cpu_baseline = {
            'value': r['pairs_per_s'], 'unit': 'pair-channel estimates/s', 'cores': granted, 'cores_is': how, 'threads': r['threads'],
            'host_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': '%d packets one by one, naive fp32 net (torch-CPU sgemm) real+imag + numpy LS; median per packet' % r['packets'],
            'sample_detail': 'batch = Nt*Nr rows per packet (massiveMIMO_CSI_prediction_DNN.py:339-346), un-shared layer 0, real then imag model; '
                             'threads = the torch intra-op count that was fastest on this box (oracle/cpu_baseline.py), cores = CPUs the container may use',
            'dnn_only': r['dnn_pairs_per_s'], 'ls_only': r['ls_pairs_per_s'], 'cpu_model': r['cpu_model'],
            'dnn_one_large_batch': r['batched_dnn_pairs_per_s'], 'dnn_one_large_batch_packets': r['batched_packets'],
            'gpu_over_cpu': value / r['pairs_per_s']}

    # the same steps on the fp32 MFMA kernels, for comparison (outside the headline timed region)
    native = None
    if split_engine and world == 1:
        eng.set_option('f32_engine', 0)
        step(); eng.synchronize()
        eng.profile_enable(True); eng.profile_reset()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        eng.synchronize()
        dn = time.perf_counter() - t1
        pn = eng.profile()['pair_dense_gemm']
        eng.profile_enable(False)
        ntf = pn['flops'] / max(pn['ms'], 1e-9) / 1e9
        native = {'value': pairs_per_step * args.steps / dn, 'ms_per_step': dn / args.steps * 1e3,
                  'pair_dense_gemm_tflops': ntf, 'frac_of_fp32_mfma_peak': ntf / FP32_MATRIX_PEAK_TFLOPS,
                  'what': "same steps with f32_engine = 0 (v_mfma_f32_32x32x2_f32 kernels)"}
        if args.check > 0:
            native['dnn_rel_err'] = max(o.row_rel_err(take(d_ore), r_re), o.row_rel_err(take(d_oim), r_im))
        eng.set_option('f32_engine', {'auto': -1, 'native': 0, 'split': 1}[args.engine])

    # one-packet latency (the reference's literal per-packet call, DNN.py:346), device-resident, outside the timed region
    latency = None
    if rank == 0 and world == 1 and not args.no_latency:
        ts = []
        for i in range(40):
            eng.synchronize()
            t1 = time.perf_counter()
            if not args.no_ls:
                eng.estimate_device(d_re, d_im, 1, d_ore, d_oim, d_hre, d_him)
            else:
                eng.predict_device(d_re, d_im, 1, d_ore, d_oim)
            eng.synchronize()
            ts.append(time.perf_counter() - t1)
        latency = {'one_packet_us': float(np.median(ts[10:]) * 1e6),
                   'what': 'LS + DNN(real) + DNN(imag) of one packet as one csi_estimate_device call + csi_synchronize, device-resident, median of 30 calls'}

    # N > 1: BASELINE.json's two multi-GPU configurations, by ALL ranks, each as a fresh N-rank job of this script (round-4 verdict,
    # next 4): configs[3] Nt=64 Nr=8 50000 packets sharded, configs[4] Nt=128 Nr=16 100000 packets, one hipGraph per step
    legs = None
    if world > 1 and (default_headline(args) or args.legs_packets) and not args.no_other_configs and not args.graph and not args.option:
        del d_re, d_im, d_ore, d_oim, d_hre, d_him
        try:
            eng.comm_destroy()
        except Exception:                           # noqa: BLE001
            pass
        eng.close()
        try:
            legs = run_multi_gpu_legs(pkg, args, rank, world, local)
        except Exception as e:                      # noqa: BLE001 - the side legs never take the headline's line down
            legs = [{'error': 'multi-GPU legs: ' + repr(e)}]
    if rank != 0:
        return

    # HBM traffic per launch comes from the committed rocprofv3 PMC summary of this same command ON THIS WORKLOAD
    # (profiles/rNN_<workload>_traffic.json, written by tools/profile_config.sh + tools/profile_summarize.py; the newest round that
    # has one); it cannot be measured from inside the process.  The line names the files its fractions can be recomputed from.
    workload_tag = 'nt%d_nr%d_%s' % (nt, nr, args.dtype)
    traffic, prof_files = {}, {}
    pdir = os.path.join(REPO, 'profiles')
    if os.path.isdir(pdir):
        for kind in ('traffic.json', 'kernel_stats.txt', 'pmc.txt'):
            files = sorted(f for f in os.listdir(pdir) if f.endswith('_%s_%s' % (workload_tag, kind)))
            if not files and workload_tag == 'nt32_nr4_f32':     # rounds 1-4 named the headline's summaries rNN_<kind>
                files = sorted(f for f in os.listdir(pdir) if len(f) == len('r00_' + kind) and f.endswith('_' + kind))
            if files:
                prof_files[kind.split('.')[0]] = 'profiles/' + files[-1]
        if 'traffic' in prof_files:
            with open(os.path.join(REPO, prof_files['traffic'])) as f:
                traffic = json.load(f)
            traffic['_file'] = prof_files['traffic']

    def hbm_per_launch(*prefixes):
        for pre in prefixes:
            if isinstance(traffic.get(pre), dict):               # the kernel's own name first (csi_band4 is also a prefix of csi_band4_bf16)
                return traffic[pre].get('hbm_bytes_per_launch')
            for k, v in traffic.items():
                if k.startswith(pre) and isinstance(v, dict):
                    return v.get('hbm_bytes_per_launch')
        return None

    dom = prof['pair_dense_gemm']
    dom_ms = dom['ms'] / max(dom['launches'], 1)
    achieved = dom['flops'] / max(dom['ms'], 1e-9) / 1e9          # TFLOP/s
    if args.dtype == 'bf16':
        mfma_peak, peak_note = BF16_MATRIX_PEAK_TFLOPS, 'dense bf16 MFMA peak'
    elif split_engine:
        mfma_peak = BF16_MATRIX_PEAK_TFLOPS / SPLIT_PRODUCTS
        peak_note = ('split-f16 engine: dense f16 MFMA peak (2500 TFLOP/s) / 3 MFMA products per fp32-grade multiply; '
                     'achieved counts each multiply-add of the algorithm once (executed f16 flops are 3x)')
    else:
        mfma_peak, peak_note = FP32_MATRIX_PEAK_TFLOPS, 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'
    # the practical ceiling of the dominant kernel, measured here: its MFMA + barrier skeleton on the model's own operand data
    practical = None
    if band_kernel and world == 1:
        try:
            sk_ms, sk_tf = eng.band_skeleton(npkt * nr * nt, 5)
            practical = {'skeleton_ms_per_launch': sk_ms, 'skeleton_executed_tflops': sk_tf}
        except Exception as e:
            practical = {'error': repr(e)}
    kernels = {}
    for name, p in prof.items():
        if p['launches']:
            kernels[name] = dict(launches=p['launches'], ms_total=round(p['ms'], 3),
                                 ms_avg=round(p['ms'] / p['launches'], 4),
                                 tflops=round(p['flops'] / max(p['ms'], 1e-9) / 1e9, 2),
                                 algo_gbs=round(p['bytes'] / max(p['ms'], 1e-9) / 1e6, 1))
    out = {
        'metric': 'channel-estimates/sec (Nt=%d,Nr=%d)' % (nt, nr),
        'value': value,
        'unit': 'pair-channel estimates/s',
        'n_gpus': n_gpus,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': args.dtype,
        'arithmetic': ('fp32 operands, fp32 accumulate, fp32 results; GEMM products on the f16 matrix cores with split (hi + lo) '
                       'operands, 3 MFMA per product - error vs the fp64 oracle in parity_check (contract 1e-5)') if split_engine else
                      ('fp32 MFMA' if args.dtype == 'f32' else 'bf16 operands, fp32 accumulate'),
        'data': 'synthetic',
        'launch': ('one hipGraph per step, replayed (%d replays counted): LS kernel + range-guard memsets + magnitude sample + layer 0 + '
                   'per-pair layers + regressor, both models, every packet chunk' % graph_replays) if args.graph else 'eager kernel launches',
        'input': ('structured sounding packets, %d at each of {-25,-20,-15,-10,-5,0,5,10} dB in one launch (synth.mixed_snr_batch)' % (npkt // 8)) if mixed
                 else 'i.i.d. CN(0,1) generated on the device (csi_synth_white)',
        'config': {'workload': '%sNt=%d Nr=%d, %d packets/step in total (%d on this rank)%s, LS + DNN(real) + DNN(imag), FC %s + BN, 234 bins' % (
                       'configs[1]: ' if (nt, nr, args.packets, args.dtype) == (32, 4, 4000, 'f32') else
                       ('configs[2]: ' if (nt, nr, args.packets, args.dtype) == (64, 4, 5000, 'bf16') else
                        ('configs[3]: ' if (nt, nr, args.packets, args.scaling) == (64, 8, 50000, 'strong') else
                         ('configs[4]: ' if (nt, nr, args.packets, args.scaling) == (128, 16, 100000, 'strong') else ''))),
                       nt, nr, total_pkts, npkt, ' (8 SNR x %d)' % (npkt // 8) if mixed else '', 'x'.join(map(str, hidden))),
                   'pairs_per_step': pairs_per_step, 'packets_per_s': value / (nr * nt), 'ls_included': not args.no_ls,
                   'ranks': world, 'devices_visible_per_rank': ndev, 'dist_backend': backend if world > 1 else None,
                   'world_size_checked': pkg.dist.world_size(), 'weights_via': via,
                   'sharding': ('contiguous packet ranges per rank (%s scaling), weights broadcast once over %s, no collective in the step'
                                % (args.scaling, 'RCCL' if backend == 'nccl' else backend)) if world > 1 else 'single GPU'},
        'roofline': {'bound': 'mfma', 'kernel': 'pair_dense_gemm' + ((' (%s: first per-pair layer + regressor fused, h2 in registers)' % band_name) if band_any else ''), 'achieved': achieved, 'peak': mfma_peak,
                     'unit': 'TFLOP/s', 'frac': achieved / mfma_peak, 'peak_note': peak_note,
                     'executed_mfma_tflops': achieved * (SPLIT_PRODUCTS if split_engine else 1),
                     'vs_fp32_mfma_peak': achieved / FP32_MATRIX_PEAK_TFLOPS if args.dtype == 'f32' else None,
                     'traffic': (hbm_per_launch((band_name if band_kernel else 'gemm_hs_pp_pair_kernel<2') if split_engine else 'pair_gemm') if args.dtype == 'f32'
                                 else hbm_per_launch(band_name if band_any else 'gemm_bf16_pp_pair_kernel<1')), 'traffic_unit': 'HBM bytes per launch (PMC)',
                     'traffic_source': traffic.get('_file'), 'profile_files': prof_files or None,
                     'algorithmic_bytes_per_launch': dom['bytes'] / max(dom['launches'], 1),
                     'avg_launch_ms': dom_ms, 'flops_per_launch': dom['flops'] / max(dom['launches'], 1),
                     'practical_peak': (practical['skeleton_executed_tflops'] / SPLIT_PRODUCTS) if practical and 'error' not in practical else None,
                     'frac_of_practical': (achieved / (practical['skeleton_executed_tflops'] / SPLIT_PRODUCTS)) if practical and 'error' not in practical else None,
                     'practical_peak_note': ('csi_profile_band_skeleton on this box, this run: the same kernel with everything but its MFMAs, barriers and waits '
                                             'removed and its operand registers holding the model\'s own split weights (relu-like zeros in the activation fragments), '
                                             '%.3f ms per launch = %.0f TFLOP/s of executed f16 MFMA; divided by 3 products.  The part runs these kernels at its ~1.4 kW '
                                             'power limit at 1.8-2.0 GHz, not 2.4 (profiles/r04_power.txt), so the table peak is not reachable on real data'
                                             % (practical['skeleton_ms_per_launch'], practical['skeleton_executed_tflops'])) if practical and 'error' not in practical else (practical or {}).get('error')},
        'kernels': kernels,
        'parity_check': check,
        'latency': latency,
        'ranks_ms': [round(i['ms_per_step'], 4) for i in infos],
        'devices': [i['device'] for i in infos],
        'ranks': infos,
    }
    if no_events is not None:
        out['timed_region_events'] = {'ms_per_step_with_kernel_events': round(dt_rank / args.steps * 1e3, 4), 'ms_per_step_without': round(no_events, 4),
                                      'note': 'rank 0, the same %d steps again behind the timed region with csi_profile_enable(0): `value` is the run WITH the per-kernel '
                                              'HIP events (they feed `kernels` and `roofline`), this is what they cost' % args.steps}
    if guard:
        out['split_engine_range_guard'] = guard
    if native:
        out['native_fp32_engine'] = native
    if host_path:
        out['host_path_pcie_inclusive'] = host_path
    if 'ls_estimate' in kernels:
        out['roofline_ls'] = ls_roofline(prof['ls_estimate'], hbm_per_launch('ls_estimate_'), traffic.get('_file'))

    if cpu_baseline:
        out['cpu_baseline'] = cpu_baseline
    if world == 1 and not args.no_regimes and args.dtype == 'f32' and not args.graph and not args.option and (nt, nr, tuple(hidden)) == (32, 4, (1024, 1024)):
        try:
            out['regimes'] = regimes(pkg, eng, nt, nr, hidden, d_re, d_im, npkt, ls=not args.no_ls)
            big = [r for r in out['regimes'] if r['packets'] == 500]
            if big:
                big[0]['rate_vs_headline'] = round(big[0]['pairs_per_s'] / value, 4)
        except Exception as e:                      # side measurements never take the headline down
            out['regimes'] = {'error': repr(e)}
    if world == 1 and not args.no_next_rows and args.dtype == 'f32' and nt > 0:
        try:
            out['next_rows'] = next_rows(pkg, eng, nt, nr, hidden, wts, d_re, d_im, min(npkt, 1000))
        except Exception as e:                      # side measurements never take the headline down
            out['next_rows'] = {'error': repr(e)}
    default_run = default_headline(args)
    if legs is not None:
        out['other_configs'] = legs
    if world == 1 and default_run and not args.no_other_configs and not args.graph and not args.option:
        del d_re, d_im, d_ore, d_oim, d_hre, d_him
        out['other_configs'] = other_configs()
    # the whole process as rank 0 saw it (imports, input synthesis, warm-up, timed region, side measurements): what a clock around
    # the command should read, give or take the interpreter's start
    out['bench_wall_s'] = round(time.perf_counter() - t_process, 1)
    if args.full_line:
        print(json.dumps(out))
        return
    detail_file = write_detail(out, args.detail_file)
    print(compact_line(out, detail_file))

LINE_LIMIT = 4096                     # bytes: the driver's capture lost round 5's 20 KB line (BENCH_r05.parsed = null)


def _sig(x, n=6):
    """floats to n significant digits (the line is a record, bench_detail.json keeps every digit)"""
    if isinstance(x, float):
        return float('%.*g' % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def write_detail(out, path=''):
    """The whole measurement object (every note, per-kernel table, host-path legs, regimes, next rows, per-rank records) goes to a
    file, not to the line: bench_detail.json beside this script and, on a gpurun box, gpurun_out/bench_detail.json so that it
    travels back.  Never to stderr: the driver's record keeps one 8 KB tail of stdout + stderr together and the line has to be in it."""
    text = json.dumps(out, indent=1)
    written = None
    targets = [path] if path else [os.path.join(REPO, 'bench_detail.json')]
    if not path and os.path.isdir(os.path.join(REPO, 'gpurun_out')):
        targets.append(os.path.join(REPO, 'gpurun_out', 'bench_detail.json'))
    for t in targets:
        try:
            with open(t, 'w') as f:
                f.write(text)
            written = written or os.path.relpath(t, REPO)
        except OSError as e:
            print('bench.py: cannot write %s (%s)' % (t, e), file=sys.stderr)
    return written


def compact_line(out, detail_file):
    """The ONE line of the contract, below LINE_LIMIT bytes for any rank count: contract keys, `config`, `roofline` (dominant
    kernel), `roofline_ls`, `cpu_baseline`, `parity_check` and one-number summaries of the side measurements (what the reference
    itself reports is one figure per configuration, massiveMIMO_CSI_prediction_DNN.py:441-475).  Everything else is in `detail_file`."""
    rf, cfg = out['roofline'], out['config']
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                'vs_baseline', 'dtype', 'data')}
    line['config'] = {'workload': cfg['workload'], 'pairs_per_step': cfg['pairs_per_step'], 'packets_per_s': cfg['packets_per_s'],
                      'ranks': cfg['ranks'], 'weights_via': cfg['weights_via'].split(':')[0].split(' (')[0], 'ls_included': cfg['ls_included'],
                      'launch': 'hipGraph' if out['launch'].startswith('one hipGraph') else 'eager'}
    line['roofline'] = {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'flops_per_launch',
                                               'algorithmic_bytes_per_launch', 'practical_peak', 'frac_of_practical')}
    line['roofline']['kernel'] = rf['kernel'].split(' (')[-1].split(':')[0] if '(' in rf['kernel'] else rf['kernel']
    line['roofline']['profile_files'] = sorted((rf.get('profile_files') or {}).values()) or None
    if out.get('roofline_ls'):
        line['roofline_ls'] = {k: out['roofline_ls'].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')}
    if out.get('cpu_baseline'):
        cb = out['cpu_baseline']
        line['cpu_baseline'] = {k: cb.get(k) for k in ('value', 'unit', 'cores', 'cores_is', 'threads', 'host_cpu_count', 'cpu_model', 'kind', 'sample',
                                                       'dnn_only', 'ls_only', 'dnn_one_large_batch', 'gpu_over_cpu')}
    pc = dict(out.get('parity_check') or {})
    if 'packets' in pc:
        pc['packets'] = len(pc['packets'])
    line['parity_check'] = pc
    if out.get('latency'):
        line['latency_us'] = out['latency']['one_packet_us']
    line['ranks_ms'] = out['ranks_ms']
    if out.get('split_engine_range_guard'):
        line['hs_range_fallbacks'] = out['split_engine_range_guard']['hs_range_fallbacks']
    if out.get('timed_region_events'):
        line['ms_per_step_without_kernel_events'] = out['timed_region_events']['ms_per_step_without']
    if out.get('native_fp32_engine'):
        n = out['native_fp32_engine']
        line['native_fp32_engine'] = {'value': n['value'], 'ms_per_step': n['ms_per_step'], 'frac_of_fp32_mfma_peak': n['frac_of_fp32_mfma_peak']}
    hp = out.get('host_path_pcie_inclusive')
    if hp:
        c128 = (hp.get('python_c128_to_c64') or {}).get('dnn_only') or {}
        c64 = (hp.get('c64_pinned_in_and_out') or {}).get('dnn_only') or {}
        line['host_path'] = {'packets': hp.get('packets'), 'planes_pairs_per_s': hp.get('pairs_per_s'),
                             'c128_dnn_only_pairs_per_s': c128.get('pairs_per_s'), 'c128_frac_of_pcie_bound': c128.get('frac_of_pcie_bound'),
                             'c64_pinned_pairs_per_s': c64.get('pairs_per_s'), 'c64_frac_of_pcie_bound': c64.get('frac_of_pcie_bound')}
    rg = out.get('regimes')
    if isinstance(rg, list):
        # packets -> [queued us per call, latency us, bound us]
        line['regimes_us'] = {str(r['packets']): [r['pipelined_us'], r['latency_us'], r['bound_us']] for r in rg}
        big = [r for r in rg if r['packets'] == 500 and 'rate_vs_headline' in r]
        if big:
            line['regime_500_rate_vs_headline'] = big[0]['rate_vs_headline']
    elif rg:
        line['regimes_us'] = rg
    nx = out.get('next_rows')
    if isinstance(nx, dict) and 'error' not in nx:
        line['next_rows'] = {'lmmse_links_per_s': (nx.get('lmmse') or {}).get('links_per_s'), 'lmmse_frac_fp64': ((nx.get('lmmse') or {}).get('roofline') or {}).get('frac'),
                             'train_ms_per_step': (nx.get('train_step') or {}).get('ms_per_step'), 'train_loss_rel_err': (nx.get('train_step') or {}).get('loss_rel_err'),
                             'ls_vht_pilot_frac_hbm': ((nx.get('ls_vht_pilot') or {}).get('roofline_ls') or {}).get('frac')}
    elif nx:
        line['next_rows'] = nx
    if out.get('other_configs') is not None:
        oc = []
        for c in out['other_configs']:
            if 'error' in c or 'skipped' in c:
                oc.append({'config': c.get('config'), 'error': (c.get('error') or c.get('skipped'))[:160]})
                continue
            pcheck = c.get('parity_check') or c.get('parity_check_rank0_shard') or {}
            e = {'config': c['config'], 'value': c['value'], 'ms_per_step': c['ms_per_step'], 'dtype': c['dtype'],
                 'frac': (c.get('roofline') or {}).get('frac'), 'dnn_rel_err': pcheck.get('dnn_rel_err')}
            if 'dnn_rel_err_vs_bf16_emulation' in pcheck:
                e['dnn_rel_err_vs_bf16_emulation'] = pcheck['dnn_rel_err_vs_bf16_emulation']
            if 'n_gpus' in c:
                e['n_gpus'] = c['n_gpus']
                e['ranks_ms'] = c.get('ranks_ms')
            oc.append(e)
        line['other_configs'] = oc
    line['detail_file'] = detail_file
    line['bench_wall_s'] = out['bench_wall_s']
    line = _sig(line)
    line['value'], line['ms_per_step'] = out['value'], out['ms_per_step']          # the contract's two numbers keep every digit
    text = json.dumps(line, separators=(',', ':'))
    # a line that would not fit sheds its summaries, least important first, never a contract key (and says which in `dropped`)
    dropped = []
    for key in ('next_rows', 'host_path', 'native_fp32_engine', 'regimes_us', 'other_configs', 'ranks_ms'):
        if len(text) < LINE_LIMIT:
            break
        if key in line:
            line.pop(key)
            dropped.append(key)
            line['dropped'] = dropped
            text = json.dumps(line, separators=(',', ':'))
    assert len(text) < LINE_LIMIT, 'bench line of %d bytes' % len(text)
    return text


def ls_roofline(p, traffic_bytes, traffic_file):
    """HBM roofline of the LS kernel.  `achieved` = bytes the kernel MOVES (PMC: FETCH_SIZE / WRITE_SIZE of the committed counter
    pass of this workload, gfx950 correction) over the launch time measured here; the kernel never fetches the 64-sample cyclic
    prefix of a symbol, so that is less than SURVEY 8d's 2560 B in + 1872 B out per pair, which stays on the line as the labelled
    `algorithmic_gbs` / `algorithmic_frac` (round-4 verdict, weak 4).  Without a counter file for the workload `achieved` falls back
    to the bytes the kernel is known to move by construction: 2048 B in (no prefix) + 1872 B out per pair."""
    ms = p['ms'] / max(p['launches'], 1)
    algo = p['bytes'] / max(p['launches'], 1)
    moved = traffic_bytes if traffic_bytes else algo * (2048.0 + 1872.0) / (2560.0 + 1872.0)
    gbs = moved / max(ms, 1e-9) / 1e6
    return {'bound': 'hbm', 'kernel': 'ls_estimate', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
            'achieved_is': ('PMC bytes per launch (%s) / launch time by HIP events in this run' % traffic_file) if traffic_bytes
                           else 'bytes moved by construction (2048 B in without the cyclic prefixes + 1872 B out per pair) / launch time; no counter file for this workload',
            'traffic': traffic_bytes, 'traffic_source': traffic_file, 'avg_launch_ms': ms,
            'algorithmic_bytes_per_launch': algo, 'algorithmic_gbs': algo / max(ms, 1e-9) / 1e6, 'algorithmic_frac': algo / max(ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
            'algorithmic_note': 'SURVEY 8d: 2560 B in + 1872 B out per pair - counts the cyclic prefixes the kernel leaves in HBM; a speed figure, not the roofline fraction',
            'achievable_note': 'a plain float4 copy reaches 6.29 TB/s on this part (MI355X_MICROARCH.md): frac / 0.786 = fraction of the achievable rate'}


def regimes(pkg, eng, nt, nr, hidden, d_re, d_im, npkt_resident, ls=True):
    """Batch-size regimes of the SAME context, device-resident, outside the timed region (round-4 verdict, next 3): 1 packet
    (DNN.py:339-346 predicts one packet per step), 8, 64, and 500 packets (what full_pipeline_maMIMO_DNNEst.sh:44-48 hands one
    `--test` run per SNR level).  Per size: `latency_us` = one call + csi_synchronize, median; `pipelined_us` = per call when 20
    calls are queued back to back (a serving loop that does not wait between calls); pairs/s from the latter; the bound that applies
    and the fraction of it reached.  Bounds: up to 8 packets the call streams both component models' weights once (HBM at 8 TB/s;
    they also fit the 256 MiB Infinity Cache, so a loop of calls may read them from there: `weights_from`); from 64 packets the
    executed flops of the shared-layer-0 network (6 463 488 per pair) at the ceiling of the engine that serves the size -
    2500 / 3 TFLOP/s where the split-f16 engine runs (`hs_launches` moved), 157.3 TFLOP/s on the fp32 MFMA kernels - plus the LS
    kernel's 4432 B per pair at 8 TB/s."""
    h1, h2 = hidden[0], hidden[-1]
    w_bytes = 2 * 4.0 * (320 * nt * h1 + (h1 * h2 if len(hidden) > 1 else 0) + h2 * 234)
    flops_pair = 4.0 * 320 * h1 + 4.0 * ((h1 * h2 if len(hidden) > 1 else 0) + h2 * 234)
    res = []
    for n in (1, 8, 64, 500):
        if n > npkt_resident:
            continue
        o = [eng.empty((n, nr, nt, 234)) for _ in range(4)]

        def call():                       # LS + DNN as ONE call (csi_estimate_device), as a serving loop would issue it
            if ls:
                eng.estimate_device(d_re, d_im, n, o[0], o[1], o[2], o[3])
            else:
                eng.predict_device(d_re, d_im, n, o[0], o[1])
        for _ in range(5):
            call()
        eng.synchronize()
        hs0 = eng.get_option('hs_launches')
        ls0 = eng.get_option('l0_stream_launches')
        lat = []
        for _ in range(30):
            t0 = time.perf_counter()
            call()
            eng.synchronize()
            lat.append(time.perf_counter() - t0)
        pip = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                call()
            eng.synchronize()
            pip.append((time.perf_counter() - t0) / 20)
        hs_per_call = (eng.get_option('hs_launches') - hs0) / 130.0
        split = hs_per_call > 0
        # 2 split-engine launches per call = the band kernel of each component model only: layer 0 ran on the fp32 MFMA kernels
        streamed = eng.get_option('l0_stream_launches') > ls0        # layer 0 on the weight-streaming split-f16 kernel
        mixed = split and len(hidden) == 2 and hs_per_call < 3 and not streamed
        eng.profile_enable(True); eng.profile_reset()
        call(); eng.synchronize()
        prof = {k: round(v['ms'] * 1e3, 1) for k, v in eng.profile().items() if v['launches']}
        eng.profile_enable(False)
        pairs = n * nr * nt
        t_w = w_bytes / (HBM_PEAK_GBS * 1e9)
        l0_flops = 4.0 * 320 * h1
        if mixed:
            t_f = pairs * (l0_flops / (FP32_MATRIX_PEAK_TFLOPS * 1e12) + (flops_pair - l0_flops) / (BF16_MATRIX_PEAK_TFLOPS / SPLIT_PRODUCTS * 1e12))
        else:
            t_f = pairs * flops_pair / ((BF16_MATRIX_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MATRIX_PEAK_TFLOPS) * 1e12)
        t_ls = pairs * 4432.0 / (HBM_PEAK_GBS * 1e9) if ls else 0.0
        if t_w >= t_f:
            bound, t_b = 'hbm: both models\' weights (%.1f MB) streamed once at 8 TB/s' % (w_bytes / 1e6), t_w + t_ls
        else:
            bound, t_b = 'mfma: executed flops at %s' % ('157.3 TFLOP/s (layer 0, fp32 MFMA kernels) + 2500/3 TFLOP/s (per-pair layers, split-f16 band kernel)' if mixed else
                                                         '2500/3 TFLOP/s (split-f16 engine)' if split else '157.3 TFLOP/s (fp32 MFMA kernels)'), t_f + t_ls
        t_lat, t_pip = float(np.median(lat)), float(np.median(pip))
        res.append({'packets': n, 'pairs': pairs, 'latency_us': round(t_lat * 1e6, 1), 'pipelined_us': round(t_pip * 1e6, 1),
                    'pairs_per_s': pairs / t_pip, 'engine': 'fp32 MFMA layer 0 + split-f16 band kernel' if mixed else
                    ('split-f16 (weight-streaming layer 0 + column-split band kernel)' if streamed else ('split-f16' if split else 'fp32 MFMA')),
                    'bound': bound, 'bound_us': round(t_b * 1e6, 2), 'frac_of_bound': round(t_b / t_pip, 4), 'frac_of_bound_latency': round(t_b / t_lat, 4),
                    'kernels_us_one_call_with_events': prof})
        del o
    return res


FP64_VECTOR_PEAK_TFLOPS = 78.6        # MI355X datasheet (fp64 vector = fp64 matrix); the microarchitecture guide has no fp64 row

P_VHT4 = [[1, -1, 1, 1], [1, 1, -1, 1], [1, 1, 1, -1], [-1, 1, 1, 1]]


def newest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (None if there is none): the file a fraction on the line can be recomputed from"""
    pdir = os.path.join(REPO, 'profiles')
    files = sorted(f for f in os.listdir(pdir) if f.endswith('_' + suffix)) if os.path.isdir(pdir) else []
    return 'profiles/' + files[-1] if files else None


def next_rows(pkg, eng, nt, nr, hidden, wts, d_re, d_im, npkt):
    """SURVEY 8 'next' rows on the driver's record, outside the timed region, each with its own oracle check:
    f-3 LMMSE smoother (LMMSE_ce.m:23-39), f-4 one training step (DNN.py:312-316), and the LS estimate with the pilot matrix real
    pipelines carry (helperGetP, helperMIMOChannelEstimate.m:13: the 802.11 VHT 4x4 base doubled up - Hadamard, not Sylvester-ordered)."""
    import ctypes
    from oracle import csi_oracle as o
    res = {}
    rng = np.random.default_rng(77)
    lib, ctx = eng._lib, eng._ctx
    # ---- LS with a non-Sylvester +-1 pilot: the Walsh-Hadamard kernel through its symbol / antenna tables (round 4)
    P0 = wts['P']['pilot']
    Pv = np.kron(pkg.synth.hadamard(nt // 4), np.array(P_VHT4, np.float64)) if nt % 4 == 0 and nt >= 8 else None
    h_re, h_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    if Pv is not None:
        eng.set_pilot(Pv)
        for _ in range(3):
            eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
        eng.synchronize()
        eng.profile_enable(True); eng.profile_reset()
        for _ in range(10):
            eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
        eng.synchronize()
        p = eng.profile()['ls_estimate']
        eng.profile_enable(False)
        ltf = d_re.download(0, 2) + 1j * d_im.download(0, 2)
        ref = o.ls_estimate(ltf, Pv)
        got = h_re.download(0, 2) + 1j * h_im.download(0, 2)
        res['ls_vht_pilot'] = {'pilot': 'kron(H_%d, P_VHT4): Hadamard, not Sylvester-ordered' % (nt // 4), 'ls_mode': eng.get_option('ls_mode'),
                               'pilot_class': eng.get_option('ls_pilot_fast'), 'packets': npkt, 'ms_per_launch': p['ms'] / p['launches'],
                               'roofline_ls': ls_roofline(p, None, None),
                               'ls_rel_err': max(o.row_rel_err(got.real, ref.real), o.row_rel_err(got.imag, ref.imag))}
        eng.set_pilot(P0)
    # ---- LMMSE smoother of the LS estimate (fp64 Levinson solves)
    eng.ls_estimate_device(d_re, d_im, npkt, h_re, h_im)
    o_re, o_im = eng.empty((npkt, nr, nt, 234)), eng.empty((npkt, nr, nt, 234))
    hv_h = (np.sort(np.abs(rng.standard_normal((npkt, 100)))) * 1e-7).astype(np.float32)
    snr_h = rng.choice([-10.0, 0.0, 10.0, 25.0], size=(npkt, nr)).astype(np.float32)
    hv, snr = eng.to_device(hv_h), eng.to_device(snr_h)
    run = lambda: eng._check(lib.csi_lmmse_estimate_device(ctx, h_re.ptr, h_im.ptr, npkt, hv.ptr, 100, snr.ptr, o_re.ptr, o_im.ptr))
    run(); eng.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(3):
        run()
    eng.synchronize()
    p = eng.profile()['lmmse_levinson']
    eng.profile_enable(False)
    ms = p['ms'] / p['launches']
    tf = p['flops'] / max(p['ms'], 1e-9) / 1e9
    k = 1
    hls = h_re.download(0, k) + 1j * h_im.download(0, k)
    ref = o.lmmse_estimate(hls, hv_h[:k].astype(np.float64), snr_h[:k].astype(np.float64))
    got = o_re.download(0, k) + 1j * o_im.download(0, k)
    res['lmmse'] = {'what': 'LMMSE_ce.m:23-39 per link as one Hermitian-Toeplitz Levinson solve per (packet, rx) in fp64 (csrc/lmmse.hip.h)',
                    'packets': npkt, 'ms_per_launch': ms, 'links_per_s': npkt * nr * nt / (ms * 1e-3),
                    'roofline': {'bound': 'fp64 vector FMA', 'achieved': tf, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP64_VECTOR_PEAK_TFLOPS},
                    'rel_err_vs_oracle': max(o.row_rel_err(got.real, ref.real), o.row_rel_err(got.imag, ref.imag)), 'checked_packets': k,
                    'tuning': 'a "next" row (SURVEY 8f-3): correct and measured, not tuned - one Levinson recursion per (packet, rx) is a serial chain of 234 steps',
                    'profile_files': {'kernel_stats': newest_profile('next_rows_kernel_stats.txt'), 'pmc': newest_profile('next_rows_pmc.txt'),
                                      'kernel': 'lmmse_levinson_kernel'}}
    # ---- one training step of the shipped model (B = 256), resident dataset, against the fp64 oracle's loss
    B, n_rows = 256, 64
    table = (rng.standard_normal((n_rows, 320 * nt)) * 0.1).astype(np.float32)
    N = n_rows * nt
    ltf_row, itx = np.repeat(np.arange(n_rows), nt).astype(np.int32), np.tile(np.arange(nt), n_rows).astype(np.int32)
    y = (rng.standard_normal((N, 234)) * 0.5).astype(np.float32)
    w0 = {kk: np.array(v, copy=True) for kk, v in wts['real'].items()}
    eng.train_begin('real', weights=w0, lr=1e-4, dropout=0.0, seed=1)
    eng.train_set_dataset('real', table, ltf_row, itx, y)
    ids = [rng.permutation(N)[:B].astype(np.int32) for _ in range(24)]
    loss0 = eng.train_step_indexed('real', ids[0], noise_std=0.0)
    x0 = np.concatenate([table[ltf_row[ids[0]]], np.asarray(P0, np.float32)[itx[ids[0]]]], axis=1)
    refw = {kk: np.asarray(v, np.float64) for kk, v in w0.items() if kk != 'bn_eps'}
    rloss = o.train_step_reference(refw, o.adam_init(refw), x0, y[ids[0]], lr=1e-4, use_bn=True)[0]
    for i in range(1, 4):
        eng.train_step_indexed('real', ids[i], noise_std=0.1)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(4, 24):
        eng.train_step_indexed('real', ids[i], noise_std=0.1)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 20
    eng.profile_enable(True); eng.profile_reset()
    for i in range(4, 9):
        eng.train_step_indexed('real', ids[i], noise_std=0.1)
    eng.synchronize()
    pr = eng.profile()
    eng.profile_enable(False)
    eng.train_end('real', commit=False)
    g = pr['train_gemm']
    res['train_step'] = {'what': 'one optimiser step of DNN.py:312-316 (AWGN input noise, Dense+relu+BN x2, regressor, mse, Adam), B = %d, resident '
                                 'de-duplicated dataset (csi_train_indexed)' % B, 'ms_per_step': dt * 1e3, 'samples_per_s': B / dt,
                         'train_gemm_ms_per_step': g['ms'] / 5, 'train_gemm_tflops': g['flops'] / max(g['ms'], 1e-9) / 1e9,
                         'train_gemm_frac_of_fp32_mfma_peak': g['flops'] / max(g['ms'], 1e-9) / 1e9 / FP32_MATRIX_PEAK_TFLOPS,
                         'elementwise_ms_per_step': pr['train_elementwise']['ms'] / 5,
                         'first_step_loss': loss0, 'first_step_loss_oracle_fp64': rloss, 'loss_rel_err': abs(loss0 - rloss) / max(abs(rloss), 1e-30),
                         'tuning': 'a "next" row (SURVEY 8f-4): correct and measured, not tuned - B = 256 rows fill a quarter of the chip\'s fp32 MFMA tiles',
                         'profile_files': {'kernel_stats': newest_profile('next_rows_kernel_stats.txt'), 'kernels': 'gemm_f32_kernel<*> launches of grid <= 256 workgroups'}}
    return res


def cpus_granted():
    """CPUs this process may actually use (north_star: core count stated): the cgroup v2 quota when there is one (cpu.max = quota /
    period), else the affinity mask, else os.cpu_count() - on the GPU boxes os.cpu_count() says 256 while the container is granted 16."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            return round(int(quota) / int(period), 2), 'cgroup cpu.max %s/%s' % (quota, period)
    except (OSError, ValueError):
        pass
    try:                                  # cgroup v1
        quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if quota > 0 and period > 0:
            return round(quota / period, 2), 'cgroup cpu.cfs_quota_us %d/%d' % (quota, period)
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0)), 'sched_getaffinity'
    except (AttributeError, OSError):
        return os.cpu_count(), 'os.cpu_count'


def default_headline(args):
    """the driver's command: config 2 with nothing overridden (the side configurations are measured only beside that)"""
    return (args.nt, args.nr, args.packets, args.dtype, tuple(args.hidden), args.engine, args.scaling) == (32, 4, 4000, 'f32', (1024, 1024), 'auto', 'weak')


HBM_BYTES_PER_GPU = 288e9


def multi_gpu_legs(args, world):
    """BASELINE.json configs[3] / configs[4] as N-rank jobs: (name, workload, flags, fits, GB resident per rank)."""
    tot = [int(x) for x in args.legs_packets.split(',')] if args.legs_packets else [50000, 100000]
    legs = []
    for name, nt, nr, total, extra in (('configs[3]', 64, 8, tot[0], []), ('configs[4]', 128, 16, tot[1], ['--graph'])):
        per_rank = -(-total // world)
        gb = per_rank * nr * (2 * 320 * nt + 4 * nt * 234) * 4 / 1e9           # preamble planes + DNN and LS result planes
        flags = ['--scaling', 'strong', '--nt', str(nt), '--nr', str(nr), '--packets', str(total), '--input', 'white',
                 '--steps', '5', '--warmup', '4' if extra else '2'] + extra
        what = 'Nt=%d Nr=%d, %d packets sharded over %d GPUs (contiguous packet ranges, weights broadcast once over RCCL)%s' % (
            nt, nr, total, world, ', one hipGraph per step and rank' if extra else '')
        legs.append((name, what, flags, gb < 0.8 * HBM_BYTES_PER_GPU / 1e9, gb))
    return legs


def run_multi_gpu_legs(pkg, args, rank, world, local):
    """Every rank of the running job starts ONE child per leg with its own RANK / LOCAL_RANK and a fresh rendezvous port that rank 0
    found and broadcast: a fresh N-rank job of this script per configuration (own engines, weights broadcast through the library,
    own shard, rank 0 of it checks packets of its shard against the oracle).  Rank 0 returns the parsed lines."""
    import socket
    import subprocess
    import torch.distributed as tdist
    res = []
    for name, what, flags, fits, gb in multi_gpu_legs(args, world):
        if not fits:
            res.append({'config': name, 'workload': what, 'skipped': '%.0f GB per rank do not fit %d GPUs of %.0f GB' % (gb, world, HBM_BYTES_PER_GPU / 1e9)})
            continue
        box = [None]
        if rank == 0:
            with socket.socket() as so:
                so.bind(('127.0.0.1', 0))
                box[0] = so.getsockname()[1]
        tdist.broadcast_object_list(box, src=0)
        env = dict(os.environ, MASTER_PORT=str(box[0]), CSI_RCCL_ID_TOKEN='leg-%s-%d' % (name, box[0]), CSI_DIST_TIMEOUT_S='180')
        for k in [k for k in env if k.startswith('TORCHELASTIC_')]:      # under torchrun: the child job's rank 0 serves its own store
            env.pop(k)                                                     # (TORCHELASTIC_USE_AGENT_STORE would make it look for the agent's)
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--full-line', '--no-other-configs', '--no-cpu-baseline', '--no-latency',
               '--host-path', '0', '--check', '4', '--no-next-rows', '--no-regimes', '--weights-via', args.weights_via] + flags
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=420)     # (one GPU's share takes 5 / 22 s of wall time)
            rc, so_, se_ = r.returncode, r.stdout, r.stderr
        except Exception as e:                       # noqa: BLE001 - a failed side measurement never takes the headline down
            rc, so_, se_ = -1, '', repr(e)
        rcs = pkg.dist.gather_objects(rc)
        pkg.dist.barrier()
        if rank != 0:
            continue
        lines = [ln for ln in so_.splitlines() if ln.startswith('{')]
        if any(rcs) or not lines:
            res.append({'config': name, 'workload': what, 'flags': ' '.join(flags), 'error': 'rank exit codes %s: %s' % (rcs, se_[-400:])})
            continue
        j = json.loads(lines[-1])
        res.append({'config': name, 'workload': what, 'flags': ' '.join(flags), 'n_gpus': j['n_gpus'], 'value': j['value'], 'unit': j['unit'],
                    'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'scaling': j['scaling'], 'dtype': j['dtype'], 'launch': j['launch'],
                    'pairs_per_step': j['config']['pairs_per_step'], 'input': j['input'], 'ranks_ms': j['ranks_ms'],
                    'devices': [d.get('ordinal') for d in j['devices']], 'packets_per_rank': [r_['packets'] for r_ in j['ranks']],
                    'rccl_ranks': j['config']['world_size_checked'], 'weights_via': j['config']['weights_via'], 'sharding': j['config']['sharding'],
                    'roofline': {k: j['roofline'][k] for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms')},
                    'roofline_ls_frac': (j.get('roofline_ls') or {}).get('frac'),
                    'parity_check_rank0_shard': j['parity_check'], 'split_engine_range_guard': j.get('split_engine_range_guard'),
                    'wall_s': round(time.perf_counter() - t0, 1)})
    return res


OTHER_CONFIGS = [
    ('configs[2]', 'Nt=64 Nr=4, 5000 packets, bf16 MFMA, 1 GPU',
     ['--dtype', 'bf16', '--nt', '64', '--nr', '4', '--packets', '5000', '--steps', '5', '--warmup', '2']),
    ('configs[3] share', 'Nt=64 Nr=8: one GPU\'s share (6250 packets) of the 50000 packets sharded over 8 GPUs, fp32',
     ['--nt', '64', '--nr', '8', '--packets', '6250', '--steps', '5', '--warmup', '2']),
    ('configs[4] share', 'Nt=128 Nr=16: one GPU\'s share (12500 packets) of the 100000 packets over 8 GPUs, fp32, one hipGraph per step',
     ['--nt', '128', '--nr', '16', '--packets', '12500', '--steps', '5', '--warmup', '4', '--graph']),
]


def other_configs():
    """BASELINE.json configs[2..4] as far as one GPU can run them, each in a FRESH process of this script (its own engine,
    weights and buffers) after the headline's timed region: ms/step, pairs/s, the dominant kernel's roofline fraction and a
    2-packet check against the fp64 oracle.  Reported beside the headline, never part of it."""
    import subprocess
    res = []
    for name, what, flags in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--full-line', '--no-other-configs', '--no-cpu-baseline', '--no-latency',
               '--host-path', '0', '--check', '4', '--no-next-rows'] + flags
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not lines:
                res.append({'config': name, 'workload': what, 'error': 'rc %d: %s' % (r.returncode, r.stderr[-400:])})
                continue
            j = json.loads(lines[-1])
            res.append({'config': name, 'workload': what, 'flags': ' '.join(flags), 'value': j['value'], 'unit': j['unit'],
                        'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'dtype': j['dtype'], 'launch': j['launch'],
                        'pairs_per_step': j['config']['pairs_per_step'], 'input': j['input'],
                        'roofline': {k: j['roofline'].get(k) for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'flops_per_launch',
                                                                         'traffic', 'traffic_source', 'algorithmic_bytes_per_launch', 'profile_files')},
                        'kernels': j['kernels'],
                        'roofline_ls': {k: (j.get('roofline_ls') or {}).get(k) for k in ('achieved', 'frac', 'traffic', 'traffic_source', 'avg_launch_ms', 'algorithmic_frac', 'achieved_is')},
                        'parity_check': j['parity_check'], 'split_engine_range_guard': j.get('split_engine_range_guard'),
                        'wall_s': round(time.perf_counter() - t0, 1)})
        except Exception as e:                      # a failed side measurement must never take the headline line down
            res.append({'config': name, 'workload': what, 'error': repr(e)})
    return res


if __name__ == '__main__':
    main()
