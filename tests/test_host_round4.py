"""CPU tests added in round 4 (no GPU): host-only logic of the library and the bench's launch path.

  * csi_pilot_classify: the decomposition P = D1 Pi1 H Pi2 D2 that lets Hadamard-equivalent pilot matrices take the
    Walsh-Hadamard LS kernel - checked by re-assembling P from the returned tables and by running the oracle's LS through them;
  * the RCCL-not-found path returns an error with text (it used to pass NULL to std::string: ADVICE round 3);
  * `bench.py --gpus 8 --rendezvous-only`.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P_VHT4 = np.array([[1, -1, 1, 1], [1, 1, -1, 1], [1, 1, 1, -1], [-1, 1, 1, 1]], np.float64)


def _reassemble(oracle, nt, sym_src, out_row):
    """P from the tables: P[j, s] = rs[j] H[sigma(j), tau(s)] cs[s]."""
    H = oracle.hadamard(nt)
    tau = np.empty(nt, int); cs = np.empty(nt)
    sigma = np.empty(nt, int); rs = np.empty(nt)
    for u in range(nt):
        s = int(sym_src[u]) & 255
        tau[s], cs[s] = u, (-1.0 if sym_src[u] & 256 else 1.0)
        j = int(out_row[u]) & 255
        sigma[j], rs[j] = u, (-1.0 if out_row[u] & 256 else 1.0)
    return rs[:, None] * H[np.ix_(sigma, tau)] * cs[None, :]


@pytest.mark.parametrize('nt', [2, 4, 8, 16, 32, 64, 128])
def test_pilot_classify_signed_permutations_of_sylvester(pkg, oracle, nt):
    from dl_channel_estimation_mamimo_amd.engine import classify_pilot
    rng = np.random.default_rng(nt)
    H = oracle.hadamard(nt)
    kind, a, b = classify_pilot(H)
    assert kind == 1 and np.array_equal(a, np.arange(nt)) and np.array_equal(b, np.arange(nt))
    cases = [rng.choice([-1.0, 1.0], nt)[:, None] * H[rng.permutation(nt)][:, rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[None, :] for _ in range(4)]
    cases += [-H, H[::-1].copy(), H[:, ::-1].copy()]
    if nt >= 4:
        cases.append(np.kron(oracle.hadamard(nt // 4), P_VHT4) if nt > 4 else P_VHT4)
        cases.append(np.kron(P_VHT4, oracle.hadamard(nt // 4)) if nt > 4 else P_VHT4.T.copy())
    for P in cases:
        kind, a, b = classify_pilot(P)
        assert kind in (1, 2), 'a signed permutation of the Sylvester matrix was not recognised'
        assert kind == 2 or np.array_equal(P, H)
        assert sorted(a & 255) == list(range(nt)) and sorted(b & 255) == list(range(nt))
        assert np.array_equal(_reassemble(oracle, nt, a, b), P)


def test_pilot_classify_tables_reproduce_the_ls_estimate(pkg, oracle):
    """What the PERM kernel does with the tables, in numpy: fetch symbol tau^-1(u) with its sign, FWHT in Sylvester order, store row
    r to antenna sigma^-1(r) with its sign - equals the oracle's LS (helperMIMOChannelEstimate.m:24-36) for that P."""
    from dl_channel_estimation_mamimo_amd.engine import classify_pilot
    rng = np.random.default_rng(3)
    nt, nr = 16, 2
    P = np.kron(oracle.hadamard(nt // 4), P_VHT4)
    ltf, _ = oracle.make_structured_packets(rng, 2, nr, P, snr_db=15.0)
    kind, a, b = classify_pilot(P)
    assert kind == 2
    spectra = oracle.ofdm_demod(ltf, nt)                     # [npkt, nr, 234, symbol]
    G = np.stack([spectra[..., int(a[u]) & 255] * (-1.0 if a[u] & 256 else 1.0) for u in range(nt)], axis=-1)
    W = np.einsum('ru,pxku->pxrk', oracle.hadamard(nt), G)      # FWHT over the symbol index -> [npkt, nr, row, 234]
    out = np.empty_like(W)
    for r in range(nt):
        out[:, :, int(b[r]) & 255] = W[:, :, r] * (-1.0 if b[r] & 256 else 1.0)
    den = nt * oracle.vht_ltf_256()[oracle.data_carrier_indices() - 1]
    ref = oracle.ls_estimate(ltf, P)
    assert np.allclose(out / den, ref, rtol=1e-12, atol=1e-12)


def test_pilot_classify_rejects_what_is_not_equivalent(pkg, oracle):
    from dl_channel_estimation_mamimo_amd.engine import classify_pilot
    rng = np.random.default_rng(9)
    H = oracle.hadamard(16)
    assert classify_pilot(rng.choice([-1.0, 1.0], (16, 16)))[0] == 0               # not Hadamard
    assert classify_pilot(0.5 * H)[0] == 0                                        # entries must be +-1
    assert classify_pilot(rng.standard_normal((16, 16)))[0] == 0
    assert classify_pilot(oracle.hadamard(8)[:6, :6].copy())[0] == 0              # order not a power of two
    Q = H.copy(); Q[3, 5] *= -1                                                   # one flipped entry
    assert classify_pilot(Q)[0] == 0
    D = H.copy(); D[7] = D[6]                                                     # a repeated row
    assert classify_pilot(D)[0] == 0
    # a Hadamard matrix from Sylvester's by switching a closed quadruple: whatever the answer, it must be a VERIFIED one
    S = H.copy(); S[np.ix_([0, 4, 8, 12], [0, 1, 2, 3])] *= -1
    assert np.allclose(S @ S.T, 16 * np.eye(16))
    kind, a, b = classify_pilot(S)
    assert kind in (0, 2)
    if kind == 2:
        assert np.array_equal(_reassemble(oracle, 16, a, b), S)
    lib = pkg.load_library()
    assert lib.csi_pilot_classify(None, 16, None, None) == -1
    big = oracle.hadamard(256).astype(np.float32)
    assert classify_pilot(big)[0] == 0                                            # beyond the Walsh-Hadamard kernels' range


def test_rccl_missing_library_is_an_error_not_a_crash():
    """with no librccl the loader called dlerror() twice and passed NULL to std::string (SIGSEGV).
    CSI_RCCL_ONLY=1 restricts the search to CSI_RCCL_LIBRARY, so a host without RCCL can be played here."""
    code = ("import dl_channel_estimation_mamimo_amd as pkg\n"
            "from dl_channel_estimation_mamimo_amd.engine import get_unique_id\n"
            "try:\n    get_unique_id()\n    print('NO ERROR')\n"
            "except pkg.CsiError as e:\n    print('CSIERR', e.code, str(e))\n"
            "try:\n    get_unique_id()\nexcept pkg.CsiError as e:\n    print('AGAIN', e.code)\n")
    env = dict(os.environ, CSI_RCCL_LIBRARY='/nonexistent/librccl-missing.so', CSI_RCCL_ONLY='1', PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r.returncode == 0, r.stdout                                            # 139 before the fix
    assert 'CSIERR -3' in r.stdout and 'RCCL not found' in r.stdout and 'librccl-missing' in r.stdout, r.stdout
    assert 'AGAIN -3' in r.stdout                                                 # the failure is remembered, not retried into a crash


def test_bench_gpus_8_rendezvous_only():
    """`bench.py --gpus 8 --rendezvous-only`: eight ranks start, meet, shard the packets and leave - the launch path of the
    driver's 8-GPU run without GPU work (gloo)."""
    env = dict(os.environ, CSI_DIST_BACKEND='gloo', PYTHONPATH=REPO)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # (round 5: no retry here any more - the launcher itself starts the ranks again on a fresh port when the rendezvous fails, and says so)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--rendezvous-only'], env=env, cwd=REPO,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['rendezvous_only'] and out['n_gpus'] == 8 and out['ranks_seen'] == 8 and out['requested'] == 8
    assert [rk['rank'] for rk in out['ranks']] == list(range(8)) and len({rk['pid'] for rk in out['ranks']}) == 8
    assert out['scaling'] == 'weak' and out['packets_per_step'] == 8 * out['ranks'][0]['packets']
    # round-4 verdict, next 4: an 8-rank run of the driver's command also schedules BASELINE.json's two 8-GPU configurations
    legs = {l['config']: l for l in out['other_configs_scheduled']}
    assert set(legs) == {'configs[3]', 'configs[4]'} and all(l['fits'] for l in legs.values())
    assert '--scaling strong --nt 64 --nr 8 --packets 50000 --input white' in legs['configs[3]']['flags']
    assert '--nt 128 --nr 16 --packets 100000' in legs['configs[4]']['flags'] and '--graph' in legs['configs[4]']['flags']
    assert 150 < legs['configs[4]']['per_rank_gb'] < 0.8 * 288          # one GPU's share of configs[4] is resident in its 288 GB


def test_native_crc32c_matches_the_table_walk_and_known_answers(pkg):
    """csi_crc32c (SSE4.2 inside the library, ADVICE round 3: the pure-Python walk needs ~30 s for a Nt = 32 SavedModel) against the
    RFC 3720 known answers and against the Python implementation, chained calls included."""
    from dl_channel_estimation_mamimo_amd import keras_files as k
    assert k._crc32c_native(), 'the library must export csi_crc32c'
    assert k.crc32c(b'123456789') == 0xE3069283 == k.crc32c(b'123456789', native=False)
    assert k.crc32c(bytes(32)) == 0x8A9136AA and k.crc32c(b'\xff' * 32) == 0x62A8AB43
    rng = np.random.default_rng(4)
    for n in (0, 1, 7, 8, 9, 63, 4097):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert k.crc32c(d) == k.crc32c(d, native=False), n
        h = n // 3
        assert k.crc32c(d[h:], k.crc32c(d[:h])) == k.crc32c(d), n                  # continuing value
    a = np.arange(1000, dtype=np.float32)
    assert k.crc32c(a) == k.crc32c(a.tobytes(), native=False)                      # array-likes


def test_unique_id_file_carries_the_launch_token(pkg, tmp_path, monkeypatch):
    """dist.exchange_unique_id without torch.distributed: the file is accepted only with THIS launch's token -
    a leftover of another launch is ignored however fresh it is, and the file is private (0600)."""
    import hashlib
    import stat
    import struct
    from dl_channel_estimation_mamimo_amd import dist, engine
    path = tmp_path / 'id'
    monkeypatch.setenv('CSI_RCCL_ID_FILE', str(path))
    monkeypatch.setenv('CSI_RCCL_ID_TOKEN', 'launch-A')
    monkeypatch.delenv('TORCHELASTIC_RUN_ID', raising=False)
    monkeypatch.setattr(engine, 'get_unique_id', lambda: bytes(range(128)))
    uid = dist.exchange_unique_id(0, 2)
    blob = path.read_bytes()
    assert uid == bytes(range(128)) and blob[:32] == hashlib.sha256(b'launch-A').digest() and blob[40:] == uid and len(blob) == 168
    assert abs(struct.unpack('<d', blob[32:40])[0] - __import__('time').time()) < 60
    assert stat.S_IMODE(os.stat(path).st_mode) == 0o600
    assert dist.exchange_unique_id(1, 2, timeout_s=1.0) == uid
    monkeypatch.setenv('CSI_RCCL_ID_TOKEN', 'launch-B')                            # another launch finds launch A's file
    with pytest.raises(RuntimeError):
        dist.exchange_unique_id(1, 2, timeout_s=0.3)
    os.utime(path, (1, 1))                                                           # an OLD file of the right launch is fine (late rank)
    monkeypatch.setenv('CSI_RCCL_ID_TOKEN', 'launch-A')
    assert dist.exchange_unique_id(3, 4, timeout_s=1.0) == uid


def test_unique_id_file_without_a_launch_token_wants_a_fresh_file(pkg, tmp_path, monkeypatch):
    """without CSI_RCCL_ID_TOKEN / TORCHELASTIC_RUN_ID the tag is the rendezvous triple, the same for
    every launch on one port - a file left by a KILLED launch (atexit never ran) has a matching tag.  Such a
    leftover must not be handed to a non-root rank: it is refused by its age (the time rank 0 stamped INTO the file - not the mtime),
    the reader keeps polling and takes the file rank 0 of its own launch writes meanwhile.  Rank 0's exit handler removes only what
    it wrote itself."""
    import hashlib
    import struct
    import threading
    import time
    from dl_channel_estimation_mamimo_amd import dist, engine
    path = tmp_path / 'id'
    monkeypatch.setenv('CSI_RCCL_ID_FILE', str(path))
    for k in ('CSI_RCCL_ID_TOKEN', 'TORCHELASTIC_RUN_ID'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', '29511')
    tag = hashlib.sha256(b'127.0.0.1:29511:2').digest()
    stale = bytes([7]) * 128
    path.write_bytes(tag + struct.pack('<d', time.time() - 3600.0) + stale)         # an hour-old leftover with the RIGHT tag, fresh mtime
    with pytest.raises(RuntimeError, match='CSI_RCCL_ID_TOKEN'):
        dist.exchange_unique_id(1, 2, timeout_s=0.4)
    # the same leftover, and rank 0 of this launch arrives half a second later: the reader returns rank 0's id, never the stale one
    fresh = bytes(range(128))
    monkeypatch.setattr(engine, 'get_unique_id', lambda: fresh)
    got = {}
    t = threading.Thread(target=lambda: got.update(uid=dist.exchange_unique_id(1, 2, timeout_s=10.0)))
    t.start()
    time.sleep(0.5)
    assert dist.exchange_unique_id(0, 2) == fresh
    t.join(20)
    assert got.get('uid') == fresh
    # a file that carries only the old layout (tag + id, no stamp) is never accepted
    path.write_bytes(tag + fresh)
    with pytest.raises(RuntimeError):
        dist.exchange_unique_id(1, 2, timeout_s=0.3)
    # rank 0's exit handler leaves a file alone that is no longer its own
    import atexit
    calls = []
    monkeypatch.setattr(atexit, 'register', lambda fn: calls.append(fn))
    dist.exchange_unique_id(0, 2)
    other = tag + struct.pack('<d', time.time()) + bytes([9]) * 128
    path.write_bytes(other)                                                          # a later launch owns the path now
    calls[-1]()
    assert path.read_bytes() == other
    dist.exchange_unique_id(0, 2)
    calls[-1]()
    assert not path.exists()


def test_unique_id_file_is_found_by_ranks_with_different_parents(tmp_path):
    """ranks of a hand-rolled launch need not share a parent process (separate shells, ssh / docker exec
    sessions, `bash -c` wrappers, service units).  Without a launch token, path and tag derive from the rendezvous triple alone, so
    rank 1 - started here behind an extra `bash -c` + `sh -c`, i.e. with another parent than rank 0's - finds rank 0's file in the
    DEFAULT location.  (Stale files are handled by the stamp inside the file: the test above.)"""
    code = ("import sys, os; sys.path.insert(0, %r); import dl_channel_estimation_mamimo_amd as pkg; "
            "from dl_channel_estimation_mamimo_amd import dist, engine; engine.get_unique_id = lambda: bytes(range(128)); "
            "r = int(sys.argv[1]); uid = dist.exchange_unique_id(r, 2, timeout_s=30.0); "
            "import time; time.sleep(1.5 if r == 0 else 0); print('rank', r, 'ppid', os.getppid(), uid.hex()[:16])") % REPO
    env = dict(os.environ, XDG_RUNTIME_DIR=str(tmp_path), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + os.getpid() % 200))
    for k in ('CSI_RCCL_ID_TOKEN', 'TORCHELASTIC_RUN_ID', 'CSI_RCCL_ID_FILE', 'RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    script = tmp_path / 'rank.py'
    script.write_text(code)
    p1 = subprocess.Popen(['bash', '-c', 'sh -c "%s %s 1"; true' % (sys.executable, script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    p0 = subprocess.Popen([sys.executable, str(script), '0'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    o0, e0 = p0.communicate(timeout=120)
    o1, e1 = p1.communicate(timeout=120)
    assert p0.returncode == 0 and 'rank 0' in o0, e0[-2000:]
    assert 'rank 1' in o1, (o1, e1[-2000:])
    f0, f1 = o0.split(), o1.split()
    assert f0[3] != f1[3], 'the two ranks were meant to have different parents'
    assert f0[4] == f1[4] == bytes(range(128)).hex()[:16]


def test_integration_doc_indexes_every_entry_point_of_the_header():
    """INTEGRATION.md 9 names every symbol include/csi_mamimo.h declares (written out or as `csi_x` / `_y` shorthand inside one cell),
    and names nothing the header does not declare - the drop-in boundary and its documentation cannot drift apart."""
    import re
    with open(os.path.join(REPO, 'include', 'csi_mamimo.h')) as f:
        declared = set(re.findall(r'\b(csi_[a-z0-9_]+)\(', f.read()))
    with open(os.path.join(REPO, 'INTEGRATION.md')) as f:
        doc = f.read()
    sec = doc[doc.index('## 9. Entry-point index'):]
    named, unknown = set(), []
    for row in sec.splitlines():
        if not row.startswith('| `csi_'):
            continue
        cell = row.split('|')[1]
        base = None
        for tok in re.findall(r'`([a-z0-9_]+)`', cell):
            if tok.startswith('csi_'):
                base = tok
                name = tok
            else:                               # `_step` behind `csi_train_begin`: replace trailing components of the last full name
                parts = base.split('_')
                name = next(('_'.join(parts[:k]) + tok for k in range(len(parts) - 1, 0, -1) if '_'.join(parts[:k]) + tok in declared), None)
            (named.add(name) if name in declared else unknown.append((tok, name)))
    assert not unknown, unknown
    assert declared - named == set(), sorted(declared - named)


def test_host_pipeline_pools_cover_every_range_once(tmp_path):
    """tests/hostpool_check.cpp: the host pipeline's thread pools (one range on one pool, one range on two pools at once - the input
    staging of calls with pinned result arrays) visit every element exactly once for awkward sizes / alignments / thread counts, and
    the complex128 -> planes and planes -> complex64 loops driven through them give the scalar loops' bits.  Host code only."""
    exe = str(tmp_path / 'hostpool_check')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    res = subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', '-std=c++17', '-Wno-unused-value', '-pthread',
                          os.path.join(REPO, 'tests', 'hostpool_check.cpp'), '-o', exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout
    # every alignment case of the complex128 split / complex64 weave / streaming copy under each SIMD choice the CPU offers
    # (CSI_HOST_SIMD: 0 scalar, 2 AVX2, 5 AVX-512 with 64-byte streaming stores)
    for cap in ('0', '2', '5', '5'):
        run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=120, env=dict(os.environ, CSI_HOST_SIMD=cap))
        assert run.returncode == 0 and 'hostpool_check: ok' in run.stdout and 'simd: cap %s,' % cap in run.stdout, run.stdout


def test_pinned_pool_recycles_buffers_behind_fresh_arrays(pkg):
    """engine.PinnedPool (the memory behind estimate(..., pinned_results=True) / CSIPredictor(pinned_results=True)): every take() is a
    fresh array; its buffer returns to the pool only when the array AND every view of it are gone; the next take of that size reuses it
    (a serving loop pins once); other sizes allocate; the idle cap drops buffers instead of hoarding them.  Fake allocator: no GPU."""
    import ctypes
    import gc
    from dl_channel_estimation_mamimo_amd.engine import PinnedPool
    made = []

    def alloc(n):
        buf = (ctypes.c_char * n)()
        made.append(ctypes.addressof(buf))
        return buf

    pool = PinnedPool(alloc, max_idle_bytes=1 << 20)
    a = pool.take((4, 3, 5), np.complex64)
    assert a.shape == (4, 3, 5) and a.dtype == np.complex64 and a.flags['C_CONTIGUOUS'] and pool.allocated == 1
    addr_a = a.ctypes.data
    a[...] = 1 + 2j
    b = pool.take((4, 3, 5), np.complex64)                 # `a` is alive: a second buffer
    assert b.ctypes.data != addr_a and pool.allocated == 2
    view = a[1:3, :, ::2]                                  # a view keeps the buffer out of the pool after `a` itself is dropped
    del a
    gc.collect()
    assert pool.idle_bytes == 0
    c = pool.take((4, 3, 5), np.complex64)
    assert c.ctypes.data not in (addr_a, b.ctypes.data) and pool.allocated == 3
    assert np.all(view == 1 + 2j)                          # nobody overwrote what the view still shows
    del view
    gc.collect()
    assert pool.idle_bytes == 4 * 3 * 5 * 8 and pool.reused == 0
    d = pool.take((60,), np.complex64)                     # same byte count, another shape: the recycled buffer
    assert d.ctypes.data == addr_a and pool.reused == 1 and pool.allocated == 3 and pool.idle_bytes == 0
    e = pool.take((7,), np.float32)                        # another size allocates
    assert pool.allocated == 4
    big = pool.take((1 << 18,), np.complex64)              # 2 MiB > the idle cap: dropped on release, not kept
    del big, b, c, d, e
    gc.collect()
    assert pool.idle_bytes <= 1 << 20 and pool.idle_bytes == 3 * 480 + 28          # b, c, d (480 B each) and e; `big` was dropped
    pool.clear()
    assert pool.idle_bytes == 0
    assert len(made) == 5


def test_engine_objects_hold_no_reference_cycle(pkg):
    """A CsiEngine must be freed by its last reference (the suites create hundreds; device memory held until a gc pass would pile
    up): its members may not point back at it.  Checked on the class without a device: every attribute __init__ sets that could hold
    a callable is built from a weak reference."""
    import inspect
    from dl_channel_estimation_mamimo_amd import engine
    src = inspect.getsource(engine.CsiEngine.__init__)
    assert 'PinnedPool(self.' not in src and 'weakref.ref(self)' in src
    # and the pool itself keeps nothing alive but idle buffers
    import gc
    import weakref

    class Owner:
        def __init__(self):
            me = weakref.ref(self)
            self.pool = engine.PinnedPool(lambda n: me().alloc(n))

        def alloc(self, n):
            return bytearray(n)

    o = Owner()
    a = o.pool.take((8,), np.float32)
    w = weakref.ref(o)
    gc.disable()
    try:
        del o
        assert w() is None                              # freed by reference counting alone, with an array of its pool still alive
    finally:
        gc.enable()
    assert a.shape == (8,)


def test_host_pipeline_on_a_model_of_the_stream_semantics(tmp_path):
    """tests/hostpipe_mock_check.cpp: hp_run (stager / drainer threads, two pinned + two device slots, three streams chained by
    events) and both host pipelines on top of it run 688 calls (round 5: the complex64 entry point from pageable and pinned input joined them) - side threads and inline, several thread counts, slot sizes from 7
    packets to one chunk, both estimators / either alone, pageable and pinned result arrays (the launched weave kernel as a stream item) - against a small model of the HIP stream semantics (FIFO streams on their own
    threads, events as positions) with arithmetic stand-ins for the kernels: every result exact.  (Under -fsanitize=thread the same
    binary shows no race, and a pipeline with one event dependency removed shows races and 11 wrong results: tools/sanitize_host.sh.)"""
    exe = str(tmp_path / 'hostpipe_mock_check')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    res = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-Wno-unused-value', '-pthread',
                          os.path.join(REPO, 'tests', 'hostpipe_mock_check.cpp'), '-o', exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout[-3000:]
    for _ in range(2):
        run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
        assert run.returncode == 0 and 'hostpipe_mock_check: ok' in run.stdout and '688 pipelined calls' in run.stdout, run.stdout[-3000:]


def test_weight_broadcast_by_n_ranks_on_the_stream_and_rccl_models(tmp_path):
    """tests/comm_mock_check.cpp: csi_comm_init / csi_broadcast_weights run by 2, 4 and 8 RANKS (threads) on one CPU - the library's whole
    translation unit against a model of the HIP runtime (tests/mock_hip.hpp) and of RCCL (ranks meet inside ncclBroadcast / ncclAllReduce
    in call order; a rank that skips a collective its peers enter is reported as a hang).  Every receiver ends with the root's device
    buffers byte for byte (root 0 and root != 0, three network shapes incl. bf16); a receiver built for another network is refused
    THERE with the reason and reported on EVERY other rank, nobody waits, the root keeps its model; a second broadcast on the same
    communicators; a root with nothing loaded; argument checks.  22 scenarios, ~3000 collectives."""
    exe = str(tmp_path / 'comm_mock_check')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    res = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-Wno-unused-value', '-pthread',
                          os.path.join(REPO, 'tests', 'comm_mock_check.cpp'), '-o', exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout[-3000:]
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=400, env=dict(os.environ, COMM_MOCK_WATCHDOG_S='120'))
    assert run.returncode == 0 and 'comm_mock_check: ok' in run.stdout and '22 broadcast scenarios' in run.stdout, run.stdout[-3000:]


def test_every_option_of_the_library_is_documented_in_the_header():
    """Every name csi_set_option / csi_get_option accept appears (quoted) in include/csi_mamimo.h, and everything that can be set can be
    read back (the write-only development switch "ls_debug" aside) - the option surface and its documentation cannot drift apart."""
    import re
    with open(os.path.join(REPO, 'dl-channel-estimation-mamimo_amd', 'csrc', 'csi_mamimo.hip')) as f:
        src = f.read()
    with open(os.path.join(REPO, 'include', 'csi_mamimo.h')) as f:
        documented = set(re.findall(r'"([a-z0-9_]+)"', f.read()))

    def names(fn):
        i = src.index('int %s(' % fn)
        return set(re.findall(r'n == "([a-z0-9_]+)"', src[i:src.index('\n}\n', i)]))

    settable, gettable = names('csi_set_option'), names('csi_get_option')
    assert len(settable) >= 20 and len(gettable) >= 40
    assert settable - gettable == {'ls_debug'}
    assert (settable | gettable) - documented == set(), sorted((settable | gettable) - documented)


@pytest.fixture(scope='module')
def mock_so(tmp_path_factory):
    """tests/mock_library.cpp built once: the library's translation unit on the model of the HIP runtime (kernels dropped)."""
    so = str(tmp_path_factory.mktemp('mocklib') / 'libcsi_mock.so')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    res = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value', '-pthread',
                          os.path.join(REPO, 'tests', 'mock_library.cpp'), '-o', so], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout[-3000:]
    return so


def test_python_engine_on_the_mock_runtime_pinned_result_pool(mock_so):
    """The Python layer on a machine without a GPU: tests/mock_library.cpp (the library's translation unit on the model of the HIP
    runtime; kernels dropped) loaded in place of the product library in a child process.  CsiEngine.estimate(pinned_results=True) takes
    its result arrays from the recycling pool of pinned buffers (csi_host_malloc), the library downloads straight into them
    (hp_direct_out_calls), a collected result's buffer is reused by the next call, mixing pinned and pageable arrays falls back to the
    host weave, close() leaves nothing behind."""
    so = mock_so
    script = r'''
import gc, sys
import numpy as np
sys.path.insert(0, %r)
import dl_channel_estimation_mamimo_amd as pkg
from dl_channel_estimation_mamimo_amd import _lib
_lib._SO = %r
nt, nr, hidden, npkt = 8, 2, (64, 64), 50
e = pkg.CsiEngine(nt, nr, hidden=hidden)
rng = np.random.default_rng(0)
w = pkg.synth.make_weights(rng, nt, hidden)
e.load_weights('real', w); e.load_weights('imag', w); e.set_pilot(pkg.synth.hadamard(nt))
x = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
a, b = e.estimate(x)
assert a.shape == b.shape == (npkt, nr, nt, 234) and a.dtype == np.complex64 and e.get_option('hp_direct_out_calls') == 0
c, d = e.estimate(x, pinned_results=True)
assert c.shape == a.shape and c.dtype == np.complex64 and c.flags['C_CONTIGUOUS']
assert e.get_option('hp_direct_out_calls') == 1 and (e.result_pool.allocated, e.result_pool.reused) == (2, 0)
addr = {c.ctypes.data, d.ctypes.data}
c2, d2 = e.estimate(x, pinned_results=True)                      # the first results are alive: fresh buffers
assert {c2.ctypes.data, d2.ctypes.data}.isdisjoint(addr) and e.result_pool.allocated == 4
del c, d
gc.collect()
assert e.result_pool.idle_bytes == 2 * c2.nbytes
c3, _ = e.estimate(x, ls=False, pinned_results=True)             # a collected result's buffer serves the next call
assert c3.ctypes.data in addr and e.result_pool.reused == 1 and e.get_option('hp_direct_out_calls') == 3
mixed = (e.pinned_empty(a.shape, np.complex64), np.empty(a.shape, np.complex64))
e.estimate(x, out=mixed)
assert e.get_option('hp_direct_out_calls') == 3                    # one pageable array: host weave for both
e.set_option('hp_device_weave', 0)
e.estimate(x, pinned_results=True)
assert e.get_option('hp_direct_out_calls') == 3
try:
    e.estimate(x[:, :1])
    raise SystemExit('a wrong shape was accepted')
except pkg.CsiError:
    pass
# round 5: a complex64 batch is not widened - csi_estimate_c64 (pageable: staged by a plain copy; pinned: DMA'd as it is)
x64 = x.astype(np.complex64)
e.set_option('hp_device_weave', 1)
n_direct = e.get_option('hp_direct_out_calls')
a64, b64 = e.estimate(x64)
assert a64.shape == a.shape and a64.dtype == np.complex64 and b64.shape == b.shape
xp = e.pinned_empty(x64.shape, np.complex64); xp[...] = x64
c64, _ = e.estimate(xp, ls=False, pinned_results=True)
assert c64.shape == a.shape and e.get_option('hp_direct_out_calls') == n_direct + 1
try:
    e.estimate(x64[:, :1])
    raise SystemExit('a wrong complex64 shape was accepted')
except pkg.CsiError:
    pass
arr = e.to_device(np.arange(6, dtype=np.float32))
e.close()
assert arr.ptr == 0 and e.result_pool.idle_bytes == 0
print('mock-runtime engine: ok')
''' % (REPO, so)
    run = subprocess.run([sys.executable, '-c', script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert run.returncode == 0 and 'mock-runtime engine: ok' in run.stdout, run.stdout[-3000:]


def test_bench_script_runs_end_to_end_on_the_mock_runtime(mock_so, tmp_path):
    """bench.py itself, every leg a one-GPU run has except the fresh-process side configs - warm-up, timed region with kernel events,
    the steps again without them, host path incl. the pinned-result leg and the link probe, oracle check, CPU baseline, latency loop,
    practical peak, next rows - on the mock runtime in a child process: the numbers mean nothing (kernels are dropped), but every line of
    the script executes, the contract's keys are there and no side measurement reports an exception."""
    runner = ("import sys, runpy; sys.path.insert(0, %r); from dl_channel_estimation_mamimo_amd import _lib; _lib._SO = %r; "
              "sys.argv = ['bench.py'] + sys.argv[1:]; runpy.run_path(%r, run_name='__main__')") % (REPO, mock_so, os.path.join(REPO, 'bench.py'))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', runner, '--packets', '64', '--steps', '2', '--warmup', '1', '--host-path', '64', '--no-other-configs',
                        '--cpu-budget-s', '2', '--detail-file', str(tmp_path / 'bench_detail_mock.json')], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    raw = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    # round-5 verdict, next 1: the line is a record the driver can parse (BENCH_r05.parsed was null for a 20 KB line) - below 4 KB,
    # stderr quiet enough to share an 8 KB tail with it, everything else in the detail file it names
    assert len(raw) < 4096, len(raw)
    assert len(raw) + len(r.stderr) < 8000, (len(raw), len(r.stderr), r.stderr[-2000:])
    line = json.loads(raw)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'roofline_ls', 'cpu_baseline', 'parity_check', 'latency_us', 'ranks_ms', 'host_path', 'next_rows', 'regimes_us', 'detail_file', 'bench_wall_s'):
        assert key in line, key
    assert 'dropped' not in line
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['warmup'] == 1 and line['config']['pairs_per_step'] == 64 * 128
    assert 'workload' in line['config'] and not any(k in line['config'] for k in ('model', 'global_batch', 'seq_len'))
    for key in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'flops_per_launch', 'algorithmic_bytes_per_launch'):
        assert key in line['roofline'], key
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in line['roofline_ls'], key
    cb = line['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'threads', 'host_cpu_count', 'cpu_model', 'kind', 'sample'):
        assert key in cb, key
    # round-5 verdict, weak 6: `cores` = the CPUs the container grants, not torch's thread pick and not the host's count
    assert 0 < cb['cores'] <= os.cpu_count() and cb['host_cpu_count'] == os.cpu_count() and cb['threads'] >= 1
    assert sorted(line['regimes_us']) == ['1', '64', '8'] and all(len(v) == 3 and v[0] > 0 and v[2] > 0 for v in line['regimes_us'].values())
    # the detail file: the whole object, with what the line left out
    with open(os.path.join(REPO, line['detail_file'])) as f:
        detail = json.load(f)
    for key in ('kernels', 'host_path_pcie_inclusive', 'next_rows', 'regimes', 'ranks', 'devices', 'arithmetic', 'input', 'launch'):
        assert key in detail, key
    assert detail['value'] == line['value'] and detail['ms_per_step'] == line['ms_per_step']
    # round-4 verdict, next 3: the batch-size regimes that fit the resident batch (64 packets here), each with its bound
    assert [r_['packets'] for r_ in detail['regimes']] == [1, 8, 64] and all(r_['bound_us'] > 0 and 'bound' in r_ and r_['pipelined_us'] > 0 for r_ in detail['regimes'])
    assert detail['roofline_ls']['achieved_is'] and 'algorithmic_frac' in detail['roofline_ls']
    c128 = detail['host_path_pcie_inclusive']['python_c128_to_c64']
    assert 'dnn_only_pinned_result' in c128 and c128['dnn_only_pinned_result']['direct_downloads'] == 4, c128.keys()
    text = json.dumps(detail)
    assert '_error' not in text and '"error"' not in text, [k for k in ('_error', '"error"') if k in text]


def test_no_option_of_the_product_build_selects_the_two_workgroup_ringb_form(mock_so):
    """Round-4 verdict, weak 7 / next 6.  The bf16-split LS kernel's two-workgroups-per-CU form (Nt <= 32) is the one form with a known
    wrong-result signature (profiles/r04_ls_ringb_variants.txt).  The shipped library does not contain it: whatever "ls_kernel",
    "ls_ringb_min" and "ls_v2" say, kernel 7 is planned ONE workgroup per CU (get "ls_per_cu"), and "ls_overlap_cus" > 0 - the other
    arrangement that put LS workgroups beside foreign MFMA waves - is refused with text.  Runs on the mock runtime (the library's own
    translation unit and option code, kernels dropped)."""
    script = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
import dl_channel_estimation_mamimo_amd as pkg
from dl_channel_estimation_mamimo_amd import _lib
_lib._SO = %r
rng = np.random.default_rng(3)
for nt in (16, 24, 32, 48, 64, 128):
    e = pkg.CsiEngine(nt, 2, hidden=(64, 64))
    P = np.sign(rng.standard_normal((nt, nt)))              # +-1 entries, not Hadamard: one bf16 piece, the generic kernels
    e.set_pilot(P)
    assert e.get_option('ls_pilot_fast') == 0
    for ringb_min in (33, 16, 0):
        e.set_option('ls_ringb_min', ringb_min)
        for v2 in (0, 1):
            e.set_option('ls_v2', v2)
            for forced in (0, 7):
                e.set_option('ls_kernel', forced)
                mode, per_cu = e.get_option('ls_mode'), e.get_option('ls_per_cu')
                if forced == 7 or (nt >= max(ringb_min, 16) and mode == 7):
                    assert mode == 7, (nt, ringb_min, v2, forced, mode)
                if mode == 7:
                    assert per_cu == 1, (nt, ringb_min, v2, forced, per_cu)
    try:
        e.set_option('ls_overlap_cus', 32)
        raise SystemExit('ls_overlap_cus > 0 was accepted by the product build')
    except pkg.CsiError as err:
        assert 'not part of the product build' in str(err), str(err)
    assert e.get_option('ls_overlap_cus') == 0
    e.set_option('ls_overlap_cus', 0)                       # 0 stays settable
    e.close()
print('ringb gating: ok')
''' % (REPO, mock_so)
    run = subprocess.run([sys.executable, '-c', script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert run.returncode == 0 and 'ringb gating: ok' in run.stdout, run.stdout[-3000:]


def test_two_rank_bench_runs_the_multi_gpu_legs_on_the_mock_runtime(mock_so, tmp_path):
    """`torchrun ... bench.py --gpus 2` end to end without a GPU (gloo, the library's translation unit on the mock runtime): the weak-scaled config-2
    headline by two ranks AND - round-4 verdict, next 4 - the configs[3] / configs[4] legs as fresh two-rank jobs started by the
    running ranks (shrunken packet counts; configs[4] as one hipGraph per step and rank).  Numbers mean nothing here (kernels are
    dropped); what is asserted is that the first N > 1 run of the driver records the right workloads with per-rank evidence."""
    env = dict(os.environ, CSI_DIST_BACKEND='gloo', CSI_DEBUG_HOOKS='1', CSI_LIBRARY_PATH=mock_so, PYTHONPATH=REPO)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)
    # launched the driver's way - torch.distributed.run sets RANK / WORLD_SIZE and the TORCHELASTIC_* variables, which the legs'
    # child jobs must not inherit (their rank 0 serves its own store); bench.py's own launcher is covered by the rendezvous-only tests
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--check', '0',
                        '--packets', '64', '--legs-packets', '24,6', '--input', 'white',
                        '--detail-file', str(tmp_path / 'bench_detail_2rank.json')], env=env, cwd=REPO,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    raw = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    line = json.loads(raw)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['pairs_per_step'] == 2 * 64 * 128      # (64 packets per rank: the mock's allocator fills every buffer)
    # round-5 verdict, next 1: the N-rank line with both legs stays a record - below 4 KB at 2 ranks AND extrapolated to 8 (what grows
    # with the rank count is `ranks_ms` of the headline and of each leg; everything per-rank beyond that lives in the detail file)
    per_rank = (len(json.dumps(line['ranks_ms'])) + sum(len(json.dumps(l.get('ranks_ms'))) for l in line['other_configs'])) / 2.0
    assert len(raw) < 4096 and len(raw) + 6 * (per_rank + 4) < 4096, (len(raw), per_rank)
    assert 'dropped' not in line and len(line['ranks_ms']) == 2
    for key in ('roofline', 'parity_check', 'detail_file', 'bench_wall_s'):
        assert key in line, key
    brief = {l['config']: l for l in line['other_configs']}
    assert set(brief) == {'configs[3]', 'configs[4]'} and all(l['n_gpus'] == 2 and l['ms_per_step'] > 0 and len(l['ranks_ms']) == 2 for l in brief.values()), line['other_configs']
    with open(os.path.join(REPO, line['detail_file'])) as f:
        detail = json.load(f)
    assert len(detail['ranks']) == 2 and len(detail['devices']) == 2
    legs = {l['config']: l for l in detail['other_configs']}
    assert set(legs) == {'configs[3]', 'configs[4]'}, detail['other_configs']
    for name, (nt, nr, total) in {'configs[3]': (64, 8, 24), 'configs[4]': (128, 16, 6)}.items():
        l = legs[name]
        assert 'error' not in l and 'skipped' not in l, l
        assert l['n_gpus'] == 2 and l['scaling'] == 'strong' and l['pairs_per_step'] == total * nt * nr
        assert l['packets_per_rank'] == [total // 2, total // 2] and len(l['ranks_ms']) == 2 and len(l['devices']) == 2
        assert 'weights_via' in l and 'sharding' in l and 'roofline' in l and l['ms_per_step'] > 0
    assert 'hipGraph' in legs['configs[4]']['launch'] and 'eager' in legs['configs[3]']['launch']


def test_no_packed_fp32_instruction_with_a_cross_half_second_source_in_the_built_library():
    """Round 6 (profiles/r06_pk_opsel_probe.txt): on gfx950 a v_pk_add/mul/fma_f32 whose SECOND source takes its low half from the high
    register (op_sel = [0, 1, ..]) loses that operand in lanes 48-63 while another wave of the SIMD issues MFMAs - the LS transform's +-i rotations were
    such instructions and came back wrong beside a bf16 GEMM.  The compiler writes the form on its own for complex arithmetic, so the check is on the
    machine code of the built library (every kernel, the generated assembly ones included), not on the sources."""
    import shutil
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import opsel_census
    if not os.path.exists(os.path.join(opsel_census.LLVM, 'llvm-objdump')):
        pytest.skip('no llvm-objdump in this image')
    import dl_channel_estimation_mamimo_amd as pkg
    so = pkg.build_library(force=False)
    total, found = opsel_census.census(so)
    assert total > 1000, 'the disassembly found no packed-fp32 instructions at all: the census is broken'
    assert not found, 'packed-fp32 instructions with a cross-half second source: %s' % found[:5]
    # and the census does see the form where it exists
    assert opsel_census.vulnerable('v_pk_add_f32 v[4:5], v[8:9], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]')
    assert not opsel_census.vulnerable('v_pk_fma_f32 v[4:5], v[8:9], v[12:13], v[2:3] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]')
    assert not opsel_census.vulnerable('v_pk_mul_f32 v[4:5], v[8:9], v[12:13] op_sel_hi:[1,0]')
