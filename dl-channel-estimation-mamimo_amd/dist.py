"""Multi-GPU plumbing: one process per GPU, packets sharded by index, weights broadcast once.

The path has no steady-state exchange: every packet is independent and the weights are
read-only.  The only collective is the load-time broadcast of the weight blob
and the pilot matrix from rank 0 - ``torch.distributed`` with the ``nccl`` backend, which is
RCCL over xGMI on ROCm (``gloo`` on CPU-only hosts, used by the tests).

Process order matters on ROCm: call ``init_process_group`` (which imports torch and selects the GPU)
BEFORE the first ``CsiEngine`` is created - see ``_lib.load_library`` for why; with WORLD_SIZE > 1
in the environment ``load_library`` enforces it."""
import os
import time
import numpy as np

_T_PROCESS_START = time.time()        # (module import: early in the life of a rank) - freshness reference of exchange_unique_id


def env_rank_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def shard_range(n, rank, world):
    """Contiguous packet range [lo, hi) of this rank; the first n % world ranks get one extra."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    rank, world, local = env_rank_world()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    # CSI_DIST_TIMEOUT_S bounds the rendezvous and every later collective of the group (torch's default is 10-30 minutes): bench.py's
    # N > 1 side legs set it, so that a leg whose ranks cannot meet costs two minutes and an error entry, not the headline's line
    kw = {}
    if os.environ.get('CSI_DIST_TIMEOUT_S'):
        import datetime
        kw['timeout'] = datetime.timedelta(seconds=float(os.environ['CSI_DIST_TIMEOUT_S']))
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def broadcast_weights(weights, src=0, device=None):
    """Broadcast a {name: float32 ndarray} dict from ``src`` to every rank as ONE flat buffer
    (a single large collective instead of one per tensor: xGMI rings are per-link bound, so
    fewer, larger transfers).  Non-source ranks may pass ``None``; the tensor names and shapes
    travel first as a small object broadcast.  Returns the dict on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return weights
    rank = dist.get_rank()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    meta = [None]
    if rank == src:
        names = [k for k, v in weights.items() if isinstance(v, np.ndarray)]
        meta[0] = [(k, tuple(weights[k].shape)) for k in names]
    dist.broadcast_object_list(meta, src=src)
    total = int(sum(int(np.prod(s)) for _, s in meta[0]))
    if rank == src:
        flat = np.concatenate([np.ascontiguousarray(weights[k], dtype=np.float32).ravel() for k, _ in meta[0]])
        buf = torch.from_numpy(flat).to(device)
    else:
        buf = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    host = buf.cpu().numpy()
    out, off = {}, 0
    for k, s in meta[0]:
        n = int(np.prod(s))
        out[k] = host[off:off + n].reshape(s).copy()
        off += n
    return out


def exchange_unique_id(rank, world, timeout_s=120.0):
    """The RCCL unique id of the library's own communicator (engine.get_unique_id) from rank 0 to every rank.  With
    ``torch.distributed`` initialised it rides on its object broadcast; without it, through a file (``CSI_RCCL_ID_FILE``,
    default <XDG_RUNTIME_DIR or /tmp>/csi_rccl_<uid>/id_<hash of the launch token>, mode 0600) that rank 0 writes atomically
    and removes at exit (if it is still its own) - no torch in the process at all."""
    from .engine import get_unique_id
    try:
        import torch.distributed as tdist
        have_pg = tdist.is_available() and tdist.is_initialized()
    except Exception:
        have_pg = False
    if have_pg:
        box = [get_unique_id() if rank == 0 else None]
        tdist.broadcast_object_list(box, src=0)
        return box[0]
    # The file carries, in front of the 128 id bytes, a 32-byte tag of the launch token and the wall time rank 0 wrote it at.
    # Token: CSI_RCCL_ID_TOKEN (bench.py's own launcher sets a fresh one per launch), else torchrun's TORCHELASTIC_RUN_ID, else -
    # a hand-rolled launch - the rendezvous triple (address, port, world size: what ranks started from separate shells, ssh sessions
    # or service units still agree on; the launcher's pid is NOT part of it).  Only the first two are unique per launch; with the
    # fallback a file left by a KILLED launch (atexit does not run on SIGKILL) can carry the same tag, so there a reader also wants
    # the file to be younger than its own process start minus CSI_RCCL_ID_MAX_AGE_S (default 120 s: ranks of one launch start
    # within that; a rank that is later than that must be given a token) and keeps polling until rank 0 has replaced it.
    # Default location: a directory of this user (mode 0700), the file itself mode 0600.
    import hashlib
    import struct
    strong = os.environ.get('CSI_RCCL_ID_TOKEN') or os.environ.get('TORCHELASTIC_RUN_ID')
    token = (strong or '%s:%s:%d' % (os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '29500'), world)).encode()
    tag = hashlib.sha256(token).digest()                       # 32 bytes in front of the stamp and the id
    max_age = float(os.environ.get('CSI_RCCL_ID_MAX_AGE_S', '120'))
    path = os.environ.get('CSI_RCCL_ID_FILE')
    if not path:
        base = os.path.join(os.environ.get('XDG_RUNTIME_DIR') or '/tmp', 'csi_rccl_%d' % os.getuid())
        os.makedirs(base, mode=0o700, exist_ok=True)
        if os.stat(base).st_uid != os.getuid():
            raise RuntimeError('%s belongs to another user: set CSI_RCCL_ID_FILE' % base)
        path = os.path.join(base, 'id_%s' % hashlib.sha256(token).hexdigest()[:24])
    if rank == 0:
        try:
            os.remove(path)                                    # a stale file of an earlier launch with the same token
        except FileNotFoundError:
            pass
        uid = get_unique_id()
        blob = tag + struct.pack('<d', time.time()) + uid
        tmp = '%s.%d' % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(blob)
        os.replace(tmp, path)

        def _remove_own():                                     # only what THIS launch wrote: a later launch may own the path by now
            try:
                with open(path, 'rb') as f:
                    if f.read() != blob:
                        return
                os.remove(path)
            except OSError:
                pass
        import atexit
        atexit.register(_remove_own)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with open(path, 'rb') as f:
                blob = f.read()
            if len(blob) == 32 + 8 + 128 and blob[:32] == tag:
                if strong or struct.unpack('<d', blob[32:40])[0] >= _T_PROCESS_START - max_age:
                    return blob[40:]
        except FileNotFoundError:
            pass
        time.sleep(0.05)
    raise RuntimeError('no RCCL unique id of this launch at %s after %.0f s%s' % (
        path, timeout_s, '' if strong else ' (no per-launch token: set CSI_RCCL_ID_TOKEN to the same fresh value on every rank)'))


def local_device_count():
    """GPUs visible to this process (ranks are mapped onto them round-robin)."""
    import torch
    return max(torch.cuda.device_count(), 1)


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_objects(obj):
    """Every rank's (small, picklable) object, in rank order, on every rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def device_identity(ordinal=None):
    """What tells two GPUs apart in a bench line: ordinal used by this rank, marketing name, PCI address (when torch
    exposes it) and the gfx architecture."""
    import torch
    if not torch.cuda.is_available():
        return {'ordinal': None, 'name': 'cpu', 'pci': None, 'arch': None}
    d = torch.cuda.current_device() if ordinal is None else int(ordinal)
    pr = torch.cuda.get_device_properties(d)
    pci = None
    if hasattr(pr, 'pci_bus_id'):
        pci = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, getattr(pr, 'pci_device_id', 0))
    return {'ordinal': d, 'name': pr.name, 'pci': pci, 'arch': getattr(pr, 'gcnArchName', None)}


def all_reduce_max(value, device=None):
    """max over ranks of a python float (used for the max-over-ranks step time)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_min(value, device=None):
    """min over ranks of a python number."""
    return -all_reduce_max(-float(value), device)


def all_reduce_sum(value, device=None):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class _DevView:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def all_reduce_device(ptr, count, average=True):
    """In-place sum (or mean) all-reduce over the ranks of ``count`` float32 values at device address
    ``ptr`` - the flat gradient buffer of ``csi_train_grads``.  One collective for the whole model:
    over RCCL the xGMI rings are per-link bound, so one large bucket beats many small ones; the layer-0
    gradient (83 % of the bytes) is produced last by the backward pass, so there is little to overlap.
    The caller has to synchronise the library's stream first (``engine.synchronize()``)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    t = torch.as_tensor(_DevView(ptr, count), device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if average:
        t.mul_(1.0 / dist.get_world_size())
    torch.cuda.synchronize()


def all_reduce_mean_arrays(arrays):
    """Mean over the ranks of a dict of small host arrays (running BatchNormalization statistics at the
    end of a data-parallel fit)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arrays
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    names = sorted(arrays)
    flat = torch.from_numpy(np.concatenate([np.asarray(arrays[k], np.float32).ravel() for k in names])).to(dev)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    host = (flat / dist.get_world_size()).cpu().numpy()
    out, off = {}, 0
    for k in names:
        n = int(np.asarray(arrays[k]).size)
        out[k] = host[off:off + n].reshape(np.asarray(arrays[k]).shape).copy()
        off += n
    return out
