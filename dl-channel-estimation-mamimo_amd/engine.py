"""CsiEngine: one HIP context (one GPU, one stream) holding the two regressors (real, imag),
the pilot matrix and the activation workspace.  Thin object wrapper over the C-ABI."""
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import CsiError

N_DATA = 234     # data subcarriers, generate_maMIMO_LTF.m:98
SYM_LEN = 320    # FFT 256 + CP 64, generate_maMIMO_LTF.m:96-97


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class DeviceArray:
    """float32 array in HBM owned by an engine (hipMalloc through the C-ABI)."""

    def __init__(self, engine, shape):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * 4
        p = ctypes.c_void_p()
        engine._check(engine._lib.csi_device_malloc(engine._ctx, ctypes.byref(p), self.nbytes))
        self.ptr = p.value or 0
        engine._arrays.add(self)          # close() frees what is still alive: an array never outlives its engine as leaked HBM

    def upload(self, host, first=0):
        """Copy `host` into rows [first, first + len(host)) along axis 0 (default: the whole array)."""
        host = _f32c(host)
        row = self.nbytes // max(self.shape[0], 1)
        assert host.nbytes % max(row, 1) == 0 and first * row + host.nbytes <= self.nbytes, (host.shape, self.shape, first)
        if first == 0 and host.nbytes != self.nbytes:
            assert host.shape[1:] == self.shape[1:], (host.shape, self.shape)
        self.engine._check(self.engine._lib.csi_memcpy_h2d(self.engine._ctx, self.ptr + first * row, host.ctypes.data, host.nbytes))
        return self

    def download(self, first=0, count=None):
        """Copy back rows [first, first+count) along axis 0 (default: everything)."""
        n0 = self.shape[0]
        count = n0 - first if count is None else count
        row = self.nbytes // max(n0, 1)
        out = np.empty((count,) + self.shape[1:], dtype=np.float32)
        self.engine._check(self.engine._lib.csi_memcpy_d2h(self.engine._ctx, out.ctypes.data,
                                                           self.ptr + first * row, count * row))
        return out

    def free(self):
        if self.ptr and self.engine._ctx:
            self.engine._lib.csi_device_free(self.engine._ctx, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedPool:
    """Fresh numpy arrays over RECYCLED pinned host buffers.  ``take(shape, dtype)`` returns an array nobody else holds - the
    reference's wrapper returns fresh arrays (inference.py:31-32) and so does this - whose memory goes back to the pool once the
    array and every view of it are garbage-collected, so a serving loop pins its result memory once, not per call (pinning a
    gigabyte costs more than the host pass it saves).  ``alloc(nbytes)`` returns an object with the buffer protocol that frees
    its memory when it is collected (CsiEngine: csi_host_malloc behind a ctypes array with a finalizer)."""

    def __init__(self, alloc, max_idle_bytes=4 << 30):
        self._alloc, self._free, self._idle, self._max = alloc, {}, 0, int(max_idle_bytes)
        self.allocated = self.reused = 0

    def take(self, shape, dtype):
        count = int(np.prod(shape))
        n = max(count * np.dtype(dtype).itemsize, 1)
        idle = self._free.get(n)
        if idle:
            buf = idle.pop()
            self._idle -= n
            self.reused += 1
        else:
            buf = self._alloc(n)
            self.allocated += 1
        flat = np.frombuffer(buf, dtype=dtype, count=count)      # every later view has THIS array as its base: it dies last
        weakref.finalize(flat, self._give_back, n, buf)
        return flat.reshape(shape)

    def _give_back(self, n, buf):
        if self._idle + n <= self._max:                           # beyond the cap the buffer is simply dropped (and freed)
            self._free.setdefault(n, []).append(buf)
            self._idle += n

    def clear(self):
        self._free.clear()
        self._idle = 0

    @property
    def idle_bytes(self):
        return self._idle


def get_unique_id():
    """128-byte RCCL unique id (ncclGetUniqueId through the C-ABI) - create on ONE rank, hand to all."""
    lib = _lib.load_library()
    buf = ctypes.create_string_buffer(128)
    rc = lib.csi_get_unique_id(buf)
    if rc != 0:
        raise CsiError(rc, (lib.csi_last_error(None) or b'').decode())
    return buf.raw


def classify_pilot(P):
    """Which LS despread ``set_pilot(P)`` will choose (host only, no GPU needed): returns (kind, sym_src, out_row) with kind
    0 = generic real P (matrix-core despread), 1 = the Sylvester Hadamard matrix, 2 = a signed row / column permutation of it
    (e.g. the 802.11 VHT mapping matrix doubled up) - 1 and 2 take the Walsh-Hadamard kernel.  For kinds 1 / 2 the tables give
    P[j, s] = rs[j] * H[sigma(j), tau(s)] * cs[s] as sym_src[u] = tau^-1(u) | (256 if cs < 0) and out_row[r] = sigma^-1(r) | (256 if rs < 0)."""
    lib = _lib.load_library()
    P = _f32c(P)
    if P.ndim != 2 or P.shape[0] != P.shape[1]:
        raise CsiError(-1, f'P must be square, got {P.shape}')
    nt = P.shape[0]
    a, b = np.zeros(max(nt, 1), np.int32), np.zeros(max(nt, 1), np.int32)
    kind = lib.csi_pilot_classify(_fp(P), nt, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), b.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if kind < 0:
        raise CsiError(kind, 'csi_pilot_classify: bad argument')
    return kind, (a if kind else None), (b if kind else None)


class CsiEngine:
    """Owns a ``csi_ctx``.  Shapes follow the reference: nt tx antennas, nr rx antennas,
    len_ltf = 320*nt samples per rx preamble, FC hidden widths ``hidden`` (--nn), n_out outputs
    (massiveMIMO_CSI_prediction_DNN.py:18,227)."""

    def __init__(self, nt, nr, hidden=(1024, 1024), n_out=N_DATA, use_bn=True, bn_eps=1e-3,
                 device=0, workspace_bytes=0, dtype='f32', len_ltf=None):
        self._lib = _lib.load_library()
        self._ctx = None
        self.nt, self.nr = int(nt), int(nr)
        # nt == 0: single-input model without pilot input (DNN.py:180,234); only predict_samples
        self.len_ltf = SYM_LEN * self.nt if self.nt > 0 else int(len_ltf)
        self.d_in = self.len_ltf + self.nt
        self.hidden = tuple(int(h) for h in hidden)
        self.n_out = int(n_out)
        self.use_bn = bool(use_bn)
        cfg = _lib.CsiConfig()
        cfg.nt, cfg.nr, cfg.len_ltf = self.nt, self.nr, self.len_ltf
        if not 1 <= len(self.hidden) <= _lib.CSI_MAX_HIDDEN:
            raise CsiError(-1, f'between 1 and {_lib.CSI_MAX_HIDDEN} hidden layers are supported')
        cfg.n_hidden = len(self.hidden)
        for i, h in enumerate(self.hidden):
            cfg.hidden[i] = h
        cfg.n_out, cfg.use_bn, cfg.bn_eps = self.n_out, int(self.use_bn), float(bn_eps)
        cfg.dtype = {'f32': _lib.CSI_DTYPE_F32, 'bf16': _lib.CSI_DTYPE_BF16}[dtype]
        cfg.device, cfg.workspace_bytes = int(device), int(workspace_bytes)
        ctx = ctypes.c_void_p()
        rc = self._lib.csi_create(ctypes.byref(cfg), ctypes.byref(ctx))
        if rc != 0:
            raise CsiError(rc, (self._lib.csi_last_error(None) or b'').decode())
        self._ctx = ctx
        self._arrays = weakref.WeakSet()
        me = weakref.ref(self)                                  # (no engine -> pool -> engine cycle: an engine is freed when its last reference goes)
        self.result_pool = PinnedPool(lambda n: me()._pinned_buffer(n))       # estimate(..., pinned_results=True)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise CsiError(rc, (self._lib.csi_last_error(self._ctx) or b'').decode())

    def close(self):
        if self._ctx:
            for arr in list(getattr(self, '_arrays', ())):
                arr.free()
            if getattr(self, 'result_pool', None) is not None:
                self.result_pool.clear()                             # idle pinned buffers (those behind live arrays free themselves later)
            self._lib.csi_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.csi_synchronize(self._ctx))

    def set_option(self, name, value):
        """Tuning knobs of csi_set_option: 'use_graph', 'force_tile', 'xcd_order', 'ls_fft_first_max', 'ls_kernel',
        'f32_engine' (-1 automatic / 0 fp32 MFMA kernels / 1 split-f16 engine), 'hs_in_shift', 'hs_act_shift', ..."""
        self._check(self._lib.csi_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name):
        """Current value of a csi_set_option knob, or of the counters 'hs_launches' / 'hs_range_fallbacks'."""
        v = ctypes.c_int64(0)
        self._check(self._lib.csi_get_option(self._ctx, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def empty(self, shape):
        return DeviceArray(self, shape)

    def to_device(self, host):
        host = _f32c(host)
        return DeviceArray(self, host.shape).upload(host)

    # ------------------------------------------------------------------ model state
    def load_weights(self, model, weights):
        """model: 0/'real' or 1/'imag'.  weights: dict keras-name -> ndarray
        (fc_dense{i}.kernel [in,out], .bias, bn{i}.gamma/beta/moving_mean/moving_variance,
        fc_regressor.kernel/.bias)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        keep, arr = [], (_lib.CsiTensor * len(weights))()
        n = 0
        for name, val in weights.items():
            if not isinstance(val, np.ndarray):
                continue
            a = _f32c(val)
            keep.append(a)
            arr[n].name = name.encode()
            arr[n].data = _fp(a)
            arr[n].rows = a.shape[0] if a.ndim == 2 else 1
            arr[n].cols = a.shape[1] if a.ndim == 2 else a.size
            n += 1
        self._check(self._lib.csi_load_weights(self._ctx, int(idx), arr, n))

    # ------------------------------------------------------------------ on-box fine-tuning (csi_train_*)
    def _tensor_array(self, weights):
        keep, arr = [], (_lib.CsiTensor * max(1, len(weights)))()
        n = 0
        for name, val in weights.items():
            if not isinstance(val, np.ndarray):
                continue
            a = _f32c(val)
            keep.append(a)
            arr[n].name = name.encode()
            arr[n].data = _fp(a)
            arr[n].rows = a.shape[0] if a.ndim == 2 else 1
            arr[n].cols = a.shape[1] if a.ndim == 2 else a.size
            n += 1
        return keep, arr, n

    def train_begin(self, model, weights=None, lr=1e-4, dropout=0.15, seed=0, beta1=0.9, beta2=0.999, eps=1e-7,
                    bn_momentum=0.99):
        """Trainer of one component model (reference: Adam(lr), --dropout, keras defaults elsewhere;
        massiveMIMO_CSI_prediction_DNN.py:16,19,272).  weights=None: Glorot-uniform initialisation."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        tc = _lib.CsiTrainConfig(lr=lr, beta1=beta1, beta2=beta2, eps=eps, bn_momentum=bn_momentum, dropout=dropout, seed=seed)
        keep, arr, n = self._tensor_array(weights or {})
        self._check(self._lib.csi_train_begin(self._ctx, int(idx), ctypes.byref(tc), arr if n else None, n))

    def train_step(self, model, x, y, noise_std=0.0):
        """One optimiser step on the rows x [B, len_ltf+nt], labels y [B, n_out]; returns the batch loss."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        x, y = _f32c(x), _f32c(y)
        if x.ndim != 2 or x.shape[1] != self.d_in or y.shape != (x.shape[0], self.n_out):
            raise CsiError(-1, f'x must be [B,{self.d_in}] and y [B,{self.n_out}], got {x.shape} / {y.shape}')
        loss = ctypes.c_float()
        self._check(self._lib.csi_train_step(self._ctx, int(idx), _fp(x), _fp(y), x.shape[0], float(noise_std), ctypes.byref(loss)))
        return float(loss.value)

    def train_backward(self, model, x, y, noise_std=0.0):
        """Loss and gradients of one batch without the parameter update (data-parallel step, part 1)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        x, y = _f32c(x), _f32c(y)
        if x.ndim != 2 or x.shape[1] != self.d_in or y.shape != (x.shape[0], self.n_out):
            raise CsiError(-1, f'x must be [B,{self.d_in}] and y [B,{self.n_out}], got {x.shape} / {y.shape}')
        loss = ctypes.c_float()
        self._check(self._lib.csi_train_backward(self._ctx, int(idx), _fp(x), _fp(y), x.shape[0], float(noise_std), ctypes.byref(loss)))
        return float(loss.value)

    def train_grads(self, model):
        """(device pointer, element count) of the flat gradient buffer (the all-reduce operand)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        p, n = ctypes.POINTER(ctypes.c_float)(), ctypes.c_int64()
        self._check(self._lib.csi_train_grads(self._ctx, int(idx), ctypes.byref(p), ctypes.byref(n)))
        return ctypes.cast(p, ctypes.c_void_p).value, int(n.value)

    def train_apply(self, model):
        """Adam on the (all-reduced) gradients (data-parallel step, part 2)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        self._check(self._lib.csi_train_apply(self._ctx, int(idx)))

    def train_set_dataset(self, model, ltf_table, ltf_row, itx, y):
        """Upload a training set once (csi_train_set_dataset): ltf_table [n_rows, len_ltf] every rx preamble
        of this component once, per sample its table row, tx index and labels [N, n_out]."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        table, y = _f32c(ltf_table), _f32c(y)
        row = np.ascontiguousarray(ltf_row, dtype=np.int32)
        itx = np.ascontiguousarray(itx, dtype=np.int32)
        if table.ndim != 2 or table.shape[1] != self.len_ltf or y.shape != (row.size, self.n_out) or itx.shape != row.shape:
            raise CsiError(-1, f'dataset shapes: table [n,{self.len_ltf}], ltf_row/itx [N], y [N,{self.n_out}]')
        ip = ctypes.POINTER(ctypes.c_int32)
        self._check(self._lib.csi_train_set_dataset(self._ctx, int(idx), _fp(table), table.shape[0], row.ctypes.data_as(ip),
                                                    itx.ctypes.data_as(ip), _fp(y), row.size))

    def _train_indexed(self, model, mode, ids, noise_std):
        idx = {'real': 0, 'imag': 1}.get(model, model)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        loss = ctypes.c_float()
        self._check(self._lib.csi_train_indexed(self._ctx, int(idx), mode, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ids.size,
                                                float(noise_std), ctypes.byref(loss)))
        return float(loss.value)

    def train_step_indexed(self, model, ids, noise_std=0.0):
        """One optimiser step on the samples ``ids`` of the resident dataset."""
        return self._train_indexed(model, 0, ids, noise_std)

    def train_backward_indexed(self, model, ids, noise_std=0.0):
        return self._train_indexed(model, 1, ids, noise_std)

    def train_eval_indexed(self, model, ids):
        return self._train_indexed(model, 2, ids, 0.0)

    def train_eval(self, model, x, y):
        """mse of the current trainer parameters in inference mode (the reference's val_loss)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        x, y = _f32c(x), _f32c(y)
        if x.ndim != 2 or x.shape[1] != self.d_in or y.shape != (x.shape[0], self.n_out):
            raise CsiError(-1, f'x must be [B,{self.d_in}] and y [B,{self.n_out}], got {x.shape} / {y.shape}')
        loss = ctypes.c_float()
        self._check(self._lib.csi_train_eval(self._ctx, int(idx), _fp(x), _fp(y), x.shape[0], ctypes.byref(loss)))
        return float(loss.value)

    def train_set_lr(self, model, lr):
        idx = {'real': 0, 'imag': 1}.get(model, model)
        self._check(self._lib.csi_train_set_lr(self._ctx, int(idx), float(lr)))

    def train_tensor_names(self):
        names = []
        for i, _ in enumerate(self.hidden):
            names += [f'fc_dense{i}.kernel', f'fc_dense{i}.bias']
            if self.use_bn:
                names += [f'bn{i}.gamma', f'bn{i}.beta', f'bn{i}.moving_mean', f'bn{i}.moving_variance']
        return names + ['fc_regressor.kernel', 'fc_regressor.bias']

    def _train_shape(self, name):
        base = name[5:] if name.startswith('grad:') else name
        widths = (self.d_in,) + self.hidden
        if base.startswith('fc_regressor'):
            fan_in, out = self.hidden[-1], self.n_out
        else:
            i = int(''.join(ch for ch in base.split('.')[0] if ch.isdigit()))
            fan_in, out = widths[i], self.hidden[i]
        return (fan_in, out) if base.endswith('.kernel') else (out,)

    def train_get(self, model, name):
        """One trainer tensor by keras name ('grad:<name>': gradient of the last step)."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        if (name[5:] if name.startswith('grad:') else name) not in self.train_tensor_names():
            raise CsiError(-1, f"train_get: unknown tensor '{name}'")
        out = np.empty(self._train_shape(name), np.float32)
        self._check(self._lib.csi_train_get(self._ctx, int(idx), name.encode(), _fp(out), out.size))
        return out

    def train_weights(self, model):
        return {n: self.train_get(model, n) for n in self.train_tensor_names()}

    def train_end(self, model, commit=True):
        idx = {'real': 0, 'imag': 1}.get(model, model)
        self._check(self._lib.csi_train_end(self._ctx, int(idx), int(bool(commit))))

    def set_pilot(self, P):
        """P [nt, nt], row j = pilot sequence of tx j (= dataset['P'][:, j],
        massiveMIMO_dataGenerator.py:311)."""
        P = _f32c(P)
        if P.shape != (self.nt, self.nt):
            raise CsiError(-1, f'P must be [{self.nt},{self.nt}], got {P.shape}')
        self._check(self._lib.csi_set_pilot(self._ctx, _fp(P)))

    # ------------------------------------------------------------------ host-buffer calls
    def _split(self, ltf, ltf_im=None):
        if ltf_im is None:
            ltf = np.asarray(ltf)
            re, im = _f32c(ltf.real), _f32c(ltf.imag)
        else:
            re, im = _f32c(ltf), _f32c(ltf_im)
        if re.ndim != 3 or re.shape[1:] != (self.nr, self.len_ltf) or im.shape != re.shape:
            raise CsiError(-1, f'preambles must be [npkt,{self.nr},{self.len_ltf}], got {re.shape}')
        return re, im

    def _out_planes(self, out, npkt, width):
        shape = (npkt, self.nr, self.nt, width)
        if out is None:
            return np.empty(shape, dtype=np.float32), np.empty(shape, dtype=np.float32)
        o_re, o_im = out
        for o in (o_re, o_im):
            if o.dtype != np.float32 or o.shape != shape or not o.flags['C_CONTIGUOUS']:
                raise CsiError(-1, f'out planes must be C-contiguous float32 {shape}')
        return o_re, o_im

    def predict(self, ltf, ltf_im=None, out=None):
        """DNN estimate.  ltf complex [npkt,nr,len_ltf] (or two float planes).
        Returns (out_real, out_imag) float32 [npkt,nr,nt,n_out]; ``out=(o_re, o_im)`` reuses caller
        buffers (pinned ones from ``pinned_empty`` are DMA'd directly)."""
        re, im = self._split(ltf, ltf_im)
        npkt = re.shape[0]
        o_re, o_im = self._out_planes(out, npkt, self.n_out)
        self._check(self._lib.csi_predict(self._ctx, _fp(re), _fp(im), npkt, _fp(o_re), _fp(o_im)))
        return o_re, o_im

    def ls_estimate(self, ltf, ltf_im=None, out=None):
        """LS estimate, complex64 [npkt,nr,nt,234]; with ``out=(h_re, h_im)`` the two float32 planes
        are returned instead (no complex assembly on the host)."""
        re, im = self._split(ltf, ltf_im)
        npkt = re.shape[0]
        h_re, h_im = self._out_planes(out, npkt, N_DATA)
        self._check(self._lib.csi_ls_estimate(self._ctx, _fp(re), _fp(im), npkt, _fp(h_re), _fp(h_im)))
        if out is not None:
            return h_re, h_im
        h = np.empty(h_re.shape, dtype=np.complex64)
        h.real = h_re
        h.imag = h_im
        return h

    def estimate(self, ltf, dnn=True, ls=True, out=None, pinned_results=False):
        """Both estimators on the arrays of the reference's deployment wrapper (inference.py:24-32): ``ltf``
        complex128 [npkt, nr, len_ltf] in, complex64 [npkt, nr, nt, n_out] (DNN) and / or [npkt, nr, nt, 234] (LS)
        out - one upload for both, the real / imag split and the complex assembly done inside the library's
        staging copies (csi_estimate_c128).  Returns (dnn, ls); an estimator that was not asked for is None.
        ``out=(dnn_buf, ls_buf)`` reuses complex64 arrays; arrays from ``pinned_empty(shape, np.complex64)`` receive the
        downloads directly (complex values assembled on the device, no host pass on the result side; same bits).
        ``pinned_results=True`` takes the result arrays from ``self.result_pool``: fresh arrays over recycled pinned buffers.
        A complex64 ``ltf`` is NOT widened: it goes through csi_estimate_c64 (uploaded as it is - straight from the array when it
        came from ``pinned_empty(shape, np.complex64)`` - and split on the device); same bits as the complex128 call on such values."""
        c64_in = isinstance(ltf, np.ndarray) and ltf.dtype == np.complex64
        ltf = np.ascontiguousarray(ltf, dtype=np.complex64 if c64_in else np.complex128)
        if ltf.ndim != 3 or ltf.shape[1:] != (self.nr, self.len_ltf):
            raise CsiError(-1, f'preambles must be [npkt,{self.nr},{self.len_ltf}], got {ltf.shape}')
        npkt = ltf.shape[0]
        bufs = []
        for want, width, given in ((dnn, self.n_out, out[0] if out else None), (ls, N_DATA, out[1] if out else None)):
            if not want:
                bufs.append(None)
                continue
            shape = (npkt, self.nr, self.nt, width)
            if given is None:
                given = self.result_pool.take(shape, np.complex64) if pinned_results else np.empty(shape, dtype=np.complex64)
            elif given.dtype != np.complex64 or given.shape != shape or not given.flags['C_CONTIGUOUS']:
                raise CsiError(-1, f'out arrays must be C-contiguous complex64 {shape}')
            bufs.append(given)
        if bufs[0] is None and bufs[1] is None:
            raise CsiError(-1, 'estimate: nothing asked for')
        self._check((self._lib.csi_estimate_c64 if c64_in else self._lib.csi_estimate_c128)(self._ctx, ltf.ctypes.data, npkt,
                                                bufs[0].ctypes.data if bufs[0] is not None else None,
                                                bufs[1].ctypes.data if bufs[1] is not None else None))
        return bufs[0], bufs[1]

    def _pinned_buffer(self, nbytes):
        """ctypes array over ``nbytes`` of pinned host memory (csi_host_malloc), freed when the ctypes array is collected."""
        p = ctypes.c_void_p()
        self._check(self._lib.csi_host_malloc(self._ctx, ctypes.byref(p), max(int(nbytes), 1)))
        buf = (ctypes.c_char * max(int(nbytes), 1)).from_address(p.value)
        lib, addr = self._lib, p.value
        weakref.finalize(buf, lambda: lib.csi_host_free(None, ctypes.c_void_p(addr)))      # (no context: the buffer may outlive the engine)
        return buf

    def pinned_empty(self, shape, dtype=np.float32):
        """numpy array in pinned host memory (csi_host_malloc); freed with the array."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self._check(self._lib.csi_host_malloc(self._ctx, ctypes.byref(p), max(n, 1)))
        buf = (ctypes.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        lib, addr = self._lib, p.value
        import weakref
        # no context in the finalizer: the array may outlive the engine (csi_host_free accepts NULL for that)
        weakref.finalize(buf, lambda: lib.csi_host_free(None, ctypes.c_void_p(addr)))
        return arr

    def lmmse_estimate(self, h_ls, hvec, snr_db):
        """LMMSE smoothing (LMMSE_ce.m) of an LS estimate.  h_ls complex [npkt,nr,nt,234]; hvec
        [npkt, L] (the reference's 'h' argument); snr_db [npkt, nr].  Returns complex64."""
        h_ls = np.asarray(h_ls)
        re, im = _f32c(h_ls.real), _f32c(h_ls.imag)
        npkt = re.shape[0]
        if re.shape != (npkt, self.nr, self.nt, N_DATA):
            raise CsiError(-1, f'h_ls must be [npkt,{self.nr},{self.nt},{N_DATA}], got {re.shape}')
        hvec, snr_db = _f32c(hvec), _f32c(snr_db)
        if hvec.ndim != 2 or hvec.shape[0] != npkt or snr_db.shape != (npkt, self.nr):
            raise CsiError(-1, 'hvec must be [npkt, L] and snr_db [npkt, nr]')
        o_re, o_im = np.empty_like(re), np.empty_like(re)
        self._check(self._lib.csi_lmmse_estimate(self._ctx, _fp(re), _fp(im), npkt, _fp(hvec), hvec.shape[1], _fp(snr_db),
                                                 _fp(o_re), _fp(o_im)))
        return o_re + 1j * o_im

    def predict_samples(self, model, x):
        """Literal Model.predict of one component: x [B, len_ltf+nt] -> float32 [B, n_out]."""
        idx = {'real': 0, 'imag': 1}.get(model, model)
        x = _f32c(x)
        if x.ndim != 2 or x.shape[1] != self.d_in:
            raise CsiError(-1, f'x must be [B,{self.d_in}], got {x.shape}')
        y = np.empty((x.shape[0], self.n_out), dtype=np.float32)
        self._check(self._lib.csi_predict_samples(self._ctx, int(idx), _fp(x), x.shape[0], _fp(y)))
        return y

    # ------------------------------------------------------------------ multi-GPU: RCCL inside the library
    def comm_init(self, rank, world, unique_id):
        """ncclCommInitRank on this engine's GPU (collective over the ranks); ``unique_id`` = the 128 bytes
        ``get_unique_id()`` produced on one rank (see ``dist.exchange_unique_id``)."""
        uid = bytes(unique_id)
        if len(uid) != 128:
            raise CsiError(-1, 'unique id must be 128 bytes')
        self._check(self._lib.csi_comm_init(self._ctx, int(rank), int(world), uid))

    def broadcast_weights(self, root=0):
        """Both component models and the pilot matrix as they sit on ``root``'s GPU -> every rank's GPU (ncclBroadcast of the
        device buffers, no host copy); afterwards every engine is loaded.  Returns the bytes moved."""
        self._check(self._lib.csi_broadcast_weights(self._ctx, int(root)))
        return self.get_option('comm_bytes')

    def clone_weights_from(self, other):
        """Both component models and the pilot matrix as they sit on ``other``'s GPU memory -> this engine (device to device, same
        process; the receiver side of ``broadcast_weights`` without RCCL).  The engines must agree in nt, hidden widths, n_out,
        use_bn and dtype; a mismatch raises and leaves this engine empty."""
        self._check(self._lib.csi_clone_weights(self._ctx, other._ctx))

    def comm_destroy(self):
        self._check(self._lib.csi_comm_destroy(self._ctx))

    # ------------------------------------------------------------------ device-resident calls
    def _checked(self, launch):
        """Run ``launch()`` (device-pointer calls), synchronise, and - if the split-f16 engine's range guard reports
        (CsiError code -6: an operand left the f16 range after scaling, or a whole row sat in its denormals) - run it
        again on the fp32 MFMA kernels, as the host-buffer entry points do by themselves.  Returns 'split', or 'fp32' when
        the repeat served the call; the option is put back either way (changing it drops cached hipGraphs)."""
        launch()
        try:
            self.synchronize()
            return 'split'
        except CsiError as err:
            if err.code != -6:
                raise
        engine = self.get_option('f32_engine')
        self.set_option('f32_engine', 0)
        try:
            launch()
            self.synchronize()
        finally:
            self.set_option('f32_engine', engine)
        self.range_recoveries = getattr(self, 'range_recoveries', 0) + 1
        return 'fp32'

    def predict_device(self, d_re, d_im, npkt, d_out_re, d_out_im, checked=False):
        """Asynchronous.  On the split-f16 engine a range-guard hit is reported by the NEXT ``synchronize()`` (CsiError
        code -6): outputs must not be consumed before it returned cleanly (the host-buffer calls repeat by themselves).
        ``checked=True`` does that for the caller: synchronises, repeats the call on the fp32 MFMA kernels if the guard
        reported, and returns which engine served it ('split' / 'fp32')."""
        def launch():
            self._check(self._lib.csi_predict_device(self._ctx, d_re.ptr, d_im.ptr, int(npkt), d_out_re.ptr, d_out_im.ptr))
        if checked:
            return self._checked(launch)
        launch()

    def estimate_device(self, d_re, d_im, npkt, d_out_re, d_out_im, d_h_re, d_h_im, checked=False):
        """LS + DNN of device-resident packets as one call (one hipGraph under 'use_graph'); ``checked`` as in predict_device."""
        def launch():
            self._check(self._lib.csi_estimate_device(self._ctx, d_re.ptr, d_im.ptr, int(npkt), d_out_re.ptr, d_out_im.ptr, d_h_re.ptr, d_h_im.ptr))
        if checked:
            return self._checked(launch)
        launch()

    def ls_estimate_device(self, d_re, d_im, npkt, d_h_re, d_h_im):
        self._check(self._lib.csi_ls_estimate_device(self._ctx, d_re.ptr, d_im.ptr, int(npkt), d_h_re.ptr, d_h_im.ptr))

    def nmse(self, h_ref, h_est):
        """NMSE_subk of the reference's evaluation (BER_test_maMIMO_LTF.m:675-686): per link
        ||ref - est||^2 / ||ref||^2 over the last axis, mean over all links.  Complex arrays of equal
        shape [..., n_bins] (e.g. the true channel and ``out_real + 1j * out_imag``)."""
        h_ref, h_est = np.asarray(h_ref), np.asarray(h_est)
        if h_ref.shape != h_est.shape or h_ref.ndim < 1 or h_ref.size == 0:
            raise CsiError(-1, f'h_ref and h_est must have the same non-empty shape, got {h_ref.shape} and {h_est.shape}')
        n_bins = h_ref.shape[-1]
        planes = [_f32c(a).reshape(-1, n_bins) for a in (h_ref.real, h_ref.imag, h_est.real, h_est.imag)]
        out = ctypes.c_double(0.0)
        self._check(self._lib.csi_nmse(self._ctx, _fp(planes[0]), _fp(planes[1]), _fp(planes[2]), _fp(planes[3]),
                                       planes[0].shape[0], n_bins, ctypes.byref(out)))
        return float(out.value)

    def nmse_device(self, d_ref_re, d_ref_im, d_est_re, d_est_im, nlinks, n_bins=N_DATA, d_per_link=None):
        """The same metric on device-resident planes ([nlinks][n_bins] float32 each); synchronous.
        ``d_per_link`` (a DeviceArray of nlinks floats) also receives the per-link ratios."""
        out = ctypes.c_double(0.0)
        self._check(self._lib.csi_nmse_device(self._ctx, d_ref_re.ptr, d_ref_im.ptr, d_est_re.ptr, d_est_im.ptr, int(nlinks), int(n_bins),
                                              d_per_link.ptr if d_per_link is not None else None, ctypes.byref(out)))
        return float(out.value)

    def synth_white(self, seed, first_pkt, npkt, d_re, d_im):
        self._check(self._lib.csi_synth_white(self._ctx, int(seed), int(first_pkt), int(npkt), d_re.ptr, d_im.ptr))

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on=True):
        self._check(self._lib.csi_profile_enable(self._ctx, int(bool(on))))

    def profile_reset(self):
        self._check(self._lib.csi_profile_reset(self._ctx))

    def pcie_probe(self, h2d_bytes, d2h_bytes):
        """(ms up alone, ms down alone, ms both at once) for these byte counts between pinned host memory and the device on two
        copy streams: the floor of a host-buffer call that moves them (csi_profile_pcie)."""
        a, b, ab = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.csi_profile_pcie(self._ctx, int(h2d_bytes), int(d2h_bytes), ctypes.byref(a), ctypes.byref(b), ctypes.byref(ab)))
        return a.value, b.value, ab.value

    def band_skeleton(self, rows, iters=5):
        """(ms per launch, executed f16 TFLOP/s) of the fused per-pair kernel's MFMA + barrier skeleton on the loaded model's own
        operand data: the practical ceiling of that kernel on this part (csi_profile_band_skeleton)."""
        ms, fl = ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.csi_profile_band_skeleton(self._ctx, int(rows), int(iters), ctypes.byref(ms), ctypes.byref(fl)))
        return ms.value, fl.value / (ms.value * 1e-3) / 1e12

    def profile(self):
        """dict kernel-name -> {ms, launches, flops, bytes} since the last reset."""
        out = {}
        for k in range(self._lib.csi_profile_num_kernels()):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            fl, by = ctypes.c_double(), ctypes.c_double()
            self._check(self._lib.csi_profile_query(self._ctx, k, ctypes.byref(ms), ctypes.byref(n),
                                                    ctypes.byref(fl), ctypes.byref(by)))
            out[self._lib.csi_profile_kernel_name(k).decode()] = dict(
                ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out
