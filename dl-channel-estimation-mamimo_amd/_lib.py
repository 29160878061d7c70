"""ctypes binding of include/csi_mamimo.h (the C-ABI of the HIP library).

The binding is deliberately thin: one Python attribute per exported symbol, argument types
copied from the header.  ``load_library()`` raises ``CsiError`` when the shared object has not
been built - the product path never falls back to a CPU implementation."""
import ctypes
import os
import sys
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG_DIR)
_SO = os.path.join(_PKG_DIR, 'libcsi_mamimo.so')
_SRC = os.path.join(_PKG_DIR, 'csrc', 'csi_mamimo.hip')

CSI_MAX_HIDDEN = 8
CSI_ABI_VERSION = 1
CSI_DTYPE_F32 = 0
CSI_DTYPE_BF16 = 1

STATUS_NAMES = {0: 'CSI_OK', -1: 'CSI_ERR_INVALID_ARG', -2: 'CSI_ERR_NOT_READY', -3: 'CSI_ERR_HIP',
                -4: 'CSI_ERR_NO_DEVICE', -5: 'CSI_ERR_NOMEM', -6: 'CSI_ERR_RANGE'}


class CsiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'{STATUS_NAMES.get(code, code)}: {msg}')
        self.code = code


class CsiConfig(ctypes.Structure):
    _fields_ = [('nt', ctypes.c_int32), ('nr', ctypes.c_int32), ('len_ltf', ctypes.c_int32),
                ('n_hidden', ctypes.c_int32), ('hidden', ctypes.c_int32 * CSI_MAX_HIDDEN),
                ('n_out', ctypes.c_int32), ('use_bn', ctypes.c_int32), ('bn_eps', ctypes.c_float),
                ('dtype', ctypes.c_int32), ('device', ctypes.c_int32), ('workspace_bytes', ctypes.c_int64)]


class CsiTensor(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.POINTER(ctypes.c_float)),
                ('rows', ctypes.c_int64), ('cols', ctypes.c_int64)]


class CsiTrainConfig(ctypes.Structure):
    _fields_ = [('lr', ctypes.c_float), ('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('eps', ctypes.c_float),
                ('bn_momentum', ctypes.c_float), ('dropout', ctypes.c_float), ('seed', ctypes.c_uint64)]


_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_ctx = ctypes.c_void_p

# symbol -> (restype, argtypes); must list every function declared in include/csi_mamimo.h
SYMBOLS = {
    'csi_abi_version': (ctypes.c_int, []),
    'csi_create': (ctypes.c_int, [ctypes.POINTER(CsiConfig), ctypes.POINTER(_ctx)]),
    'csi_destroy': (None, [_ctx]),
    'csi_last_error': (ctypes.c_char_p, [_ctx]),
    'csi_load_weights': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(CsiTensor), ctypes.c_int]),
    'csi_set_pilot': (ctypes.c_int, [_ctx, _fp]),
    'csi_predict': (ctypes.c_int, [_ctx, _fp, _fp, ctypes.c_int64, _fp, _fp]),
    'csi_predict_device': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    'csi_predict_samples': (ctypes.c_int, [_ctx, ctypes.c_int, _fp, ctypes.c_int64, _fp]),
    'csi_ls_estimate': (ctypes.c_int, [_ctx, _fp, _fp, ctypes.c_int64, _fp, _fp]),
    'csi_ls_estimate_device': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64, _vp, _vp]),
    'csi_estimate_c128': (ctypes.c_int, [_ctx, _vp, ctypes.c_int64, _vp, _vp]),
    'csi_estimate_c64': (ctypes.c_int, [_ctx, _vp, ctypes.c_int64, _vp, _vp]),
    'csi_estimate_device': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    'csi_lmmse_estimate': (ctypes.c_int, [_ctx, _fp, _fp, ctypes.c_int64, _fp, ctypes.c_int, _fp, _fp, _fp]),
    'csi_lmmse_estimate_device': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int, _vp, _vp, _vp]),
    'csi_train_begin': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(CsiTrainConfig), ctypes.POINTER(CsiTensor), ctypes.c_int]),
    'csi_train_step': (ctypes.c_int, [_ctx, ctypes.c_int, _fp, _fp, ctypes.c_int64, ctypes.c_float, _fp]),
    'csi_train_backward': (ctypes.c_int, [_ctx, ctypes.c_int, _fp, _fp, ctypes.c_int64, ctypes.c_float, _fp]),
    'csi_train_grads': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(_fp), ctypes.POINTER(ctypes.c_int64)]),
    'csi_train_apply': (ctypes.c_int, [_ctx, ctypes.c_int]),
    'csi_train_set_dataset': (ctypes.c_int, [_ctx, ctypes.c_int, _fp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), _fp, ctypes.c_int64]),
    'csi_train_indexed': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, ctypes.c_float, _fp]),
    'csi_train_eval': (ctypes.c_int, [_ctx, ctypes.c_int, _fp, _fp, ctypes.c_int64, _fp]),
    'csi_train_set_lr': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_float]),
    'csi_train_get': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_char_p, _fp, ctypes.c_int64]),
    'csi_train_end': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int]),
    'csi_synchronize': (ctypes.c_int, [_ctx]),
    'csi_set_option': (ctypes.c_int, [_ctx, ctypes.c_char_p, ctypes.c_int64]),
    'csi_nmse': (ctypes.c_int, [_ctx, _fp, _fp, _fp, _fp, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    'csi_nmse_device': (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    'csi_get_option': (ctypes.c_int, [_ctx, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    'csi_device_malloc': (ctypes.c_int, [_ctx, ctypes.POINTER(_vp), ctypes.c_int64]),
    'csi_device_free': (ctypes.c_int, [_ctx, _vp]),
    'csi_host_malloc': (ctypes.c_int, [_ctx, ctypes.POINTER(_vp), ctypes.c_int64]),
    'csi_host_free': (ctypes.c_int, [_ctx, _vp]),
    'csi_memcpy_h2d': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64]),
    'csi_memcpy_d2h': (ctypes.c_int, [_ctx, _vp, _vp, ctypes.c_int64]),
    'csi_synth_white': (ctypes.c_int, [_ctx, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, _vp, _vp]),
    'csi_profile_enable': (ctypes.c_int, [_ctx, ctypes.c_int]),
    'csi_profile_reset': (ctypes.c_int, [_ctx]),
    'csi_profile_num_kernels': (ctypes.c_int, []),
    'csi_get_unique_id': (ctypes.c_int, [ctypes.c_char_p]),
    'csi_comm_init': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    'csi_comm_destroy': (ctypes.c_int, [_ctx]),
    'csi_broadcast_weights': (ctypes.c_int, [_ctx, ctypes.c_int]),
    'csi_clone_weights': (ctypes.c_int, [_ctx, _ctx]),
    'csi_crc32c': (ctypes.c_uint32, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_uint32]),
    'csi_pilot_classify': (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    'csi_profile_band_skeleton': (ctypes.c_int, [_ctx, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'csi_profile_pcie': (ctypes.c_int, [_ctx, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'csi_profile_kernel_name': (ctypes.c_char_p, [ctypes.c_int]),
    'csi_profile_query': (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def library_path():
    return _SO


def build_band_kernel(verbose=False):
    """csrc/band_kernel_gen.py + csrc/band4_kernel_gen.py (the register-blocked bf16 form; its main() writes the kernels of both
    generators into one file) -> gfx950 assembly -> code object -> csrc/band8_hsaco.inc (a C array the library embeds and loads
    with hipModuleLoadData on first use).  clang / ld.lld of the ROCm LLVM; no GPU needed."""
    import tempfile
    csrc = os.path.join(_PKG_DIR, 'csrc')
    gen8, gen, inc = os.path.join(csrc, 'band_kernel_gen.py'), os.path.join(csrc, 'band4_kernel_gen.py'), os.path.join(csrc, 'band8_hsaco.inc')
    names = ['csi_band8', 'csi_band8_cs', 'csi_band8_bf16_cs', 'csi_band8_nostage', 'csi_band8_bf16', 'csi_band8_bf16_nostage', 'csi_band8_skeleton_rnd',
             'csi_band4_bf16', 'csi_band4', 'csi_band4_cs', 'csi_band4_bf16_cs']
    tag = '// kernels: ' + ' '.join(names)
    if os.path.exists(inc) and os.path.getmtime(inc) >= max(os.path.getmtime(gen), os.path.getmtime(gen8)):
        with open(inc) as f:
            if f.readline().strip() == tag:           # same generator, same kernel list
                return inc
    llvm = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
    with tempfile.TemporaryDirectory() as tmp:
        asm, obj, co = (os.path.join(tmp, 'band8.' + e) for e in ('s', 'o', 'hsaco'))
        cmds = [[sys.executable, gen, asm] + names,
                [os.path.join(llvm, 'clang'), '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', asm, '-o', obj],
                [os.path.join(llvm, 'ld.lld'), '-shared', obj, '-o', co]]
        for cmd in cmds:
            if verbose:
                print(' '.join(cmd))
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
            if res.returncode != 0:
                raise RuntimeError('band kernel build failed:\n' + res.stdout)
        with open(co, 'rb') as f:
            blob = f.read()
    rows = [', '.join('0x%02x' % b for b in blob[i:i + 24]) for i in range(0, len(blob), 24)]
    with open(inc + '.tmp', 'w') as f:
        f.write(tag + '\n')
        f.write('// generated by _lib.build_band_kernel from band_kernel_gen.py + band4_kernel_gen.py - gfx950 code object of the band kernels (%d bytes)\n' % len(blob))
        f.write('alignas(4096) static const unsigned char band8_hsaco[] = {\n' + ',\n'.join(rows) + '};\n')
    os.replace(inc + '.tmp', inc)
    return inc


def build_library(force=False, verbose=False):
    """Compile csrc/csi_mamimo.hip for gfx950 into the in-tree shared object (hipcc
    cross-compiles without a GPU); the assembly band kernel is generated, assembled and embedded first.  Returns the path.
    Several processes may arrive here at once (the ranks of one `torch.distributed.run` launch on a tree without a built library): one of them builds
    under an exclusive file lock, into a temporary name that replaces the library atomically; the others wait and find it up to date."""
    import fcntl
    try:
        lock = open(_SO + '.lock', 'w')
    except OSError:                      # a read-only tree: nothing can be built there anyway; the up-to-date check below still answers
        return _build_library_locked(force, verbose)
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_library_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_library_locked(force, verbose):
    build_band_kernel(verbose)
    srcs = [_SRC] + [os.path.join(_PKG_DIR, 'csrc', f) for f in os.listdir(os.path.join(_PKG_DIR, 'csrc'))]
    srcs.append(os.path.join(_REPO, 'include', 'csi_mamimo.h'))
    if not force and os.path.exists(_SO):
        so_m = os.path.getmtime(_SO)
        if all(os.path.getmtime(s) <= so_m for s in srcs if os.path.exists(s)):
            return _SO
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # host side: baseline x86-64; the three AVX2 staging loops of the host pipeline carry their own target attribute and a
    # run-time CPU check (csi_hostpipe.hpp)
    tmp_so = '%s.tmp.%d' % (_SO, os.getpid())
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value', _SRC, '-o', tmp_so]
    # CSI_BUILD_DEFINES="NAME ..." adds -DNAME: CSI_LS_RACE_VARIANTS compiles the race-hunt instantiations of the LS kernel
    # (tools/ls_race_box*.sh); the product build carries none of them
    cmd[1:1] = ['-D' + d for d in os.environ.get('CSI_BUILD_DEFINES', '').split()]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if res.returncode != 0:
        if os.path.exists(tmp_so):
            os.remove(tmp_so)
        raise RuntimeError('hipcc failed:\n' + res.stdout)
    os.replace(tmp_so, _SO)
    return _SO


def load_library():
    """dlopen the HIP library and attach prototypes.  Raises CsiError if it is not built."""
    global _lib, _SO
    if _lib is not None:
        return _lib
    # test hook (multi-rank dry runs on a machine without a GPU, tests/mock_library.cpp): another build of the SAME translation unit.
    # Honoured only together with CSI_DEBUG_HOOKS=1, like the code-object hooks of the library itself.
    if os.environ.get('CSI_DEBUG_HOOKS') == '1' and os.environ.get('CSI_LIBRARY_PATH'):
        _SO = os.environ['CSI_LIBRARY_PATH']
    if not os.path.exists(_SO):
        raise CsiError(-4, f'{_SO} not found: build it with __graft_entry__.build() '
                           f'(hipcc --offload-arch=gfx950); there is no CPU fallback')
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and the library
    # links /opt/rocm's.  Whichever is loaded first serves both (same soname).  Measured on the GPU
    # box: torch first -> both work; library first -> torch.cuda reports no device, which would break
    # the RCCL weight broadcast of a multi-rank run.  So in a multi-rank job torch goes first.
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch                      # noqa: F401
    lib = ctypes.CDLL(_SO)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    ver = lib.csi_abi_version()
    if ver != CSI_ABI_VERSION:
        raise CsiError(-1, f'ABI version mismatch: library {ver}, binding {CSI_ABI_VERSION}')
    _lib = lib
    return lib
