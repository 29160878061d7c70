"""MI355X-native massive-MIMO channel estimation: LS pilot estimate + per-pair FC-DNN CSI
regressor behind the reference's ``CSIPredictor`` / ``Model.predict`` surface.

Host code is Python + ctypes over the C-ABI in ``include/csi_mamimo.h``; all arithmetic runs in
the hand-written HIP kernels of ``csrc/`` (gfx950).  There is no CPU fallback: every compute
entry point raises if ``libcsi_mamimo.so`` is missing or no gfx950 device is visible.
"""
from ._lib import CsiError, build_library, library_path, load_library   # noqa: F401
from .engine import CsiEngine, DeviceArray                              # noqa: F401
from .model import CSIModel, load_weight_file, save_weight_file         # noqa: F401
from .inference import CSIPredictor                                     # noqa: F401
from . import synth, dist, dataset, trainer                             # noqa: F401

__all__ = ['CsiEngine', 'DeviceArray', 'CSIModel', 'CSIPredictor', 'CsiError', 'build_library',
           'library_path', 'load_library', 'load_weight_file', 'save_weight_file', 'synth', 'dist', 'dataset', 'trainer']
