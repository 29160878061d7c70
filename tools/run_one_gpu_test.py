#!/usr/bin/env python3
"""Run ONE test function of the GPU suite without pytest's collection of the whole tests/ tree - for the last GPU seconds of a
round, when a fresh box's `python -m pytest tests ...` start-up alone would eat them.
    python tools/run_one_gpu_test.py tests/test_gpu_*.py test_estimate_c128_into_pinned_result_arrays
Fixtures served: pkg, oracle (as tests/conftest.py builds them).  Prints PASSED / the traceback, exit code 0 / 1."""
import importlib.util
import inspect
import os
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, REPO)


def main():
    path, name = sys.argv[1], sys.argv[2]
    t0 = time.time()
    import conftest  # noqa: F401  (thread-pool limits before numpy loads)
    spec = importlib.util.spec_from_file_location('one_test_module', os.path.join(REPO, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fn = getattr(mod, name)
    import dl_channel_estimation_mamimo_amd as pkg
    from oracle import csi_oracle
    have = {'pkg': pkg, 'oracle': csi_oracle}
    args = [have[p] for p in inspect.signature(fn).parameters]
    t1 = time.time()
    try:
        fn(*args)
    except Exception:       # noqa: BLE001
        traceback.print_exc()
        print('FAILED %s::%s (imports %.1f s, test %.1f s)' % (path, name, t1 - t0, time.time() - t1))
        return 1
    print('PASSED %s::%s (imports %.1f s, test %.1f s)' % (path, name, t1 - t0, time.time() - t1))
    return 0


if __name__ == '__main__':
    sys.exit(main())
