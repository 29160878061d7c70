// mock_library.cpp - the library's translation unit linked against the model of the HIP runtime (tests/mock_hip.hpp) as a shared object:
// the Python layer (ctypes table, CsiEngine, pinned result pool) then runs on a machine without a GPU.  Kernel launches are dropped, so
// numbers are meaningless - call flow, argument checks, buffer lifetimes and counters are what it serves (tests/test_host_round4.py).
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 -shared -fPIC -pthread tests/mock_library.cpp -o /tmp/libcsi_mock.so
#include "../dl-channel-estimation-mamimo_amd/csrc/csi_mamimo.hip"

#include "mock_hip.hpp"
