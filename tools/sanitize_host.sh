#!/bin/bash
# Build container (no GPU): the host side of the library under the sanitizers.
#   1. tests/hostpool_check.cpp (thread pools, staging loops) under -fsanitize=thread and -fsanitize=address
#   1b. tests/hostpipe_mock_check.cpp (hp_run on a model of the HIP stream semantics) under -fsanitize=thread
#   1c. tests/comm_mock_check.cpp (csi_broadcast_weights by N rank threads on the HIP + RCCL models) under -fsanitize=thread
#   2. libcsi_mamimo.so rebuilt with -Xarch_host -fsanitize=address / undefined, swapped in for the CPU tests that call host-only entry
#      points (csi_pilot_classify, csi_crc32c, the RCCL-not-found path, the symbol table), sanitizer run-time preloaded into python;
#      the product library is put back (and compared) afterwards
# Round 4: all clean (no report).  ~4 minutes.
set -u
cd "$(dirname "$0")/.."
RT=$(dirname "$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
SO=dl-channel-estimation-mamimo_amd/libcsi_mamimo.so
SEL="pilot or crc or rccl or symbol or fallback or unique"
for san in thread address; do
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Wno-unused-value -pthread -Xarch_host -fsanitize=$san tests/hostpool_check.cpp -o /tmp/hostpool_$san 2>/dev/null || { echo "build failed ($san)"; exit 1; }
  for cap in 2 5; do CSI_HOST_SIMD=$cap ASAN_OPTIONS=detect_leaks=0 /tmp/hostpool_$san 2>&1 | tail -2 | sed "s/^/hostpool_check [$san, simd $cap]: /"; done
done
# the whole host pipeline (hp_run: stager / drainer threads, slots, events) on the stream model of tests/hostpipe_mock_check.cpp
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Wno-unused-value -pthread -Xarch_host -fsanitize=thread tests/hostpipe_mock_check.cpp -o /tmp/hpmock_tsan 2>/dev/null || { echo "build failed (hostpipe mock)"; exit 1; }
for i in 1 2 3; do TSAN_OPTIONS=halt_on_error=0 /tmp/hpmock_tsan > /tmp/hpmock_tsan.log 2>&1; echo "hostpipe_mock_check [thread] run $i: $(grep -c '^WARNING: ThreadSanitizer' /tmp/hpmock_tsan.log) races; $(tail -1 /tmp/hpmock_tsan.log)"; done
# csi_broadcast_weights by 2 / 4 / 8 rank threads on the HIP + RCCL models of tests/comm_mock_check.cpp (the whole library TU)
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Wno-unused-value -pthread -Xarch_host -fsanitize=thread tests/comm_mock_check.cpp -o /tmp/comm_tsan 2>/dev/null || { echo "build failed (comm mock)"; exit 1; }
TSAN_OPTIONS=halt_on_error=0 /tmp/comm_tsan > /tmp/comm_tsan.log 2>&1; echo "comm_mock_check [thread]: $(grep -c '^WARNING: ThreadSanitizer' /tmp/comm_tsan.log) races; $(tail -1 /tmp/comm_tsan.log)"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Wno-unused-value -pthread -Xarch_host -fsanitize=address tests/comm_mock_check.cpp -o /tmp/comm_asan 2>/dev/null || { echo "build failed (comm mock, address)"; exit 1; }
ASAN_OPTIONS=detect_leaks=1 /tmp/comm_asan > /tmp/comm_asan.log 2>&1; echo "comm_mock_check [address + leak]: $(grep -c 'ERROR: \(Address\|Leak\)Sanitizer' /tmp/comm_asan.log) reports; $(tail -1 /tmp/comm_asan.log)"
cp -p $SO /tmp/libcsi_product.so
for san in address undefined; do
  extra=""; [ $san = undefined ] && extra="-Xarch_host -fno-sanitize=vptr,function"
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -Wno-unused-value -Xarch_host -fsanitize=$san $extra -shared-libsan dl-channel-estimation-mamimo_amd/csrc/csi_mamimo.hip -o /tmp/libcsi_$san.so 2>/dev/null || { echo "build failed ($san)"; continue; }
  cp /tmp/libcsi_$san.so $SO; touch $SO
  rt=$RT/libclang_rt.asan-x86_64.so; [ $san = undefined ] && rt=$RT/libclang_rt.ubsan_standalone-x86_64.so
  LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 python -m pytest tests/test_host_round4.py tests/test_host.py -x -q -s -m "not gpu" -k "$SEL" > /tmp/san_$san.log 2>&1
  echo "library [$san]: $(tail -1 /tmp/san_$san.log); reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/san_$san.log)"
done
cp -p /tmp/libcsi_product.so $SO; touch $SO
cmp /tmp/libcsi_product.so $SO && echo "product library restored"
