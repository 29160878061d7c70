// weave.hip.h - `output_real + 1j * output_imag` (inference.py:31) on the device: two float32 planes -> complex64.
// Used by csi_estimate_c128 when the caller's result arrays are pinned host memory: the complex values are then assembled in HBM
// behind the kernels of a packet chunk and the download lands in the caller's array itself - no staging copy and no host pass on
// the result side (with pageable arrays the host threads of the pipeline do the same assembly while they copy out of the staging
// buffer, csi_hostpipe.hpp: hp_weave_c64).  HBM-bound and small next to the link: 16 B per complex value moved at several TB/s
// against 8 B per value over PCIe at ~50 GB/s.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

// out[i] = (re[i], im[i]), i in [0, n): one 8-byte store per lane, 4-byte loads from each plane (both fully coalesced; the planes of
// a chunk start at arbitrary float offsets, so wider loads would need an alignment case split that 2 x 0.96 GB per 4000 packets do
// not pay for)
__global__ __launch_bounds__(256) void weave_c64_kernel(const float* __restrict__ re, const float* __restrict__ im, float2* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = make_float2(re[i], im[i]);
}

// The other direction, for csi_estimate_c64: a complex64 batch as the caller holds it - interleaved (re, im) floats - into the two float32
// planes every kernel of the path reads (X.real / X.imag, inference.py:29-30).  One 8-byte load per lane, 4-byte stores to each plane.
__global__ __launch_bounds__(256) void split_c64_kernel(const float2* __restrict__ in, float* __restrict__ re, float* __restrict__ im, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = in[i];
        re[i] = v.x;
        im[i] = v.y;
    }
}

}  // namespace csi
