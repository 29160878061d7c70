// metrics.hip.h - the reference's accuracy metric on the device.
//   NMSE_subk(real, pred), BER_test_maMIMO_LTF.m:675-686: per (tx, rx) link
//       || real(:,t,r) - pred(:,t,r) ||^2 / || real(:,t,r) ||^2      over the 234 data bins,
//   averaged over all links ('mean(subK_nmse, "all")'; snr_loop_testing.m:44,51,58 then averages the
//   per-packet values, which is the same number when every packet has Nt x Nr links).
// One wave per link (two planes of the reference and of the estimate, [link][n_bins] fp32), fp32
// products summed in fp64 per lane and across the wave; the per-link ratios are then summed by ONE
// workgroup in a fixed order (deterministic), in fp64.  HBM-bound: 16 * n_bins bytes per link.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csi {

__global__ __launch_bounds__(256) void nmse_links_kernel(const float* __restrict__ ref_re, const float* __restrict__ ref_im,
                                                         const float* __restrict__ est_re, const float* __restrict__ est_im,
                                                         int64_t nlinks, int n_bins, float* __restrict__ ratio) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t l = wave; l < nlinks; l += nwaves) {
        const size_t base = (size_t)l * n_bins;
        double num = 0.0, den = 0.0;
        for (int k = lane; k < n_bins; k += 64) {
            const float rr = ref_re[base + k], ri = ref_im[base + k];
            const float dr = rr - est_re[base + k], di = ri - est_im[base + k];
            num += (double)dr * dr + (double)di * di;
            den += (double)rr * rr + (double)ri * ri;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            num += __shfl_down(num, off, 64);
            den += __shfl_down(den, off, 64);
        }
        if (lane == 0) ratio[l] = (float)(num / den);
    }
}

// out[0] += sum of ratio[0..n) in a fixed order: thread t owns the strided subsequence t, t+1024, ...;
// the 1024 partial sums are combined by a binary tree in LDS
__global__ __launch_bounds__(1024) void nmse_sum_kernel(const float* __restrict__ ratio, int64_t n, double* __restrict__ out) {
    __shared__ double part[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)ratio[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += part[0];
}

}  // namespace csi
