// pkadd_mfma_probe.hip - does a packed-fp32 add with a half-swapped source (v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]) ever
// return a wrong value in its last quarter-wave when the SIMD's other wave, of ANOTHER workgroup, issues bf16 MFMAs?
//
// Why: DESIGN.md 4.2 / profiles/r04_ls_ringb_variants.txt.  Every one of the rare bad items of ls_estimate_ringb_kernel<1, 4, 1, NPP, 2>
// (two workgroups per CU) is the result of ONE such instruction (pk_add_mi / pk_add_pi of a radix-4 butterfly) wrong in lanes 48-63,
// at the first launch after another kernel, on some boxes of the pool.  The LS kernel executes a few hundred of these operations per
// item; this probe executes ~10^9 per second per CU under the suspected conditions and checks every result in the kernel:
//   * grid of 2 x 256 workgroups of 4 waves, 78 KiB of LDS each, so that two share a CU (one wave of each per SIMD);
//   * "matrix" workgroups issue chains of v_mfma_f32_32x32x16_bf16 (mode bit 1: v_mfma_f32_32x32x2_f32 instead, the fp32 ring kernel's
//     instruction; mode bit 2: no MFMA at all, plain VALU filler) on register data;
//   * "check" workgroups run radix-4 butterflies with the LS kernel's own inline-asm strings (twiddle product = v_pk_mul_f32 +
//     v_pk_fma_f32 with op_sel, outputs 1 / 3 = v_pk_add_f32 with op_sel) and, on the same inputs, scalar v_add / v_sub / v_mul / v_fma
//     that round identically; any bit that differs is counted and the first 256 differences are logged (block, wave, lane, output,
//     repetition, both values);
//   * the host alternates the probe with a DIFFERENT kernel and an optional idle gap (the events of the LS kernel need "the first
//     launch after another kernel"; on three of four boxes they came in the first seconds after the box had been idle).
// Build / run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkadd_mfma_probe.hip -o /tmp/pkadd_probe && /tmp/pkadd_probe [seconds] [mode] [idle_ms]
// Prints one line per 1000 launches and a summary: launches, checked results, differences by output and by lane quarter.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HIP_OK(e)                                                                                  \
    do {                                                                                           \
        hipError_t r_ = (e);                                                                       \
        if (r_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__);            \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

// ---- the LS kernel's operations, string for string (ls_estimate.hip.h: pk_add_mi, pk_add_pi, pk_cmul)
__device__ __forceinline__ f32x2 pk_add_mi(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add_pi(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_cmul(f32x2 x, f32x2 w) {
    f32x2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(x), "v"(w), "v"(t));
    return d;
}
// ---- the same arithmetic on scalar instructions (one rounding per add; the product term rounded once before the fma)
__device__ __forceinline__ float s_add(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_sub(float a, float b) { float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_mul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float s_fnma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 sc_cmul(f32x2 x, f32x2 w) {
    return f32x2{s_fnma(x[1], w[1], s_mul(x[0], w[0])), s_fma(x[0], w[1], s_mul(x[1], w[0]))};
}

struct ErrRec { uint32_t block, wave, lane, out, rep, it, got, want; };
struct ProbeOut {
    unsigned long long checked;        // butterflies checked
    uint32_t n_err;                    // differing floats
    uint32_t by_out[8];                // output m (0..3) x component (re, im)
    uint32_t by_quarter[4];            // lane >> 4
    uint32_t n_log;
    ErrRec log[256];
    float sink;
};

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(int32_t)h * (1.0f / 2147483648.0f); }     // (-1, 1)

template <int NS>
__device__ void check_role(ProbeOut* o, int iters, uint32_t rep) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t seed = hash32(blockIdx.x * 1024u + threadIdx.x) ^ hash32(rep * 2654435761u);
    unsigned long long mine = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 x[NS][4], w[4], y[NS][4], r[NS][4];
#pragma unroll
        for (int m = 1; m < 4; ++m) {
            const float ang = unit(seed = hash32(seed + m)) * 3.14159265f;
            w[m] = f32x2{__cosf(ang), -__sinf(ang)};
        }
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) x[n][m] = f32x2{unit(seed = hash32(seed + 17u)) * 8.0f, unit(seed = hash32(seed + 29u)) * 8.0f};
        // packed form: the statement order of lsc_fft_stages
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            f32x2 t[4];
            t[0] = x[n][0];
#pragma unroll
            for (int m = 1; m < 4; ++m) t[m] = pk_cmul(x[n][m], w[m]);
            const f32x2 a = t[0] + t[2], b = t[0] - t[2], c = t[1] + t[3], d = t[1] - t[3];
            y[n][0] = a + c;
            y[n][1] = pk_add_mi(b, d);
            y[n][2] = a - c;
            y[n][3] = pk_add_pi(b, d);
        }
        // scalar form
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            f32x2 t[4];
            t[0] = x[n][0];
#pragma unroll
            for (int m = 1; m < 4; ++m) t[m] = sc_cmul(x[n][m], w[m]);
            const f32x2 a = f32x2{s_add(t[0][0], t[2][0]), s_add(t[0][1], t[2][1])}, b = f32x2{s_sub(t[0][0], t[2][0]), s_sub(t[0][1], t[2][1])};
            const f32x2 c = f32x2{s_add(t[1][0], t[3][0]), s_add(t[1][1], t[3][1])}, d = f32x2{s_sub(t[1][0], t[3][0]), s_sub(t[1][1], t[3][1])};
            r[n][0] = f32x2{s_add(a[0], c[0]), s_add(a[1], c[1])};
            r[n][1] = f32x2{s_add(b[0], d[1]), s_sub(b[1], d[0])};
            r[n][2] = f32x2{s_sub(a[0], c[0]), s_sub(a[1], c[1])};
            r[n][3] = f32x2{s_sub(b[0], d[1]), s_add(b[1], d[0])};
        }
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int cpt = 0; cpt < 2; ++cpt) {
                    const uint32_t g = __builtin_bit_cast(uint32_t, (float)y[n][m][cpt]), e = __builtin_bit_cast(uint32_t, (float)r[n][m][cpt]);
                    if (g != e) {
                        atomicAdd(&o->n_err, 1u);
                        atomicAdd(&o->by_out[2 * m + cpt], 1u);
                        atomicAdd(&o->by_quarter[lane >> 4], 1u);
                        const uint32_t k = atomicAdd(&o->n_log, 1u);
                        if (k < 256) o->log[k] = ErrRec{blockIdx.x, wave, lane, (uint32_t)(2 * m + cpt), rep, (uint32_t)it, g, e};
                    }
                }
        mine += NS;
    }
    // one add per wave
    for (int s = 32; s; s >>= 1) mine += __shfl_xor((long long)mine, s);
    if (lane == 0) atomicAdd(&o->checked, mine);
}

__device__ void matrix_role(ProbeOut* o, int iters, int mode) {
    const uint32_t lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    uint32_t h = hash32(blockIdx.x * 977u + threadIdx.x);
    if (mode & 2) {            // no matrix instruction: VALU filler of about the same duration
        float f = unit(h);
        for (int it = 0; it < iters * 64; ++it) f = __builtin_fmaf(f, 0.999f, 0.001f);
        if (f == 123.456f) o->sink = f;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        if (mode & 1) {
            const float pa = unit(h = hash32(h + 1)), pb = unit(h = hash32(h + 2));
#pragma unroll
            for (int rpt = 0; rpt < 4; ++rpt)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, pb, acc[k], 0, 0, 0);
        } else {
            bf16x8 pa, pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pa[e] = (__bf16)unit(h = hash32(h + e));
                pb[e] = (__bf16)unit(h = hash32(h + 31u * e));
            }
            // the LS kernel's despread of one chunk: 2 bins x 2 planes x 3 piece products = 12 MFMAs of 8 passes; here 16 per iteration
#pragma unroll
            for (int rpt = 0; rpt < 4; ++rpt)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[k], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    if (s == 123.456f && lane == 0) o->sink = s;
}

// mode bit 1: fp32 MFMA, bit 2: no MFMA, bit 4: every workgroup checks (no matrix workgroups), bit 8: roles swapped (low blocks check)
__global__ __launch_bounds__(256, 2) void probe_kernel(ProbeOut* o, int iters, int mode, uint32_t rep) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0 && iters < 0) smem[0] = 1.f;          // (the allocation is what matters: two workgroups per CU)
    const bool high = blockIdx.x >= gridDim.x / 2;
    const bool check = (mode & 4) || (high != ((mode & 8) != 0));
    if (check) check_role<2>(o, iters, rep);
    else matrix_role(o, iters * 2, mode);
}

// "another kernel" between two probe launches (the LS events need it): streams through a buffer
__global__ void other_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
    const int mode = argc > 2 ? (int)strtol(argv[2], nullptr, 0) : 0;
    const int idle_ms = argc > 3 ? atoi(argv[3]) : 0;
    const int iters = argc > 4 ? atoi(argv[4]) : 64;
    const size_t lds = 78 * 1024;
    HIP_OK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProbeOut* d_o;
    HIP_OK(hipMalloc(&d_o, sizeof(ProbeOut)));
    HIP_OK(hipMemset(d_o, 0, sizeof(ProbeOut)));
    float* d_buf;
    const size_t nbuf = (size_t)8 << 20;
    HIP_OK(hipMalloc(&d_buf, nbuf * sizeof(float)));
    HIP_OK(hipMemset(d_buf, 0, nbuf * sizeof(float)));
    std::vector<ProbeOut> h(1);
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t rep = 0, last_err = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(other_kernel, dim3(1024), dim3(256), 0, 0, d_buf, nbuf);
        if (idle_ms > 0 && (rep & 15) == 0) {
            HIP_OK(hipDeviceSynchronize());
            std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
        }
        hipLaunchKernelGGL(probe_kernel, dim3(512), dim3(256), lds, 0, d_o, iters, mode, rep);
        ++rep;
        if (rep % 1000 == 0) {
            HIP_OK(hipMemcpy(h.data(), d_o, sizeof(ProbeOut), hipMemcpyDeviceToHost));
            if (h[0].n_err != last_err) printf("launch %u: %u differing values so far\n", rep, h[0].n_err);
            last_err = h[0].n_err;
        }
    }
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h.data(), d_o, sizeof(ProbeOut), hipMemcpyDeviceToHost));
    const ProbeOut& r = h[0];
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d idle %d ms: %u launches in %.1f s, %.3e butterflies checked (x 3 twiddle products + 2 op_sel adds each), %u differing values\n", mode,
           idle_ms, rep, dt, (double)r.checked, r.n_err);
    printf("  by output (m, re|im): ");
    for (int k = 0; k < 8; ++k) printf("%u ", r.by_out[k]);
    printf("\n  by lane quarter (0-15, 16-31, 32-47, 48-63): %u %u %u %u\n", r.by_quarter[0], r.by_quarter[1], r.by_quarter[2], r.by_quarter[3]);
    for (uint32_t k = 0; k < r.n_log && k < 256 && k < 24; ++k)
        printf("  block %u wave %u lane %u output %u.%s launch %u iteration %u: got %08x want %08x\n", r.log[k].block, r.log[k].wave, r.log[k].lane,
               r.log[k].out >> 1, (r.log[k].out & 1) ? "im" : "re", r.log[k].rep, r.log[k].it, r.log[k].got, r.log[k].want);
    return r.n_err ? 1 : 0;
}
