// pk_opsel_probe.hip - does a packed-fp32 instruction with op_sel (a cross-half operand) return a wrong result while ANOTHER kernel (bf16 MFMAs, on its
// own stream) starts and stops beside it?  The LS transform's +-i rotations were `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] neg_*`; in calls whose
// second stream ran a bf16 MFMA GEMM beside the LS kernel they came back wrong in lanes 48-63 (profiles/r06_small_calls.txt (4), (5)).  This probe has
// no LS code in it: victim waves run chains of packed operations in several forms next to the same arithmetic in single operations (inline asm, so the
// compiler cannot pack them) and count the differences per form and per 16-lane quarter; the aggressor is a loop of v_mfma_f32_32x32x16_bf16.
// build: hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_probe.hip -o tools/pk_opsel_probe      run: tools/pk_opsel_probe [rounds] [lds_kib] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NFORM = 8;
// forms: 0 v_pk_add op_sel swap + neg_hi (a - i b)   1 v_pk_add op_sel swap + neg_lo (a + i b)   2 v_pk_add op_sel swap, no neg
//        3 v_pk_mul op_sel:[0,0] op_sel_hi:[1,0] (broadcast lo)   4 v_pk_fma op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]
//        5 v_pk_mov_b32 op_sel:[1,0] (swap halves)   6 v_pk_add plain (no op_sel: control)   7 v_pk_add neg_lo/neg_hi only (control)
__device__ __forceinline__ unsigned bits(float x) { return __builtin_bit_cast(unsigned, x); }

__global__ __launch_bounds__(256, 2) void victim(unsigned* counts, unsigned* samples, int iters, unsigned seed, int lds_floats) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // per-lane operands: distinct halves, changed every iteration by an LCG kept in integer registers
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u);
    unsigned bad[NFORM] = {0, 0, 0, 0, 0, 0, 0, 0};
    // some LDS traffic as in the transform (the stage images): written and read back each iteration
    float* my = lds + (lds_floats >= 256 * 4 ? tid * 4 : 0);
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        f32x2 a = {(float)(int)(s >> 8) * (1.0f / 65536.0f) - 100.f, (float)(int)((s * 7u) >> 8) * (1.0f / 65536.0f) + 3.f};
        s = s * 1664525u + 1013904223u;
        f32x2 b = {(float)(int)(s >> 8) * (1.0f / 32768.0f) + 17.f, (float)(int)((s * 13u) >> 8) * (1.0f / 65536.0f) - 41.f};
        if (lds_floats >= 256 * 4) {
            *reinterpret_cast<f32x2*>(my) = a; *reinterpret_cast<f32x2*>(my + 2) = b;
            __builtin_amdgcn_wave_barrier();
            a = *reinterpret_cast<f32x2*>(my); b = *reinterpret_cast<f32x2*>(my + 2);
        }
        f32x2 p, t;
        float r0, r1;
        // 0
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) { if (!bad[0]) { samples[0 * 64 + lane] = bits(p[0]); } ++bad[0]; }
        // 1
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
        asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[1];
        // 2
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[2];
        // 3
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(t) : "v"(a), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
        if (bits(t[0]) != bits(r0) || bits(t[1]) != bits(r1)) ++bad[3];
        // 4   d = (-a.hi b.hi + t.lo, a.lo b.hi + t.hi)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(p) : "v"(a), "v"(b), "v"(t));
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=&v"(r0) : "v"(a[1]), "v"(b[1]), "v"(t[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r1) : "v"(a[0]), "v"(b[1]), "v"(t[1]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[4];
        // 5
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(p) : "v"(a), "v"(a));
        if (bits(p[0]) != bits(a[1]) || bits(p[1]) != bits(a[0])) ++bad[5];
        // 6
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(p) : "v"(a), "v"(b));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[1]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[6];
        // 7
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
        asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[1]));
        if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[7];
    }
#pragma unroll
    for (int f = 0; f < NFORM; ++f)
        if (bad[f]) atomicAdd(&counts[f * 4 + (lane >> 4)], bad[f]);
    if (tid == 0) atomicAdd(&counts[NFORM * 4], 1u);          // workgroups that ran to the end
}

// aggressor: bf16 MFMAs back to back, 8 waves per workgroup, one workgroup per CU and more
__global__ __launch_bounds__(512, 1) void aggressor(float* sink, int iters, int lds_bytes_used) {
    extern __shared__ float lds[];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (lds_bytes_used) lds[threadIdx.x] = s;
    if (s == 123.456f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const int lds_kib = argc > 2 ? atoi(argv[2]) : 70;          // the victim's LDS allocation (70 KiB: two workgroups per CU, as the LS kernel at Nt = 64)
    const int mode = argc > 3 ? atoi(argv[3]) : 0;              // 0: aggressor launched behind the victim on a second stream; 1: no aggressor; 2: aggressor first
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    unsigned *counts, *samples; float* sink;
    CHECK(hipMalloc(&counts, (NFORM * 4 + 4) * 4)); CHECK(hipMalloc(&samples, NFORM * 64 * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(counts, 0, (NFORM * 4 + 4) * 4)); CHECK(hipMemset(samples, 0, NFORM * 64 * 4));
    CHECK(hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024));
    CHECK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t fork; CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    const int viters = 3000, aiters = 20000;
    for (int r = 0; r < rounds; ++r) {
        if (mode == 2) hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), 128 * 1024, sb, sink, aiters, 1);
        if (mode == 0) { CHECK(hipEventRecord(fork, sa)); CHECK(hipStreamWaitEvent(sb, fork, 0)); }
        // 2000 workgroups over 512 slots: four rounds, as the 1000-packet LS launch
        hipLaunchKernelGGL(victim, dim3(2000), dim3(256), lds_kib * 1024, sa, counts, samples, viters, 12345u + r, lds_kib * 256);
        CHECK(hipGetLastError());
        if (mode == 0) hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), 128 * 1024, sb, sink, aiters, 1);
        CHECK(hipGetLastError());
        CHECK(hipStreamSynchronize(sa)); CHECK(hipStreamSynchronize(sb));
    }
    std::vector<unsigned> h(NFORM * 4 + 4);
    CHECK(hipMemcpy(h.data(), counts, (NFORM * 4 + 4) * 4, hipMemcpyDeviceToHost));
    printf("victim workgroups completed: %u of %d\n", h[NFORM * 4], rounds * 2000);
    const char* names[NFORM] = {"v_pk_add_f32 op_sel swap neg_hi", "v_pk_add_f32 op_sel swap neg_lo", "v_pk_add_f32 op_sel swap", "v_pk_mul_f32 op_sel_hi:[1,0]",
                                "v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_add_f32 plain", "v_pk_add_f32 neg only"};
    printf("rounds %d, victim LDS %d KiB, mode %d: wrong packed results per 16-lane quarter (lanes 0-15, 16-31, 32-47, 48-63) of %.3g per form\n", rounds, lds_kib, mode,
           (double)rounds * 2000 * 256 * viters);
    for (int f = 0; f < NFORM; ++f) printf("  %-58s %10u %10u %10u %10u\n", names[f], h[f * 4], h[f * 4 + 1], h[f * 4 + 2], h[f * 4 + 3]);
    return 0;
}
