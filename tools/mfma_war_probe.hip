// mfma_war_probe.hip - which hazard did the "drain" of ls_estimate_ringb_kernel hide (DESIGN.md 4.2, round-3 verdict item 5)?
//
// Hypothesis: an MFMA that a wave has issued while the SIMD's matrix pipe is busy with the OTHER wave's MFMAs reads its A / B
// source registers only when it starts - later than "issue" - and the return of an LDS read issued right behind it may land in
// those registers first (LDS data returns asynchronously: no interlock protects registers an in-flight MFMA still has to read;
// hipcc's hazard recogniser inserts no wait for "MFMA reads A/B, then a load writes them").
//
// Test: a 512-thread workgroup = two waves per SIMD.  "Victim" waves run [chain of L dependent MFMAs on A, B] -> [P idle
// wait states] -> [ds_read_b128 into the A and B registers of other values] -> wait -> compare the accumulator with the exact
// expected value.  "Hammer" waves (their SIMD partners) issue independent MFMAs back to back, or stay idle.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_war_probe.hip -o tools/mfma_war_probe.bin      Run: no arguments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CHAIN, int PAD, int OVER>      // OVER: 0 nothing, 1 ds_read_b128 into A and B, 2 ds_read into A only, 3 ds_read into B only
__global__ __launch_bounds__(512) void probe(unsigned* errs, unsigned* first_bad, int iters, int hammer) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 4; i += 512) lds[i] = 0x40004000u;       // bf16 (2.0, 2.0)
    __syncthreads();
    // waves w and w + 4 share a SIMD (dispatch order 0 -> 2 -> 1 -> 3 per wave quadruple): waves 0-3 hammer, 4-7 are the victims
    if (wave < 4) {
        if (!hammer) return;
        f32x16 h0 = {}, h1 = {}, h2 = {}, h3 = {};
        u32x4 x = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        for (int it = 0; it < iters * (CHAIN + 8); ++it) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %4, %0\n\t"
                         "v_mfma_f32_32x32x16_bf16 %1, %4, %4, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %4, %2\n\t"
                         "v_mfma_f32_32x32x16_bf16 %3, %4, %4, %3\n\t"
                         : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(x));
        }
        if (h0[0] + h1[0] + h2[0] + h3[0] == 12345.f) errs[1] = 1;          // keep the chain alive
        return;
    }
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds + lane * 16;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 A = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};       // bf16 1.0 everywhere: every product 1, every output CHAIN * 16
        u32x4 B = A;
        f32x16 acc = {};
        asm volatile("" : "+v"(A), "+v"(B), "+v"(acc));
        asm volatile(
            ".rept %c4\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
            ".endr\n\t"
            ".rept %c5\n\t"
            "s_nop 0\n\t"
            ".endr\n\t"
            ".if %c6 == 1 || %c6 == 2\n\t"
            "ds_read_b128 %1, %3\n\t"
            ".endif\n\t"
            ".if %c6 == 1 || %c6 == 3\n\t"
            "ds_read_b128 %2, %3\n\t"
            ".endif\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t"
            "s_nop 15\n\t"
            "s_nop 15\n\t"
            "s_nop 15\n\t"
            : "+v"(acc), "+v"(A), "+v"(B) : "v"(addr), "n"(CHAIN), "n"(PAD), "n"(OVER) : "memory");
        const float want = 16.f * CHAIN;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (acc[e] != want) { ++bad; if (!atomicAdd(&first_bad[0], 1u)) { first_bad[1] = __builtin_bit_cast(unsigned, acc[e]); first_bad[2] = it; first_bad[3] = wave * 64 + lane; } }
        if (OVER && (A[0] != 0x40004000u && (OVER == 1 || OVER == 2))) ++bad;       // the overwrite itself must have happened
    }
    if (bad) atomicAdd(&errs[0], bad);
}

// the same with a VALU overwrite (v_mov_b32 into every A and B register right behind the chain): physical registers named in the asm
template <int CHAIN, int PAD>
__global__ __launch_bounds__(512) void probe_valu(unsigned* errs, unsigned* first_bad, int iters, int hammer) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave < 4) {
        if (!hammer) return;
        f32x16 h0 = {}, h1 = {}, h2 = {}, h3 = {};
        u32x4 x = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        for (int it = 0; it < iters * (CHAIN + 8); ++it) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %4, %0\n\t"
                         "v_mfma_f32_32x32x16_bf16 %1, %4, %4, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %4, %2\n\t"
                         "v_mfma_f32_32x32x16_bf16 %3, %4, %4, %3\n\t"
                         : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(x));
        }
        if (h0[0] + h1[0] + h2[0] + h3[0] == 12345.f) errs[1] = 1;
        return;
    }
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r15;
        asm volatile(
            "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t"
            "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t"
            "s_nop 4\n\t"
            "v_mfma_f32_32x32x16_bf16 v[110:125], v[100:103], v[104:107], 0\n\t"
            ".rept %c2 - 1\n\t"
            "v_mfma_f32_32x32x16_bf16 v[110:125], v[100:103], v[104:107], v[110:125]\n\t"
            ".endr\n\t"
            ".rept %c3\n\t"
            "s_nop 0\n\t"
            ".endr\n\t"
            "v_mov_b32 v100, 0x40004000\n\tv_mov_b32 v101, 0x40004000\n\tv_mov_b32 v102, 0x40004000\n\tv_mov_b32 v103, 0x40004000\n\t"
            "v_mov_b32 v104, 0x40004000\n\tv_mov_b32 v105, 0x40004000\n\tv_mov_b32 v106, 0x40004000\n\tv_mov_b32 v107, 0x40004000\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v110\n\t"
            "v_mov_b32 %1, v125\n\t"
            : "=v"(r0), "=v"(r15) : "n"(CHAIN), "n"(PAD)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",
              "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125");
        const float want = 16.f * CHAIN;
        if (r0 != want || r15 != want) { ++bad; if (!atomicAdd(&first_bad[0], 1u)) { first_bad[1] = __builtin_bit_cast(unsigned, r0 != want ? r0 : r15); first_bad[2] = it; first_bad[3] = wave * 64 + lane; } }
    }
    if (bad) atomicAdd(&errs[0], bad);
}

template <int CHAIN, int PAD>
static void run_valu(unsigned* d, int hammer, int blocks) {
    CK(hipMemset(d, 0, 64));
    hipLaunchKernelGGL((probe_valu<CHAIN, PAD>), dim3(blocks), dim3(512), 0, 0, d, d + 4, 2000, hammer);
    CK(hipDeviceSynchronize());
    unsigned h[8];
    CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("chain %d  pad %2d wait states  overwrite A + B by v_mov_b32  partner %-9s : %8u wrong accumulator values", CHAIN, PAD, hammer ? "hammering" : "idle", h[0]);
    if (h[0]) printf("   (first: %g instead of %g, iteration %u, thread %u)", __builtin_bit_cast(float, h[5]), 16.f * CHAIN, h[6], h[7]);
    printf("\n");
}

template <int CHAIN, int PAD, int OVER>
static void run(unsigned* d, int hammer, int blocks) {
    CK(hipMemset(d, 0, 64));
    hipLaunchKernelGGL((probe<CHAIN, PAD, OVER>), dim3(blocks), dim3(512), 0, 0, d, d + 4, 2000, hammer);
    CK(hipDeviceSynchronize());
    unsigned h[8];
    CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("chain %d  pad %2d wait states  overwrite %s  partner %-9s : %8u wrong accumulator values", CHAIN, PAD,
           OVER == 0 ? "none " : OVER == 1 ? "A + B" : OVER == 2 ? "A    " : "B    ", hammer ? "hammering" : "idle", h[0]);
    if (h[0]) printf("   (first: %g instead of %g, iteration %u, thread %u)", __builtin_bit_cast(float, h[5]), 16.f * CHAIN, h[6], h[7]);
    printf("\n");
}

int main() {
    unsigned* d;
    CK(hipMalloc(&d, 64));
    const int blocks = 1024;       // 4 rounds of 256 CUs
    for (int hammer = 0; hammer < 2; ++hammer) {
        run<6, 0, 0>(d, hammer, blocks);
        run<1, 0, 1>(d, hammer, blocks);
        run<6, 0, 1>(d, hammer, blocks);
        run<6, 0, 2>(d, hammer, blocks);
        run<6, 0, 3>(d, hammer, blocks);
        run<6, 1, 1>(d, hammer, blocks);
        run<6, 2, 1>(d, hammer, blocks);
        run<6, 4, 1>(d, hammer, blocks);
        run<6, 8, 1>(d, hammer, blocks);
        run<6, 16, 1>(d, hammer, blocks);
        run<6, 32, 1>(d, hammer, blocks);
        run<6, 64, 1>(d, hammer, blocks);
        run<12, 0, 1>(d, hammer, blocks);
        run_valu<1, 0>(d, hammer, blocks);
        run_valu<6, 0>(d, hammer, blocks);
        run_valu<6, 1>(d, hammer, blocks);
        run_valu<6, 4>(d, hammer, blocks);
        run_valu<12, 0>(d, hammer, blocks);
    }
    return 0;
}
