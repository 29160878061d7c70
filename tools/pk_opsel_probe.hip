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

constexpr int NFORM = 44;
// forms: 0 v_pk_add op_sel swap + neg_hi (a - i b)   1 v_pk_add op_sel swap + neg_lo (a + i b)   2 v_pk_add op_sel swap, no neg
//        3 v_pk_mul op_sel:[0,0] op_sel_hi:[1,0] (broadcast lo)   4 v_pk_fma op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]
//        5 v_pk_mov_b32 op_sel:[1,0] (swap halves)   6 v_pk_add plain (no op_sel: control)   7 v_pk_add neg_lo/neg_hi only (control)
__device__ __forceinline__ unsigned bits(float x) { return __builtin_bit_cast(unsigned, x); }

__global__ __launch_bounds__(256, 2) void victim(unsigned* counts, unsigned* samples, int iters, unsigned seed, int lds_floats, float* scratch, const float* dma_src) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // per-lane operands: distinct halves, changed every iteration by an LCG kept in integer registers
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (tid * 40503u);
    unsigned bad[NFORM] = {};
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
        // 26-33: which operand / which modifier.  Each against single operations.
        {
            f32x2 c = {a[1] * 0.5f + 1.f, b[0] - 3.f};
            // 26 v_pk_add: src0 swapped
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[1]), "v"(b[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[0]), "v"(b[1]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[26];
            // 27 v_pk_add: src1 hi for both halves (op_sel:[0,1] op_sel_hi:[1,1])
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[1]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[27];
            // 28 v_pk_add: src1 lo for both halves (op_sel:[0,0] op_sel_hi:[1,0]) - what the compiler writes for vector + scalar
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[28];
            // 29 v_pk_mul: src1 swapped
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[29];
            // 30 v_pk_fma: src1 swapped
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(p) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r0) : "v"(a[0]), "v"(b[1]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r1) : "v"(a[1]), "v"(b[0]), "v"(c[1]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[30];
            // 31 v_pk_fma: src2 swapped
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(p) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r0) : "v"(a[0]), "v"(b[0]), "v"(c[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r1) : "v"(a[1]), "v"(b[1]), "v"(c[0]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[31];
            // 32 v_pk_add with src1 swapped, the operands in the other order (swap on the FIRST source of a commutative add: op_sel:[1,0] op_sel_hi:[0,1] is 26)
            //    here: an SGPR-free check of the swap with a constant second source
            asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(p) : "v"(a));
            asm volatile("v_add_f32 %0, 1.0, %1" : "=&v"(r0) : "v"(a[1]));
            asm volatile("v_add_f32 %0, 1.0, %1" : "=&v"(r1) : "v"(a[0]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[32];
            // 33 v_pk_mul: src0 swapped
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[1]), "v"(b[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[0]), "v"(b[1]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[33];
        }
        // 34-43: 16-bit packed forms, the mix instructions and the compiler's 64-bit copy.  References: the same instruction WITHOUT the select on a
        // pre-arranged operand (v_perm / v_alignbit build the swapped register in single operations).
        {
            const unsigned ua = bits(a[0]) ^ (bits(b[1]) << 3), ub = bits(b[0]) ^ (bits(a[1]) >> 5);      // two f16 pairs (any bit pattern: compare bits)
            unsigned ub_sw, up, ur;
            asm volatile("v_alignbit_b32 %0, %1, %1, 16" : "=&v"(ub_sw) : "v"(ub));                          // halves of ub swapped
            // 34 v_pk_add_f16 src1 swapped
            asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(up) : "v"(ua), "v"(ub));
            asm volatile("v_pk_add_f16 %0, %1, %2" : "=&v"(ur) : "v"(ua), "v"(ub_sw));
            if (up != ur) ++bad[34];
            // 35 v_pk_mul_f16 src1 swapped
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(up) : "v"(ua), "v"(ub));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=&v"(ur) : "v"(ua), "v"(ub_sw));
            if (up != ur) ++bad[35];
            // 36 v_pk_fma_f16 src1 swapped
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(up) : "v"(ua), "v"(ub), "v"(ua));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=&v"(ur) : "v"(ua), "v"(ub_sw), "v"(ua));
            if (up != ur) ++bad[36];
            // 37 v_pk_add_u16 src1 swapped
            asm volatile("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(up) : "v"(ua), "v"(ub));
            asm volatile("v_pk_add_u16 %0, %1, %2" : "=&v"(ur) : "v"(ua), "v"(ub_sw));
            if (up != ur) ++bad[37];
            // 38 v_pk_max_i16 src1 swapped
            asm volatile("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(up) : "v"(ua), "v"(ub));
            asm volatile("v_pk_max_i16 %0, %1, %2" : "=&v"(ur) : "v"(ua), "v"(ub_sw));
            if (up != ur) ++bad[38];
            // 39 v_fma_mix_f32: src1 = the HIGH f16 of ub  against  src1 = the low f16 of the swapped register
            float fm, fr;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=&v"(fm) : "v"(a[0]), "v"(ub), "v"(b[0]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=&v"(fr) : "v"(a[0]), "v"(ub_sw), "v"(b[0]));
            if (bits(fm) != bits(fr)) ++bad[39];
            // 40 v_fma_mixlo_f16 with the high f16 of src1
            unsigned m0 = 0, m1 = 0;
            asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(m0) : "v"(a[0]), "v"(ub), "v"(b[0]));
            asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "+v"(m1) : "v"(a[0]), "v"(ub_sw), "v"(b[0]));
            if (m0 != m1) ++bad[40];
            // 41 the compiler's 64-bit copy: v_pk_mov_b32 d, s, s op_sel:[0,1]  (D.lo = src0.lo, D.hi = src1.hi)
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=&v"(p) : "v"(a), "v"(b));
            if (bits(p[0]) != bits(a[0]) || bits(p[1]) != bits(b[1])) ++bad[41];
            // 42 v_pk_add_f32 with op_sel on src0 AND src1 (both halves swapped)
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=&v"(p) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[1]), "v"(b[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[0]), "v"(b[0]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[42];
            // 43 v_pk_add_f32 with the SAME register pair as both sources, second one swapped ( (x + y, y + x) )
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(p) : "v"(a));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(a[1]));
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r0)) ++bad[43];
        }
        // 8-11: the butterfly's dependent chain in ONE asm statement - two packed differences (the producers), then the op_sel rotation that reads them,
        // with 0 / 1 / 2 / 4 idle issue slots in between; the reference from single operations
        {
            f32x2 x0 = a, x1 = b, x2 = {b[1] * 0.5f, a[0] + 1.f}, x3 = {a[1] - 2.f, b[0] * 0.25f};
            float bb0, bb1, dd0, dd1;
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(bb0) : "v"(x0[0]), "v"(x2[0]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(bb1) : "v"(x0[1]), "v"(x2[1]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(dd0) : "v"(x1[0]), "v"(x3[0]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(dd1) : "v"(x1[1]), "v"(x3[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(bb0), "v"(dd1));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(bb1), "v"(dd0));
            f32x2 pb, pd;
#define CHAIN(GAP)                                                                                                                        \
            asm volatile("v_pk_add_f32 %1, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                          \
                         "v_pk_add_f32 %2, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t" GAP                                                       \
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"                                               \
                         : "=&v"(p), "=&v"(pb), "=&v"(pd) : "v"(x0), "v"(x1), "v"(x2), "v"(x3))
            CHAIN("");
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[8];
            CHAIN("s_nop 0\n\t");
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[9];
            CHAIN("s_nop 1\n\t");
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[10];
            CHAIN("s_nop 3\n\t");
            if (bits(p[0]) != bits(r0) || bits(p[1]) != bits(r1)) ++bad[11];
#undef CHAIN
        }
        // 12-19: the same rotation while LDS reads issued just before it RETURN into other registers of the wave (the transform keeps the next row's
        // ds_read_b64 in flight under the butterflies): eight reads, N x 16 idle cycles, eight rotations back to back, then the wait
        if (lds_floats >= 256 * 4) {
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)my;
            f32x2 d0, d1, d2, d3, d4, d5, d6, d7, p0, p1, p2, p3, p4, p5, p6, p7;
#define RET(DELAY, SLOT)                                                                                                                     \
            asm volatile("ds_read_b64 %8, %18\n\tds_read_b64 %9, %18 offset:8\n\tds_read_b64 %10, %18\n\tds_read_b64 %11, %18 offset:8\n\t"          \
                         "ds_read_b64 %12, %18\n\tds_read_b64 %13, %18 offset:8\n\tds_read_b64 %14, %18\n\tds_read_b64 %15, %18 offset:8\n\t" DELAY \
                         "v_pk_add_f32 %0, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
                         "v_pk_add_f32 %2, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
                         "v_pk_add_f32 %4, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %5, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
                         "v_pk_add_f32 %6, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %7, %16, %17 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
                         "s_waitcnt lgkmcnt(0)"                                                                                              \
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3),  \
                           "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)                                                                       \
                         : "v"(a), "v"(b), "v"(la) : "memory");                                                                              \
            if (bits(p0[0]) != bits(r0) || bits(p1[0]) != bits(r0) || bits(p2[0]) != bits(r0) || bits(p3[0]) != bits(r0) || bits(p4[0]) != bits(r0) ||   \
                bits(p5[0]) != bits(r0) || bits(p6[0]) != bits(r0) || bits(p7[0]) != bits(r0) || bits(p0[1]) != bits(r1) || bits(p7[1]) != bits(r1)) ++bad[SLOT]
            RET("", 12);
            RET("s_nop 15\n\t", 13);
            RET("s_nop 15\n\ts_nop 15\n\t", 14);
            RET("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t", 15);
            RET("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t", 16);
            RET("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t", 17);
            RET("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t", 18);
            RET("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t", 19);
#undef RET
        }
        // 20-23: the rotation while STORES issued just before it still fetch their data registers (the LS kernel stores a finished item - 128
        // global_store_dword per thread - and goes straight on to the next chunk's butterflies): 16 / 32 stores, then rotations back to back for a while
        {
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
            float* sp = scratch + ((size_t)blockIdx.x * 256 + tid) * 4;
            f32x2 q0, q1, q2, q3;
#define ST4 "global_store_dword %4, %6, off\n\tglobal_store_dword %4, %7, off offset:4\n\tglobal_store_dword %4, %8, off offset:8\n\tglobal_store_dword %4, %9, off offset:12\n\t"
#define ROT4 "v_pk_add_f32 %0, %10, %11 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %10, %11 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
             "v_pk_add_f32 %2, %10, %11 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %10, %11 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
#define STORES(NST, NROT, SLOT)                                                                                                       \
            asm volatile(NST NROT : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(sp), "v"(0), "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "v"(a), "v"(b) : "memory"); \
            if (bits(q0[0]) != bits(r0) || bits(q1[0]) != bits(r0) || bits(q2[0]) != bits(r0) || bits(q3[0]) != bits(r0) || bits(q0[1]) != bits(r1) || bits(q3[1]) != bits(r1)) ++bad[SLOT]
            STORES(ST4 ST4 ST4 ST4, ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4, 20);
            STORES(ST4 ST4 ST4 ST4 ST4 ST4 ST4 ST4, ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4, 21);
            STORES(ST4 ST4 ST4 ST4 ST4 ST4 ST4 ST4, "s_nop 15\n\ts_nop 15\n\t" ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4, 22);
            STORES(ST4 ST4 ST4 ST4 ST4 ST4 ST4 ST4, "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t" ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4, 23);
#undef STORES
        }
        // 24-25: LDS-DMA (global_load_lds_dwordx4, M0 = the destination) issued by this wave AND by the CU's other waves while rotations run: the LS kernel's ring
        if (lds_floats >= 256 * 4 + 4 * 1024) {
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(a[0]), "v"(b[1]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=&v"(r1) : "v"(a[1]), "v"(b[0]));
            const float* src = dma_src + (((size_t)blockIdx.x * 4 + (tid >> 6)) * 4096 + (size_t)(it & 63) * 256 + lane * 4) % ((size_t)(64 << 20) / 4);
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + 256 * 4 + (tid >> 6) * 1024));
            f32x2 q0, q1, q2, q3;
#define DMA1 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
#define ROT4 "v_pk_add_f32 %0, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t" \
             "v_pk_add_f32 %2, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            asm volatile(DMA1 ROT4 ROT4 DMA1 ROT4 ROT4 DMA1 ROT4 ROT4 DMA1 ROT4 ROT4 "s_waitcnt vmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(src), "s"(dst), "v"(a), "v"(b) : "memory");
            if (bits(q0[0]) != bits(r0) || bits(q1[0]) != bits(r0) || bits(q2[0]) != bits(r0) || bits(q3[0]) != bits(r0) || bits(q0[1]) != bits(r1) || bits(q3[1]) != bits(r1)) ++bad[24];
            asm volatile(DMA1 DMA1 DMA1 DMA1 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 ROT4 "s_waitcnt vmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(src), "s"(dst), "v"(a), "v"(b) : "memory");
            if (bits(q0[0]) != bits(r0) || bits(q1[0]) != bits(r0) || bits(q2[0]) != bits(r0) || bits(q3[0]) != bits(r0) || bits(q0[1]) != bits(r1) || bits(q3[1]) != bits(r1)) ++bad[25];
#undef DMA1
#undef ROT4
        }
    }
#pragma unroll
    for (int f = 0; f < NFORM; ++f)
        if (bad[f]) atomicAdd(&counts[f * 4 + (lane >> 4)], bad[f]);
    if (tid == 0) atomicAdd(&counts[NFORM * 4], 1u);          // workgroups that ran to the end
}

// aggressor: bf16 MFMAs back to back, 8 waves per workgroup, one workgroup per CU and more
__global__ __launch_bounds__(512) void aggressor(float* sink, int iters, int lds_bytes_used) {
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

// co-resident aggressors (modes 10 + kind): 256 threads, 16 KiB of LDS - its waves sit on the victims' SIMDs.  kind bits: 1 MFMA bf16, 2 ds_read_b128 feeding
// the MFMA operands, 4 s_barrier per iteration, 8 global_load_dwordx4 per iteration, 16 ds_write_b128, 32 MFMA f16 instead of bf16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void aggressor2(float* sink, const float4* gsrc, int iters, int kind) {
    __shared__ float4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = float4{0.001f * i, 0.002f, 0.003f, 0.004f};
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float4 g = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (kind & 2) {
            const float4 va = sh[(threadIdx.x + it) & 1023], vb = sh[(threadIdx.x * 3 + it) & 1023];
            a = __builtin_bit_cast(bf16x8, va); b = __builtin_bit_cast(bf16x8, vb);
        }
        if (kind & 8) g = gsrc[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)it * 65536) & ((1u << 22) - 1)];
        if (kind & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        }
        if (kind & 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[k], 0, 0, 0);
        }
        if (kind & 16) sh[(threadIdx.x + it) & 1023] = float4{acc[0][0], g.x, acc[1][1], g.y};
        if (kind & 4) __syncthreads();
    }
    float s = g.x + g.y;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 123.456f) sink[0] = s + sh[threadIdx.x].x;
}

// modes 5 / 6: the same tight loop of four independent MFMAs, written in asm so that ONLY the register file of the accumulators differs:
// 5 = accumulators in AGPRs (a[..]: what the product's GEMM kernels and hipcc's large kernels use), 6 = accumulators in arch VGPRs (v[..])
template <bool AGPR>
__global__ __launch_bounds__(256) void aggressor3(float* sink, int iters) {
    __shared__ float pad[4096];          // 16 KiB: one workgroup per CU beside the victims (one MFMA wave per SIMD)
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        if (AGPR)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 123.456f) { pad[threadIdx.x] = s; sink[0] = s + pad[(threadIdx.x + 1) & 4095]; }
}

// memory aggressor (mode 3 / 4): a streaming copy with the GEMM's LDS footprint (it cannot share a CU with a victim workgroup either)
__global__ __launch_bounds__(512, 1) void mem_aggressor(const float4* src, float4* dst, size_t n4, int reps) {
    extern __shared__ float lds[];
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) dst[i] = src[i];
    if (n4 == 1) lds[threadIdx.x] = 0.f;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const int lds_kib = argc > 2 ? atoi(argv[2]) : 70;          // the victim's LDS allocation (70 KiB: two workgroups per CU, as the LS kernel at Nt = 64)
    const int mode = argc > 3 ? atoi(argv[3]) : 0;              // 0: MFMA aggressor (128 KiB of LDS: never on a victim's CU) behind the victim on a second stream; 1: no aggressor; 2: aggressor first; 3: memory aggressor; 4: MFMA aggressor without LDS (shares CUs with the victim)
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
    unsigned *counts, *samples; float* sink;
    CHECK(hipMalloc(&counts, (NFORM * 4 + 4) * 4)); CHECK(hipMalloc(&samples, NFORM * 64 * 4)); CHECK(hipMalloc(&sink, 64));
    float *scratch, *big; CHECK(hipMalloc(&scratch, (size_t)2000 * 256 * 16)); CHECK(hipMalloc(&big, (size_t)512 << 20)); CHECK(hipMemset(big, 0, (size_t)512 << 20));
    CHECK(hipFuncSetAttribute((const void*)mem_aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipMemset(counts, 0, (NFORM * 4 + 4) * 4)); CHECK(hipMemset(samples, 0, NFORM * 64 * 4));
    CHECK(hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024));
    CHECK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t fork; CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    const int viters = argc > 4 ? atoi(argv[4]) : 300, aiters = 20000;
    for (int r = 0; r < rounds; ++r) {
        if (mode == 2) hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), 128 * 1024, sb, sink, aiters, 1);
        if (mode == 0 || mode >= 3) { CHECK(hipEventRecord(fork, sa)); CHECK(hipStreamWaitEvent(sb, fork, 0)); }
        // 2000 workgroups over 512 slots: four rounds, as the 1000-packet LS launch
        hipLaunchKernelGGL(victim, dim3(2000), dim3(256), lds_kib * 1024, sa, counts, samples, viters, 12345u + r, lds_kib * 256, scratch, big);
        CHECK(hipGetLastError());
        if (mode == 0) hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), 128 * 1024, sb, sink, aiters, 1);
        // mode 4: the same MFMA loop WITHOUT an LDS footprint: its waves share CUs (and SIMDs) with the victim's
        if (mode == 4) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, sb, sink, aiters / 8, 0);
        if (mode == 5) hipLaunchKernelGGL(aggressor3<true>, dim3(2048), dim3(256), 0, sb, sink, aiters / 16);
        if (mode == 6) hipLaunchKernelGGL(aggressor3<false>, dim3(2048), dim3(256), 0, sb, sink, aiters / 16);
        if (mode >= 10) hipLaunchKernelGGL(aggressor2, dim3(2048), dim3(256), 0, sb, sink, (const float4*)big, aiters / 16, mode - 10);
        if (mode == 3) hipLaunchKernelGGL(mem_aggressor, dim3(256), dim3(512), 128 * 1024, sb, (const float4*)big, (float4*)(big + (size_t)(256 << 20) / 4), (size_t)(256 << 20) / 16, 4);
        CHECK(hipGetLastError());
        CHECK(hipStreamSynchronize(sa)); CHECK(hipStreamSynchronize(sb));
    }
    std::vector<unsigned> h(NFORM * 4 + 4);
    CHECK(hipMemcpy(h.data(), counts, (NFORM * 4 + 4) * 4, hipMemcpyDeviceToHost));
    printf("victim workgroups completed: %u of %d\n", h[NFORM * 4], rounds * 2000);
    const char* names[NFORM] = {"v_pk_add_f32 op_sel swap neg_hi", "v_pk_add_f32 op_sel swap neg_lo", "v_pk_add_f32 op_sel swap", "v_pk_mul_f32 op_sel_hi:[1,0]",
                                "v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_add_f32 plain", "v_pk_add_f32 neg only",
                                "chain: 2 packed differences -> op_sel rotation, back to back", "chain, 1 idle slot", "chain, 2 idle slots", "chain, 4 idle slots",
                                "8 LDS reads in flight, rotations at once", "... after 16 cycles", "... 32", "... 48", "... 64", "... 80", "... 96", "... 128",
                                "16 stores in flight, 32 rotations", "32 stores in flight, 32 rotations", "32 stores, 32 cycles, rotations", "32 stores, 96 cycles, rotations",
                                "LDS-DMA interleaved with rotations", "4 LDS-DMA, then 64 rotations",
                                "v_pk_add_f32 src0 swapped (op_sel:[1,0] op_sel_hi:[0,1])", "v_pk_add_f32 src1 hi for both halves", "v_pk_add_f32 src1 lo for both halves",
                                "v_pk_mul_f32 src1 swapped", "v_pk_fma_f32 src1 swapped", "v_pk_fma_f32 src2 swapped", "v_pk_add_f32 src0 swapped + constant", "v_pk_mul_f32 src0 swapped",
                                "v_pk_add_f16 src1 halves swapped", "v_pk_mul_f16 src1 halves swapped", "v_pk_fma_f16 src1 halves swapped", "v_pk_add_u16 src1 halves swapped",
                                "v_pk_max_i16 src1 halves swapped", "v_fma_mix_f32 src1 = high f16", "v_fma_mixlo_f16 src1 = high f16", "v_pk_mov_b32 op_sel:[0,1] (64-bit copy)",
                                "v_pk_add_f32 src0 AND src1 swapped", "v_pk_add_f32 d, a, a with the second swapped"};
    printf("rounds %d, victim LDS %d KiB, mode %d: wrong packed results per 16-lane quarter (lanes 0-15, 16-31, 32-47, 48-63) of %.3g per form\n", rounds, lds_kib, mode,
           (double)rounds * 2000 * 256 * viters);
    for (int f = 0; f < NFORM; ++f) printf("  %-58s %10u %10u %10u %10u\n", names[f], h[f * 4], h[f * 4 + 1], h[f * 4 + 2], h[f * 4 + 3]);
    return 0;
}
