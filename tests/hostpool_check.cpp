// hostpool_check.cpp - CPU check of the host pipeline's thread pools (csrc/csi_hostpipe.hpp): HpPool::parallel_range and
// hp_parallel_range2 (one range on two pools at once) cover [0, n) exactly once for awkward sizes, and hp_split_c128 / hp_weave_c64
// driven through them give the bits of the scalar loops.  No HIP call is made: it runs in the build container (tests/test_host_round4.py
// compiles it with hipcc and runs it).  Clean under `-Xarch_host -fsanitize=thread` and `-fsanitize=address` as well (round 4, by hand).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dl-channel-estimation-mamimo_amd/csrc/csi_hostpipe.hpp"

static size_t up64(size_t b) { return (b + 63) / 64 * 64; }      // aligned_alloc: the size is a multiple of the alignment

int main() {
    int bad = 0;
    for (int ta : {0, 1, 3, 10}) {
        for (int tb : {0, 2, 4}) {
            HpPool a, b;
            a.start(ta);
            b.start(tb);
            for (size_t n : {(size_t)0, (size_t)1, (size_t)15, (size_t)65536, (size_t)131071, (size_t)131072, (size_t)1000003, (size_t)5242880}) {
                for (size_t align : {(size_t)1, (size_t)16, (size_t)64}) {
                    std::vector<unsigned char> hit(n, 0);
                    hp_parallel_range2(a, b, n, (size_t)1 << 16, align, [&](size_t lo, size_t hi) {
                        for (size_t i = lo; i < hi; ++i) ++hit[i];
                    });
                    for (size_t i = 0; i < n; ++i)
                        if (hit[i] != 1) { ++bad; std::printf("range2: ta %d tb %d n %zu align %zu: element %zu visited %d times\n", ta, tb, n, align, i, hit[i]); break; }
                }
            }
            // the staging loops through both pools against the scalar loops
            const size_t n = 777777;
            std::vector<double> src(2 * n);
            for (size_t i = 0; i < 2 * n; ++i) src[i] = (double)((i * 2654435761u) % 1000003) / 977.0 - 500.0;
            std::vector<float> re(n + 8, -1.f), im(n + 8, -1.f), c64(2 * n + 8, -1.f);
            hp_parallel_range2(a, b, n, (size_t)1 << 16, 16, [&](size_t lo, size_t hi) { hp_split_c128(src.data(), re.data() + 3, im.data() + 5, lo, hi); });
            for (size_t i = 0; i < n; ++i)
                if (re[3 + i] != (float)src[2 * i] || im[5 + i] != (float)src[2 * i + 1]) { ++bad; std::printf("split: ta %d tb %d element %zu\n", ta, tb, i); break; }
            a.parallel_range(n, (size_t)1 << 16, [&](size_t lo, size_t hi) { hp_weave_c64(re.data() + 3, im.data() + 5, c64.data() + 2, lo, hi); });
            for (size_t i = 0; i < n; ++i)
                if (c64[2 + 2 * i] != re[3 + i] || c64[3 + 2 * i] != im[5 + i]) { ++bad; std::printf("weave: ta %d tb %d element %zu\n", ta, tb, i); break; }
            if (re[2] != -1.f || re[3 + n] != -1.f || im[4] != -1.f || im[5 + n] != -1.f || c64[1] != -1.f || c64[2 + 2 * n] != -1.f) { ++bad; std::printf("out of range write: ta %d tb %d\n", ta, tb); }
        }
    }
    // every alignment case of the staging loops (64-byte lines -> AVX-512 where the CPU has it, 32-byte -> AVX2, neither -> scalar),
    // ranges that start and end anywhere, against the scalar statement of the same loop; CSI_HOST_SIMD=0 / 2 / 5 caps the choice
    {
        const size_t n = 100003, pad = 64;
        float* R = static_cast<float*>(std::aligned_alloc(64, up64((n + 2 * pad) * 4)));
        float* M = static_cast<float*>(std::aligned_alloc(64, up64((n + 2 * pad) * 4)));
        float* C = static_cast<float*>(std::aligned_alloc(64, up64((2 * n + 2 * pad) * 4)));
        char* B = static_cast<char*>(std::aligned_alloc(64, up64(8 * n + 2 * pad)));
        std::vector<double> src(2 * n);
        for (size_t i = 0; i < 2 * n; ++i) src[i] = (double)((i * 2246822519u) % 999983) / 1013.0 - 490.0;
        const int offs[][2] = {{0, 0}, {3, 3}, {3, 5}, {8, 0}, {16, 0}, {1, 17}, {7, 8}};
        for (const auto& o : offs)
            for (size_t b : {(size_t)0, (size_t)1, (size_t)13, (size_t)64})
                for (size_t e : {n, n - 1, n - 17, b + 5, b + 40}) {
                    if (e > n || e < b) continue;
                    for (size_t i = 0; i < n + 2 * pad; ++i) R[i] = M[i] = -1.f;
                    float *re = R + pad + o[0] - 0, *im = M + pad + o[1];
                    hp_split_c128(src.data(), re, im, b, e);
                    for (size_t i = 0; i < n; ++i) {
                        const bool in = i >= b && i < e;
                        if (re[i] != (in ? (float)src[2 * i] : -1.f) || im[i] != (in ? (float)src[2 * i + 1] : -1.f)) { ++bad; std::printf("split align (%d, %d) [%zu, %zu): element %zu\n", o[0], o[1], b, e, i); break; }
                    }
                    for (int doff : {0, 2, 1, 8, 16}) {
                        for (size_t i = 0; i < 2 * n + 2 * pad; ++i) C[i] = -1.f;
                        float* dst = C + pad + doff;
                        for (size_t i = 0; i < n; ++i) { re[i] = (float)i * 0.5f; im[i] = -(float)i; }
                        hp_weave_c64(re, im, dst, b, e);
                        for (size_t i = 0; i < n; ++i) {
                            const bool in = i >= b && i < e;
                            if (dst[2 * i] != (in ? re[i] : -1.f) || dst[2 * i + 1] != (in ? im[i] : -1.f)) { ++bad; std::printf("weave align %d [%zu, %zu): element %zu\n", doff, b, e, i); break; }
                        }
                        if (dst[-1] != -1.f || dst[2 * n] != -1.f) { ++bad; std::printf("weave align %d: out of range write\n", doff); }
                    }
                }
        for (size_t doff : {(size_t)0, (size_t)1, (size_t)31, (size_t)32, (size_t)63})
            for (size_t soff : {(size_t)0, (size_t)5})
                for (size_t bytes : {(size_t)0, (size_t)100, (size_t)4095, (size_t)4096 + 70, (size_t)400000 + 13}) {
                    std::memset(B, 0x5a, 8 * n + 2 * pad);
                    const char* sp = reinterpret_cast<const char*>(src.data()) + soff;
                    hp_stream_copy(B + pad + doff, sp, bytes);
                    if (std::memcmp(B + pad + doff, sp, bytes) || B[pad + doff - 1] != 0x5a || B[pad + doff + bytes] != 0x5a) { ++bad; std::printf("copy doff %zu soff %zu bytes %zu\n", doff, soff, bytes); }
                }
        std::free(R); std::free(M); std::free(C); std::free(B);
    }
#ifdef CSI_HOST_AVX2      // (hipcc parses this file once more for the device, where the host SIMD helpers do not exist)
    std::printf("simd: cap %d, avx2 %d, avx512 %d\n", hp_simd_cap(), (int)hp_have_avx2(), (int)hp_have_avx512());
#endif
    std::printf(bad ? "FAILED (%d)\n" : "hostpool_check: ok\n", bad);
    return bad ? 1 : 0;
}
