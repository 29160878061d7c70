// hostpool_check.cpp - CPU check of the host pipeline's thread pools (csrc/csi_hostpipe.hpp): HpPool::parallel_range and
// hp_parallel_range2 (one range on two pools at once) cover [0, n) exactly once for awkward sizes, and hp_split_c128 / hp_weave_c64
// driven through them give the bits of the scalar loops.  No HIP call is made: it runs in the build container (tests/test_host_round4.py
// compiles it with hipcc and runs it).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../dl-channel-estimation-mamimo_amd/csrc/csi_hostpipe.hpp"

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
    std::printf(bad ? "FAILED (%d)\n" : "hostpool_check: ok\n", bad);
    return bad ? 1 : 0;
}
