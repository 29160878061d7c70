// csi_hostpipe.hpp - the host-buffer entry points (csi_predict / csi_ls_estimate) as a pipeline.
//
// The reference hands numpy arrays to Model.predict (massiveMIMO_CSI_prediction_DNN.py:346); the
// drop-in calls therefore take plain (pageable) host pointers.  A synchronous
// copy -> kernels -> copy loop moves them at ~5 GB/s (the runtime's internal single-threaded
// staging) and leaves the GPU idle 95 % of the time (1.1 M pairs/s against 19 M device-resident).
// Here packets flow through two slots:
//
//   host threads   user -> pinned[slot]                                   pinned[slot] -> user
//   copy-in stream            pinned -> device[slot]
//   compute stream                        LS / DNN kernels on device[slot]
//   copy-out stream                                        device -> pinned[slot]
//
// so chunk i+1 is staged and uploaded while chunk i computes and chunk i-1 drains.  The
// user <-> pinned copies are split over a small pool of host threads (one memcpy stream per
// thread reaches the DRAM bandwidth a single thread cannot).  Buffers the caller has already
// pinned (hipHostMalloc / hipHostRegister) are detected and DMA'd directly.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>

#include "csi_context.hpp"
#include "weave.hip.h"

// AVX2 / AVX-512 only inside the three staging loops below, each compiled for that ISA by a target attribute and entered behind a
// run-time CPU check - the rest of the host code is built for the baseline x86-64 ISA (a host or VM without AVX2 would
// otherwise die with SIGILL anywhere in the library, not only here: round-2 advice).  The AVX-512 forms exist for their 64-byte
// streaming stores: one instruction writes a whole cache line, the write-combining buffer never holds half of one (complex128 split,
// one thread of the build container: 17.4 against 15.2 GB/s; the hosts of the MI355X boxes are Zen 5 parts with full-width AVX-512).
// "CSI_HOST_SIMD" in the environment caps the choice (0 scalar, 2 AVX2, 5 AVX-512 = default: the best the CPU has) - A/B and tests.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#define CSI_HOST_AVX2 1
#define CSI_AVX2_FN __attribute__((target("avx2")))
#define CSI_AVX512_FN __attribute__((target("avx512f")))
#endif

namespace {

#ifdef CSI_HOST_AVX2
inline int hp_simd_cap() {
    static const int cap = [] {
        const char* e = std::getenv("CSI_HOST_SIMD");
        return e && *e ? std::atoi(e) : 5;
    }();
    return cap;
}
inline bool hp_have_avx2() {
    static const bool have = __builtin_cpu_supports("avx2") && hp_simd_cap() >= 2;
    return have;
}
inline bool hp_have_avx512() {
    static const bool have = __builtin_cpu_supports("avx512f") && hp_simd_cap() >= 5;
    return have;
}
// AVX-512 forms: 64-byte aligned destinations (the callers' head loops see to it), each returns the first element it did not handle
CSI_AVX512_FN inline size_t hp_split_c128_avx512(const double* __restrict__ src, float* __restrict__ re, float* __restrict__ im, size_t j, size_t e) {
    const __m512i i_re = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i i_im = _mm512_setr_epi32(1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31);
    for (; j + 16 <= e; j += 16) {
        const double* s = src + 2 * j;
        const __m256 c0 = _mm512_cvtpd_ps(_mm512_loadu_pd(s)), c1 = _mm512_cvtpd_ps(_mm512_loadu_pd(s + 8));
        const __m256 c2 = _mm512_cvtpd_ps(_mm512_loadu_pd(s + 16)), c3 = _mm512_cvtpd_ps(_mm512_loadu_pd(s + 24));
        const __m512 v01 = _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(c0)), _mm256_castps_pd(c1), 1));   // values 0-7: re, im alternating
        const __m512 v23 = _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(c2)), _mm256_castps_pd(c3), 1));   // values 8-15
        _mm512_stream_ps(re + j, _mm512_permutex2var_ps(v01, i_re, v23));
        _mm512_stream_ps(im + j, _mm512_permutex2var_ps(v01, i_im, v23));
    }
    _mm_sfence();
    return j;
}
CSI_AVX512_FN inline size_t hp_weave_c64_avx512(const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ dst, size_t j, size_t e) {
    const __m512i i_lo = _mm512_setr_epi32(0, 16, 1, 17, 2, 18, 3, 19, 4, 20, 5, 21, 6, 22, 7, 23);
    const __m512i i_hi = _mm512_setr_epi32(8, 24, 9, 25, 10, 26, 11, 27, 12, 28, 13, 29, 14, 30, 15, 31);
    for (; j + 16 <= e; j += 16) {
        const __m512 r = _mm512_loadu_ps(re + j), m = _mm512_loadu_ps(im + j);
        _mm512_stream_ps(dst + 2 * j, _mm512_permutex2var_ps(r, i_lo, m));
        _mm512_stream_ps(dst + 2 * j + 16, _mm512_permutex2var_ps(r, i_hi, m));
    }
    _mm_sfence();
    return j;
}
CSI_AVX512_FN inline size_t hp_stream_copy_avx512(char* __restrict__ d, const char* __restrict__ s, size_t bytes) {
    size_t i = 0;
    for (; i + 256 <= bytes; i += 256) {
        const __m512 a = _mm512_loadu_ps(s + i), b = _mm512_loadu_ps(s + i + 64), c = _mm512_loadu_ps(s + i + 128), e = _mm512_loadu_ps(s + i + 192);
        _mm512_stream_ps(reinterpret_cast<float*>(d + i), a);
        _mm512_stream_ps(reinterpret_cast<float*>(d + i + 64), b);
        _mm512_stream_ps(reinterpret_cast<float*>(d + i + 128), c);
        _mm512_stream_ps(reinterpret_cast<float*>(d + i + 192), e);
    }
    _mm_sfence();
    return i;
}
// each returns the first element it did not handle
CSI_AVX2_FN inline size_t hp_split_c128_avx2(const double* __restrict__ src, float* __restrict__ re, float* __restrict__ im, size_t j, size_t e) {
    for (; j + 8 <= e; j += 8) {
        const double* s = src + 2 * j;
        const __m128 c0 = _mm256_cvtpd_ps(_mm256_loadu_pd(s)), c1 = _mm256_cvtpd_ps(_mm256_loadu_pd(s + 4));
        const __m128 c2 = _mm256_cvtpd_ps(_mm256_loadu_pd(s + 8)), c3 = _mm256_cvtpd_ps(_mm256_loadu_pd(s + 12));
        const __m256 v01 = _mm256_set_m128(c1, c0), v23 = _mm256_set_m128(c3, c2);       // [re0 im0 re1 im1 | re2 im2 re3 im3], [4 5 | 6 7]
        const __m256 r = _mm256_shuffle_ps(v01, v23, 0x88), m = _mm256_shuffle_ps(v01, v23, 0xDD);   // [0 1 4 5 | 2 3 6 7]
        _mm256_stream_ps(re + j, _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(r), 0xD8)));
        _mm256_stream_ps(im + j, _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(m), 0xD8)));
    }
    _mm_sfence();
    return j;
}
CSI_AVX2_FN inline size_t hp_weave_c64_avx2(const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ dst, size_t j, size_t e) {
    for (; j + 8 <= e; j += 8) {
        const __m256 r = _mm256_loadu_ps(re + j), m = _mm256_loadu_ps(im + j);
        const __m256 lo = _mm256_unpacklo_ps(r, m), hi = _mm256_unpackhi_ps(r, m);      // [r0 m0 r1 m1 | r4 m4 r5 m5], [r2 m2 r3 m3 | r6 m6 r7 m7]
        _mm256_stream_ps(dst + 2 * j, _mm256_permute2f128_ps(lo, hi, 0x20));
        _mm256_stream_ps(dst + 2 * j + 8, _mm256_permute2f128_ps(lo, hi, 0x31));
    }
    _mm_sfence();
    return j;
}
CSI_AVX2_FN inline size_t hp_stream_copy_avx2(char* __restrict__ d, const char* __restrict__ s, size_t bytes) {
    size_t i = 0;
    for (; i + 128 <= bytes; i += 128) {
        const __m256 a = _mm256_loadu_ps(reinterpret_cast<const float*>(s + i)), b = _mm256_loadu_ps(reinterpret_cast<const float*>(s + i + 32));
        const __m256 c = _mm256_loadu_ps(reinterpret_cast<const float*>(s + i + 64)), e = _mm256_loadu_ps(reinterpret_cast<const float*>(s + i + 96));
        _mm256_stream_ps(reinterpret_cast<float*>(d + i), a);
        _mm256_stream_ps(reinterpret_cast<float*>(d + i + 32), b);
        _mm256_stream_ps(reinterpret_cast<float*>(d + i + 64), c);
        _mm256_stream_ps(reinterpret_cast<float*>(d + i + 96), e);
    }
    _mm_sfence();
    return i;
}
#endif

// complex128 (re, im doubles interleaved) -> two float32 planes, elements [b, e).  The planes live in pinned staging
// memory that this core never reads again (the DMA engine does): streaming stores spare the read-for-ownership of
// every destination line - a third of the DRAM traffic of this loop.
inline void hp_split_c128(const double* __restrict__ src, float* __restrict__ re, float* __restrict__ im, size_t b, size_t e) {
    size_t j = b;
#ifdef CSI_HOST_AVX2
    if (hp_have_avx2()) {
        // both planes' addresses share their offset in a 64-byte line (pinned planes: always): walk to the line, then whole lines
        const bool lines = hp_have_avx512() && !((reinterpret_cast<uintptr_t>(re) ^ reinterpret_cast<uintptr_t>(im)) & 63);
        const uintptr_t mask = lines ? 63 : 31;
        while (j < e && ((reinterpret_cast<uintptr_t>(re + j) | reinterpret_cast<uintptr_t>(im + j)) & mask)) {
            re[j] = (float)src[2 * j];
            im[j] = (float)src[2 * j + 1];
            ++j;
        }
        if (lines && j < e) j = hp_split_c128_avx512(src, re, im, j, e);
        if (!((reinterpret_cast<uintptr_t>(re + j) | reinterpret_cast<uintptr_t>(im + j)) & 31)) j = hp_split_c128_avx2(src, re, im, j, e);
    }
#endif
    for (; j < e; ++j) {
        re[j] = (float)src[2 * j];
        im[j] = (float)src[2 * j + 1];
    }
}

// two float32 planes -> complex64 (re, im floats interleaved), elements [b, e); the caller's result array is written once
inline void hp_weave_c64(const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ dst, size_t b, size_t e) {
    size_t j = b;
#ifdef CSI_HOST_AVX2
    if (hp_have_avx2()) {
        const bool lines = hp_have_avx512() && !(reinterpret_cast<uintptr_t>(dst) & 7);      // complex64 values on 8-byte addresses reach a line boundary
        const uintptr_t mask = lines ? 63 : 31;
        while (j < e && (reinterpret_cast<uintptr_t>(dst + 2 * j) & mask)) {
            dst[2 * j] = re[j];
            dst[2 * j + 1] = im[j];
            ++j;
        }
        if (lines && j < e) j = hp_weave_c64_avx512(re, im, dst, j, e);
        if (!(reinterpret_cast<uintptr_t>(dst + 2 * j) & 31)) j = hp_weave_c64_avx2(re, im, dst, j, e);
    }
#endif
    for (; j < e; ++j) {
        dst[2 * j] = re[j];
        dst[2 * j + 1] = im[j];
    }
}

// plain copy whose destination is not read again by this core (user -> pinned staging, pinned staging -> the caller's
// result array): streaming stores; glibc's memcpy switches to them only beyond a threshold tied to the (huge) L3 here
inline void hp_stream_copy(void* __restrict__ dst, const void* __restrict__ src, size_t bytes) {
#ifdef CSI_HOST_AVX2
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    const size_t head = (64 - (reinterpret_cast<uintptr_t>(d) & 63)) & 63;
    if (bytes >= 4096 + head && hp_have_avx2()) {
        std::memcpy(d, s, head);
        d += head; s += head; bytes -= head;
        size_t i = hp_have_avx512() ? hp_stream_copy_avx512(d, s, bytes) : 0;
        i += hp_stream_copy_avx2(d + i, s + i, bytes - i);
        std::memcpy(d + i, s + i, bytes - i);
        return;
    }
#endif
    std::memcpy(dst, src, bytes);
}

}  // namespace

// a small pool of host threads for the DRAM-bound staging loops (parallel memcpy / split / weave)
struct HpPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    int job_n = 0, job_next = 0, job_left = 0;
    uint64_t job_gen = 0;
    bool stop = false;

    void worker_loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || (job_gen != seen && job_next < job_n); });
            if (stop) return;
            seen = job_gen;
            while (job_next < job_n) {
                const int i = job_next++;
                lk.unlock();
                job(i);
                lk.lock();
                if (--job_left == 0) cv_done.notify_all();
            }
        }
    }
    void start(int n_threads) {
        for (int i = 0; i < n_threads; ++i) workers.emplace_back([this] { worker_loop(); });
    }
    // f(0..n-1) on the pool and the calling thread; returns when all are done
    void parallel(int n, const std::function<void(int)>& f) {
        if (n <= 1 || workers.empty()) {
            for (int i = 0; i < n; ++i) f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            job_n = n;
            job_next = 0;
            job_left = n;
            ++job_gen;
        }
        cv_job.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        while (job_next < job_n) {
            const int i = job_next++;
            lk.unlock();
            f(i);
            lk.lock();
            if (--job_left == 0) cv_done.notify_all();
        }
        cv_done.wait(lk, [&] { return job_left == 0; });
    }
    void copy(void* dst, const void* src, size_t bytes) {
        const int parts = (int)std::min<size_t>(workers.size() + 1, std::max<size_t>(1, bytes >> 20));     // >= 1 MiB per part
        const size_t per = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
        parallel(parts, [&](int i) {
            const size_t o = (size_t)i * per;
            if (o < bytes) hp_stream_copy((char*)dst + o, (const char*)src + o, std::min(per, bytes - o));
        });
    }
    // f(begin, end) over [0, n) in contiguous parts of >= min_part elements on the pool
    void parallel_range(size_t n, size_t min_part, const std::function<void(size_t, size_t)>& f) {
        const int parts = (int)std::min<size_t>(workers.size() + 1, std::max<size_t>(1, n / std::max<size_t>(min_part, 1)));
        const size_t per = (n + parts - 1) / parts;
        parallel(parts, [&](int i) {
            const size_t b = (size_t)i * per, e = std::min(n, b + per);
            if (b < e) f(b, e);
        });
    }
    ~HpPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_job.notify_all();
        for (auto& t : workers) t.join();
    }
};

// f(begin, end) over [0, n) on TWO pools at once, the range divided in proportion to their thread counts: `a` is driven by the calling
// thread, `b` by a helper thread that lives for the call (pools block their caller).  For the input staging of calls whose result side
// needs no host pass (pinned result arrays): the output pool's threads would idle while the input side is what the call waits for.
inline void hp_parallel_range2(HpPool& a, HpPool& b, size_t n, size_t min_part, size_t align, const std::function<void(size_t, size_t)>& f) {
    const size_t ta = a.workers.size() + 1, tb = b.workers.size() + 1;
    size_t na = n * ta / (ta + tb);
    na -= na % std::max<size_t>(align, 1);
    if (n < 2 * min_part || na == 0 || na == n) {
        a.parallel_range(n, min_part, f);
        return;
    }
    std::thread helper;
    try {
        helper = std::thread([&] { b.parallel_range(n - na, min_part, [&](size_t lo, size_t hi) { f(na + lo, na + hi); }); });
    } catch (const std::system_error&) {          // no thread to be had (pids limit): the first pool does all of it
        a.parallel_range(n, min_part, f);
        return;
    }
    a.parallel_range(na, min_part, f);
    helper.join();
}

struct csi_hostpipe {
    // Two pools: the input side (user -> pinned staging) runs AHEAD on its own thread, beside the calling thread's output side
    // (pinned staging -> user), so that neither waits for the other's DRAM-bound loop (round 4; they used to alternate on one pool)
    HpPool pool_in, pool_out;
    HpPool& pool() { return pool_out; }      // the single-pool users (plane entry points' copies)

    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    char* pin_in[2] = {nullptr, nullptr};
    char* pin_out[2] = {nullptr, nullptr};
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
    char* dev[2] = {nullptr, nullptr};
    size_t dev_bytes = 0;

    // where the time of the last pipelined call went (microseconds; "hp_*_us" options, tools/hostpath_probe.py)
    std::atomic<int64_t> us_stage{0};        // stager thread busy converting / copying user -> pinned
    int64_t us_wait_stage = 0;               // calling thread waiting for a staged chunk
    int64_t us_wait_out = 0;                 // ... for a download to finish
    std::atomic<int64_t> us_weave_thread{0}; // drainer thread busy converting / copying pinned -> user
    int64_t us_weave = 0;                    // (copy of it when the call ends)
    int64_t us_total = 0;
    void clock_reset() { us_stage = 0; us_weave_thread = 0; us_wait_stage = us_wait_out = us_weave = us_total = 0; }
    static int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    void copy(void* dst, const void* src, size_t bytes) { pool_out.copy(dst, src, bytes); }
    void copy_in(void* dst, const void* src, size_t bytes) { pool_in.copy(dst, src, bytes); }
    ~csi_hostpipe() {
        for (int s = 0; s < 2; ++s) {
            if (pin_in[s]) hipHostFree(pin_in[s]);
            if (pin_out[s]) hipHostFree(pin_out[s]);
            if (dev[s]) hipFree(dev[s]);
            if (ev_in[s]) hipEventDestroy(ev_in[s]);
            if (ev_comp[s]) hipEventDestroy(ev_comp[s]);
            if (ev_out[s]) hipEventDestroy(ev_out[s]);
        }
        if (s_in) hipStreamDestroy(s_in);
        if (s_out) hipStreamDestroy(s_out);
    }
};

namespace {

bool hp_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();          // a plain malloc pointer is "invalid value" for this query
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int hp_get(csi_ctx* c, csi_hostpipe** out) {
    if (!c->hostpipe) {
        auto* h = new csi_hostpipe();
        c->hostpipe = h;
        HIP_TRY(c, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
        for (int s = 0; s < 2; ++s) {
            HIP_TRY(c, hipEventCreateWithFlags(&h->ev_in[s], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&h->ev_comp[s], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&h->ev_out[s], hipEventDisableTiming));
        }
        int nt = c->host_threads;
        // copies and complex128 / complex64 conversions are DRAM-bound streams: one thread moves ~5-10 GB/s, the PCIe link
        // wants ~100 GB/s of staging in both directions together
        if (nt <= 0) {
            nt = (int)std::min<unsigned>(32, std::max<unsigned>(2, std::thread::hardware_concurrency() / 6));
            // a container's CPU quota (cgroup v2 cpu.max "<quota> <period>"): more runnable threads than that are throttled, not faster
            // (measured on the pool's boxes: 256 hardware threads visible, quota 16 CPUs - profiles/r04_hostpath_probe.txt)
            if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
                long long quota = 0, period = 0;
                if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                    nt = std::max(2, std::min(nt, (int)((quota + period - 1) / period)));
                std::fclose(f);
            }
        }
        // the input side moves twice the bytes of the output side (complex128 read + two float planes written, against two planes
        // read + complex64 written): two thirds of the threads
        const int n_in = std::max(1, (2 * nt + 1) / 3), n_out = std::max(1, nt - n_in);
        if (c->hp_side_threads == 0) { h->pool_in.start(nt - 1); h->pool_out.start(nt - 1); }      // used in turn: each the full count
        else { h->pool_in.start(n_in - 1); h->pool_out.start(n_out - 1); }                         // + the stager / drainer thread itself
    }
    *out = c->hostpipe;
    return CSI_OK;
}

// dev_extra: bytes of device staging beyond the in + out regions (csi_estimate_c64: the interleaved chunk as it was uploaded)
int hp_reserve(csi_ctx* c, csi_hostpipe* h, size_t in_bytes, size_t out_bytes, bool need_pin_in, bool need_pin_out, size_t dev_extra = 0) {
    if ((!need_pin_in || h->pin_in_bytes >= in_bytes) && (!need_pin_out || h->pin_out_bytes >= out_bytes) &&
        h->dev_bytes >= in_bytes + out_bytes + dev_extra)
        return CSI_OK;                                    // every call ends drained: nothing in flight uses the slots
    HIP_TRY(c, hipStreamSynchronize(h->s_in));
    HIP_TRY(c, hipStreamSynchronize(h->s_out));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < 2; ++s) {
        if (need_pin_in && h->pin_in_bytes < in_bytes) {
            if (h->pin_in[s]) hipHostFree(h->pin_in[s]);
            h->pin_in[s] = nullptr;
            HIP_TRY(c, hipHostMalloc((void**)&h->pin_in[s], in_bytes, hipHostMallocDefault));
        }
        if (need_pin_out && h->pin_out_bytes < out_bytes) {
            if (h->pin_out[s]) hipHostFree(h->pin_out[s]);
            h->pin_out[s] = nullptr;
            HIP_TRY(c, hipHostMalloc((void**)&h->pin_out[s], out_bytes, hipHostMallocDefault));
        }
        if (h->dev_bytes < in_bytes + out_bytes + dev_extra) {
            if (h->dev[s]) hipFree(h->dev[s]);
            h->dev[s] = nullptr;
            if (hipMalloc((void**)&h->dev[s], in_bytes + out_bytes + dev_extra + 1024) != hipSuccess)
                return fail(c, CSI_ERR_NOMEM, "host pipeline: device staging allocation failed");
        }
    }
    if (need_pin_in) h->pin_in_bytes = std::max(h->pin_in_bytes, in_bytes);
    if (need_pin_out) h->pin_out_bytes = std::max(h->pin_out_bytes, out_bytes);
    h->dev_bytes = std::max(h->dev_bytes, in_bytes + out_bytes + dev_extra);
    return CSI_OK;
}

// Packet ranges of the pipelined calls.  They are bound by the slower copy direction (the download: 1.92 GB for DNN + LS at config 2;
// rocprofv3 --memory-copy-trace shows the D2H engine busy 88 % of the span at ~50 GB/s, profiles/r03_c128_copy_trace.txt), so what is left
// to win is the head and the tail of that stream: a SHORT first chunk lets the first download start after ~1.5 ms instead of ~4.5, a short
// last one leaves little to copy out behind the last download.  `chunk` stays the slot size; every host-buffer entry point uses the same
// schedule (the plane calls and csi_estimate_c128 return identical bits for identical packets).
void hp_schedule(int64_t npkt, int64_t chunk, std::vector<int64_t>& first_of, std::vector<int64_t>& size_of) {
    const int64_t small = std::max<int64_t>(1, chunk / 4);
    int64_t p = 0;
    auto push = [&](int64_t n) { first_of.push_back(p); size_of.push_back(n); p += n; };
    if (npkt > chunk) push(small);
    while (npkt - p > chunk + small) push(chunk);
    if (npkt - p > chunk) { push(npkt - p - small); push(small); }
    else if (npkt - p > 2 * small && npkt > chunk) { push(npkt - p - small); push(small); }
    else push(npkt - p);
}

// A pipelined call is run by three threads (round 4; one thread used to do all of it in turn, and the H2D stream stood still
// whenever that thread was converting results):
//   stager   stage(i, slot): user memory -> pinned_in[slot]      at most two chunks ahead of the uploads (the slot is free again
//                                                                once the H2D of chunk i - 2 has completed)
//   caller   enqueues H2D(i), the kernels of chunk i, D2H(i) and records the events; blocks only on real dependencies
//   drainer  weave(i, slot): pinned_out[slot] -> user memory     behind the D2H of chunk i; the caller enqueues D2H(i + 2) into the
//                                                                same slot only after that
// Each side owns a pool of helper threads for its DRAM-bound loop (csi_hostpipe::pool_in / pool_out).
struct HpSide {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int64_t done = 0;           // chunks this side has finished (staged / drained)
    int64_t released = 0;       // chunks the caller has handed over (H2D enqueued / D2H enqueued)
    bool abort = false;
    hipError_t err = hipSuccess;

    // stager: needs_release_of(i) = i - 2 (the upload that frees its slot); drainer: = i (its own download)
    // false: no thread to be had (pids limit, EAGAIN) - the caller falls back to the inline arrangement for this call (ADVICE round 4:
    // the std::system_error used to leave through the extern "C" entry point and end in std::terminate)
    bool start(int device, int64_t nchunks, int lag, hipEvent_t* ev, std::atomic<int64_t>* busy_us, const std::function<void(int64_t, int)>& work) {
      try {
        th = std::thread([this, device, nchunks, lag, ev, busy_us, work] {
            (void)hipSetDevice(device);
            for (int64_t i = 0; i < nchunks; ++i) {
                const int s = (int)(i & 1);
                if (i - lag >= 0) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return abort || released > i - lag; });
                        if (abort) return;
                    }
                    const hipError_t e = hipEventSynchronize(ev[s]);
                    if (e != hipSuccess) {
                        std::lock_guard<std::mutex> lk(mu);
                        err = e;
                        abort = true;
                        cv.notify_all();
                        return;
                    }
                }
                const int64_t t0 = csi_hostpipe::now_us();
                work(i, s);
                if (busy_us) *busy_us += csi_hostpipe::now_us() - t0;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    done = i + 1;
                }
                cv.notify_all();
            }
        });
      } catch (const std::system_error&) {
        return false;
      }
      return true;
    }
    bool wait_done(int64_t n) {               // until chunks 0 .. n - 1 are through this side
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return abort || done >= n; });
        return !abort;
    }
    void release(int64_t i) {
        {
            std::lock_guard<std::mutex> lk(mu);
            released = i + 1;
        }
        cv.notify_all();
    }
    void cancel() {
        {
            std::lock_guard<std::mutex> lk(mu);
            abort = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void join() { if (th.joinable()) th.join(); }
    ~HpSide() { cancel(); }
};

struct HpPipe {
    int64_t nchunks = 0;
    std::function<void(int64_t, int)> stage;          // empty: the caller's inputs are pinned, H2D straight from them
    std::function<int(int64_t, int)> enqueue_in;      // H2D of chunk i on h->s_in
    std::function<int(int64_t, int)> compute;         // kernels of chunk i on c->stream
    std::function<int(int64_t, int)> enqueue_out;     // D2H of chunk i on h->s_out
    std::function<void(int64_t, int)> weave;          // empty: the caller's outputs are pinned, D2H straight into them
};

int hp_run(csi_ctx* c, csi_hostpipe* h, const HpPipe& p) {
    const int dev = c->cfg.device;
    h->clock_reset();
    const int64_t t_begin = csi_hostpipe::now_us();
    HpSide stager, drainer;                            // their destructors stop and join them on every return path
    bool threads = c->hp_side_threads != 0;            // 0: stage / weave inline on the calling thread, in turn (the round-3 arrangement; A/B)
    if (threads && p.stage && !stager.start(dev, p.nchunks, 2, h->ev_in, &h->us_stage, p.stage)) threads = false;
    if (threads && p.weave && !drainer.start(dev, p.nchunks, 0, h->ev_out, &h->us_weave_thread, p.weave)) {
        stager.cancel();                               // (it has not been released a chunk yet: it stops at its first wait, or after chunks 0 / 1, which the inline path stages again)
        threads = false;
    }
    auto inline_stage = [&](int64_t i, int s) -> int {
        if (i >= 2) HIP_TRY(c, hipEventSynchronize(h->ev_in[s]));
        const int64_t t0 = csi_hostpipe::now_us();
        p.stage(i, s);
        h->us_stage += csi_hostpipe::now_us() - t0;
        return CSI_OK;
    };
    auto inline_drain = [&](int64_t i) -> int {
        const int64_t t0 = csi_hostpipe::now_us();
        HIP_TRY(c, hipEventSynchronize(h->ev_out[i & 1]));
        const int64_t t1 = csi_hostpipe::now_us();
        p.weave(i, (int)(i & 1));
        h->us_wait_out += t1 - t0;
        h->us_weave_thread += csi_hostpipe::now_us() - t1;
        return CSI_OK;
    };
    auto timed_wait = [&](HpSide& side, int64_t n, int64_t* acc) {
        const int64_t t0 = csi_hostpipe::now_us();
        const bool ok = side.wait_done(n);
        *acc += csi_hostpipe::now_us() - t0;
        return ok;
    };
    for (int64_t i = 0; i < p.nchunks; ++i) {
        const int s = (int)(i & 1);
        // device[s] inputs are free once the kernels of chunk i - 2 are done
        if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(h->s_in, h->ev_comp[s], 0));
        if (p.stage && !threads) { const int r0 = inline_stage(i, s); if (r0) return r0; }
        if (p.stage && threads && !timed_wait(stager, i + 1, &h->us_wait_stage))
            return fail(c, CSI_ERR_HIP, "host pipeline: input staging failed: %s", hipGetErrorString(stager.err));
        int rc = p.enqueue_in(i, s);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(h->ev_in[s], h->s_in));
        if (p.stage && threads) stager.release(i);
        // kernels: behind the upload, and behind the download of chunk i - 2 that releases device[s]'s output half
        HIP_TRY(c, hipStreamWaitEvent(c->stream, h->ev_in[s], 0));
        if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, h->ev_out[s], 0));
        c->in_host_pipeline = true;                    // (the chunk's kernels stay on the one stream: csi_predict_device's two-stream fork is for device-resident calls)
        rc = p.compute(i, s);
        c->in_host_pipeline = false;
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(h->ev_comp[s], c->stream));
        // pinned_out[s] must have been handed to the user (chunk i - 2) before the next download lands in it
        if (p.weave && !threads && i >= 2) { const int r0 = inline_drain(i - 2); if (r0) return r0; }
        if (p.weave && threads && i >= 2 && !timed_wait(drainer, i - 1, &h->us_wait_out))
            return fail(c, CSI_ERR_HIP, "host pipeline: result staging failed: %s", hipGetErrorString(drainer.err));
        HIP_TRY(c, hipStreamWaitEvent(h->s_out, h->ev_comp[s], 0));
        rc = p.enqueue_out(i, s);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(h->ev_out[s], h->s_out));
        if (p.weave && threads) drainer.release(i);
    }
    if (p.weave && !threads) {
        for (int64_t i = std::max<int64_t>(0, p.nchunks - 2); i < p.nchunks; ++i) { const int r0 = inline_drain(i); if (r0) return r0; }
    } else if (p.weave) {
        if (!timed_wait(drainer, p.nchunks, &h->us_wait_out))
            return fail(c, CSI_ERR_HIP, "host pipeline: result staging failed: %s", hipGetErrorString(drainer.err));
    } else {
        HIP_TRY(c, hipStreamSynchronize(h->s_out));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    h->us_weave = (int64_t)h->us_weave_thread;
    h->us_total = csi_hostpipe::now_us() - t_begin;
    return CSI_OK;
}

// run(d_re, d_im, np, d_ore, d_oim) enqueues the kernels of np packets on c->stream
int hp_packets_impl(csi_ctx* c, csi_hostpipe* h, const float* re, const float* im, int64_t npkt, float* o_re, float* o_im, int n_out,
                    const std::function<int(const float*, const float*, int64_t, float*, float*)>& run);

int hp_packets(csi_ctx* c, const float* re, const float* im, int64_t npkt, float* o_re, float* o_im, int n_out,
               const std::function<int(const float*, const float*, int64_t, float*, float*)>& run) {
    csi_hostpipe* h = nullptr;
    int rc = hp_get(c, &h);
    if (rc) return rc;
    rc = hp_packets_impl(c, h, re, im, npkt, o_re, o_im, n_out, run);
    if (rc) {
        // an error left copies / kernels in flight that still reference the caller's buffers and the
        // slots: drain everything before handing control back (the error text of the failure is kept)
        const std::string keep = c->err;
        hipStreamSynchronize(h->s_in);
        hipStreamSynchronize(c->stream);
        hipStreamSynchronize(h->s_out);
        (void)hipGetLastError();
        c->err = keep;
    }
    return rc;
}

int hp_packets_impl(csi_ctx* c, csi_hostpipe* h, const float* re, const float* im, int64_t npkt, float* o_re, float* o_im, int n_out,
                    const std::function<int(const float*, const float*, int64_t, float*, float*)>& run) {
    const csi_config& cf = c->cfg;
    int rc = CSI_OK;
    const size_t in_pkt = (size_t)cf.nr * cf.len_ltf * sizeof(float);            // one plane
    const size_t out_pkt = (size_t)cf.nr * cf.nt * n_out * sizeof(float);        // one plane
    // chunk: large enough for full-size GEMM tiles (>= 64k pair rows), small enough to pipeline
    int64_t chunk = std::max<int64_t>(1, (int64_t)65536 / std::max(1, cf.nr * cf.nt));
    chunk = std::max<int64_t>(chunk, ((int64_t)8 << 20) / (int64_t)in_pkt);       // >= 8 MiB per upload
    chunk = std::min(chunk, npkt);
    const bool in_pinned = hp_is_pinned(re) && hp_is_pinned(im);
    const bool out_pinned = hp_is_pinned(o_re) && hp_is_pinned(o_im);
    rc = hp_reserve(c, h, 2 * in_pkt * chunk, 2 * out_pkt * chunk, !in_pinned, !out_pinned);
    if (rc) return rc;

    std::vector<int64_t> first_of, size_of;
    hp_schedule(npkt, chunk, first_of, size_of);
    const int64_t nchunks = (int64_t)size_of.size();
    if (nchunks == 1 && 2 * (in_pkt + out_pkt) * (size_t)npkt <= ((size_t)8 << 20)) {
        // small call (the reference's per-packet loop): nothing to pipeline - one stream, no events, no
        // thread hand-over; the staging copies are a few hundred KB
        float* d_re = reinterpret_cast<float*>(h->dev[0]);
        float* d_im = reinterpret_cast<float*>(h->dev[0] + in_pkt * chunk);
        float* d_ore = reinterpret_cast<float*>(h->dev[0] + 2 * in_pkt * chunk);
        float* d_oim = reinterpret_cast<float*>(h->dev[0] + 2 * in_pkt * chunk + out_pkt * chunk);
        const float* s_re = re;
        const float* s_im = im;
        if (!in_pinned) {
            std::memcpy(h->pin_in[0], re, in_pkt * npkt);
            std::memcpy(h->pin_in[0] + in_pkt * chunk, im, in_pkt * npkt);
            s_re = reinterpret_cast<const float*>(h->pin_in[0]);
            s_im = reinterpret_cast<const float*>(h->pin_in[0] + in_pkt * chunk);
        }
        HIP_TRY(c, hipMemcpyAsync(d_re, s_re, in_pkt * npkt, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_im, s_im, in_pkt * npkt, hipMemcpyHostToDevice, c->stream));
        rc = run(d_re, d_im, npkt, d_ore, d_oim);
        if (rc) return rc;
        float* t_re = out_pinned ? o_re : reinterpret_cast<float*>(h->pin_out[0]);
        float* t_im = out_pinned ? o_im : reinterpret_cast<float*>(h->pin_out[0] + out_pkt * chunk);
        HIP_TRY(c, hipMemcpyAsync(t_re, d_ore, out_pkt * npkt, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(t_im, d_oim, out_pkt * npkt, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!out_pinned) {
            std::memcpy(o_re, t_re, out_pkt * npkt);
            std::memcpy(o_im, t_im, out_pkt * npkt);
        }
        return CSI_OK;
    }
    auto np_of = [&](int64_t i) { return size_of[(size_t)i]; };
    HpPipe p;
    p.nchunks = nchunks;
    if (!in_pinned)
        p.stage = [&](int64_t i, int s) {
            const size_t ioff = (size_t)first_of[(size_t)i] * cf.nr * cf.len_ltf;
            h->copy_in(h->pin_in[s], re + ioff, in_pkt * np_of(i));
            h->copy_in(h->pin_in[s] + in_pkt * chunk, im + ioff, in_pkt * np_of(i));
        };
    p.enqueue_in = [&](int64_t i, int s) -> int {
        const size_t ioff = (size_t)first_of[(size_t)i] * cf.nr * cf.len_ltf;
        const float* s_re = in_pinned ? re + ioff : reinterpret_cast<const float*>(h->pin_in[s]);
        const float* s_im = in_pinned ? im + ioff : reinterpret_cast<const float*>(h->pin_in[s] + in_pkt * chunk);
        HIP_TRY(c, hipMemcpyAsync(h->dev[s], s_re, in_pkt * np_of(i), hipMemcpyHostToDevice, h->s_in));
        HIP_TRY(c, hipMemcpyAsync(h->dev[s] + in_pkt * chunk, s_im, in_pkt * np_of(i), hipMemcpyHostToDevice, h->s_in));
        return CSI_OK;
    };
    p.compute = [&](int64_t i, int s) -> int {
        return run(reinterpret_cast<float*>(h->dev[s]), reinterpret_cast<float*>(h->dev[s] + in_pkt * chunk), np_of(i),
                   reinterpret_cast<float*>(h->dev[s] + 2 * in_pkt * chunk), reinterpret_cast<float*>(h->dev[s] + 2 * in_pkt * chunk + out_pkt * chunk));
    };
    p.enqueue_out = [&](int64_t i, int s) -> int {
        const size_t ooff = (size_t)first_of[(size_t)i] * cf.nr * cf.nt * n_out;
        float* t_re = out_pinned ? o_re + ooff : reinterpret_cast<float*>(h->pin_out[s]);
        float* t_im = out_pinned ? o_im + ooff : reinterpret_cast<float*>(h->pin_out[s] + out_pkt * chunk);
        HIP_TRY(c, hipMemcpyAsync(t_re, h->dev[s] + 2 * in_pkt * chunk, out_pkt * np_of(i), hipMemcpyDeviceToHost, h->s_out));
        HIP_TRY(c, hipMemcpyAsync(t_im, h->dev[s] + 2 * in_pkt * chunk + out_pkt * chunk, out_pkt * np_of(i), hipMemcpyDeviceToHost, h->s_out));
        return CSI_OK;
    };
    if (!out_pinned)
        p.weave = [&](int64_t i, int s) {
            const size_t off = (size_t)first_of[(size_t)i] * cf.nr * cf.nt * n_out;
            h->copy(o_re + off, h->pin_out[s], out_pkt * np_of(i));
            h->copy(o_im + off, h->pin_out[s] + out_pkt * chunk, out_pkt * np_of(i));
        };
    return hp_run(c, h, p);
}

// ---------------------------------------------------------------------------------------------------
// complex128 in, complex64 out: the arrays the reference's deployment wrapper receives and returns
// (inference.py:24-32: np.complex128 batch in, ``output_real + 1j*output_imag`` out).  The split into the two
// float32 planes the kernels read, and the interleave of the two result planes, happen in the staging copies
// of the pipeline (host threads, chunk by chunk, beside the uploads / kernels / downloads of the neighbouring
// chunks) instead of as whole-array numpy passes in front of and behind the call.  One upload serves both
// estimators: dnn_c64 and / or ls_c64 may be null.
//
// in_c64 (csi_estimate_c64): the batch arrives as complex64 - half the bytes to read on the host and nothing to convert there.  The
// interleaved chunk is uploaded AS IT IS (straight from the caller's array when that is pinned host memory, else through a plain copy
// into the pinned slot) and split into the two planes on the device (split_c64_kernel, HBM-bound, in front of the chunk's kernels).
int hp_estimate_c128_impl(csi_ctx* c, csi_hostpipe* h, const void* in_any, bool in_c64, int64_t npkt, float* dnn_c64, float* ls_c64) {
    const double* in = static_cast<const double*>(in_any);
    const float* in32 = static_cast<const float*>(in_any);
    const csi_config& cf = c->cfg;
    int rc = CSI_OK;
    const size_t in_n = (size_t)cf.nr * cf.len_ltf;                     // samples per packet
    const size_t in_pkt = in_n * sizeof(float);                          // one plane
    const size_t dnn_n = dnn_c64 ? (size_t)cf.nr * cf.nt * cf.n_out : 0, ls_n = ls_c64 ? (size_t)cf.nr * cf.nt * LS_NDATA : 0;
    const size_t out_pkt = (dnn_n + ls_n) * sizeof(float);              // one plane of everything that comes back
    int64_t chunk = std::max<int64_t>(1, (int64_t)65536 / std::max(1, cf.nr * cf.nt));
    chunk = std::max<int64_t>(chunk, ((int64_t)8 << 20) / (int64_t)in_pkt);
    if (c->hp_chunk_packets > 0) chunk = c->hp_chunk_packets;
    chunk = std::min(chunk, npkt);
    // Result arrays in pinned host memory (csi_host_malloc; engine.pinned_empty(shape, np.complex64)): the complex values are
    // assembled on the device behind the chunk's kernels (weave_c64_kernel) and downloaded straight into the caller's arrays - no
    // staging buffer, no host pass on the result side ("hp_device_weave" = 0: the host threads weave as for pageable arrays).
    const bool direct_out = c->hp_device_weave != 0 && (!dnn_c64 || hp_is_pinned(dnn_c64)) && (!ls_c64 || hp_is_pinned(ls_c64));
    // device[s]: in re | in im | result planes (dnn re | dnn im | ls re | ls im) | direct_out only: dnn complex64 | ls complex64
    //            | in_c64 only: the chunk's interleaved (re, im) samples as uploaded
    const bool in_pinned = in_c64 && hp_is_pinned(in_any);
    const size_t raw_off = 2 * in_pkt * chunk + (direct_out ? 4 : 2) * out_pkt * chunk;
    rc = hp_reserve(c, h, 2 * in_pkt * chunk, (direct_out ? 4 : 2) * out_pkt * chunk, !in_pinned, !direct_out, in_c64 ? 2 * in_pkt * chunk : 0);
    if (rc) return rc;
    if (direct_out) ++c->hp_direct_out_calls;
    std::vector<int64_t> first_of, size_of;
    hp_schedule(npkt, chunk, first_of, size_of);
    const int64_t nchunks = (int64_t)size_of.size();
    auto np_of = [&](int64_t i) { return size_of[(size_t)i]; };
    HpPipe p;
    p.nchunks = nchunks;
    if (in_c64 && !in_pinned)
        p.stage = [&](int64_t i, int s) {                                // complex64, pageable: a plain copy into pinned_in[s]
            const float* src = in32 + (size_t)first_of[(size_t)i] * in_n * 2;
            const auto cp = [&](size_t b, size_t e) { hp_stream_copy(h->pin_in[s] + b, reinterpret_cast<const char*>(src) + b, e - b); };
            const size_t bytes = (size_t)np_of(i) * in_n * 2 * sizeof(float);
            if (direct_out && c->hp_side_threads != 0) hp_parallel_range2(h->pool_in, h->pool_out, bytes, (size_t)1 << 18, 64, cp);
            else h->pool_in.parallel_range(bytes, (size_t)1 << 18, cp);
        };
    if (!in_c64)
    p.stage = [&](int64_t i, int s) {                                    // complex128 -> two float32 planes in pinned_in[s]
        const double* src = in + (size_t)first_of[(size_t)i] * in_n * 2;
        float* p_re = reinterpret_cast<float*>(h->pin_in[s]);
        float* p_im = reinterpret_cast<float*>(h->pin_in[s] + in_pkt * chunk);
        const auto split = [&](size_t b, size_t e) { hp_split_c128(src, p_re, p_im, b, e); };
        // no host pass on the result side: the output pool's threads join the input side (side-thread arrangement only: with
        // hp_side_threads = 0 each pool already holds the full thread count)
        if (direct_out && c->hp_side_threads != 0) hp_parallel_range2(h->pool_in, h->pool_out, (size_t)np_of(i) * in_n, (size_t)1 << 16, 16, split);
        else h->pool_in.parallel_range((size_t)np_of(i) * in_n, (size_t)1 << 16, split);
    };
    p.enqueue_in = [&](int64_t i, int s) -> int {
        if (in_c64) {                                                    // ONE upload of the interleaved chunk
            const void* src = in_pinned ? static_cast<const void*>(in32 + (size_t)first_of[(size_t)i] * in_n * 2) : static_cast<const void*>(h->pin_in[s]);
            HIP_TRY(c, hipMemcpyAsync(h->dev[s] + raw_off, src, 2 * in_pkt * np_of(i), hipMemcpyHostToDevice, h->s_in));
            return CSI_OK;
        }
        HIP_TRY(c, hipMemcpyAsync(h->dev[s], h->pin_in[s], in_pkt * np_of(i), hipMemcpyHostToDevice, h->s_in));
        HIP_TRY(c, hipMemcpyAsync(h->dev[s] + in_pkt * chunk, h->pin_in[s] + in_pkt * chunk, in_pkt * np_of(i), hipMemcpyHostToDevice, h->s_in));
        return CSI_OK;
    };
    // device[s] / pinned_out[s] output layout: dnn re | dnn im | ls re | ls im, each sized for `chunk` packets
    p.compute = [&](int64_t i, int s) -> int {
        float* d_re = reinterpret_cast<float*>(h->dev[s]);
        float* d_im = reinterpret_cast<float*>(h->dev[s] + in_pkt * chunk);
        float* d_out = reinterpret_cast<float*>(h->dev[s] + 2 * in_pkt * chunk);
        int r = CSI_OK;
        if (in_c64) {                                                    // interleaved chunk -> the two planes, same stream, in front of the kernels
            const size_t n = (size_t)np_of(i) * in_n;
            const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
            hipLaunchKernelGGL(split_c64_kernel, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const float2*>(h->dev[s] + raw_off), d_re, d_im, n);
            HIP_TRY(c, hipGetLastError());
        }
        if (ls_c64) r = csi_ls_estimate_device(c, d_re, d_im, np_of(i), d_out + 2 * dnn_n * chunk, d_out + 2 * dnn_n * chunk + ls_n * chunk);
        if (!r && dnn_c64) r = csi_predict_device(c, d_re, d_im, np_of(i), d_out, d_out + dnn_n * chunk);
        if (!r && direct_out) {                                          // planes -> complex64 behind them, same stream
            float* d_c64 = d_out + 2 * (dnn_n + ls_n) * chunk;
            auto weave = [&](const float* re, const float* im, float* dst, size_t n) -> int {
                const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
                hipLaunchKernelGGL(weave_c64_kernel, dim3(blocks), dim3(256), 0, c->stream, re, im, reinterpret_cast<float2*>(dst), n);
                HIP_TRY(c, hipGetLastError());
                return CSI_OK;
            };
            if (dnn_c64) r = weave(d_out, d_out + dnn_n * chunk, d_c64, (size_t)np_of(i) * dnn_n);
            if (!r && ls_c64) r = weave(d_out + 2 * dnn_n * chunk, d_out + 2 * dnn_n * chunk + ls_n * chunk, d_c64 + 2 * dnn_n * chunk, (size_t)np_of(i) * ls_n);
        }
        return r;
    };
    p.enqueue_out = [&](int64_t i, int s) -> int {                      // one D2H per plane actually filled (np of chunk packets)
        const float* d_out = reinterpret_cast<const float*>(h->dev[s] + 2 * in_pkt * chunk);
        const int64_t np = np_of(i);
        if (direct_out) {                                                // the chunk's complex64 values into the caller's pinned arrays
            const float* d_c64 = d_out + 2 * (dnn_n + ls_n) * chunk;
            const size_t first = (size_t)first_of[(size_t)i];
            if (dnn_c64) HIP_TRY(c, hipMemcpyAsync(dnn_c64 + first * dnn_n * 2, d_c64, dnn_n * np * 2 * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
            if (ls_c64) HIP_TRY(c, hipMemcpyAsync(ls_c64 + first * ls_n * 2, d_c64 + 2 * dnn_n * chunk, ls_n * np * 2 * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
            return CSI_OK;
        }
        float* t = reinterpret_cast<float*>(h->pin_out[s]);
        if (dnn_c64) {
            HIP_TRY(c, hipMemcpyAsync(t, d_out, dnn_n * np * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
            HIP_TRY(c, hipMemcpyAsync(t + dnn_n * chunk, d_out + dnn_n * chunk, dnn_n * np * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
        }
        if (ls_c64) {
            const size_t o = 2 * dnn_n * chunk;
            HIP_TRY(c, hipMemcpyAsync(t + o, d_out + o, ls_n * np * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
            HIP_TRY(c, hipMemcpyAsync(t + o + ls_n * chunk, d_out + o + ls_n * chunk, ls_n * np * sizeof(float), hipMemcpyDeviceToHost, h->s_out));
        }
        return CSI_OK;
    };
    if (!direct_out)
        p.weave = [&](int64_t i, int s) {                                // two float32 planes -> complex64 in the caller's arrays
            const int64_t np = np_of(i);
            const float* t = reinterpret_cast<const float*>(h->pin_out[s]);
            auto weave = [&](const float* re, const float* im, float* dst, size_t n) {
                h->pool_out.parallel_range(n, (size_t)1 << 16, [&](size_t b, size_t e) { hp_weave_c64(re, im, dst, b, e); });
            };
            if (dnn_c64) weave(t, t + dnn_n * chunk, dnn_c64 + (size_t)first_of[(size_t)i] * dnn_n * 2, (size_t)np * dnn_n);
            if (ls_c64) weave(t + 2 * dnn_n * chunk, t + 2 * dnn_n * chunk + ls_n * chunk, ls_c64 + (size_t)first_of[(size_t)i] * ls_n * 2, (size_t)np * ls_n);
        };
    return hp_run(c, h, p);
}

// The link itself: `up` bytes host -> device and `down` bytes device -> host between PINNED host memory and device memory, in
// 32 MiB pieces on the pipeline's two copy streams - each direction alone, then both at once (what a perfectly overlapped call of
// these byte counts would need: the ceiling the host-buffer entry points are measured against).
int hp_pcie_probe(csi_ctx* c, int64_t up, int64_t down, double* ms_up, double* ms_down, double* ms_both) {
    csi_hostpipe* h = nullptr;
    int rc = hp_get(c, &h);
    if (rc) return rc;
    const size_t piece = (size_t)32 << 20;
    char *pu = nullptr, *pd = nullptr, *du = nullptr, *dd = nullptr;
    auto release = [&] {
        if (pu) hipHostFree(pu);
        if (pd) hipHostFree(pd);
        if (du) hipFree(du);
        if (dd) hipFree(dd);
    };
    if (hipHostMalloc((void**)&pu, 2 * piece, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&pd, 2 * piece, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&du, 2 * piece) != hipSuccess || hipMalloc((void**)&dd, 2 * piece) != hipSuccess) {
        release();
        (void)hipGetLastError();
        return fail(c, CSI_ERR_NOMEM, "csi_profile_pcie: staging allocation failed");
    }
    std::memset(pu, 1, 2 * piece);
    std::memset(pd, 1, 2 * piece);
    hipError_t e = hipMemset(dd, 0, 2 * piece);
    auto run = [&](int64_t nu, int64_t nd, double* ms) {
        if (e != hipSuccess) return;
        e = hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        int k = 0;
        for (int64_t o = 0; o < nu && e == hipSuccess; o += (int64_t)piece, ++k)
            e = hipMemcpyAsync(du + (k & 1) * piece, pu + (k & 1) * piece, (size_t)std::min<int64_t>(piece, nu - o), hipMemcpyHostToDevice, h->s_in);
        k = 0;
        for (int64_t o = 0; o < nd && e == hipSuccess; o += (int64_t)piece, ++k)
            e = hipMemcpyAsync(pd + (k & 1) * piece, dd + (k & 1) * piece, (size_t)std::min<int64_t>(piece, nd - o), hipMemcpyDeviceToHost, h->s_out);
        if (e == hipSuccess) e = hipStreamSynchronize(h->s_in);
        if (e == hipSuccess) e = hipStreamSynchronize(h->s_out);
        *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    double w = 0, a = 0, b = 0, ab = 0;
    run(std::min<int64_t>(up, 4 * (int64_t)piece), std::min<int64_t>(down, 4 * (int64_t)piece), &w);      // warm-up
    run(up, 0, &a);
    run(0, down, &b);
    run(up, down, &ab);
    release();
    if (e != hipSuccess) return fail(c, CSI_ERR_HIP, "csi_profile_pcie: %s", hipGetErrorString(e));
    if (ms_up) *ms_up = a;
    if (ms_down) *ms_down = b;
    if (ms_both) *ms_both = ab;
    return CSI_OK;
}

int hp_estimate_c128(csi_ctx* c, const void* in, bool in_c64, int64_t npkt, float* dnn_c64, float* ls_c64) {
    csi_hostpipe* h = nullptr;
    int rc = hp_get(c, &h);
    if (rc) return rc;
    rc = hp_estimate_c128_impl(c, h, in, in_c64, npkt, dnn_c64, ls_c64);
    if (rc) {
        const std::string keep = c->err;
        hipStreamSynchronize(h->s_in);
        hipStreamSynchronize(c->stream);
        hipStreamSynchronize(h->s_out);
        (void)hipGetLastError();
        c->err = keep;
    }
    return rc;
}

}  // namespace
