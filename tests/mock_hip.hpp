// mock_hip.hpp - a small model of the HIP runtime for CPU harnesses of OUR host code (tests/comm_mock_check.cpp): include it AFTER the
// code under test (which has included <hip/hip_runtime.h>).  The definitions take precedence over libamdhip64's at link time; nothing of
// the real runtime is called.
//   * a stream is a FIFO of work items run by its own thread (the null stream is one more such stream); copies, memsets and whatever a
//     harness pushes are items on it;
//   * hipEventRecord marks a position; hipStreamWaitEvent makes a stream wait for the position recorded at call time;
//     hipEventSynchronize / hipStreamSynchronize / hipDeviceSynchronize block the caller - the happens-before edges HIP guarantees;
//   * "device" and pinned memory are host memory; hipMalloc fills a buffer with a pattern derived from a global counter, so that two
//     allocations never look alike by accident;
//   * kernel launches (hipLaunchKernel, hipModuleLaunchKernel) go to mock::launch_hook when it is set and are otherwise counted and
//     dropped - a harness of host logic does not need what the kernels compute;
//   * one "node" of mock::n_devices gfx950 devices.
// Test scaffolding only; it is not a stand-in for anything of the reference.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
namespace mock {

struct Stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t submitted = 0, completed = 0;
    bool stop = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            f();
            {
                std::lock_guard<std::mutex> lk(mu);
                ++completed;
            }
            cv.notify_all();
        }
    }
    std::vector<std::function<void()>>* capture = nullptr;      // stream capture (hipStreamBeginCapture): items are recorded, not run
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (capture) { capture->push_back(std::move(f)); return; }
            q.push_back(std::move(f));
            ++submitted;
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t want = submitted;
        cv.wait(lk, [&] { return completed >= want; });
    }
    ~Stream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
};

struct Event {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, done = 0;
};

inline std::mutex g_mu;
inline std::set<Stream*> g_streams;
inline std::map<const char*, size_t> g_pinned;          // base -> bytes
inline std::atomic<long> g_copies{0}, g_peer_copies{0}, g_waits{0}, g_launches{0}, g_allocs{0};
inline int n_devices = 8;
inline thread_local long fail_alloc_in = -1;        // > 0: the n-th hipMalloc of THIS thread from now on fails (out of memory), once
inline std::function<hipError_t(const void* fn, void** args, hipStream_t st)> launch_hook;

inline Stream* null_stream() {
    static Stream* s = [] {
        auto* p = new Stream();
        std::lock_guard<std::mutex> lk(g_mu);
        g_streams.insert(p);
        return p;
    }();
    return s;
}
inline bool is_pinned(const void* p) {                                // anywhere inside a hipHostMalloc'ed range (g_mu held by the caller)
    auto it = g_pinned.upper_bound(static_cast<const char*>(p));
    if (it == g_pinned.begin()) return false;
    --it;
    return static_cast<const char*>(p) < it->first + it->second;
}
inline Stream* S(hipStream_t s) { return s ? reinterpret_cast<Stream*>(s) : null_stream(); }
inline Event* E(hipEvent_t e) { return reinterpret_cast<Event*>(e); }

}  // namespace mock

extern "C" {
hipError_t hipSetDevice(int d) { return d >= 0 && d < mock::n_devices ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDeviceCount(int* n) { *n = mock::n_devices; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= mock::n_devices) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "mock MI355X %d", d);
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock HIP error"; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    auto* p = new mock::Stream();
    std::lock_guard<std::mutex> lk(mock::g_mu);
    mock::g_streams.insert(p);
    *s = reinterpret_cast<hipStream_t>(p);
    return hipSuccess;
}
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(mock::g_mu);
        mock::g_streams.erase(mock::S(s));
    }
    delete mock::S(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    mock::S(s)->sync();
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
    std::vector<mock::Stream*> all;
    {
        std::lock_guard<std::mutex> lk(mock::g_mu);
        all.assign(mock::g_streams.begin(), mock::g_streams.end());
    }
    for (auto* s : all) s->sync();
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = reinterpret_cast<hipEvent_t>(new mock::Event());
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete mock::E(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    mock::Event* ev = mock::E(e);
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(ev->mu);
        g = ++ev->recorded;
    }
    mock::S(s)->push([ev, g] {
        std::lock_guard<std::mutex> lk(ev->mu);      // (notified under the lock: whoever sees `done` may destroy the event at once, as HIP allows)
        if (ev->done < g) ev->done = g;
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    mock::Event* ev = mock::E(e);
    std::unique_lock<std::mutex> lk(ev->mu);
    const uint64_t g = ev->recorded;
    ev->cv.wait(lk, [&] { return ev->done >= g; });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.25f; return hipSuccess; }      // (a fixed, non-zero "duration": callers divide by it)
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    mock::Event* ev = mock::E(e);
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(ev->mu);
        g = ev->recorded;
    }
    ++mock::g_waits;
    if (g)
        mock::S(s)->push([ev, g] {
            std::unique_lock<std::mutex> lk(ev->mu);
            ev->cv.wait(lk, [&] { return ev->done >= g; });
        });
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    if (mock::fail_alloc_in > 0 && --mock::fail_alloc_in == 0) { mock::fail_alloc_in = -1; *p = nullptr; return hipErrorOutOfMemory; }
    const size_t bytes = (n + 255) / 256 * 256;
    *p = std::aligned_alloc(256, bytes ? bytes : 256);
    if (!*p) return hipErrorOutOfMemory;
    const unsigned seed = (unsigned)(++mock::g_allocs) * 2654435761u;
    unsigned* w = static_cast<unsigned*>(*p);
    for (size_t i = 0; i < bytes / 4; ++i) w[i] = seed + (unsigned)i * 40503u;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    *p = std::aligned_alloc(4096, (n + 4095) / 4096 * 4096);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(mock::g_mu);
    mock::g_pinned[static_cast<const char*>(*p)] = (n + 4095) / 4096 * 4096;
    return hipSuccess;
}
hipError_t hipHostFree(void* p) {
    {
        std::lock_guard<std::mutex> lk(mock::g_mu);
        mock::g_pinned.erase(static_cast<const char*>(p));
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> lk(mock::g_mu);
    if (!mock::is_pinned(p)) return hipErrorInvalidValue;
    std::memset(a, 0, sizeof *a);
    a->type = hipMemoryTypeHost;
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) {
    ++mock::g_copies;
    bool pageable_src = false;
    if (k == hipMemcpyHostToDevice || k == hipMemcpyHostToHost) {
        std::lock_guard<std::mutex> lk(mock::g_mu);
        pageable_src = !mock::is_pinned(s);
    }
    if (pageable_src) {
        // HIP reads a PAGEABLE source before the call returns (staged copy): the caller may reuse it at once - modelled by copying it now;
        // a pinned source is read when the stream gets there
        auto keep = std::make_shared<std::vector<char>>(static_cast<const char*>(s), static_cast<const char*>(s) + n);
        mock::S(st)->push([d, keep] { std::memcpy(d, keep->data(), keep->size()); });
    } else {
        mock::S(st)->push([d, s, n] { std::memcpy(d, s, n); });
    }
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t st) {
    ++mock::g_copies;
    ++mock::g_peer_copies;
    mock::S(st)->push([d, s, n] { std::memcpy(d, s, n); });
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    mock::null_stream()->sync();
    std::memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
    mock::null_stream()->sync();
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
    mock::S(st)->push([d, v, n] { std::memset(d, v, n); });
    return hipSuccess;
}
hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t w, size_t h, hipStream_t st) {
    mock::S(st)->push([d, pitch, v, w, h] { for (size_t r = 0; r < h; ++r) std::memset(static_cast<char*>(d) + r * pitch, v, w); });
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipModuleLoadData(hipModule_t* m, const void*) { *m = reinterpret_cast<hipModule_t>(new int(1)); return hipSuccess; }
hipError_t hipModuleLoad(hipModule_t* m, const char*) { *m = reinterpret_cast<hipModule_t>(new int(1)); return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { delete reinterpret_cast<int*>(m); return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t, const char*) { *f = reinterpret_cast<hipFunction_t>(new int(2)); return hipSuccess; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void**, void**) {
    ++mock::g_launches;
    return hipSuccess;
}
static thread_local struct { dim3 g, b; size_t sh; hipStream_t st; } mock_cfg_;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) {
    mock_cfg_.g = g; mock_cfg_.b = b; mock_cfg_.sh = sh; mock_cfg_.st = st;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
    *g = mock_cfg_.g; *b = mock_cfg_.b; *sh = mock_cfg_.sh; *st = mock_cfg_.st;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3, dim3, void** args, size_t, hipStream_t st) {
    ++mock::g_launches;
    return mock::launch_hook ? mock::launch_hook(fn, args, st) : hipSuccess;
}
// stream capture / graphs, as far as csi_estimate_device + "use_graph" needs them: while a stream captures, what is pushed onto it
// (kernel launches are dropped anyway; memsets, copies, event records) is recorded into the graph instead of run; an instantiated
// graph is that list, a launch pushes its items onto the stream in order.  One capturing stream at a time per graph, no cross-stream
// joins (the library captures single-stream sequences only: csi_context.hpp, in_graph_call).
hipError_t hipStreamBeginCapture(hipStream_t st, hipStreamCaptureMode) {
    mock::Stream* s = mock::S(st);
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->capture) return hipErrorIllegalState;
    s->capture = new std::vector<std::function<void()>>();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t st, hipGraph_t* g) {
    mock::Stream* s = mock::S(st);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->capture) return hipErrorIllegalState;
    *g = reinterpret_cast<hipGraph_t>(s->capture);
    s->capture = nullptr;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) {
    *e = reinterpret_cast<hipGraphExec_t>(new std::vector<std::function<void()>>(*reinterpret_cast<std::vector<std::function<void()>>*>(g)));
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t st) {
    for (auto& f : *reinterpret_cast<std::vector<std::function<void()>>*>(e)) mock::S(st)->push(f);
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete reinterpret_cast<std::vector<std::function<void()>>*>(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete reinterpret_cast<std::vector<std::function<void()>>*>(e); return hipSuccess; }
}  // extern "C"
#endif
