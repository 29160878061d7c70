// hostpipe_mock_check.cpp - the host pipeline of the host-buffer entry points (csrc/csi_hostpipe.hpp: hp_run with its stager / drainer
// threads, two pinned + two device slots, three streams chained by events) executed WITHOUT a GPU against a small model of the HIP
// stream semantics, so that ThreadSanitizer can see every hand-over of a buffer:
//   * a stream is a FIFO of work items run by its own thread; hipMemcpyAsync and the "kernels" are items on it;
//   * hipEventRecord marks a position, hipStreamWaitEvent makes a stream wait for the position recorded at call time,
//     hipEventSynchronize / hipStreamSynchronize block the caller - the happens-before edges HIP guarantees and no others;
//   * "device memory" is host memory, csi_predict_device / csi_ls_estimate_device are stand-ins that enqueue a simple arithmetic map
//     on the context's stream (what the pipeline needs from them: stream order, input -> output).
// A dependency the pipeline forgets (a slot reused before its download ended, a staging buffer rewritten under a running upload) is a
// data race here, and a wrong result.  The model is tests/mock_hip.hpp (its definitions take precedence over libamdhip64's at link
// time; nothing of the real runtime is called).  This is test scaffolding for OUR host code - it is not a stand-in for anything of the
// reference.
//   hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -pthread [-Xarch_host -fsanitize=thread] tests/hostpipe_mock_check.cpp -o /tmp/hpmock && /tmp/hpmock
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dl-channel-estimation-mamimo_amd/csrc/csi_hostpipe.hpp"

#include "mock_hip.hpp"          // the model of the HIP runtime (streams as FIFO threads, events as positions, "device" memory on the host)

#if !defined(__HIP_DEVICE_COMPILE__)
extern "C" {
// the two device entry points the pipeline calls, as stream-ordered arithmetic maps (input chunk -> output planes)
static inline float dnn_map(const float* x, size_t len, int t, int k) { return x[(size_t)(t * 7 + k) % len] * 0.5f + (float)t; }
int csi_predict_device(csi_ctx* c, const float* d_re, const float* d_im, int64_t np, float* o_re, float* o_im) {
    const csi_config cf = c->cfg;
    mock::S(c->stream)->push([=] {
        const size_t len = (size_t)cf.len_ltf;
        for (int64_t p = 0; p < np; ++p)
            for (int r = 0; r < cf.nr; ++r)
                for (int t = 0; t < cf.nt; ++t)
                    for (int k = 0; k < cf.n_out; ++k) {
                        const size_t o = (((size_t)p * cf.nr + r) * cf.nt + t) * cf.n_out + k, i = ((size_t)p * cf.nr + r) * len;
                        o_re[o] = dnn_map(d_re + i, len, t, k);
                        o_im[o] = dnn_map(d_im + i, len, t, k) - 3.f;
                    }
    });
    return CSI_OK;
}
int csi_ls_estimate_device(csi_ctx* c, const float* d_re, const float* d_im, int64_t np, float* h_re, float* h_im) {
    const csi_config cf = c->cfg;
    mock::S(c->stream)->push([=] {
        const size_t len = (size_t)cf.len_ltf;
        for (int64_t p = 0; p < np; ++p)
            for (int r = 0; r < cf.nr; ++r)
                for (int t = 0; t < cf.nt; ++t)
                    for (int k = 0; k < LS_NDATA; ++k) {
                        const size_t o = (((size_t)p * cf.nr + r) * cf.nt + t) * LS_NDATA + k, i = ((size_t)p * cf.nr + r) * len + (size_t)(k + 11 * t) % len;
                        h_re[o] = d_re[i] - d_im[i];
                        h_im[o] = d_re[i] + 2.f * d_im[i];
                    }
    });
    return CSI_OK;
}
}  // extern "C"

// the kernels the pipeline launches itself, as stream items: weave_c64_kernel (result arrays in pinned memory) - args (re, im, out, n) -
// and, round 5, split_c64_kernel (csi_estimate_c64: the interleaved chunk as uploaded -> the two planes) - args (in, re, im, n)
static hipError_t weave_launch(const void* fn, void** args, hipStream_t st) {
    if (fn == reinterpret_cast<const void*>(&csi::split_c64_kernel)) {
        const float* in = reinterpret_cast<const float*>(*static_cast<const float2**>(args[0]));
        float* re = *static_cast<float**>(args[1]);
        float* im = *static_cast<float**>(args[2]);
        const size_t n = *static_cast<size_t*>(args[3]);
        mock::S(st)->push([=] { for (size_t i = 0; i < n; ++i) { re[i] = in[2 * i]; im[i] = in[2 * i + 1]; } });
        return hipSuccess;
    }
    if (fn != reinterpret_cast<const void*>(&csi::weave_c64_kernel)) return hipErrorInvalidDeviceFunction;
    const float* re = *static_cast<const float**>(args[0]);
    const float* im = *static_cast<const float**>(args[1]);
    float* out = reinterpret_cast<float*>(*static_cast<float2**>(args[2]));
    const size_t n = *static_cast<size_t*>(args[3]);
    mock::S(st)->push([=] { for (size_t i = 0; i < n; ++i) { out[2 * i] = re[i]; out[2 * i + 1] = im[i]; } });
    return hipSuccess;
}

int main() {
    int bad = 0;
    mock::launch_hook = weave_launch;
    csi_ctx* c = new csi_ctx();
    c->cfg.nt = 2; c->cfg.nr = 2; c->cfg.len_ltf = 640; c->cfg.n_out = 52; c->cfg.device = 0;
    hipStreamCreateWithFlags(&c->stream, 0);
    const int64_t npkt = 203;
    const size_t in_n = (size_t)c->cfg.nr * c->cfg.len_ltf, dnn_n = (size_t)c->cfg.nr * c->cfg.nt * c->cfg.n_out, ls_n = (size_t)c->cfg.nr * c->cfg.nt * LS_NDATA;
    std::vector<double> x(2 * in_n * npkt);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (double)((i * 2654435761u) % 100003) / 97.0 - 500.0;
    // what the call has to return, straight from the input
    std::vector<float> re(in_n * npkt), im(in_n * npkt), want_dnn(2 * dnn_n * npkt), want_ls(2 * ls_n * npkt);
    for (size_t i = 0; i < in_n * npkt; ++i) { re[i] = (float)x[2 * i]; im[i] = (float)x[2 * i + 1]; }
    std::vector<float> x32(x.size());                     // the batch as complex64: the values the complex128 call converts to
    for (size_t i = 0; i < x.size(); ++i) x32[i] = (float)x[i];
    for (int64_t p = 0; p < npkt; ++p)
        for (int r = 0; r < c->cfg.nr; ++r)
            for (int t = 0; t < c->cfg.nt; ++t) {
                const size_t i = ((size_t)p * c->cfg.nr + r) * c->cfg.len_ltf, len = (size_t)c->cfg.len_ltf;
                for (int k = 0; k < c->cfg.n_out; ++k) {
                    const size_t o = (((size_t)p * c->cfg.nr + r) * c->cfg.nt + t) * c->cfg.n_out + k;
                    want_dnn[2 * o] = dnn_map(re.data() + i, len, t, k);
                    want_dnn[2 * o + 1] = dnn_map(im.data() + i, len, t, k) - 3.f;
                }
                for (int k = 0; k < LS_NDATA; ++k) {
                    const size_t o = (((size_t)p * c->cfg.nr + r) * c->cfg.nt + t) * LS_NDATA + k, j = i + (size_t)(k + 11 * t) % len;
                    want_ls[2 * o] = re[j] - im[j];
                    want_ls[2 * o + 1] = re[j] + 2.f * im[j];
                }
            }
    std::vector<float> dnn(2 * dnn_n * npkt), ls(2 * ls_n * npkt);
    int calls = 0;
    for (int side : {1, 0})
        for (int threads : {0, 3, 6})
            for (int chunk : {0, 7, 32, 50, 203, 500}) {
                if (c->hostpipe) { delete c->hostpipe; c->hostpipe = nullptr; }      // (what csi_set_option "host_threads" does)
                c->hp_side_threads = side;
                c->host_threads = threads;
                c->hp_chunk_packets = chunk;
                for (int what = 0; what < 3; ++what) {                               // both estimators, DNN alone, LS alone
                    std::fill(dnn.begin(), dnn.end(), -7.f);
                    std::fill(ls.begin(), ls.end(), -7.f);
                    const int rc = hp_estimate_c128(c, x.data(), false, npkt, what != 2 ? dnn.data() : nullptr, what != 1 ? ls.data() : nullptr);
                    ++calls;
                    if (rc) { ++bad; std::printf("side %d threads %d chunk %d what %d: rc %d (%s)\n", side, threads, chunk, what, rc, c->err.c_str()); continue; }
                    if (what != 2 && std::memcmp(dnn.data(), want_dnn.data(), dnn.size() * 4)) { ++bad; std::printf("side %d threads %d chunk %d what %d: DNN result differs\n", side, threads, chunk, what); }
                    if (what != 1 && std::memcmp(ls.data(), want_ls.data(), ls.size() * 4)) { ++bad; std::printf("side %d threads %d chunk %d what %d: LS result differs\n", side, threads, chunk, what); }
                }
                // result arrays in "pinned" memory: complex64 assembled by the launched kernel, downloads into the arrays themselves
                {
                    float *pd = nullptr, *pl = nullptr;
                    hipHostMalloc(reinterpret_cast<void**>(&pd), dnn.size() * 4, 0);
                    hipHostMalloc(reinterpret_cast<void**>(&pl), ls.size() * 4, 0);
                    for (int what = 0; what < 3; ++what) {
                        std::fill(pd, pd + dnn.size(), -7.f);
                        std::fill(pl, pl + ls.size(), -7.f);
                        const int64_t before = c->hp_direct_out_calls;
                        const int rc = hp_estimate_c128(c, x.data(), false, npkt, what != 2 ? pd : nullptr, what != 1 ? pl : nullptr);
                        ++calls;
                        if (rc || c->hp_direct_out_calls != before + 1) { ++bad; std::printf("side %d threads %d chunk %d what %d pinned: rc %d (%s), direct %lld\n", side, threads, chunk, what, rc, c->err.c_str(), (long long)(c->hp_direct_out_calls - before)); continue; }
                        if (what != 2 && std::memcmp(pd, want_dnn.data(), dnn.size() * 4)) { ++bad; std::printf("side %d threads %d chunk %d what %d: pinned DNN result differs\n", side, threads, chunk, what); }
                        if (what != 1 && std::memcmp(pl, want_ls.data(), ls.size() * 4)) { ++bad; std::printf("side %d threads %d chunk %d what %d: pinned LS result differs\n", side, threads, chunk, what); }
                    }
                    // round 5, csi_estimate_c64: the same batch as complex64 - from pageable memory (staged by a plain copy) and from
                    // "pinned" memory (uploaded from the caller's array itself), into pageable and into pinned result arrays
                    float* px = nullptr;
                    hipHostMalloc(reinterpret_cast<void**>(&px), x32.size() * 4, 0);
                    std::memcpy(px, x32.data(), x32.size() * 4);
                    for (int src = 0; src < 2; ++src)
                        for (int what = 0; what < 3; ++what) {
                            std::fill(pd, pd + dnn.size(), -7.f); std::fill(pl, pl + ls.size(), -7.f);
                            std::fill(dnn.begin(), dnn.end(), -7.f); std::fill(ls.begin(), ls.end(), -7.f);
                            const float* in64 = src ? px : x32.data();
                            int rc = hp_estimate_c128(c, in64, true, npkt, what != 2 ? pd : nullptr, what != 1 ? pl : nullptr);
                            if (!rc) rc = hp_estimate_c128(c, in64, true, npkt, what != 2 ? dnn.data() : nullptr, what != 1 ? ls.data() : nullptr);
                            calls += 2;
                            if (rc) { ++bad; std::printf("side %d threads %d chunk %d what %d c64 src %d: rc %d (%s)\n", side, threads, chunk, what, src, rc, c->err.c_str()); continue; }
                            if (what != 2 && (std::memcmp(pd, want_dnn.data(), dnn.size() * 4) || std::memcmp(dnn.data(), want_dnn.data(), dnn.size() * 4))) { ++bad; std::printf("side %d threads %d chunk %d what %d c64 src %d: DNN result differs\n", side, threads, chunk, what, src); }
                            if (what != 1 && (std::memcmp(pl, want_ls.data(), ls.size() * 4) || std::memcmp(ls.data(), want_ls.data(), ls.size() * 4))) { ++bad; std::printf("side %d threads %d chunk %d what %d c64 src %d: LS result differs\n", side, threads, chunk, what, src); }
                        }
                    hipHostFree(px);
                    hipHostFree(pd);
                    hipHostFree(pl);
                }
                // the plane entry points' pipeline (hp_packets) on the same model: DNN planes
                std::vector<float> o_re(dnn_n * npkt, -7.f), o_im(dnn_n * npkt, -7.f);
                const int rc = hp_packets(c, re.data(), im.data(), npkt, o_re.data(), o_im.data(), c->cfg.n_out,
                                          [c](const float* a, const float* b, int64_t np, float* p, float* q) { return csi_predict_device(c, a, b, np, p, q); });
                ++calls;
                bool same = rc == 0;
                for (size_t i = 0; same && i < dnn_n * npkt; ++i) same = o_re[i] == want_dnn[2 * i] && o_im[i] == want_dnn[2 * i + 1];
                if (!same) { ++bad; std::printf("side %d threads %d chunk %d: plane pipeline rc %d or result differs\n", side, threads, chunk, rc); }
            }
    // the plane pipeline over several chunks (its slots are >= 8 MiB of input): pageable and "pinned" caller buffers, both arrangements
    {
        if (c->hostpipe) { delete c->hostpipe; c->hostpipe = nullptr; }
        c->cfg.nt = 16; c->cfg.nr = 8; c->cfg.len_ltf = 512; c->cfg.n_out = 20;
        c->host_threads = 4;
        const int64_t np2 = 1700;                                                    // 512-packet slots: 128 | 512 | 512 | 420 | 128
        const size_t in2 = (size_t)c->cfg.nr * c->cfg.len_ltf * np2, out2 = (size_t)c->cfg.nr * c->cfg.nt * c->cfg.n_out * np2;
        float *bufs[4];
        for (int pinned = 0; pinned < 2; ++pinned) {
            for (int k = 0; k < 4; ++k) {
                const size_t bytes = (k < 2 ? in2 : out2) * 4;
                if (pinned) hipHostMalloc(reinterpret_cast<void**>(&bufs[k]), bytes, 0);
                else bufs[k] = static_cast<float*>(std::malloc(bytes));
            }
            for (size_t i = 0; i < in2; ++i) { bufs[0][i] = (float)((i * 40503u) % 8191) * 0.25f - 1000.f; bufs[1][i] = (float)((i * 12289u) % 4093) - 2000.f; }
            for (int side : {1, 0}) {
                c->hp_side_threads = side;
                std::fill(bufs[2], bufs[2] + out2, -7.f);
                std::fill(bufs[3], bufs[3] + out2, -7.f);
                const int rc = hp_packets(c, bufs[0], bufs[1], np2, bufs[2], bufs[3], c->cfg.n_out,
                                          [c](const float* a, const float* b, int64_t np, float* p, float* q) { return csi_predict_device(c, a, b, np, p, q); });
                ++calls;
                size_t wrong = rc ? 1 : 0;
                const size_t len = (size_t)c->cfg.len_ltf;
                for (int64_t p = 0; p < np2 && !wrong; ++p)
                    for (int r = 0; r < c->cfg.nr && !wrong; ++r)
                        for (int t = 0; t < c->cfg.nt && !wrong; ++t)
                            for (int k = 0; k < c->cfg.n_out; ++k) {
                                const size_t o = (((size_t)p * c->cfg.nr + r) * c->cfg.nt + t) * c->cfg.n_out + k, i = ((size_t)p * c->cfg.nr + r) * len;
                                if (bufs[2][o] != dnn_map(bufs[0] + i, len, t, k) || bufs[3][o] != dnn_map(bufs[1] + i, len, t, k) - 3.f) { ++wrong; break; }
                            }
                if (wrong) { ++bad; std::printf("plane pipeline, %s buffers, side %d: rc %d or a result differs\n", pinned ? "pinned" : "pageable", side, rc); }
            }
            for (int k = 0; k < 4; ++k) { if (pinned) hipHostFree(bufs[k]); else std::free(bufs[k]); }
        }
    }
    std::printf("%d pipelined calls on the stream model, %ld copies, %ld stream waits, %ld kernel launches\n", calls, mock::g_copies.load(), mock::g_waits.load(), mock::g_launches.load());
    std::printf(bad ? "FAILED (%d)\n" : "hostpipe_mock_check: ok\n", bad);
    return bad ? 1 : 0;
}
#else
int main() { return 0; }
#endif
