// comm_mock_check.cpp - csi_comm_init / csi_broadcast_weights (the one collective of the path) executed by N RANKS ON ONE
// CPU: the library's whole translation unit is compiled into this harness against a model of the HIP runtime (tests/mock_hip.hpp:
// streams as FIFO threads, "device" memory on the host, kernel launches dropped) and a model of RCCL (below: ranks are threads of this
// process, a communicator is a shared rendezvous, ncclBroadcast / ncclAllReduce are stream items that meet their peers in call order -
// a rank that skips a collective its peers enter leaves them waiting, which the watchdog reports as a hang).  What it shows:
//   * world = 2, 4, 8, root 0 and root != 0: every receiver ends with the root's device buffers BYTE FOR BYTE (the blobs of both
//     component models, P, Ppad, Pbf), the root's flags, its own derived tables, and rc 0 on every rank;
//   * a receiver whose csi_config differs: IT gets CSI_ERR_INVALID_ARG with the reason, EVERY other rank gets the "another rank
//     refused" error, nobody hangs, the refusing receiver holds nothing, the root keeps its model, and a second broadcast among
//     well-configured contexts on the same communicators then succeeds (the group was closed, the communicator is usable);
//   * a receiver whose device allocation fails part-way: CSI_ERR_NOMEM there, the refusal everywhere else, and the SAME contexts and
//     communicators complete the next broadcast;
//   * a root with nothing loaded: receivers end empty, rc 0.
// The kernels do not run here (launches are dropped), so a "loaded" model's re-laid-out buffers hold allocation patterns instead of
// weights - which is all the transfer protocol needs: distinct bytes on the root that must arrive unchanged.
// Test scaffolding for OUR host code; not a stand-in for anything of the reference.
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 -pthread tests/comm_mock_check.cpp -o /tmp/comm_mock && /tmp/comm_mock
#include "../dl-channel-estimation-mamimo_amd/csrc/csi_mamimo.hip"

#include "mock_hip.hpp"

#if !defined(__HIP_DEVICE_COMPILE__)
#include <chrono>
#include <map>

namespace mnccl {

struct Group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int joined = 0;
    // one collective at a time per communicator (stream order is call order on every rank): a two-phase barrier around the data movement
    uint64_t phase = 0;
    int arrived = 0;
    std::vector<const void*> send;
    std::vector<void*> recv;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t ph = phase;
        if (++arrived == world) { arrived = 0; ++phase; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph; });
    }
};
struct Comm { Group* g; int rank; };

std::mutex g_mu;
std::map<std::string, Group*> g_groups;
std::atomic<int> g_uid{0}, g_open_groups{0}, g_collectives{0};

int GetUniqueId(nccl_uid* u) {
    std::memset(u, 0, sizeof *u);
    std::snprintf(u->internal, sizeof u->internal, "mock-uid-%d", ++g_uid);
    return 0;
}
int CommInitRank(nccl_comm* out, int world, nccl_uid u, int rank) {
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Group*& slot = g_groups[std::string(u.internal)];
        if (!slot) { slot = new Group(); slot->world = world; slot->send.resize(world); slot->recv.resize(world); }
        g = slot;
    }
    if (g->world != world) return 5;
    {
        std::unique_lock<std::mutex> lk(g->mu);                 // ncclCommInitRank returns when every rank of the world has called it
        ++g->joined;
        g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= world; });
    }
    *out = new Comm{g, rank};
    return 0;
}
int CommDestroy(nccl_comm c) { delete static_cast<Comm*>(c); return 0; }
const char* GetErrorString(int) { return "mock RCCL error"; }
int GroupStart() { ++g_open_groups; return 0; }
int GroupEnd() { --g_open_groups; return 0; }
int Broadcast(const void* send, void* recv, size_t count, int dtype, int root, nccl_comm c, hipStream_t st) {
    Comm* cm = static_cast<Comm*>(c);
    const size_t bytes = count * (dtype == NCCL_INT32 ? 4 : 1);
    ++g_collectives;
    mock::S(st)->push([=] {
        Group* g = cm->g;
        g->send[cm->rank] = send;
        g->recv[cm->rank] = recv;
        g->barrier();                                            // everybody is here, the root's buffer is final
        if (cm->rank != root) std::memcpy(recv, g->send[root], bytes);
        g->barrier();                                            // everybody has its copy: the root may go on
    });
    return 0;
}
int AllReduce(const void* send, void* recv, size_t count, int dtype, int op, nccl_comm c, hipStream_t st) {
    Comm* cm = static_cast<Comm*>(c);
    if (dtype != NCCL_INT32 || op != NCCL_MIN) return 4;
    ++g_collectives;
    mock::S(st)->push([=] {
        Group* g = cm->g;
        g->send[cm->rank] = send;
        g->barrier();
        std::vector<int32_t> out(count);
        for (size_t i = 0; i < count; ++i) {
            int32_t m = static_cast<const int32_t*>(g->send[0])[i];
            for (int r = 1; r < g->world; ++r) m = std::min(m, static_cast<const int32_t*>(g->send[r])[i]);
            out[i] = m;
        }
        g->barrier();                                            // all inputs read before anybody overwrites its (in-place) buffer
        std::memcpy(recv, out.data(), count * 4);
        g->barrier();
    });
    return 0;
}

void install() {
    setenv("CSI_RCCL_ONLY", "1", 1);
    setenv("CSI_RCCL_LIBRARY", "/nonexistent/librccl-mock.so", 1);
    RcclApi& a = rccl();                                         // the loader finds nothing ...
    a.why.clear();                                               // ... and gets the model instead
    a.lib = reinterpret_cast<void*>(1);
    a.GetUniqueId = &GetUniqueId;
    a.CommInitRank = &CommInitRank;
    a.CommDestroy = &CommDestroy;
    a.Broadcast = &Broadcast;
    a.AllReduce = &AllReduce;
    a.GroupStart = &GroupStart;
    a.GroupEnd = &GroupEnd;
    a.GetErrorString = &GetErrorString;
}

}  // namespace mnccl

namespace {

struct Shape { int nt, nr, h0, h1, n_out, use_bn, dtype; };

csi_ctx* make_ctx(const Shape& s, int device) {
    csi_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.nt = s.nt; cfg.nr = s.nr; cfg.len_ltf = 320 * s.nt; cfg.n_hidden = s.h1 ? 2 : 1; cfg.hidden[0] = s.h0; cfg.hidden[1] = s.h1;
    cfg.n_out = s.n_out; cfg.use_bn = s.use_bn; cfg.bn_eps = 1e-3f; cfg.dtype = (csi_dtype)s.dtype; cfg.device = device;
    csi_ctx* c = nullptr;
    if (csi_create(&cfg, &c) != CSI_OK) { std::printf("csi_create: %s\n", csi_last_error(nullptr)); std::exit(3); }
    return c;
}

int load(csi_ctx* c, const Shape& s, unsigned seed) {
    std::deque<std::vector<float>> keep;
    std::deque<std::string> names;
    std::vector<csi_tensor> t;
    auto add = [&](const std::string& name, int rows, int cols) {
        keep.emplace_back((size_t)rows * cols);
        std::vector<float>& v = keep.back();
        const float lift = name.find("variance") != std::string::npos || name.find("gamma") != std::string::npos ? 1.f : 0.f;
        for (size_t i = 0; i < v.size(); ++i) v[i] = (float)((i * 2654435761u + seed) % 2003) / 2003.f - 0.5f + lift;
        names.push_back(name);
        t.push_back(csi_tensor{names.back().c_str(), v.data(), rows, cols});
    };
    const int widths[2] = {s.h0, s.h1};
    int in = 321 * s.nt;
    for (int l = 0; l < (s.h1 ? 2 : 1); ++l) {
        add("fc_dense" + std::to_string(l) + ".kernel", in, widths[l]);
        add("fc_dense" + std::to_string(l) + ".bias", 1, widths[l]);
        if (s.use_bn)
            for (const char* k : {".gamma", ".beta", ".moving_mean", ".moving_variance"}) add("bn" + std::to_string(l) + k, 1, widths[l]);
        in = widths[l];
    }
    add("fc_regressor.kernel", in, s.n_out);
    add("fc_regressor.bias", 1, s.n_out);
    for (int d = 0; d < 2; ++d) {
        const int rc = csi_load_weights(c, d, t.data(), (int)t.size());
        if (rc) return rc;
    }
    std::vector<float> P((size_t)s.nt * s.nt);
    for (int j = 0; j < s.nt; ++j)
        for (int q = 0; q < s.nt; ++q) P[(size_t)j * s.nt + q] = (__builtin_popcount(j & q) & 1) ? -1.f : 1.f;
    return csi_set_pilot(c, P.data());
}

// the device buffers a context holds for its models and pilot, as (pointer, bytes), in the protocol's order
std::vector<std::pair<const void*, size_t>> held(csi_ctx* c) {
    WireMeta w;
    wire_fill(c, w);
    std::vector<WBlob> b;
    wire_blobs(c, w, b);
    std::vector<std::pair<const void*, size_t>> out;
    for (WBlob& x : b) out.push_back({*x.p, x.bytes});
    return out;
}

int bad = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++bad; std::printf("FAILED %s:%d: %s  ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// one broadcast over `world` rank threads; shapes[r] is rank r's csi_config; returns the rcs; contexts stay alive in `ctx`
std::vector<int> run_world(int world, int root, const std::vector<Shape>& shapes, bool root_loaded, std::vector<csi_ctx*>& ctx, bool reuse,
                           int oom_rank = -1, long oom_at = 0) {
    std::vector<int> rcs(world, -99);
    char uid[CSI_UNIQUE_ID_BYTES];
    if (!reuse) {
        ctx.assign(world, nullptr);
        EXPECT(csi_get_unique_id(uid) == CSI_OK, "unique id");
    }
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            if (!reuse) {
                ctx[r] = make_ctx(shapes[r], r);
                int rc = csi_comm_init(ctx[r], r, world, uid);
                if (rc) { rcs[r] = rc; return; }
                if (r == root && root_loaded) {
                    rc = load(ctx[r], shapes[r], 17u + (unsigned)r);
                    if (rc) { std::printf("rank %d load: %s\n", r, csi_last_error(ctx[r])); rcs[r] = rc; return; }
                }
            }
            if (r == oom_rank) mock::fail_alloc_in = oom_at;                    // this rank's oom_at-th device allocation inside the call fails
            rcs[r] = csi_broadcast_weights(ctx[r], root);
            mock::fail_alloc_in = -1;
        });
    for (auto& t : th) t.join();
    return rcs;
}

}  // namespace

int main() {
    mnccl::install();
    std::atomic<bool> finished{false};
    std::thread watchdog([&] {
        const char* e = std::getenv("COMM_MOCK_WATCHDOG_S");
        const int limit = 10 * (e && *e ? std::atoi(e) : 180);
        for (int i = 0; i < limit && !finished; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!finished) { std::printf("HANG: a rank waits inside a collective its peers never entered (watchdog, %d s)\n", limit / 10); std::fflush(stdout); std::_Exit(2); }
    });
    const Shape base{8, 2, 64, 128, 234, 1, CSI_DTYPE_F32};
    int scenarios = 0;
    // ---- every receiver ends with the root's bytes
    for (int world : {2, 4, 8})
        for (int root : {0, world - 1}) {
            for (const Shape& s : {base, Shape{16, 2, 128, 0, 52, 0, CSI_DTYPE_F32}, Shape{32, 4, 256, 256, 234, 1, CSI_DTYPE_BF16}}) {
                if (world == 8 && s.nt != 8) continue;
                std::vector<csi_ctx*> ctx;
                const std::vector<int> rcs = run_world(world, root, std::vector<Shape>(world, s), true, ctx, false);
                ++scenarios;
                const auto want = held(ctx[root]);
                EXPECT(!want.empty(), "world %d root %d nt %d: the root holds nothing", world, root, s.nt);
                for (int r = 0; r < world; ++r) {
                    EXPECT(rcs[r] == CSI_OK, "world %d root %d rank %d: rc %d (%s)", world, root, r, rcs[r], csi_last_error(ctx[r]));
                    if (r == root || rcs[r]) continue;
                    const auto got = held(ctx[r]);
                    EXPECT(got.size() == want.size(), "world %d root %d rank %d: %zu buffers, root %zu", world, root, r, got.size(), want.size());
                    for (size_t i = 0; i < got.size() && i < want.size(); ++i) {
                        EXPECT(got[i].second == want[i].second && got[i].first != want[i].first, "world %d rank %d buffer %zu: size / aliasing", world, r, i);
                        EXPECT(!std::memcmp(got[i].first, want[i].first, want[i].second), "world %d root %d rank %d nt %d: buffer %zu of %zu bytes differs from the root's", world, root, r, s.nt, i, want[i].second);
                    }
                    EXPECT(ctx[r]->model[0].loaded && ctx[r]->model[1].loaded && ctx[r]->pilot_ok, "world %d rank %d: flags", world, r);
                    EXPECT(ctx[r]->p_sylvester == ctx[root]->p_sylvester && ctx[r]->p_pieces == ctx[root]->p_pieces, "world %d rank %d: pilot class", world, r);
                    int64_t blobs = 0, bytes = 0;
                    csi_get_option(ctx[r], "comm_blobs", &blobs);
                    csi_get_option(ctx[r], "comm_bytes", &bytes);
                    size_t total = 0;
                    for (auto& w : want) total += w.second;
                    EXPECT(blobs == (int64_t)want.size() && bytes == (int64_t)total, "world %d rank %d: counters %lld blobs %lld bytes", world, r, (long long)blobs, (long long)bytes);
                }
                for (csi_ctx* c : ctx) csi_destroy(c);
            }
        }
    // ---- a receiver built for another network: refused there, reported everywhere, nobody waits, then the same communicators work
    for (int world : {2, 4}) {
        std::vector<Shape> shapes(world, base);
        const int odd = world - 1;
        shapes[odd].h1 = 64;                                          // second hidden layer 64 wide instead of 128
        std::vector<csi_ctx*> ctx;
        std::vector<int> rcs = run_world(world, 0, shapes, true, ctx, false);
        ++scenarios;
        for (int r = 0; r < world; ++r) {
            EXPECT(rcs[r] == CSI_ERR_INVALID_ARG, "refusal, world %d rank %d: rc %d", world, r, rcs[r]);
            const std::string why = csi_last_error(ctx[r]);
            if (r == odd) EXPECT(why.find("hidden layer 1 is 128 wide on the sender, 64 here") != std::string::npos, "refusing rank's text: %s", why.c_str());
            else EXPECT(why.find("another rank refused") != std::string::npos, "rank %d text: %s", r, why.c_str());
            if (r != 0) EXPECT(held(ctx[r]).empty() && !ctx[r]->model[0].loaded && !ctx[r]->pilot_ok, "refusal, rank %d still holds something", r);
        }
        EXPECT(held(ctx[0]).size() > 0 && ctx[0]->model[0].loaded, "the root lost its model");
        EXPECT(mnccl::g_open_groups == 0, "a group was left open");
        // the well-configured ranks again, same communicators: ranks 0 .. world-2 only would leave the odd rank out of a collective
        // of its communicator, so the odd context is replaced by calling the collective with a record it accepts: same world, all ranks
        csi_destroy(ctx[odd]);
        ctx[odd] = nullptr;
        // (a fresh communicator for the repaired world - what a launcher does after fixing the configuration)
        std::vector<csi_ctx*> ctx2;
        rcs = run_world(world, 0, std::vector<Shape>(world, base), true, ctx2, false);
        for (int r = 0; r < world; ++r) EXPECT(rcs[r] == CSI_OK, "after the refusal, world %d rank %d: rc %d", world, r, rcs[r]);
        for (csi_ctx* c : ctx) if (c) csi_destroy(c);
        for (csi_ctx* c : ctx2) csi_destroy(c);
    }
    // ---- a receiver that runs out of device memory in the middle of its allocations: the same agreement, nothing leaks into a later call
    for (long at : {1L, 7L, 19L}) {
        std::vector<csi_ctx*> ctx;
        std::vector<int> rcs = run_world(4, 0, std::vector<Shape>(4, base), true, ctx, false, 2, at);
        ++scenarios;
        for (int r = 0; r < 4; ++r) {
            const std::string why = csi_last_error(ctx[r]);
            if (r == 2) EXPECT(rcs[r] == CSI_ERR_NOMEM && why.find("device allocation") != std::string::npos && held(ctx[r]).empty(), "out of memory at %ld: rank 2 rc %d (%s)", at, rcs[r], why.c_str());
            else EXPECT(rcs[r] == CSI_ERR_INVALID_ARG && why.find("another rank refused") != std::string::npos, "out of memory at %ld: rank %d rc %d (%s)", at, r, rcs[r], why.c_str());
        }
        rcs = run_world(4, 0, std::vector<Shape>(4, base), true, ctx, true);          // the same contexts and communicators, memory back
        const auto want = held(ctx[0]);
        for (int r = 0; r < 4; ++r) {
            EXPECT(rcs[r] == CSI_OK, "after the failed allocation, rank %d: rc %d (%s)", r, rcs[r], csi_last_error(ctx[r]));
            const auto got = held(ctx[r]);
            EXPECT(got.size() == want.size(), "after the failed allocation, rank %d: buffer count", r);
            for (size_t i = 0; i < got.size() && i < want.size(); ++i) EXPECT(!std::memcmp(got[i].first, want[i].first, want[i].second), "after the failed allocation, rank %d buffer %zu", r, i);
        }
        for (csi_ctx* c : ctx) csi_destroy(c);
    }
    // ---- a second broadcast on the SAME communicators (the receivers are replaced wholesale), then a root with nothing loaded
    {
        std::vector<csi_ctx*> ctx;
        std::vector<int> rcs = run_world(4, 1, std::vector<Shape>(4, base), true, ctx, false);
        for (int r = 0; r < 4; ++r) EXPECT(rcs[r] == CSI_OK, "first of two, rank %d: rc %d", r, rcs[r]);
        rcs = run_world(4, 1, std::vector<Shape>(4, base), true, ctx, true);
        ++scenarios;
        const auto want = held(ctx[1]);
        for (int r = 0; r < 4; ++r) {
            EXPECT(rcs[r] == CSI_OK, "second of two, rank %d: rc %d (%s)", r, rcs[r], csi_last_error(ctx[r]));
            const auto got = held(ctx[r]);
            EXPECT(got.size() == want.size(), "second of two, rank %d: buffer count", r);
            for (size_t i = 0; i < got.size() && i < want.size(); ++i) EXPECT(!std::memcmp(got[i].first, want[i].first, want[i].second), "second of two, rank %d buffer %zu", r, i);
        }
        for (csi_ctx* c : ctx) csi_destroy(c);
        std::vector<csi_ctx*> empty;
        rcs = run_world(2, 0, std::vector<Shape>(2, base), false, empty, false);
        ++scenarios;
        for (int r = 0; r < 2; ++r) EXPECT(rcs[r] == CSI_OK && held(empty[r]).empty() && !empty[r]->model[0].loaded, "empty root, rank %d: rc %d", r, rcs[r]);
        for (csi_ctx* c : empty) csi_destroy(c);
    }
    // ---- the in-process transport of the same protocol, ACROSS devices (hipMemcpyPeerAsync): csi_clone_weights(dst on device 3, src on device 0)
    {
        csi_ctx *src = make_ctx(base, 0), *dst = make_ctx(base, 3), *other = make_ctx(Shape{8, 2, 64, 128, 52, 1, CSI_DTYPE_F32}, 1);
        EXPECT(load(src, base, 5u) == CSI_OK, "clone: load (%s)", csi_last_error(src));
        const long peer_before = mock::g_peer_copies.load();
        EXPECT(csi_clone_weights(dst, src) == CSI_OK, "clone across devices: %s", csi_last_error(dst));
        ++scenarios;
        const auto want = held(src), got = held(dst);
        EXPECT(got.size() == want.size() && !want.empty() && mock::g_peer_copies.load() - peer_before == (long)want.size(), "clone: %zu buffers, source %zu, %ld peer copies", got.size(), want.size(), mock::g_peer_copies.load() - peer_before);
        for (size_t i = 0; i < got.size() && i < want.size(); ++i) EXPECT(got[i].first != want[i].first && !std::memcmp(got[i].first, want[i].first, want[i].second), "clone: buffer %zu", i);
        EXPECT(csi_clone_weights(other, src) == CSI_ERR_INVALID_ARG && std::string(csi_last_error(other)).find("n_out 234") != std::string::npos && held(other).empty(), "clone into another network: %s", csi_last_error(other));
        EXPECT(csi_clone_weights(dst, dst) == CSI_ERR_INVALID_ARG && csi_clone_weights(dst, nullptr) == CSI_ERR_INVALID_ARG, "clone: argument checks");
        EXPECT(held(dst).size() == want.size(), "a refused argument emptied the destination");
        csi_destroy(src); csi_destroy(dst); csi_destroy(other);
    }
    // ---- arguments
    {
        csi_ctx* c = make_ctx(base, 0);
        EXPECT(csi_broadcast_weights(c, 0) == CSI_ERR_NOT_READY, "broadcast without a communicator");
        char uid[CSI_UNIQUE_ID_BYTES];
        csi_get_unique_id(uid);
        EXPECT(csi_comm_init(c, 2, 2, uid) == CSI_ERR_INVALID_ARG && csi_comm_init(c, 0, 0, uid) == CSI_ERR_INVALID_ARG, "rank / world checks");
        EXPECT(csi_comm_init(c, 0, 1, uid) == CSI_OK && csi_broadcast_weights(c, 1) == CSI_ERR_INVALID_ARG, "root outside the world");
        EXPECT(csi_broadcast_weights(c, 0) == CSI_OK, "world of one");
        csi_destroy(c);
    }
    finished = true;
    watchdog.join();
    for (auto& kv : mnccl::g_groups) delete kv.second;              // (the model's own rendezvous objects: a leak check then shows the library's leaks only)
    mnccl::g_groups.clear();
    std::printf("%d broadcast scenarios, %d collectives on the RCCL model, %ld copies and %ld dropped kernel launches on the HIP model\n", scenarios, mnccl::g_collectives.load(),
                mock::g_copies.load(), mock::g_launches.load());
    std::printf(bad ? "FAILED (%d)\n" : "comm_mock_check: ok\n", bad);
    return bad ? 1 : 0;
}
#else
int main() { return 0; }
#endif
