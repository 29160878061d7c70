// csi_comm.hpp - the ONE collective of the path, inside the C-ABI: the load-time broadcast of the shared weights over RCCL
// (SURVEY.md 5 "Distributed communication backend", 8e: packets shard over the GPUs of a node, weights are read-only and
// shared).  csi_broadcast_weights sends what csi_load_weights / csi_set_pilot left on the ROOT's device - the re-laid-out
// fp32 matrices, their split-f16 (hi | lo) forms, bias / BatchNormalization vectors, the pilot rows of layer 0, P - device to
// device with ncclBroadcast on the context's stream: no host detour, no torch in the data path.  The few host-side scalars that
// go with them (leading dimensions, power-of-two shifts of the split engine, which buffers exist) travel first as one
// fixed-size record through the same communicator.  The pilot tables T / T_hs are rebuilt on every rank from what arrived.
//
// RCCL is loaded with dlopen at csi_comm_init (librccl.so.1, the library `torch.distributed` backend "nccl" wraps on ROCm):
// a single-GPU user of libcsi_mamimo.so needs no RCCL installed, and a process that already carries torch's copy shares it.
#pragma once
#include <dlfcn.h>

#include "csi_context.hpp"

namespace {

struct nccl_uid { char internal[CSI_UNIQUE_ID_BYTES]; };
typedef void* nccl_comm;
enum { NCCL_CHAR = 0 };

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

RcclApi& rccl() {
    static RcclApi api;
    if (api.lib || !api.why.empty()) return api;
    const char* names[] = {getenv("CSI_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) {
        api.why = std::string("RCCL not found (librccl.so.1; set CSI_RCCL_LIBRARY): ") + (dlerror() ? dlerror() : "");
        return api;
    }
    auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) api.why = std::string("RCCL lacks ") + s; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!api.why.empty()) { dlclose(api.lib); api.lib = nullptr; }
    return api;
}

#define NCCL_TRY(ctx, expr)                                                                                   \
    do {                                                                                                      \
        int r_ = (expr);                                                                                      \
        if (r_ != 0)                                                                                          \
            return fail(ctx, CSI_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
    } while (0)

// host-side scalars of the device blobs, one record for both component models
struct WireLayer {
    int32_t in, out, ldw, ldwb, ldwh, wshift, wshift_f, ashift, ashift_pre;
    int32_t has_Wt, has_Wb, has_Wb_p, has_Wh, has_Wh_f, has_Wh_p, has_bias, has_bias_hs, has_scale, has_shift;
};
struct WireMeta {
    int32_t magic, n_layers;
    int32_t loaded[2], has_W0p[2], has_W0rm[2];
    int32_t pilot_ok, p_sylvester, p_pieces;
    WireLayer layer[2][CSI_MAX_HIDDEN + 1];
};
constexpr int32_t WIRE_MAGIC = 0x43534931;      // "CSI1"

struct WBlob {
    void** p;
    size_t bytes;
};

// every device buffer of a component model that csi_load_weights fills, with the size it allocates (same formulas)
void model_blobs(const csi_config& cf, Model& m, const WireLayer* wl, const int32_t has_W0p, const int32_t has_W0rm, std::vector<WBlob>& v) {
    const size_t slack = G_SLACK_FLOATS * sizeof(float);
    for (size_t i = 0; i < m.layers.size(); ++i) {
        Layer& L = m.layers[i];
        const WireLayer& w = wl[i];
        if (w.has_Wt) v.push_back({(void**)&L.Wt, (size_t)L.out * L.ldw * 4 + slack});
        if (w.has_Wb) v.push_back({(void**)&L.Wb, (size_t)L.out * L.ldwb * 2 + 256});
        if (w.has_Wb_p) v.push_back({(void**)&L.Wb_p, (size_t)256 * L.ldwb * 2 + 256});
        if (w.has_Wh) v.push_back({(void**)&L.Wh, (size_t)L.out * L.ldwh * 2 + 4096});
        if (w.has_Wh_f) v.push_back({(void**)&L.Wh_f, (size_t)L.out * L.ldwh * 2 + 4096});
        if (w.has_Wh_p) v.push_back({(void**)&L.Wh_p, (size_t)256 * L.ldwh * 2 + 4096});
        if (w.has_bias) v.push_back({(void**)&L.bias, (size_t)L.out * 4 + slack});
        if (w.has_bias_hs) v.push_back({(void**)&L.bias_hs, (size_t)L.out * 4 + slack});
        if (w.has_scale) v.push_back({(void**)&L.scale, (size_t)L.out * 4 + slack});
        if (w.has_shift) v.push_back({(void**)&L.shift, (size_t)L.out * 4 + slack});
    }
    const size_t h1 = m.layers.empty() ? 0 : (size_t)m.layers[0].out;
    if (has_W0p) v.push_back({(void**)&m.W0p, (size_t)cf.nt * h1 * 4 + slack});
    if (has_W0rm) v.push_back({(void**)&m.W0rm, (size_t)cf.len_ltf * h1 * 4 + slack});
}

}  // namespace

struct csi_comm {
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    void* wire = nullptr;           // device staging of the WireMeta record
    int64_t bytes_broadcast = 0;    // of the last csi_broadcast_weights
    int64_t blobs_broadcast = 0;
};

namespace {

void comm_free(csi_ctx* c) {
    if (!c->comm) return;
    if (c->comm->comm && rccl().CommDestroy) rccl().CommDestroy(c->comm->comm);
    if (c->comm->wire) hipFree(c->comm->wire);
    delete c->comm;
    c->comm = nullptr;
}

}  // namespace
