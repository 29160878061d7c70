// csi_comm.hpp - the ONE collective of the path, inside the C-ABI: the load-time broadcast of the shared weights over RCCL
// (SURVEY.md 5 "Distributed communication backend", 8e: packets shard over the GPUs of a node, weights are read-only and
// shared).  csi_broadcast_weights sends what csi_load_weights / csi_set_pilot left on the ROOT's device - the re-laid-out
// fp32 matrices, their split-f16 (hi | lo) forms, bias / BatchNormalization vectors, the pilot rows of layer 0, P - device to
// device with ncclBroadcast on the context's stream: no host detour, no torch in the data path.  The few host-side scalars that
// go with them (leading dimensions, power-of-two shifts of the split engine, which buffers exist) travel first as one
// fixed-size record through the same communicator.  The pilot tables T / T_hs are rebuilt on every rank from what arrived.
//
// The protocol is written once and has two transports.  wire_fill (root: context -> record), wire_receive (receiver: check
// the record against the own csi_config, drop the old model, take the scalars, allocate), wire_blobs (the device buffers
// the record names, in one fixed order), wire_finish (receiver: flags, LS kernel attributes, pilot tables) are what BOTH
//   csi_broadcast_weights  - RCCL: record and buffers by ncclBroadcast, one ncclAllReduce(min) of a status word between
//                            them so that a rank that refuses the record takes every rank out BEFORE the grouped broadcast
//   csi_clone_weights      - one process: record by value, buffers by hipMemcpy[Peer]Async from the source context
// run.  The second one exists so that a single-GPU box executes every line of the receiver side (tests/test_gpu_*.py).
//
// RCCL is loaded with dlopen at csi_comm_init (librccl.so.1, the library `torch.distributed` backend "nccl" wraps on ROCm):
// a single-GPU user of libcsi_mamimo.so needs no RCCL installed, and a process that already carries torch's copy shares it.
#pragma once
#include <dlfcn.h>

#include "csi_context.hpp"

namespace {

struct nccl_uid { char internal[CSI_UNIQUE_ID_BYTES]; };
typedef void* nccl_comm;
enum { NCCL_CHAR = 0, NCCL_INT32 = 2, NCCL_MIN = 3 };

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

RcclApi& rccl() {
    static RcclApi api;
    if (api.lib || !api.why.empty()) return api;
    // CSI_RCCL_LIBRARY names the library to try first; CSI_RCCL_ONLY=1 makes it the ONLY candidate (tests: a host without RCCL)
    const char* only = getenv("CSI_RCCL_ONLY");
    const bool restricted = only && *only && *only != '0';
    const char* names[] = {getenv("CSI_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string last;
    for (size_t i = 0; i < (restricted ? (size_t)1 : sizeof names / sizeof names[0]); ++i) {
        const char* n = names[i];
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
        const char* e = dlerror();          // ONE call: dlerror() clears the message it returns
        last = e ? e : "";
    }
    if (!api.lib) {
        api.why = std::string("RCCL not found (librccl.so.1; set CSI_RCCL_LIBRARY): ") + last;
        return api;
    }
    auto sym = [&](const char* s) { void* p = dlsym(api.lib, s); if (!p) api.why = std::string("RCCL lacks ") + s; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
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
    // the sender's csi_config, as far as it shapes the buffers: a receiver built for anything else refuses the record
    int32_t cfg_nt, cfg_len_ltf, cfg_n_out, cfg_dtype, cfg_use_bn, cfg_hidden[CSI_MAX_HIDDEN];
    int32_t loaded[2], has_W0p[2], has_W0rm[2], hs_repr_ok[2];
    double hs_repr_err[2];                             // the sender's load-time measurement behind hs_repr_ok ("hs_weight_err_e12" reads the same on every rank)
    int32_t pilot_ok, p_sylvester, p_pieces;
    int32_t p_fast_ok, p_perm[2][CSI_WIRE_MAX_NT];     // Hadamard-equivalent pilot: output row / symbol permutation with signs (csi_set_pilot)
    WireLayer layer[2][CSI_MAX_HIDDEN + 1];
};
constexpr int32_t WIRE_MAGIC = 0x43534932;      // "CSI2"
constexpr size_t WIRE_STATUS_OFF = (sizeof(WireMeta) + 63) / 64 * 64;     // the ranks' status word sits behind the record in csi_comm::wire

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

int ls_prepare(csi_ctx* c);                     // csi_mamimo.hip
int pilot_fast_tables(csi_ctx* c);              // csi_mamimo.hip: device tables of a Hadamard-equivalent pilot from c->p_perm

// ---- the protocol, transport-independent -------------------------------------------------------------------------------------
// sender: what the context holds -> record
void wire_fill(const csi_ctx* c, WireMeta& w) {
    const csi_config& cf = c->cfg;
    std::memset(&w, 0, sizeof w);
    w.magic = WIRE_MAGIC;
    w.n_layers = cf.n_hidden + 1;
    w.cfg_nt = cf.nt; w.cfg_len_ltf = cf.len_ltf; w.cfg_n_out = cf.n_out; w.cfg_dtype = cf.dtype; w.cfg_use_bn = cf.use_bn != 0;
    for (int i = 0; i < cf.n_hidden; ++i) w.cfg_hidden[i] = cf.hidden[i];
    w.pilot_ok = c->pilot_ok;
    w.p_sylvester = c->p_sylvester;
    w.p_pieces = c->p_pieces;
    w.p_fast_ok = c->p_fast_ok;
    if (c->p_fast_ok)
        for (int k = 0; k < 2; ++k)
            for (int i = 0; i < cf.nt && i < CSI_WIRE_MAX_NT; ++i) w.p_perm[k][i] = c->p_perm[k][i];
    for (int d = 0; d < 2; ++d) {
        const Model& m = c->model[d];
        w.loaded[d] = m.loaded && (int)m.layers.size() == cf.n_hidden + 1;
        if (!w.loaded[d]) continue;
        w.hs_repr_ok[d] = m.hs_repr_ok;
        w.hs_repr_err[d] = m.hs_repr_err;
        w.has_W0p[d] = m.W0p != nullptr;
        w.has_W0rm[d] = m.W0rm != nullptr;
        for (int i = 0; i <= cf.n_hidden; ++i) {
            const Layer& L = m.layers[i];
            WireLayer& x = w.layer[d][i];
            x.in = L.in; x.out = L.out; x.ldw = L.ldw; x.ldwb = L.ldwb; x.ldwh = L.ldwh;
            x.wshift = L.wshift; x.wshift_f = L.wshift_f; x.ashift = L.ashift; x.ashift_pre = L.ashift_pre;
            x.has_Wt = L.Wt != nullptr; x.has_Wb = L.Wb != nullptr; x.has_Wb_p = L.Wb_p != nullptr; x.has_Wh = L.Wh != nullptr; x.has_Wh_f = L.Wh_f != nullptr;
            x.has_Wh_p = L.Wh_p != nullptr; x.has_bias = L.bias != nullptr; x.has_bias_hs = L.bias_hs != nullptr;
            x.has_scale = L.scale != nullptr; x.has_shift = L.shift != nullptr;
        }
    }
}

// the device buffers the record names, in the one order both sides walk (models, then P / Ppad / Pbf)
void wire_blobs(csi_ctx* c, const WireMeta& w, std::vector<WBlob>& blobs) {
    const csi_config& cf = c->cfg;
    for (int d = 0; d < 2; ++d)
        if (w.loaded[d]) model_blobs(cf, c->model[d], w.layer[d], w.has_W0p[d], w.has_W0rm[d], blobs);
    if (w.pilot_ok && cf.nt > 0) {
        const size_t ldp = (size_t)(cf.nt + 31) / 32 * 32, slack = G_SLACK_FLOATS * sizeof(float);
        blobs.push_back({(void**)&c->P, (size_t)cf.nt * cf.nt * 4 + slack});
        blobs.push_back({(void**)&c->Ppad, ldp * ldp * 4 + slack});
        blobs.push_back({(void**)&c->Pbf, (size_t)((cf.nt + 15) / 16) * 3 * ((cf.nt + 31) / 32) * LSB_BLOCK * 2 + slack});
    }
}

// a receiver that could not take the record holds nothing afterwards (never half a model with stale flags)
void wire_drop_receiver(csi_ctx* c) {
    for (int d = 0; d < 2; ++d) free_model(c->model[d]);
    for (float** p : {&c->P, &c->Ppad, &c->Pbf})
        if (*p) { hipFree(*p); *p = nullptr; }
    if (c->p_tables) { hipFree(c->p_tables); c->p_tables = nullptr; }
    c->pilot_ok = false;
    c->p_sylvester = false;
    c->p_fast_ok = false;
    c->p_fast_identity = true;
}

// receiver, step 1: the record against the own csi_config, then the old model out, the scalars in, buffers of the sender's sizes.
// Any refusal leaves the context empty (wire_drop_receiver) and the reason in csi_last_error.
int wire_receive(csi_ctx* c, const WireMeta& w, std::vector<WBlob>& blobs, const char* fn) {
    const csi_config& cf = c->cfg;
    auto refuse = [&](int code) { wire_drop_receiver(c); return code; };
    if (w.magic != WIRE_MAGIC)
        return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: the sender's record is not a CSI2 record (magic %08x): libraries of different versions?", fn, (unsigned)w.magic));
    if (w.n_layers != cf.n_hidden + 1)
        return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: the sender has %d hidden layers, this context %d", fn, w.n_layers - 1, cf.n_hidden));
    if (w.cfg_nt != cf.nt || w.cfg_len_ltf != cf.len_ltf || w.cfg_n_out != cf.n_out || w.cfg_dtype != cf.dtype || w.cfg_use_bn != (cf.use_bn != 0))
        return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: csi_config differs - sender nt %d len_ltf %d n_out %d dtype %d use_bn %d, here nt %d len_ltf %d n_out %d dtype %d use_bn %d",
                           fn, w.cfg_nt, w.cfg_len_ltf, w.cfg_n_out, w.cfg_dtype, w.cfg_use_bn, cf.nt, cf.len_ltf, cf.n_out, cf.dtype, cf.use_bn != 0));
    for (int i = 0; i < cf.n_hidden; ++i)
        if (w.cfg_hidden[i] != cf.hidden[i])
            return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: hidden layer %d is %d wide on the sender, %d here", fn, i, w.cfg_hidden[i], cf.hidden[i]));
    if (w.p_fast_ok && cf.nt > CSI_WIRE_MAX_NT)
        return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: nt %d exceeds the record's permutation tables", fn, cf.nt));
    wire_drop_receiver(c);
    for (int d = 0; d < 2; ++d) {
        if (!w.loaded[d]) continue;
        Model& m = c->model[d];
        m.layers.resize(cf.n_hidden + 1);
        for (int i = 0; i <= cf.n_hidden; ++i) {
            Layer& L = m.layers[i];
            const WireLayer& x = w.layer[d][i];
            const int want_out = i == cf.n_hidden ? cf.n_out : cf.hidden[i];
            if (x.out != want_out || x.in <= 0 || x.ldw < 0 || x.ldwb < 0 || x.ldwh < 0)
                return refuse(fail(c, CSI_ERR_INVALID_ARG, "%s: layer %d of the %s model is [%d -> %d] on the sender, %d wide here", fn, i, d ? "imag" : "real", x.in, x.out, want_out));
            L.in = x.in; L.out = x.out; L.ldw = x.ldw; L.ldwb = x.ldwb; L.ldwh = x.ldwh;
            L.wshift = x.wshift; L.wshift_f = x.wshift_f; L.ashift = x.ashift; L.ashift_pre = x.ashift_pre;
        }
    }
    wire_blobs(c, w, blobs);
    for (WBlob& b : blobs)
        if (hipMalloc(b.p, b.bytes) != hipSuccess) {
            *b.p = nullptr;
            (void)hipGetLastError();
            return refuse(fail(c, CSI_ERR_NOMEM, "%s: device allocation of %zu bytes failed", fn, b.bytes));
        }
    return CSI_OK;
}

// receiver, last step (the buffers hold the sender's bytes): flags, LS kernel attributes, what is derived locally
int wire_finish(csi_ctx* c, const WireMeta& w) {
    const csi_config& cf = c->cfg;
    c->pilot_ok = w.pilot_ok != 0;
    c->p_sylvester = w.p_sylvester != 0;
    c->p_pieces = w.p_pieces;
    c->p_fast_ok = w.p_fast_ok != 0;
    if (c->p_fast_ok) {
        for (int k = 0; k < 2; ++k)
            for (int i = 0; i < cf.nt; ++i) c->p_perm[k][i] = w.p_perm[k][i];
        int rc = pilot_fast_tables(c);
        if (rc) return rc;
    }
    if (c->pilot_ok) {
        int rc = ls_prepare(c);
        if (rc) return rc;
    }
    for (int d = 0; d < 2; ++d) {
        Model& m = c->model[d];
        m.loaded = w.loaded[d] != 0;
        m.hs_repr_ok = !m.loaded || w.hs_repr_ok[d] != 0;
        m.hs_repr_err = m.loaded ? w.hs_repr_err[d] : 0.0;
        if (m.loaded && !m.hs_repr_ok) ++c->hs_weight_pins;      // a receiver counts a pinned model like the context that loaded it
        m.table_ok = false;
        if (m.loaded && c->pilot_ok) {
            int rc = build_pilot_table(c, m);
            if (rc) return rc;
        }
    }
    return CSI_OK;
}

}  // namespace

struct csi_comm {
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    void* wire = nullptr;           // device staging of the WireMeta record, followed by the status word of the ranks' agreement
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
