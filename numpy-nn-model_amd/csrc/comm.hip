// comm.hip -- the data-parallel exchange behind the C ABI: thin status-code wrappers over RCCL (SURVEY 8b "add AllReduce*",
// 8e: one SUM all-reduce of the flat gradient bucket per step, over xGMI).
//
// The reference is single-device (neunet/autograd.py:8-14: `device` is "cpu" | "cuda"; nothing in it names a collective), so
// these entry points replace nothing -- they are what a reference-side binder (CuPy arrays through
// neunet/nn/experimental/utils.py:64-92, no torch in the process) needs for the one exchange step the path has.
//
// librccl is NOT a link-time dependency of libneunet_hip.so: it is bound with dlopen on the first communicator call, and a
// copy that is already mapped into the process wins (a host that also imports torch has torch's own librccl.so.1 loaded --
// two RCCL instances in one process would each bootstrap and each own channels/IPC handles).  Every kernel entry point of
// the library therefore keeps working on a box without RCCL; only these calls fail, with NNHIP_ECOMM and a message.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>) && !defined(NNHIP_FORCE_NO_RCCL_HEADERS)
#include <rccl/rccl.h>   // types and prototypes only (decltype below); no symbol of it is linked
#define NNHIP_HAVE_RCCL_HEADERS 1
#else
#define NNHIP_HAVE_RCCL_HEADERS 0   // a box without the RCCL headers still builds the library: the nnhipComm* entry points answer NNHIP_ECOMM
#endif
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

#if NNHIP_HAVE_RCCL_HEADERS
struct nnhipComm {
    ncclComm_t comm;
    int rank, world;
};

namespace nnhip {
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    char path[256] = "";
};

Rccl g_rccl;
std::mutex g_rccl_mu;

template <class F>
bool bind(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

// 0 or NNHIP_ECOMM (last-error says which library / symbol was missing)
int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return 0;
    const char* env = getenv("NNHIP_RCCL_LIB");
    const char* tries[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    const char* got = nullptr;
    // a copy that is already mapped (by soname) first: share the host process's RCCL instead of loading a second one
    for (const char* n : {"librccl.so.1", "librccl.so"}) {
        if (env) break;
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) { got = n; break; }
    }
    for (size_t i = 0; !h && i < sizeof(tries) / sizeof(tries[0]); ++i) {
        if (!tries[i] || !*tries[i]) continue;
        h = dlopen(tries[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) got = tries[i];
    }
    if (!h) {
        set_last_error("RCCL is not available: dlopen(librccl.so.1) failed (%s); set NNHIP_RCCL_LIB to its path", dlerror());
        return NNHIP_ECOMM;
    }
    Rccl r;
    r.handle = h;
    snprintf(r.path, sizeof(r.path), "%s", got);
    if (!bind(h, "ncclGetUniqueId", r.GetUniqueId) || !bind(h, "ncclCommInitRank", r.CommInitRank) ||
        !bind(h, "ncclCommDestroy", r.CommDestroy) || !bind(h, "ncclAllReduce", r.AllReduce) ||
        !bind(h, "ncclBroadcast", r.Broadcast) || !bind(h, "ncclGetErrorString", r.GetErrorString)) {
        set_last_error("%s does not export the RCCL entry points (ncclGetUniqueId / ncclCommInitRank / ncclAllReduce ...)", got);
        dlclose(h);
        return NNHIP_ECOMM;
    }
    bind(h, "ncclGetVersion", r.GetVersion);   // optional
    g_rccl = r;
    return 0;
}

int rccl_status(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    set_last_error("%s: %s (ncclResult_t %d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", (int)r);
    return NNHIP_ECOMM;
}

int reduce_f32(nnhipComm_t c, float* buf, int64_t n, ncclRedOp_t op, void* stream, const char* what) {
    NNHIP_CHECK_ARG(c && c->comm, NNHIP_EINVAL, "%s: null communicator", what);
    NNHIP_CHECK_ARG(n >= 0, NNHIP_EINVAL, "%s: negative count", what);
    if (n == 0) return 0;
    NNHIP_CHECK_ARG(buf, NNHIP_EINVAL, "%s: null buffer", what);
    NNHIP_CHECK_ARG(aligned4(buf), NNHIP_EALIGN, "%s: buffer not 4-byte aligned", what);
    return rccl_status(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, op, c->comm, static_cast<hipStream_t>(stream)), what);
}

}  // namespace
}  // namespace nnhip

extern "C" int nnhipCommUniqueId(void* id) {
    NNHIP_CHECK_ARG(id, NNHIP_EINVAL, "nnhipCommUniqueId: null id buffer");
    static_assert(sizeof(ncclUniqueId) == NNHIP_UNIQUE_ID_BYTES, "NNHIP_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");
    if (int rc = nnhip::load_rccl()) return rc;
    ncclUniqueId u;
    if (int rc = nnhip::rccl_status(nnhip::g_rccl.GetUniqueId(&u), "ncclGetUniqueId")) return rc;
    memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int nnhipCommInitRank(nnhipComm_t* comm, const void* id, int rank, int world) {
    NNHIP_CHECK_ARG(comm && id, NNHIP_EINVAL, "nnhipCommInitRank: null argument");
    NNHIP_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, NNHIP_EINVAL, "nnhipCommInitRank: rank %d of %d", rank, world);
    if (int rc = nnhip::load_rccl()) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    // binds the communicator to the calling thread's current device (one process per GPU: hipSetDevice(LOCAL_RANK) first)
    if (int rc = nnhip::rccl_status(nnhip::g_rccl.CommInitRank(&c, world, u, rank), "ncclCommInitRank")) return rc;
    nnhipComm* h = new (std::nothrow) nnhipComm{c, rank, world};
    if (!h) {
        nnhip::g_rccl.CommDestroy(c);
        nnhip::set_last_error("nnhipCommInitRank: out of host memory");
        return NNHIP_ENOMEM;
    }
    *comm = h;
    return 0;
}

extern "C" int nnhipCommDestroy(nnhipComm_t comm) {
    if (!comm) return 0;
    int rc = 0;
    if (comm->comm) rc = nnhip::rccl_status(nnhip::g_rccl.CommDestroy(comm->comm), "ncclCommDestroy");
    delete comm;
    return rc;
}

extern "C" int nnhipCommRank(nnhipComm_t comm, int* rank, int* world) {
    NNHIP_CHECK_ARG(comm, NNHIP_EINVAL, "nnhipCommRank: null communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return 0;
}

extern "C" int nnhipCommLibrary(char* path, int64_t path_bytes, int* version) {
    if (int rc = nnhip::load_rccl()) return rc;
    if (path && path_bytes > 0) snprintf(path, (size_t)path_bytes, "%s", nnhip::g_rccl.path);
    if (version) {
        int v = 0;
        if (nnhip::g_rccl.GetVersion) nnhip::g_rccl.GetVersion(&v);
        *version = v;
    }
    return 0;
}

extern "C" int nnhipAllReduceSumF32(nnhipComm_t comm, float* buf, int64_t n, nnhipStream_t stream) {
    return nnhip::reduce_f32(comm, buf, n, ncclSum, stream, "nnhipAllReduceSumF32");
}

extern "C" int nnhipAllReduceAvgF32(nnhipComm_t comm, float* buf, int64_t n, nnhipStream_t stream) {
    return nnhip::reduce_f32(comm, buf, n, ncclAvg, stream, "nnhipAllReduceAvgF32");
}

extern "C" int nnhipBroadcastF32(nnhipComm_t comm, float* buf, int64_t n, int root, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(comm && comm->comm, NNHIP_EINVAL, "nnhipBroadcastF32: null communicator");
    NNHIP_CHECK_ARG(n >= 0 && root >= 0 && root < comm->world, NNHIP_EINVAL, "nnhipBroadcastF32: bad count / root");
    if (n == 0) return 0;
    NNHIP_CHECK_ARG(buf && nnhip::aligned4(buf), NNHIP_EINVAL, "nnhipBroadcastF32: null or misaligned buffer");
    return nnhip::rccl_status(nnhip::g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, comm->comm,
                                                      static_cast<hipStream_t>(stream)), "nnhipBroadcastF32");
}
#else   // no RCCL headers on the build box: the entry points exist (the ABI is the same) and say why they cannot work
#define NNHIP_NO_RCCL(name) do { nnhip::set_last_error(name ": libneunet_hip.so was built without the RCCL headers"); return NNHIP_ECOMM; } while (0)
extern "C" int nnhipCommUniqueId(void*) { NNHIP_NO_RCCL("nnhipCommUniqueId"); }
extern "C" int nnhipCommInitRank(nnhipComm_t*, const void*, int, int) { NNHIP_NO_RCCL("nnhipCommInitRank"); }
extern "C" int nnhipCommDestroy(nnhipComm_t) { NNHIP_NO_RCCL("nnhipCommDestroy"); }
extern "C" int nnhipCommRank(nnhipComm_t, int*, int*) { NNHIP_NO_RCCL("nnhipCommRank"); }
extern "C" int nnhipCommLibrary(char*, int64_t, int*) { NNHIP_NO_RCCL("nnhipCommLibrary"); }
extern "C" int nnhipAllReduceSumF32(nnhipComm_t, float*, int64_t, nnhipStream_t) { NNHIP_NO_RCCL("nnhipAllReduceSumF32"); }
extern "C" int nnhipAllReduceAvgF32(nnhipComm_t, float*, int64_t, nnhipStream_t) { NNHIP_NO_RCCL("nnhipAllReduceAvgF32"); }
extern "C" int nnhipBroadcastF32(nnhipComm_t, float*, int64_t, int, nnhipStream_t) { NNHIP_NO_RCCL("nnhipBroadcastF32"); }
#endif
