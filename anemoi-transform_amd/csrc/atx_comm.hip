// Collective entry points of the C ABI (atx_comm_init / atx_bcast): RCCL over xGMI, one process per GPU.
//
// The regrid path shards on TARGET POINTS with no collective in the data path; the only exchange is
// the SOURCE stack reaching every rank once.  Three shapes of that exchange are exported:
//   atx_bcast          — the whole pitched stack from the rank that holds it (one ncclBroadcast)
//   atx_all_gather     — every rank's stack onto every rank in ONE collective (ncclAllGather): the N-broadcast exchange of a job in
//                        which every rank contributes a stack, with every xGMI link busy at once
//   atx_exchange       — band-limited: every rank sends each peer only the slab of source columns that peer's target
//                        slice references and receives its own slabs (grouped ncclSend / ncclRecv)
//   atx_gather_shards  — the per-rank target slices assembled on every rank for callers that want the full field
//                        (grouped ncclBroadcast of contiguous byte ranges; the slices differ in size because the
//                        shard boundaries are balanced by traffic, not by count)
// The reference has no counterpart (single-process loop, R: filters/fields/regrid.py:204-208).
//
// RCCL is bound at first use with dlopen, not at link time: libatx.so keeps loading on boxes without RCCL, and in a
// process that already holds an RCCL (PyTorch ships one) the same copy is reused by SONAME instead of a second one
// being mapped.
#include "atx_common.hpp"
#include "atx_nccl_abi.h"  // the NCCL / RCCL public C types and signatures bound below (shared with the test stand-in)

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace atx {

constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar
static_assert(ATX_NCCL_UNIQUE_ID_BYTES == ATX_COMM_ID_BYTES, "include/atx.h and the NCCL ABI disagree on the unique id size");

struct Rccl {
    void* handle = nullptr;
    atx_ncclGetVersion_t GetVersion = nullptr;
    atx_ncclGetUniqueId_t GetUniqueId = nullptr;
    atx_ncclCommInitRank_t CommInitRank = nullptr;
    atx_ncclCommDestroy_t CommDestroy = nullptr;
    atx_ncclGetErrorString_t GetErrorString = nullptr;
    atx_ncclBroadcast_t Broadcast = nullptr;
    atx_ncclAllGather_t AllGather = nullptr;
    atx_ncclSend_t Send = nullptr;
    atx_ncclRecv_t Recv = nullptr;
    atx_ncclGroup_t GroupStart = nullptr;
    atx_ncclGroup_t GroupEnd = nullptr;
    char error[256] = {0};
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;

template <typename F>
static bool bind(void* handle, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(handle, name));
    return fn != nullptr;
}

static void load_rccl() {
    const char* override_path = std::getenv("ATX_RCCL_LIBRARY");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
        if (!name || !*name) continue;
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        snprintf(g_rccl.error, sizeof(g_rccl.error), "RCCL not found (librccl.so.1; set ATX_RCCL_LIBRARY): %s", dlerror());
        return;
    }
    Rccl& r = g_rccl;
    const bool ok = bind(r.handle, "ncclGetVersion", r.GetVersion) && bind(r.handle, "ncclGetUniqueId", r.GetUniqueId) &&
                    bind(r.handle, "ncclCommInitRank", r.CommInitRank) && bind(r.handle, "ncclCommDestroy", r.CommDestroy) &&
                    bind(r.handle, "ncclGetErrorString", r.GetErrorString) && bind(r.handle, "ncclBroadcast", r.Broadcast) &&
                    bind(r.handle, "ncclAllGather", r.AllGather) &&
                    bind(r.handle, "ncclSend", r.Send) && bind(r.handle, "ncclRecv", r.Recv) &&
                    bind(r.handle, "ncclGroupStart", r.GroupStart) && bind(r.handle, "ncclGroupEnd", r.GroupEnd);
    if (!ok) {
        snprintf(g_rccl.error, sizeof(g_rccl.error), "the RCCL library lacks a required symbol: %s", dlerror());
        dlclose(r.handle);
        r.handle = nullptr;
    }
}

static Rccl* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

static int comm_status(ncclResult_t e, const char* what) {
    if (e == 0) return ATX_OK;
    set_error("%s: RCCL error %d (%s)", what, (int)e, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
    return ATX_ECOMM;
}

#define ATX_RCCL_OR_FAIL(r)                                   \
    Rccl* r = atx::rccl();                                    \
    do {                                                      \
        if (!r) {                                             \
            atx::set_error("%s", atx::g_rccl.error);          \
            return ATX_ECOMM;                                 \
        }                                                     \
    } while (0)

}  // namespace atx

struct atx_comm {
    ncclComm_t comm;
    int32_t world;
    int32_t rank;
    int device;
};

using namespace atx;

extern "C" int atx_comm_version(void) {
    ATX_RCCL_OR_FAIL(r);
    int v = 0;
    const int st = comm_status(r->GetVersion(&v), "ncclGetVersion");
    return st == ATX_OK ? v : st;
}

extern "C" int atx_comm_unique_id(void* id) {
    ATX_REQUIRE(id, ATX_EINVAL, "atx_comm_unique_id: null id buffer");
    ATX_RCCL_OR_FAIL(r);
    ncclUniqueId uid;
    const int st = comm_status(r->GetUniqueId(&uid), "ncclGetUniqueId");
    if (st != ATX_OK) return st;
    std::memcpy(id, &uid, ATX_COMM_ID_BYTES);
    return ATX_OK;
}

extern "C" int atx_comm_init(atx_comm** comm, int32_t world, int32_t rank, const void* id) {
    ATX_REQUIRE(comm && id, ATX_EINVAL, "atx_comm_init: null pointer");
    ATX_REQUIRE(world >= 1 && rank >= 0 && rank < world, ATX_EINVAL, "atx_comm_init: rank %d outside a world of %d", rank, world);
    ATX_RCCL_OR_FAIL(r);
    int device = -1;
    int st = hip_status(hipGetDevice(&device), "atx_comm_init: hipGetDevice");
    if (st != ATX_OK) return st;
    ncclUniqueId uid;
    std::memcpy(&uid, id, ATX_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    st = comm_status(r->CommInitRank(&c, world, uid, rank), "ncclCommInitRank");
    if (st != ATX_OK) return st;
    *comm = new atx_comm{c, world, rank, device};
    return ATX_OK;
}

extern "C" int atx_comm_destroy(atx_comm* comm) {
    if (!comm) return ATX_OK;
    ATX_RCCL_OR_FAIL(r);
    const int st = comm_status(r->CommDestroy(comm->comm), "ncclCommDestroy");
    delete comm;
    return st;
}

extern "C" int atx_comm_rank(const atx_comm* comm) { return comm ? comm->rank : ATX_EINVAL; }
extern "C" int atx_comm_world(const atx_comm* comm) { return comm ? comm->world : ATX_EINVAL; }

extern "C" int atx_bcast(atx_comm* comm, void* buf, int64_t n_bytes, int32_t root, void* stream) {
    ATX_REQUIRE(comm, ATX_EINVAL, "atx_bcast: null communicator");
    ATX_REQUIRE(n_bytes >= 0 && (buf || n_bytes == 0), ATX_EINVAL, "atx_bcast: bad buffer (n_bytes=%lld)", (long long)n_bytes);
    ATX_REQUIRE(root >= 0 && root < comm->world, ATX_EINVAL, "atx_bcast: root %d outside a world of %d", root, comm->world);
    if (n_bytes == 0) return ATX_OK;
    ATX_RCCL_OR_FAIL(r);
    return comm_status(r->Broadcast(buf, buf, (size_t)n_bytes, kNcclChar, root, comm->comm, static_cast<hipStream_t>(stream)), "ncclBroadcast");
}

extern "C" int atx_all_gather(atx_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    ATX_REQUIRE(comm, ATX_EINVAL, "atx_all_gather: null communicator");
    ATX_REQUIRE(bytes_per_rank >= 0 && ((send && recv) || bytes_per_rank == 0), ATX_EINVAL, "atx_all_gather: bad buffers (bytes_per_rank=%lld)",
                (long long)bytes_per_rank);
    if (bytes_per_rank == 0) return ATX_OK;
    ATX_RCCL_OR_FAIL(r);
    return comm_status(r->AllGather(send, recv, (size_t)bytes_per_rank, kNcclChar, comm->comm, static_cast<hipStream_t>(stream)), "ncclAllGather");
}

extern "C" int atx_exchange(atx_comm* comm, const void* const* send_ptrs, const int64_t* send_bytes, void* const* recv_ptrs,
                            const int64_t* recv_bytes, void* stream) {
    ATX_REQUIRE(comm && send_ptrs && send_bytes && recv_ptrs && recv_bytes, ATX_EINVAL, "atx_exchange: null pointer");
    for (int32_t p = 0; p < comm->world; ++p) {
        ATX_REQUIRE(send_bytes[p] >= 0 && recv_bytes[p] >= 0, ATX_EINVAL, "atx_exchange: negative byte count for peer %d", p);
        ATX_REQUIRE((send_ptrs[p] || send_bytes[p] == 0) && (recv_ptrs[p] || recv_bytes[p] == 0), ATX_EINVAL,
                    "atx_exchange: null buffer for peer %d", p);
    }
    ATX_RCCL_OR_FAIL(r);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the own slab never leaves the device: a plain copy on the same stream
    if (send_bytes[comm->rank] > 0 || recv_bytes[comm->rank] > 0) {
        ATX_REQUIRE(send_bytes[comm->rank] == recv_bytes[comm->rank], ATX_ESHAPE, "atx_exchange: own slab sizes differ (%lld vs %lld)",
                    (long long)send_bytes[comm->rank], (long long)recv_bytes[comm->rank]);
        if (send_ptrs[comm->rank] != recv_ptrs[comm->rank]) {
            const int st = hip_status(hipMemcpyAsync(recv_ptrs[comm->rank], send_ptrs[comm->rank], (size_t)send_bytes[comm->rank],
                                                     hipMemcpyDeviceToDevice, s), "atx_exchange: own slab copy");
            if (st != ATX_OK) return st;
        }
    }
    int st = comm_status(r->GroupStart(), "ncclGroupStart");
    if (st != ATX_OK) return st;
    for (int32_t p = 0; p < comm->world && st == ATX_OK; ++p) {
        if (p == comm->rank) continue;
        if (send_bytes[p] > 0) st = comm_status(r->Send(send_ptrs[p], (size_t)send_bytes[p], kNcclChar, p, comm->comm, s), "ncclSend");
        if (st == ATX_OK && recv_bytes[p] > 0)
            st = comm_status(r->Recv(recv_ptrs[p], (size_t)recv_bytes[p], kNcclChar, p, comm->comm, s), "ncclRecv");
    }
    const int end = comm_status(r->GroupEnd(), "ncclGroupEnd");
    return st != ATX_OK ? st : end;
}

extern "C" int atx_gather_shards(atx_comm* comm, void* buf, const int64_t* byte_offsets, void* stream) {
    ATX_REQUIRE(comm && buf && byte_offsets, ATX_EINVAL, "atx_gather_shards: null pointer");
    for (int32_t p = 0; p < comm->world; ++p)
        ATX_REQUIRE(byte_offsets[p] >= 0 && byte_offsets[p + 1] >= byte_offsets[p], ATX_EINVAL,
                    "atx_gather_shards: offsets must be non-negative and non-decreasing (rank %d)", p);
    ATX_RCCL_OR_FAIL(r);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int st = comm_status(r->GroupStart(), "ncclGroupStart");
    if (st != ATX_OK) return st;
    for (int32_t p = 0; p < comm->world && st == ATX_OK; ++p) {
        const int64_t n = byte_offsets[p + 1] - byte_offsets[p];
        if (n == 0) continue;
        char* part = static_cast<char*>(buf) + byte_offsets[p];
        st = comm_status(r->Broadcast(part, part, (size_t)n, kNcclChar, p, comm->comm, s), "ncclBroadcast");
    }
    const int end = comm_status(r->GroupEnd(), "ncclGroupEnd");
    return st != ATX_OK ? st : end;
}
