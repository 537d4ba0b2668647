// Sharded solves: the batch is embarrassingly parallel, so the only exchange is one all-gather of
// the control outputs over RCCL/xGMI (BASELINE.json north_star).
//
// One handle per GPU.  The handles of one job -- one per process under torch.distributed.run / MPI,
// or G of them inside one C++ host process (reference host: src/interface.cpp:3) -- join a
// communicator the NCCL way: rank 0 makes a 128-byte id (tpc_mpc_comm_unique_id), the host passes it
// to the others by whatever means it has, everybody calls tpc_mpc_comm_init_rank.
// tpc_mpc_solve_batch_compact_sharded then solves this rank's contiguous block of the batch straight
// into its slot of the full-size output arrays and all-gathers the slots, all on the caller's
// stream: afterwards every GPU holds the control outputs of all instances.
//
// RCCL is loaded with dlopen on first use, not linked: a single-GPU host never needs it, and inside a
// process that already carries an RCCL (PyTorch bundles one under the same soname) the loader hands
// out that copy instead of a second one.
#include "tpc_mpc_context.h"
#include "tpc_mpc_experimental.h"

#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#include <rccl/rccl.h>   // types and prototypes only; the symbols are resolved by dlsym below

namespace tpc {

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = "";
};

// Resolved once per process (std::call_once: handles of different threads may race to be first); a
// failed attempt is remembered with its reason.
Rccl g_rccl;
void rccl_load() {
    Rccl& r = g_rccl;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        std::snprintf(r.why, sizeof(r.why), "cannot load librccl.so.1: %s", dlerror());
        return;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(r.lib, name);
        if (!p) { ok = false; std::snprintf(r.why, sizeof(r.why), "librccl lacks %s", name); }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
}
Rccl* rccl() {
    static std::once_flag once;
    std::call_once(once, rccl_load);
    return g_rccl.lib ? &g_rccl : nullptr;
}

int rccl_fail(tpc_mpc_context* h, Rccl* r, ncclResult_t e, const char* what) {
    return fail(h, TPC_MPC_ERR_COMM, "%s: %s", what, r->GetErrorString(e));
}
#define RCCL_TRY(h, r, call)                                          \
    do {                                                              \
        ncclResult_t e__ = (call);                                    \
        if (e__ != ncclSuccess) return rccl_fail(h, r, e__, #call);   \
    } while (0)

// ncclGroupStart .. ncclGroupEnd around a set of collectives.  The group is closed on EVERY path out of the
// scope: a call that fails in between must not leave the thread's group depth above zero, or every later
// collective of the thread would be queued and never launched (and the other ranks would wait for ever).
// The first error is the one reported.
struct GroupScope {
    Rccl* r;
    bool open = false;
    explicit GroupScope(Rccl* rr) : r(rr) {}
    ncclResult_t begin() { const ncclResult_t e = r->GroupStart(); open = e == ncclSuccess; return e; }
    ncclResult_t end() { open = false; return r->GroupEnd(); }
    ~GroupScope() { if (open) (void)r->GroupEnd(); }
};

}  // namespace

void comm_destroy(tpc_mpc_context* h) {
    if (!h || !h->comm) return;
    Rccl* r = rccl();
    if (r && h->comm->comm) (void)r->CommDestroy(h->comm->comm);
    delete h->comm;
    h->comm = nullptr;
}

// first instance and count of rank's contiguous block; blocks differ by at most one instance
static void shard_range(int64_t n, int rank, int world, int64_t* first, int64_t* count) {
    const int64_t base = n / world, rem = n % world;
    *count = base + (rank < rem ? 1 : 0);
    *first = rank * base + (rank < rem ? rank : rem);
}
// ... or, interleaved (SURVEY.md section 8e names it first): rank r owns the instances i = r (mod world), so that a batch
// sorted by speed -- the iteration count is a function of the speed alone -- spreads its long instances over all ranks
static void shard_map(int64_t n, int rank, int world, int split, int64_t* first, int64_t* count, int64_t* stride) {
    if (split == TPC_MPC_SPLIT_INTERLEAVED) {
        *first = rank;
        *stride = world;
        *count = rank < n ? (n - rank + world - 1) / world : 0;
    } else {
        shard_range(n, rank, world, first, count);
        *stride = 1;
    }
}
// slot size of the interleaved exchange: every rank contributes `cap` elements (the last ones padding where n_total does
// not divide), staged as [world][cap]
static int64_t interleaved_cap(int64_t n, int world) { return (n + world - 1) / world; }

// The exchange of one output row as a list of collectives on ONE buffer (the full-size row for the block split, the
// [world][cap] staging array for the interleaved one) -- separated from the RCCL calls so that the slot arithmetic of all
// `world` owners can be executed and checked on the host (tests/test_host_logic.py through tpc_mpc_x_exchange_plan).
struct XOp {
    int kind;                 // 0: all-gather (every rank sends [send_off, +count), receives world * count at recv_off); 1: in-place broadcast from `root`
    int root;
    int64_t send_off, recv_off, count;   // elements
};
static int exchange_plan(int64_t n_total, int world, int rank, int split, bool ragged, XOp* ops, int max_ops) {
    int n = 0;
    auto push = [&](XOp op) { if (n < max_ops) ops[n] = op; ++n; };
    if (split == TPC_MPC_SPLIT_INTERLEAVED) {
        const int64_t cap = interleaved_cap(n_total, world);
        push({0, 0, rank * cap, 0, cap});
        return n;
    }
    int64_t first = 0, count = 0;
    shard_range(n_total, rank, world, &first, &count);
    if (n_total % world == 0 && !ragged) {
        push({0, 0, first, 0, count});          // equal blocks: in place (the send buffer IS this rank's slot of the receive buffer)
    } else {
        for (int q = 0; q < world; ++q) {      // ragged blocks: the all-gather spelled as one in-place broadcast per owner
            int64_t qf = 0, qc = 0;
            shard_range(n_total, q, world, &qf, &qc);
            if (qc == 0) continue;
            push({1, q, qf, qf, qc});
        }
    }
    return n;
}

// out[i] = stage[(i mod world) * cap + i / world]: the interleaved exchange's [world][cap] staging back into instance order
template <typename E>
__global__ void unpermute_kernel(const E* __restrict__ stage, E* __restrict__ out, int64_t n, int world, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = stage[(i % world) * cap + i / world];
}
static hipError_t unpermute(const void* stage, void* out, int64_t n, int world, int64_t cap, int elem_bytes, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int bs = 256;
    const unsigned grid = (unsigned)((n + bs - 1) / bs);
    if (elem_bytes == 8) hipLaunchKernelGGL(unpermute_kernel<uint64_t>, dim3(grid), dim3(bs), 0, s, (const uint64_t*)stage, (uint64_t*)out, n, world, cap);
    else hipLaunchKernelGGL(unpermute_kernel<uint32_t>, dim3(grid), dim3(bs), 0, s, (const uint32_t*)stage, (uint32_t*)out, n, world, cap);
    return hipGetLastError();
}

// runs a plan's collectives on `buf` (elements of elem_bytes) inside the caller's RCCL group
static int run_plan(tpc_mpc_context* h, Rccl* r, const XOp* ops, int n_ops, char* buf, int elem_bytes, hipStream_t s) {
    const ncclDataType_t dt = elem_bytes == 8 ? ncclFloat64 : ncclFloat32;   // (moved as bit patterns: int32 rows too)
    ncclComm_t c = h->comm->comm;
    const int64_t es = elem_bytes;
    for (int i = 0; i < n_ops; ++i) {
        const XOp& op = ops[i];
        if (op.kind == 0) RCCL_TRY(h, r, r->AllGather(buf + op.send_off * es, buf + op.recv_off * es, (size_t)op.count, dt, c, s));
        else RCCL_TRY(h, r, r->Broadcast(buf + op.send_off * es, buf + op.recv_off * es, (size_t)op.count, dt, op.root, c, s));
    }
    return TPC_MPC_OK;
}
constexpr int kMaxOps = 1024;   // (a communicator of more ranks than this is refused at init)

}  // namespace tpc

using namespace tpc;

extern "C" {

int tpc_mpc_comm_unique_id(void* id, size_t len) {
    return guarded(nullptr, [&]() -> int {
        if (!id || len < TPC_MPC_COMM_ID_BYTES) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "id buffer must hold %d bytes", TPC_MPC_COMM_ID_BYTES);
        static_assert(TPC_MPC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
        Rccl* r = rccl();
        if (!r) return fail(nullptr, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
        ncclUniqueId u;
        RCCL_TRY(nullptr, r, r->GetUniqueId(&u));
        std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
        return TPC_MPC_OK;
    });
}

int tpc_mpc_comm_init_rank(tpc_mpc_handle h, const void* id, size_t len, int rank, int world) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
        if (world < 1 || rank < 0 || rank >= world) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= rank < world");
        if (world > kMaxOps) return fail(h, TPC_MPC_ERR_BAD_ARG, "at most %d ranks", kMaxOps);
        comm_destroy(h);
        // a world of one needs no communicator -- unless a test asked for one (tpc_mpc_comm_test_mode: lets a
        // one-GPU box exercise the RCCL calls themselves)
        if (world == 1 && !h->comm_test_force) return TPC_MPC_OK;
        if (!id || len < TPC_MPC_COMM_ID_BYTES) return fail(h, TPC_MPC_ERR_BAD_ARG, "id must hold %d bytes", TPC_MPC_COMM_ID_BYTES);
        Rccl* r = rccl();
        if (!r) return fail(h, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
        HIP_TRY(h, hipSetDevice(h->device));
        Comm* c = new (std::nothrow) Comm;
        if (!c) return fail(h, TPC_MPC_ERR_ALLOC, "out of host memory");
        ncclUniqueId u;
        std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t e = r->CommInitRank(&c->comm, world, u, rank);
        if (e != ncclSuccess) { delete c; return rccl_fail(h, r, e, "ncclCommInitRank"); }
        c->rank = rank;
        c->world = world;
        h->comm = c;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_comm_test_mode(tpc_mpc_handle h, int force_communicator, int force_ragged) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        h->comm_test_force = force_communicator != 0;
        h->comm_test_ragged = force_ragged != 0;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_comm_destroy(tpc_mpc_handle h) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        (void)hipSetDevice(h->device);
        comm_destroy(h);
        return TPC_MPC_OK;
    });
}

int tpc_mpc_group_begin(void) {
    return guarded(nullptr, [&]() -> int {
        Rccl* r = rccl();
        if (!r) return fail(nullptr, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
        RCCL_TRY(nullptr, r, r->GroupStart());
        return TPC_MPC_OK;
    });
}

int tpc_mpc_group_end(void) {
    return guarded(nullptr, [&]() -> int {
        Rccl* r = rccl();
        if (!r) return fail(nullptr, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
        RCCL_TRY(nullptr, r, r->GroupEnd());
        return TPC_MPC_OK;
    });
}

int tpc_mpc_shard_range(int64_t n_total, int rank, int world, int64_t* first, int64_t* count) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) return TPC_MPC_ERR_BAD_ARG;
    shard_range(n_total, rank, world, first, count);
    return TPC_MPC_OK;
}

int tpc_mpc_shard_map(int64_t n_total, int rank, int world, int split, int64_t* first, int64_t* count, int64_t* stride) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !first || !count || !stride) return TPC_MPC_ERR_BAD_ARG;
    if (split != TPC_MPC_SPLIT_BLOCK && split != TPC_MPC_SPLIT_INTERLEAVED) return TPC_MPC_ERR_BAD_ARG;
    shard_map(n_total, rank, world, split, first, count, stride);
    return TPC_MPC_OK;
}

int tpc_mpc_x_exchange_plan(int64_t n_total, int world, int rank, int split, int force_ragged, int64_t* ops5, int max_ops,
                            int64_t* buffer_elems) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || (max_ops > 0 && !ops5)) return -1;
    if (split != TPC_MPC_SPLIT_BLOCK && split != TPC_MPC_SPLIT_INTERLEAVED) return -1;
    XOp ops[kMaxOps];
    const int n = exchange_plan(n_total, world, rank, split, force_ragged != 0, ops, kMaxOps);
    for (int i = 0; i < n && i < max_ops && i < kMaxOps; ++i) {
        ops5[5 * i] = ops[i].kind; ops5[5 * i + 1] = ops[i].root; ops5[5 * i + 2] = ops[i].send_off;
        ops5[5 * i + 3] = ops[i].recv_off; ops5[5 * i + 4] = ops[i].count;
    }
    if (buffer_elems) *buffer_elems = split == TPC_MPC_SPLIT_INTERLEAVED ? interleaved_cap(n_total, world) * world : n_total;
    return n;
}

int tpc_mpc_solve_batch_compact_sharded_split(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n_total, int split,
                                              const void* v_shard, const void* delta_y_shard, const void* delta_phi_shard,
                                              void* steering_front_all, void* steering_rear_all, int32_t* iters_shard,
                                              uint32_t* flags_out, void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_compact_model(h, p);
        if (rc) return rc;
        if (split != TPC_MPC_SPLIT_BLOCK && split != TPC_MPC_SPLIT_INTERLEAVED) return fail(h, TPC_MPC_ERR_BAD_ARG, "split is TPC_MPC_SPLIT_BLOCK or TPC_MPC_SPLIT_INTERLEAVED");
        if (n_total < 0 || n_total > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n_total < 2^31");
        if (n_total == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!steering_front_all || !steering_rear_all) return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        const int rank = h->comm ? h->comm->rank : 0, world = h->comm ? h->comm->world : 1;
        int64_t first = 0, count = 0, stride = 1;
        shard_map(n_total, rank, world, split, &first, &count, &stride);
        if (count > 0 && (!v_shard || !delta_y_shard || !delta_phi_shard)) return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(p->dtype);
        char* front = (char*)steering_front_all;
        char* rear = (char*)steering_rear_all;
        const bool exchange = world > 1 || h->comm;
        // interleaved: the shard is solved into this rank's slot of a [world][cap] staging array per output, the slots are
        // all-gathered there, and one kernel per output puts them back into instance order (a world of one without a
        // communicator owns every instance in order already: solved straight into the outputs)
        const bool staged = split == TPC_MPC_SPLIT_INTERLEAVED && exchange;
        const int64_t cap = interleaved_cap(n_total, world);
        char *sf = nullptr, *sr = nullptr;
        if (staged) {
            rc = ensure(h, &h->gather, &h->gather_bytes, 2 * pad256(world * cap * es));   // (before anything is enqueued: growing frees)
            if (rc) return rc;
            sf = (char*)h->gather;
            sr = sf + pad256(world * cap * es);
        }
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        if (count > 0) {
            char* of = staged ? sf + rank * cap * es : front + first * es;
            char* orr = staged ? sr + rank * cap * es : rear + first * es;
            rc = compact_launch(h, p, count, v_shard, delta_y_shard, delta_phi_shard, of, orr, iters_shard, s);
            if (rc) return rc;
        }
        if (exchange) {
            Rccl* r = rccl();
            if (!r) return fail(h, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
            XOp ops[kMaxOps];
            const int n_ops = exchange_plan(n_total, world, rank, split, h->comm_test_ragged, ops, kMaxOps);
            GroupScope group(r);
            RCCL_TRY(h, r, group.begin());
            rc = run_plan(h, r, ops, n_ops, staged ? sf : front, (int)es, s);
            if (rc) return rc;
            rc = run_plan(h, r, ops, n_ops, staged ? sr : rear, (int)es, s);
            if (rc) return rc;
            RCCL_TRY(h, r, group.end());
            if (staged) {
                HIP_TRY(h, unpermute(sf, front, n_total, world, cap, (int)es, s));
                HIP_TRY(h, unpermute(sr, rear, n_total, world, cap, (int)es, s));
            }
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_solve_batch_compact_sharded(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n_total,
                                        const void* v_shard, const void* delta_y_shard, const void* delta_phi_shard,
                                        void* steering_front_all, void* steering_rear_all, int32_t* iters_shard,
                                        uint32_t* flags_out, void* stream) {
    return tpc_mpc_solve_batch_compact_sharded_split(h, p, n_total, TPC_MPC_SPLIT_BLOCK, v_shard, delta_y_shard, delta_phi_shard,
                                                     steering_front_all, steering_rear_all, iters_shard, flags_out, stream);
}

int tpc_mpc_solve_batch_general_sharded(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_general_io* io_all,
                                        uint32_t* flags_out, void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_general_device_io(h, io_all);
        if (rc) return rc;
        const int64_t n_total = io_all->n, ld = io_all->ld;
        if (n_total == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        const int rank = h->comm ? h->comm->rank : 0, world = h->comm ? h->comm->world : 1;
        int64_t first = 0, count = 0;
        shard_range(n_total, rank, world, &first, &count);
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(p->dtype);
        const int I = io_all->inputs;
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        // this rank's block: columns [first, first + count) of the full-size arrays, solved in place
        if (count > 0) {
            tpc_mpc_general_io blk = *io_all;
            auto at = [&](const void* base) -> const void* { return base ? (const char*)base + first * es : nullptr; };
            blk.n = count;
            blk.A = at(io_all->A); blk.B = at(io_all->B); blk.C = at(io_all->C); blk.Q = at(io_all->Q); blk.R = at(io_all->R);
            blk.lower = at(io_all->lower); blk.upper = at(io_all->upper); blk.x0 = at(io_all->x0); blk.targets = at(io_all->targets);
            blk.controls_inout = (void*)at(io_all->controls_inout); blk.v_inout = (void*)at(io_all->v_inout);
            blk.u0 = (void*)at(io_all->u0);
            blk.iters = io_all->iters ? io_all->iters + first : nullptr;
            rc = general_launch(h, p, &blk, s);
            if (rc) return rc;
        }
        if (world > 1 || h->comm) {
            Rccl* r = rccl();
            if (!r) return fail(h, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
            const ncclDataType_t dt = p->dtype == TPC_MPC_F64 ? ncclFloat64 : ncclFloat32;
            ncclComm_t c = h->comm->comm;
            GroupScope group(r);
            RCCL_TRY(h, r, group.begin());
            // u0 is I rows of ld: each row's slots are exchanged in place like the compact form's two outputs
            for (int j = 0; j < I; ++j) {
                char* row = (char*)io_all->u0 + (int64_t)j * ld * es;
                if (n_total % world == 0 && !h->comm_test_ragged) {
                    RCCL_TRY(h, r, r->AllGather(row + first * es, row, (size_t)count, dt, c, s));
                } else {
                    for (int q = 0; q < world; ++q) {
                        int64_t qf = 0, qc = 0;
                        shard_range(n_total, q, world, &qf, &qc);
                        if (qc == 0) continue;
                        RCCL_TRY(h, r, r->Broadcast(row + qf * es, row + qf * es, (size_t)qc, dt, q, c, s));
                    }
                }
            }
            RCCL_TRY(h, r, group.end());
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_gather_shards_split(tpc_mpc_handle h, int64_t n_total, int split, void* const* rows, int n_rows, int elem_bytes,
                                void* stream) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "a host-only handle has no device arrays to exchange");
        if (split != TPC_MPC_SPLIT_BLOCK && split != TPC_MPC_SPLIT_INTERLEAVED) return fail(h, TPC_MPC_ERR_BAD_ARG, "split is TPC_MPC_SPLIT_BLOCK or TPC_MPC_SPLIT_INTERLEAVED");
        if (n_total < 0 || n_total > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n_total < 2^31");
        if (n_rows < 0 || (n_rows > 0 && !rows)) return fail(h, TPC_MPC_ERR_BAD_ARG, "need n_rows >= 0 and a row table");
        if (elem_bytes != 4 && elem_bytes != 8) return fail(h, TPC_MPC_ERR_BAD_ARG, "elem_bytes is 4 or 8");
        for (int i = 0; i < n_rows; ++i)
            if (!rows[i]) return fail(h, TPC_MPC_ERR_BAD_ARG, "null row pointer");
        if (n_total == 0 || n_rows == 0 || !h->comm) return TPC_MPC_OK;   // a world of one holds everything already, in order
        const int rank = h->comm->rank, world = h->comm->world;
        int64_t first = 0, count = 0, stride = 1;
        shard_map(n_total, rank, world, split, &first, &count, &stride);
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = elem_bytes;
        const bool staged = split == TPC_MPC_SPLIT_INTERLEAVED;
        const int64_t cap = interleaved_cap(n_total, world);
        if (staged) {
            const int rc0 = ensure(h, &h->gather, &h->gather_bytes, pad256(world * cap * es));
            if (rc0) return rc0;
        }
        StreamOrderScope order(h, s);   // (the rows were written by this handle's own solve on this or another stream)
        int rc = order.begin();
        if (rc) return rc;
        Rccl* r = rccl();
        if (!r) return fail(h, TPC_MPC_ERR_COMM, "RCCL is not available in this process: %s", g_rccl.why);
        XOp ops[kMaxOps];
        const int n_ops = exchange_plan(n_total, world, rank, split, h->comm_test_ragged, ops, kMaxOps);
        if (!staged) {
            GroupScope group(r);
            RCCL_TRY(h, r, group.begin());
            for (int i = 0; i < n_rows; ++i) {
                rc = run_plan(h, r, ops, n_ops, (char*)rows[i], elem_bytes, s);
                if (rc) return rc;
            }
            RCCL_TRY(h, r, group.end());
        } else {
            // interleaved: the row's first `count` elements are this rank's shard (element j = instance rank + j * world); one
            // row at a time through the staging array (copy in, all-gather, back in instance order)
            char* st = (char*)h->gather;
            for (int i = 0; i < n_rows; ++i) {
                if (count > 0) HIP_TRY(h, hipMemcpyAsync(st + rank * cap * es, rows[i], (size_t)(count * es), hipMemcpyDeviceToDevice, s));
                GroupScope group(r);
                RCCL_TRY(h, r, group.begin());
                rc = run_plan(h, r, ops, n_ops, st, elem_bytes, s);
                if (rc) return rc;
                RCCL_TRY(h, r, group.end());
                HIP_TRY(h, unpermute(st, rows[i], n_total, world, cap, elem_bytes, s));
            }
        }
        return order.end();
    });
}

int tpc_mpc_gather_shards(tpc_mpc_handle h, int64_t n_total, void* const* rows, int n_rows, int elem_bytes, void* stream) {
    return tpc_mpc_gather_shards_split(h, n_total, TPC_MPC_SPLIT_BLOCK, rows, n_rows, elem_bytes, stream);
}

}  // extern "C"
