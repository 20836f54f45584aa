// Multi-GPU exchange: one context per rank, RCCL over xGMI.
//
// The reference has no communication layer; the only exchange this path
// needs is one all-gather of the per-shard CSR rows before the pairwise step.
// librccl is opened lazily with dlopen so that single-GPU users never pay its load time.
#include <dlfcn.h>

#include <cstring>

#include "skm_common.h"

namespace {

// Minimal mirror of the RCCL C API we call (rccl.h: ncclUniqueId is 128 opaque bytes).
struct nccl_uid {
    char internal[128];
};
typedef int (*fn_get_uid)(nccl_uid *);
typedef int (*fn_init_rank)(void **, int, nccl_uid, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);

struct rccl_api {
    void *lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_bcast bcast = nullptr;
    fn_allgather allgather = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_errstr errstr = nullptr;
};
rccl_api g_rccl;

int load_rccl()
{
    if (g_rccl.lib)
        return SKM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);  // LOCAL: the host process may carry its own RCCL (PyTorch does)
        if (lib)
            break;
    }
    if (!lib) {
        skm_set_error("cannot load librccl: %s", dlerror());
        return SKM_E_COMM;
    }
    g_rccl.get_uid = (fn_get_uid)dlsym(lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.bcast = (fn_bcast)dlsym(lib, "ncclBroadcast");
    g_rccl.allgather = (fn_allgather)dlsym(lib, "ncclAllGather");
    g_rccl.send = (fn_send)dlsym(lib, "ncclSend");
    g_rccl.recv = (fn_recv)dlsym(lib, "ncclRecv");
    g_rccl.group_start = (fn_group)dlsym(lib, "ncclGroupStart");
    g_rccl.group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
    g_rccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.bcast || !g_rccl.allgather || !g_rccl.send || !g_rccl.recv || !g_rccl.group_start ||
        !g_rccl.group_end) {
        skm_set_error("librccl lacks a required symbol");
        dlclose(lib);
        return SKM_E_COMM;
    }
    g_rccl.lib = lib;
    return SKM_OK;
}

#define SKM_NCCL(expr)                                                                              \
    do {                                                                                            \
        int _r = (expr);                                                                            \
        if (_r != 0) {                                                                              \
            skm_set_error("%s -> RCCL error %d (%s)", #expr, _r, g_rccl.errstr ? g_rccl.errstr(_r) : "?"); \
            return SKM_E_COMM;                                                                      \
        }                                                                                           \
    } while (0)

constexpr int NCCL_INT8 = 0;  // ncclInt8 / ncclChar

struct seg_table {
    int64_t off[SKM_MAX_RANKS + 1];
};

// Padded slots (one per rank, `maxb` apart) -> segments back to back; blockIdx.y = rank.  One launch
// instead of one device copy per rank.
__global__ __launch_bounds__(256) void k_unpack_slots(const uint8_t *__restrict__ slots, int64_t maxb, seg_table t,
                                                      uint8_t *__restrict__ out)
{
    const int r = blockIdx.y;
    const int64_t bytes = t.off[r + 1] - t.off[r];
    const uint8_t *src = slots + maxb * r;
    uint8_t *dst = out + t.off[r];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int64_t nvec = bytes >> 4;
        for (int64_t i = tid; i < nvec; i += nthreads)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        done = nvec << 4;
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
        const int64_t nw = bytes >> 2;
        for (int64_t i = tid; i < nw; i += nthreads)
            reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        done = nw << 2;
    }
    for (int64_t i = done + tid; i < bytes; i += nthreads)
        dst[i] = src[i];
}

}  // namespace

extern "C" int skm_comm_unique_id(uint8_t *h_id)
{
    SKM_REQUIRE(h_id, SKM_E_BADARG, "skm_comm_unique_id: null output");
    SKM_TRY(load_rccl());
    nccl_uid uid;
    SKM_NCCL(g_rccl.get_uid(&uid));
    static_assert(sizeof(uid) == SKM_COMM_ID_BYTES, "unique id size");
    memcpy(h_id, &uid, sizeof(uid));
    return SKM_OK;
}

extern "C" int skm_comm_init(skm_ctx *ctx, int nranks, int rank, const uint8_t *h_id)
{
    SKM_REQUIRE(ctx && h_id && nranks >= 1 && nranks <= SKM_MAX_RANKS && rank >= 0 && rank < nranks, SKM_E_BADARG,
                "skm_comm_init: bad argument (1 <= nranks <= %d)", SKM_MAX_RANKS);
    SKM_REQUIRE(!ctx->comm, SKM_E_BADARG, "skm_comm_init: communicator already initialised");
    SKM_TRY(load_rccl());
    SKM_HIP(hipSetDevice(ctx->device));
    nccl_uid uid;
    memcpy(&uid, h_id, sizeof(uid));
    SKM_NCCL(g_rccl.init_rank(&ctx->comm, nranks, uid, rank));
    ctx->nranks = nranks;
    ctx->rank = rank;
    return SKM_OK;
}

extern "C" int skm_comm_destroy(skm_ctx *ctx)
{
    if (ctx && ctx->comm && g_rccl.destroy) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        g_rccl.destroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        ctx->rank = 0;
    }
    return SKM_OK;
}

// Variable-size all-gather.  Contributions are padded to the largest one and exchanged with ONE
// ncclAllGather (RCCL's ring/direct all-gather uses all xGMI links at once; a group of per-rank
// broadcasts does not), then the pieces are packed back to back with device-to-device copies.
// The padding costs nothing here: row-block shards differ by well under 1 % in size.
extern "C" int skm_allgatherv(skm_ctx *ctx, const void *d_send, const int64_t *h_bytes, void *d_recv)
{
    SKM_REQUIRE(ctx && h_bytes && d_recv, SKM_E_BADARG, "skm_allgatherv: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm) {
        SKM_REQUIRE(ctx->nranks == 1, SKM_E_COMM, "skm_allgatherv: communicator not initialised");
        if (h_bytes[0] > 0 && d_send != d_recv)
            SKM_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)h_bytes[0], hipMemcpyDeviceToDevice, ctx->stream));
        return SKM_OK;
    }
    int64_t maxb = 0;
    for (int r = 0; r < ctx->nranks; ++r) {
        SKM_REQUIRE(h_bytes[r] >= 0, SKM_E_BADARG, "skm_allgatherv: negative size for rank %d", r);
        maxb = h_bytes[r] > maxb ? h_bytes[r] : maxb;
    }
    if (maxb == 0)
        return SKM_OK;
    maxb = (maxb + 255) & ~(int64_t)255;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_J, (size_t)maxb * (size_t)ctx->nranks, &p));
    uint8_t *slots = (uint8_t *)p;
    uint8_t *mine = slots + (size_t)maxb * (size_t)ctx->rank;
    SKM_PROF(ctx, "rccl_allgatherv");
    if (h_bytes[ctx->rank] > 0)
        SKM_HIP(hipMemcpyAsync(mine, d_send, (size_t)h_bytes[ctx->rank], hipMemcpyDeviceToDevice, ctx->stream));
    SKM_NCCL(g_rccl.allgather(mine, slots, (size_t)maxb, NCCL_INT8, ctx->comm, ctx->stream));
    seg_table t = {};
    for (int r = 0; r < ctx->nranks; ++r)
        t.off[r + 1] = t.off[r] + h_bytes[r];
    const unsigned gx = (unsigned)skm_grid_cap(ctx, skm_ceil_div(maxb, 256 * 16), 4);
    k_unpack_slots<<<dim3(gx, (unsigned)ctx->nranks), 256, 0, ctx->stream>>>(slots, maxb, t, (uint8_t *)d_recv);
    return skm_check_launch("k_unpack_slots");
}

// Variable-size all-to-all as one group of point-to-point transfers (what RCCL's own all-to-all
// does); xGMI is point to point, so every pair uses its own link.
extern "C" int skm_alltoallv(skm_ctx *ctx, const void *d_send, const int64_t *h_send_bytes, void *d_recv,
                             const int64_t *h_recv_bytes)
{
    SKM_REQUIRE(ctx && h_send_bytes && h_recv_bytes, SKM_E_BADARG, "skm_alltoallv: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    const int nr = ctx->comm ? ctx->nranks : 1, me = ctx->comm ? ctx->rank : 0;
    SKM_REQUIRE(ctx->comm || ctx->nranks == 1, SKM_E_COMM, "skm_alltoallv: communicator not initialised");
    int64_t soff = 0, roff = 0, my_soff = 0, my_roff = 0;
    for (int p = 0; p < nr; ++p) {
        SKM_REQUIRE(h_send_bytes[p] >= 0 && h_recv_bytes[p] >= 0, SKM_E_BADARG, "skm_alltoallv: negative size for rank %d", p);
        if (p == me) {
            my_soff = soff;
            my_roff = roff;
        }
        soff += h_send_bytes[p];
        roff += h_recv_bytes[p];
    }
    SKM_REQUIRE(h_send_bytes[me] == h_recv_bytes[me], SKM_E_BADARG, "skm_alltoallv: self segment sizes differ");
    SKM_REQUIRE((soff == 0 || d_send) && (roff == 0 || d_recv), SKM_E_BADARG, "skm_alltoallv: null buffer");
    SKM_PROF(ctx, "rccl_alltoallv");
    if (h_send_bytes[me] > 0)
        SKM_HIP(hipMemcpyAsync((uint8_t *)d_recv + my_roff, (const uint8_t *)d_send + my_soff, (size_t)h_send_bytes[me],
                               hipMemcpyDeviceToDevice, ctx->stream));
    if (nr == 1)
        return SKM_OK;
    SKM_NCCL(g_rccl.group_start());
    soff = roff = 0;
    int status = 0;
    for (int p = 0; p < nr && status == 0; ++p) {
        if (p != me) {
            if (h_send_bytes[p] > 0)
                status = g_rccl.send((const uint8_t *)d_send + soff, (size_t)h_send_bytes[p], NCCL_INT8, p, ctx->comm, ctx->stream);
            if (status == 0 && h_recv_bytes[p] > 0)
                status = g_rccl.recv((uint8_t *)d_recv + roff, (size_t)h_recv_bytes[p], NCCL_INT8, p, ctx->comm, ctx->stream);
        }
        soff += h_send_bytes[p];
        roff += h_recv_bytes[p];
    }
    const int end_status = g_rccl.group_end();  // always close the group, also after a failed call
    SKM_NCCL(status);
    SKM_NCCL(end_status);
    return SKM_OK;
}

// ------------------------------------------------------------------------------- several arrays, one group
// The distributed basis moves struct-of-arrays data (codes + posting words; postings + column starts +
// table keys + table values + row norms).  One RCCL group holds the point-to-point transfers of ALL the
// arrays to ALL peers: RCCL fuses a group into one launch, every pair uses its own xGMI link, and each
// piece lands at its final offset (no padded scratch, no unpack pass).  The byte ranges come from two
// pure host functions, exported so that tests can execute the same plan without RCCL.

// (skm_plan_alltoallv / skm_plan_allgatherv: skm_host.cpp)

namespace {
// Executes a plan: the self pieces are device copies, the rest one RCCL group.
int run_p2p_plan(skm_ctx *ctx, const char *what, int narrays, const void *const *d_send, void *const *d_recv,
                 const skm_p2p_op *ops)
{
    const int nr = ctx->comm ? ctx->nranks : 1, me = ctx->comm ? ctx->rank : 0;
    SKM_REQUIRE(ctx->comm || ctx->nranks == 1, SKM_E_COMM, "%s: communicator not initialised", what);
    for (int a = 0; a < narrays; ++a) {
        const skm_p2p_op &op = ops[me * narrays + a];
        SKM_REQUIRE(op.send_bytes == op.recv_bytes, SKM_E_BADARG, "%s: self segment sizes differ (array %d)", what, a);
        if (op.send_bytes > 0) {
            SKM_REQUIRE(d_send[a] && d_recv[a], SKM_E_BADARG, "%s: null buffer (array %d)", what, a);
            if ((const uint8_t *)d_send[a] + op.send_off != (uint8_t *)d_recv[a] + op.recv_off)
                SKM_HIP(hipMemcpyAsync((uint8_t *)d_recv[a] + op.recv_off, (const uint8_t *)d_send[a] + op.send_off,
                                       (size_t)op.send_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    if (nr == 1)
        return SKM_OK;
    SKM_NCCL(g_rccl.group_start());
    int status = 0;
    for (int p = 0; p < nr && status == 0; ++p) {
        if (p == me)
            continue;
        for (int a = 0; a < narrays && status == 0; ++a) {
            const skm_p2p_op &op = ops[p * narrays + a];
            if (op.send_bytes > 0)
                status = g_rccl.send((const uint8_t *)d_send[a] + op.send_off, (size_t)op.send_bytes, NCCL_INT8, p, ctx->comm,
                                     ctx->stream);
            if (status == 0 && op.recv_bytes > 0)
                status = g_rccl.recv((uint8_t *)d_recv[a] + op.recv_off, (size_t)op.recv_bytes, NCCL_INT8, p, ctx->comm,
                                     ctx->stream);
        }
    }
    const int end_status = g_rccl.group_end();  // always close the group, also after a failed call
    SKM_NCCL(status);
    SKM_NCCL(end_status);
    return SKM_OK;
}
}  // namespace

extern "C" int skm_alltoallv_multi(skm_ctx *ctx, int narrays, const void *const *d_send, void *const *d_recv,
                                   const int64_t *h_elem_bytes, const int64_t *h_send_counts, const int64_t *h_recv_counts)
{
    SKM_REQUIRE(ctx && d_send && d_recv, SKM_E_BADARG, "skm_alltoallv_multi: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    skm_p2p_op ops[SKM_MAX_RANKS * SKM_MAX_ARRAYS];
    SKM_TRY(skm_plan_alltoallv(ctx->comm ? ctx->nranks : 1, narrays, h_elem_bytes, h_send_counts, h_recv_counts, ops));
    SKM_PROF(ctx, "rccl_alltoallv");
    return run_p2p_plan(ctx, "skm_alltoallv_multi", narrays, d_send, d_recv, ops);
}

extern "C" int skm_allgatherv_multi(skm_ctx *ctx, int narrays, const void *const *d_send, void *const *d_recv,
                                    const int64_t *h_elem_bytes, const int64_t *h_counts)
{
    SKM_REQUIRE(ctx && d_send && d_recv, SKM_E_BADARG, "skm_allgatherv_multi: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    skm_p2p_op ops[SKM_MAX_RANKS * SKM_MAX_ARRAYS];
    SKM_TRY(skm_plan_allgatherv(ctx->comm ? ctx->nranks : 1, ctx->comm ? ctx->rank : 0, narrays, h_elem_bytes, h_counts, ops));
    SKM_PROF(ctx, "rccl_allgatherv");
    return run_p2p_plan(ctx, "skm_allgatherv_multi", narrays, d_send, d_recv, ops);
}
