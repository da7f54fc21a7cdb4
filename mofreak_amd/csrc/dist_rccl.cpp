// libmofreak_dist.so: include/mofreak_dist.h over RCCL.  One process per GPU; no exception leaves this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mofreak_dist.h"
#include "dist_gather.h"

static_assert(sizeof(ncclUniqueId) == MOFREAK_UNIQUE_ID_BYTES, "MOFREAK_UNIQUE_ID_BYTES");

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &m)
{
    g_err = m;
    return code;
}
#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(MOFREAK_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define HIPD_TRY(expr)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? MOFREAK_ERR_OOM : MOFREAK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
}  // namespace

struct mofreak_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = -1;
    bool owned = false;
    int64_t *d_scratch = nullptr;  // counts / reductions
    size_t scratch_bytes = 0;
};

namespace {
int scratch(mofreak_comm *c, size_t bytes)
{
    if (c->scratch_bytes >= bytes) return MOFREAK_OK;
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIPD_TRY(hipMalloc((void **)&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return MOFREAK_OK;
}

struct RcclTransport {  // the interface dist_gather.h is written against
    mofreak_comm *c;
    int group_start() { return ncclGroupStart() == ncclSuccess ? 0 : fail(MOFREAK_ERR_HIP, "ncclGroupStart"); }
    int group_end()
    {
        const ncclResult_t r = ncclGroupEnd();
        return r == ncclSuccess ? 0 : fail(MOFREAK_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
    }
    int send(const void *p, int64_t bytes, int peer)
    {
        const ncclResult_t r = ncclSend(p, (size_t)bytes, ncclUint8, peer, c->comm, c->stream);
        return r == ncclSuccess ? 0 : fail(MOFREAK_ERR_HIP, std::string("ncclSend: ") + ncclGetErrorString(r));
    }
    int recv(void *p, int64_t bytes, int peer)
    {
        const ncclResult_t r = ncclRecv(p, (size_t)bytes, ncclUint8, peer, c->comm, c->stream);
        return r == ncclSuccess ? 0 : fail(MOFREAK_ERR_HIP, std::string("ncclRecv: ") + ncclGetErrorString(r));
    }
    int copy_local(void *dst, const void *src, int64_t bytes)
    {
        const hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream);
        return e == hipSuccess ? 0 : fail(MOFREAK_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    int sync()
    {
        const hipError_t e = hipStreamSynchronize(c->stream);
        return e == hipSuccess ? 0 : fail(MOFREAK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
};
}  // namespace

extern "C" {

int mofreak_dist_abi_version(void) { return MOFREAK_DIST_ABI_VERSION; }
const char *mofreak_dist_last_error(void) { return g_err.c_str(); }

int mofreak_shard_lpt(const int64_t *costs, int n, int world, int32_t *rank_of_out)
{
    if (n < 0 || world < 1 || (n > 0 && (!costs || !rank_of_out))) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_shard_lpt: bad argument");
    try {
        mofreak_dist::shard_lpt(costs, n, world, rank_of_out);
    } catch (const std::bad_alloc &) {
        return fail(MOFREAK_ERR_OOM, "out of host memory");
    }
    return MOFREAK_OK;
}

int mofreak_comm_unique_id(void *id128_out)
{
    if (!id128_out) return fail(MOFREAK_ERR_BAD_ARG, "null id");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id128_out, &id, sizeof id);
    return MOFREAK_OK;
}

int mofreak_comm_create(const void *id128, int rank, int world, int device, mofreak_comm **out)
{
    if (!out) return fail(MOFREAK_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!id128 || world < 1 || rank < 0 || rank >= world || device < 0) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_comm_create: bad argument");
    HIPD_TRY(hipSetDevice(device));
    mofreak_comm *c = new (std::nothrow) mofreak_comm;
    if (!c) return fail(MOFREAK_ERR_OOM, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->owned = true;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(MOFREAK_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)ncclCommDestroy(c->comm);
        delete c;
        return fail(MOFREAK_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return MOFREAK_OK;
}

int mofreak_comm_wrap(void *nccl_comm, void *hip_stream, int rank, int world, mofreak_comm **out)
{
    if (!out) return fail(MOFREAK_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_comm_wrap: bad argument");
    mofreak_comm *c = new (std::nothrow) mofreak_comm;
    if (!c) return fail(MOFREAK_ERR_OOM, "out of host memory");
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->rank = rank;
    c->world = world;
    *out = c;
    return MOFREAK_OK;
}

void mofreak_comm_destroy(mofreak_comm *c)
{
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->owned) {
        if (c->stream) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamDestroy(c->stream);
        }
        if (c->comm) (void)ncclCommDestroy(c->comm);
    }
    delete c;
}

int mofreak_comm_rank(const mofreak_comm *c) { return c ? c->rank : -1; }
int mofreak_comm_world(const mofreak_comm *c) { return c ? c->world : 0; }

int mofreak_gather_counts(mofreak_comm *c, int64_t n_rows, int64_t *counts_out)
{
    if (!c || !counts_out || n_rows < 0) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_gather_counts: bad argument");
    int rc = scratch(c, ((size_t)c->world + 1) * sizeof(int64_t));
    if (rc) return rc;
    HIPD_TRY(hipMemcpyAsync(c->d_scratch, &n_rows, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(ncclAllGather(c->d_scratch, c->d_scratch + 1, 1, ncclInt64, c->comm, c->stream));
    HIPD_TRY(hipMemcpyAsync(counts_out, c->d_scratch + 1, (size_t)c->world * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPD_TRY(hipStreamSynchronize(c->stream));
    return MOFREAK_OK;
}

int mofreak_allreduce_sum_i64(mofreak_comm *c, int64_t *values, int n)
{
    if (!c || n < 0 || (n > 0 && !values)) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_allreduce_sum_i64: bad argument");
    if (n == 0) return MOFREAK_OK;
    int rc = scratch(c, (size_t)n * sizeof(int64_t));
    if (rc) return rc;
    HIPD_TRY(hipMemcpyAsync(c->d_scratch, values, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(ncclAllReduce(c->d_scratch, c->d_scratch, (size_t)n, ncclInt64, ncclSum, c->comm, c->stream));
    HIPD_TRY(hipMemcpyAsync(values, c->d_scratch, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPD_TRY(hipStreamSynchronize(c->stream));
    return MOFREAK_OK;
}

int mofreak_gather_rows(mofreak_comm *c, const mofreak_row *d_rows, const int64_t *counts, int root, mofreak_row *d_out)
{
    if (!c || !counts || root < 0 || root >= c->world) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_gather_rows: bad argument");
    if (counts[c->rank] > 0 && !d_rows) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_gather_rows: null rows");
    if (c->rank == root) {
        int64_t total = 0;
        for (int r = 0; r < c->world; ++r) total += counts[r];
        if (total > 0 && !d_out) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_gather_rows: null output on the root");
    }
    RcclTransport t{c};
    return mofreak_dist::gather_rows(t, c->rank, c->world, d_rows, counts, root, d_out, (int64_t)sizeof(mofreak_row));
}

int mofreak_comm_self_exchange(mofreak_comm *c, int64_t n_bytes)
{
    constexpr int64_t kMaxBytes = (int64_t)1 << 30;  // a self-test, not a transport: a gigabyte is plenty
    if (!c || n_bytes <= 0 || n_bytes > kMaxBytes) return fail(MOFREAK_ERR_BAD_ARG, "mofreak_comm_self_exchange: bad argument");
    uint8_t *d = nullptr;
    int rc = MOFREAK_OK;
    try {  // (no exception leaves this file: the vectors' allocations included)
        std::vector<uint8_t> h((size_t)n_bytes), back((size_t)n_bytes, 0);
        for (int64_t i = 0; i < n_bytes; ++i) h[(size_t)i] = (uint8_t)(i * 131 + 7);
        if (hipMalloc((void **)&d, (size_t)2 * n_bytes) != hipSuccess) return fail(MOFREAK_ERR_OOM, "self exchange: device allocation failed");
        RcclTransport t{c};
        if (hipMemcpyAsync(d, h.data(), (size_t)n_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(d + n_bytes, 0, (size_t)n_bytes, c->stream) != hipSuccess)
            rc = fail(MOFREAK_ERR_HIP, "self exchange: upload failed");
        if (!rc && !(rc = t.group_start())) {
            int a = t.send(d, n_bytes, c->rank), b = t.recv(d + n_bytes, n_bytes, c->rank), e = t.group_end();
            rc = a ? a : b ? b : e;
        }
        if (!rc) rc = t.sync();
        if (!rc && hipMemcpy(back.data(), d + n_bytes, (size_t)n_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MOFREAK_ERR_HIP, "self exchange: download failed");
        if (!rc && back != h) rc = fail(MOFREAK_ERR_HIP, "self exchange: the bytes that came back differ");
    } catch (const std::bad_alloc &) {
        rc = fail(MOFREAK_ERR_OOM, "self exchange: host allocation failed");
    } catch (...) {
        rc = fail(MOFREAK_ERR_HIP, "self exchange: unexpected exception");
    }
    if (d) (void)hipFree(d);
    return rc;
}

}  // extern "C"
