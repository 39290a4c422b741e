// pie_comm.hip — the multi-GPU exchange behind the C ABI (include/pie_scan.h, "communicator" section): user-hash shards
// of one session table on the GPUs of one node, one batched scan per shard, and the reassembly of cross-user feeds by
// RCCL over xGMI — callable from a host that is not Python (the Node addon binds these entry points).
//
// Built on the public ABI only (a pie_ctx per GPU) plus RCCL, which is opened at run time (dlopen "librccl.so.1": the
// scan library itself stays loadable on a box without RCCL; pie_comm_create then fails with PIE_E_NODEVICE).
// xGMI is point-to-point (7 links per GPU), so the exchange is the DIRECT pattern: inside one ncclGroupStart / End every
// rank posts one ncclSend per peer and one ncclRecv per peer — each shard's message crosses its own link to each peer
// once — rather than a ring that would bound the step by one link.
#include "../../include/pie_scan.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace {

// the slice of the RCCL API this file uses (signatures of /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclInt32 = 2;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

bool load_rccl(char* err, size_t errlen)
{
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        snprintf(err, errlen, "RCCL not found (%s)", dlerror());
        return false;
    }
    Rccl r;
    r.handle = h;
#define PIE_SYM(field, name)                                            \
    *(void**)(&r.field) = dlsym(h, name);                               \
    if (!r.field) {                                                     \
        snprintf(err, errlen, "RCCL symbol %s missing", name);          \
        return false;                                                   \
    }
    PIE_SYM(GetUniqueId, "ncclGetUniqueId")
    PIE_SYM(CommInitRank, "ncclCommInitRank")
    PIE_SYM(CommInitAll, "ncclCommInitAll")
    PIE_SYM(CommDestroy, "ncclCommDestroy")
    PIE_SYM(Send, "ncclSend")
    PIE_SYM(Recv, "ncclRecv")
    PIE_SYM(GroupStart, "ncclGroupStart")
    PIE_SYM(GroupEnd, "ncclGroupEnd")
    PIE_SYM(GetErrorString, "ncclGetErrorString")
#undef PIE_SYM
    g_rccl = r;
    return true;
}

thread_local char g_comm_create_error[256] = "";

} // namespace

// One member per LOCAL rank: a single-process communicator (pie_comm_create) holds all ranks of the node, a
// process-per-GPU communicator (pie_comm_create_rank) holds one.
struct pie_comm {
    int world = 0;                 // ranks in the communicator
    int n_local = 0;               // ranks this process drives
    std::vector<int> rank_of;      // local index -> rank
    std::vector<int> device;       // local index -> HIP device
    std::vector<pie_ctx*> ctx;     // local index -> scan context (owned)
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;   // the context's stream: results are produced and exchanged in stream order
    std::vector<int*> msg;         // local index -> this rank's messages   [q_max][L]
    std::vector<int*> gath;        // local index -> gathered messages      [world][q_max][L]
    int q_max = 0;
    long long u_pad = 0, cap = 0, L = 0;
    int last_nq = 0;
    char err[512] = "";
};

namespace {

int cfail(pie_comm* c, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (c) vsnprintf(c->err, sizeof c->err, fmt, ap);
    else vsnprintf(g_comm_create_error, sizeof g_comm_create_error, fmt, ap);
    va_end(ap);
    return code;
}

#define PIE_CHIP(c, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return cfail((c), e_ == hipErrorOutOfMemory ? PIE_E_NOMEM : PIE_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define PIE_CNCCL(c, call)                                                                                      \
    do {                                                                                                        \
        ncclResult_t r_ = (call);                                                                               \
        if (r_ != 0) return cfail((c), PIE_E_HIP, "%s: %s", #call, g_rccl.GetErrorString(r_));                  \
    } while (0)
#define PIE_CCTX(c, k, call)                                                                                    \
    do {                                                                                                        \
        int rc_ = (call);                                                                                       \
        if (rc_ != PIE_OK) return cfail((c), rc_, "rank %d: %s", (c)->rank_of[k], pie_last_error((c)->ctx[k])); \
    } while (0)

void free_buffers(pie_comm* c)
{
    for (int k = 0; k < c->n_local; ++k) {
        (void)hipSetDevice(c->device[k]);
        if (k < (int)c->msg.size() && c->msg[k]) (void)hipFree(c->msg[k]);
        if (k < (int)c->gath.size() && c->gath[k]) (void)hipFree(c->gath[k]);
    }
    c->msg.assign((size_t)c->n_local, nullptr);
    c->gath.assign((size_t)c->n_local, nullptr);
    c->cap = c->L = 0;
}

int ensure_buffers(pie_comm* c, int n_q, long long u_pad, long long cap)
{
    if (c->msg.size() == (size_t)c->n_local && c->msg[0] && n_q <= c->q_max && u_pad == c->u_pad && cap <= c->cap) return PIE_OK;
    free_buffers(c);
    c->q_max = n_q > c->q_max ? n_q : c->q_max;
    c->u_pad = u_pad;
    c->cap = cap;
    c->L = u_pad + 2 + cap;
    const size_t words = (size_t)c->q_max * (size_t)c->L;
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        PIE_CHIP(c, hipMalloc(&c->msg[k], words * 4));
        PIE_CHIP(c, hipMalloc(&c->gath[k], words * 4 * (size_t)c->world));
        PIE_CHIP(c, hipMemset(c->msg[k], 0, words * 4));
    }
    return PIE_OK;
}

pie_comm* new_comm(int world, int n_local)
{
    pie_comm* c = new (std::nothrow) pie_comm();
    if (!c) return nullptr;
    c->world = world;
    c->n_local = n_local;
    c->rank_of.assign((size_t)n_local, 0);
    c->device.assign((size_t)n_local, 0);
    c->ctx.assign((size_t)n_local, nullptr);
    c->comm.assign((size_t)n_local, nullptr);
    c->stream.assign((size_t)n_local, nullptr);
    c->msg.assign((size_t)n_local, nullptr);
    c->gath.assign((size_t)n_local, nullptr);
    return c;
}

int local_index(const pie_comm* c, int rank)
{
    for (int k = 0; k < c->n_local; ++k)
        if (c->rank_of[k] == rank) return k;
    return -1;
}

} // namespace

extern "C" {

const char* pie_comm_last_error(const pie_comm* c) { return c ? c->err : g_comm_create_error; }

int pie_comm_create(const int32_t* device_ids, int32_t n, pie_comm** comm_out)
{
    if (!comm_out) return cfail(nullptr, PIE_E_INVAL, "comm_out is NULL");
    *comm_out = nullptr;
    if (!device_ids || n < 1 || n > 64) return cfail(nullptr, PIE_E_INVAL, "1..64 devices expected (got %d)", n);
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (device_ids[a] == device_ids[b]) return cfail(nullptr, PIE_E_INVAL, "device %d listed twice: one shard per GPU", device_ids[a]);
    char why[200];
    if (!load_rccl(why, sizeof why)) return cfail(nullptr, PIE_E_NODEVICE, "%s", why);
    pie_comm* c = new_comm(n, n);
    if (!c) return cfail(nullptr, PIE_E_NOMEM, "out of host memory");
    for (int k = 0; k < n; ++k) {
        c->rank_of[k] = k;
        c->device[k] = device_ids[k];
        int rc = pie_ctx_create(device_ids[k], &c->ctx[k]);
        if (rc != PIE_OK) {
            cfail(nullptr, rc, "shard %d on device %d: %s", k, device_ids[k], pie_last_error(nullptr));
            pie_comm_destroy(c);
            return rc;
        }
        void* s = nullptr;
        (void)pie_ctx_aux_stream(c->ctx[k], &s);
        c->stream[k] = (hipStream_t)s;
    }
    std::vector<int> devs(device_ids, device_ids + n);
    ncclResult_t r = g_rccl.CommInitAll(c->comm.data(), n, devs.data());
    if (r != 0) {
        cfail(nullptr, PIE_E_HIP, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
        for (auto& cm : c->comm) cm = nullptr;
        pie_comm_destroy(c);
        return PIE_E_HIP;
    }
    *comm_out = c;
    return PIE_OK;
}

int pie_comm_unique_id(void* id_out_128)
{
    if (!id_out_128) return cfail(nullptr, PIE_E_INVAL, "id_out is NULL");
    char why[200];
    if (!load_rccl(why, sizeof why)) return cfail(nullptr, PIE_E_NODEVICE, "%s", why);
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != 0) return cfail(nullptr, PIE_E_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    memcpy(id_out_128, id.internal, sizeof id.internal);
    return PIE_OK;
}

int pie_comm_create_rank(const void* id_128, int32_t rank, int32_t world, int32_t device_id, pie_comm** comm_out)
{
    if (!comm_out) return cfail(nullptr, PIE_E_INVAL, "comm_out is NULL");
    *comm_out = nullptr;
    if (!id_128 || world < 1 || rank < 0 || rank >= world) return cfail(nullptr, PIE_E_INVAL, "bad id / rank %d of %d", rank, world);
    char why[200];
    if (!load_rccl(why, sizeof why)) return cfail(nullptr, PIE_E_NODEVICE, "%s", why);
    pie_comm* c = new_comm(world, 1);
    if (!c) return cfail(nullptr, PIE_E_NOMEM, "out of host memory");
    c->rank_of[0] = rank;
    c->device[0] = device_id;
    int rc = pie_ctx_create(device_id, &c->ctx[0]);
    if (rc != PIE_OK) {
        cfail(nullptr, rc, "rank %d on device %d: %s", rank, device_id, pie_last_error(nullptr));
        pie_comm_destroy(c);
        return rc;
    }
    void* s = nullptr;
    (void)pie_ctx_aux_stream(c->ctx[0], &s);
    c->stream[0] = (hipStream_t)s;
    ncclUniqueId id;
    memcpy(id.internal, id_128, sizeof id.internal);
    if (hipSetDevice(device_id) != hipSuccess) {
        cfail(nullptr, PIE_E_NODEVICE, "hipSetDevice(%d) failed", device_id);
        pie_comm_destroy(c);
        return PIE_E_NODEVICE;
    }
    ncclResult_t r = g_rccl.CommInitRank(&c->comm[0], world, id, rank);
    if (r != 0) {
        cfail(nullptr, PIE_E_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
        c->comm[0] = nullptr;
        pie_comm_destroy(c);
        return PIE_E_HIP;
    }
    *comm_out = c;
    return PIE_OK;
}

int pie_comm_destroy(pie_comm* c)
{
    if (!c) return PIE_OK;
    for (int k = 0; k < c->n_local; ++k) {
        (void)hipSetDevice(c->device[k]);
        if (c->stream[k]) (void)hipStreamSynchronize(c->stream[k]);
    }
    free_buffers(c);
    for (int k = 0; k < c->n_local; ++k) {
        (void)hipSetDevice(c->device[k]);
        if (c->comm[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm[k]);
        if (c->ctx[k]) (void)pie_ctx_destroy(c->ctx[k]);
    }
    delete c;
    return PIE_OK;
}

int32_t pie_comm_world(const pie_comm* c) { return c ? c->world : 0; }
int32_t pie_comm_local_ranks(const pie_comm* c) { return c ? c->n_local : 0; }

pie_ctx* pie_comm_ctx(pie_comm* c, int32_t rank)
{
    if (!c) return nullptr;
    const int k = local_index(c, rank);
    return k >= 0 ? c->ctx[k] : nullptr;
}

int pie_comm_gen_synthetic_sharded(pie_comm* c, uint64_t seed, int64_t n_total, int32_t n_users, int32_t n_disc, uint32_t flags)
{
    if (!c) return PIE_E_INVAL;
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CCTX(c, k, pie_gen_synthetic(c->ctx[k], seed, n_total, 0, n_total, n_users, n_disc, flags));
        size_t rows = 0;
        int32_t users = 0;
        PIE_CCTX(c, k, pie_shard_table(c->ctx[k], c->rank_of[k], c->world, &rows, &users));
    }
    return PIE_OK;
}

int pie_comm_scan_batch_gather(pie_comm* c, const pie_query* queries, int32_t n_q, int32_t u_pad_in, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (!queries || n_q < 1 || n_q > PIE_BATCH_MAX) return cfail(c, PIE_E_INVAL, "a batch holds 1..%d queries (got %d)", PIE_BATCH_MAX, n_q);
    // u_pad: the largest shard's user count — every message has one length.  A single-process communicator sees every
    // shard; ranks of a process-per-GPU communicator agree on it beforehand and pass it in (u_pad_in > 0).
    long long u_pad = u_pad_in;
    std::vector<pie_stats> st((size_t)c->n_local);
    for (int k = 0; k < c->n_local; ++k) {
        st[k].struct_size = sizeof(pie_stats);
        PIE_CCTX(c, k, pie_stats_get(c->ctx[k], &st[k]));
        if (u_pad_in <= 0 && (long long)st[k].users > u_pad) u_pad = (long long)st[k].users;
        if ((long long)st[k].users > u_pad) return cfail(c, PIE_E_INVAL, "rank %d holds %llu users, above u_pad %lld", c->rank_of[k], (unsigned long long)st[k].users, u_pad);
    }
    if (c->n_local != c->world && u_pad_in <= 0) return cfail(c, PIE_E_INVAL, "process-per-GPU communicator: pass the agreed u_pad");
    std::vector<size_t> m((size_t)c->n_local * (size_t)n_q, 0);
    long long cap = c->cap > 0 ? c->cap : 1024;
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = ensure_buffers(c, n_q, u_pad, cap);
        if (rc) return rc;
        // 1. every shard scans its batch; the scan's own kernels write the n_q messages back to back
        for (int k = 0; k < c->n_local; ++k)
            PIE_CCTX(c, k, pie_scan_batch_begin_packed(c->ctx[k], queries, n_q, c->msg[k], (size_t)c->L, (size_t)c->u_pad, (size_t)c->cap, nullptr, 0));
        long long need = 0;
        for (int k = 0; k < c->n_local; ++k) {
            int ready = 0;
            PIE_CCTX(c, k, pie_scan_batch_finish_packed(c->ctx[k], &m[(size_t)k * n_q], &ready));
            for (int q = 0; q < n_q; ++q)
                if ((long long)m[(size_t)k * n_q + q] > need) need = (long long)m[(size_t)k * n_q + q];
        }
        if (need <= c->cap) break;
        // a row list outgrew the messages (in a process-per-GPU communicator every rank must see the same `need`: the
        // caller keeps the capacity in step through pie_comm_reserve)
        if (c->n_local != c->world) return cfail(c, PIE_E_CAPACITY, "row list of %lld rows exceeds the reserved capacity %lld: pie_comm_reserve", need, c->cap);
        cap = need + need / 16 + 64;
    }
    // 2. the exchange, direct pattern: one send and one receive per peer, all inside one group
    const size_t count = (size_t)n_q * (size_t)c->L;
    PIE_CNCCL(c, g_rccl.GroupStart());
    for (int k = 0; k < c->n_local; ++k) {
        const int me = c->rank_of[k];
        for (int p = 0; p < c->world; ++p) {
            if (p == me) continue;
            ncclResult_t r1 = g_rccl.Send(c->msg[k], count, kNcclInt32, p, c->comm[k], c->stream[k]);
            ncclResult_t r2 = g_rccl.Recv(c->gath[k] + (size_t)p * (size_t)c->q_max * (size_t)c->L, count, kNcclInt32, p, c->comm[k], c->stream[k]);
            if (r1 != 0 || r2 != 0) {
                (void)g_rccl.GroupEnd();
                return cfail(c, PIE_E_HIP, "ncclSend/ncclRecv: %s", g_rccl.GetErrorString(r1 != 0 ? r1 : r2));
            }
        }
    }
    PIE_CNCCL(c, g_rccl.GroupEnd());
    for (int k = 0; k < c->n_local; ++k) { // own message: a local copy, same stream
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        PIE_CHIP(c, hipMemcpyAsync(c->gath[k] + (size_t)c->rank_of[k] * (size_t)c->q_max * (size_t)c->L, c->msg[k], count * 4, hipMemcpyDeviceToDevice, c->stream[k]));
    }
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        PIE_CHIP(c, hipStreamSynchronize(c->stream[k]));
    }
    c->last_nq = n_q;
    if (m_out) memcpy(m_out, m.data(), m.size() * sizeof(size_t));
    return PIE_OK;
}

int pie_comm_reserve(pie_comm* c, int32_t n_q, int32_t u_pad, size_t idx_cap)
{
    if (!c) return PIE_E_INVAL;
    if (n_q < 1 || n_q > PIE_BATCH_MAX || u_pad < 1) return cfail(c, PIE_E_INVAL, "bad reservation");
    return ensure_buffers(c, n_q, u_pad, (long long)idx_cap);
}

int pie_comm_gathered_device_ptr(pie_comm* c, int32_t at_rank, void** base_out, size_t* rank_stride_words, size_t* query_stride_words,
                                 size_t* u_pad_out)
{
    if (!c) return PIE_E_INVAL;
    const int k = local_index(c, at_rank);
    if (k < 0 || !c->gath[k]) return cfail(c, PIE_E_STATE, "rank %d is not local to this communicator or nothing was gathered yet", at_rank);
    if (base_out) *base_out = c->gath[k];
    if (rank_stride_words) *rank_stride_words = (size_t)c->q_max * (size_t)c->L;
    if (query_stride_words) *query_stride_words = (size_t)c->L;
    if (u_pad_out) *u_pad_out = (size_t)c->u_pad;
    return PIE_OK;
}

int pie_comm_read_gathered(pie_comm* c, int32_t at_rank, int32_t src_rank, int32_t qi, int32_t* offsets_out, int32_t* idx_out, size_t idx_cap,
                           size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    const int k = local_index(c, at_rank);
    if (k < 0 || !c->gath[k]) return cfail(c, PIE_E_STATE, "rank %d is not local to this communicator or nothing was gathered yet", at_rank);
    if (src_rank < 0 || src_rank >= c->world || qi < 0 || qi >= c->last_nq) return cfail(c, PIE_E_INVAL, "source rank / query outside the last exchange");
    PIE_CHIP(c, hipSetDevice(c->device[k]));
    const int* msg = c->gath[k] + ((size_t)src_rank * (size_t)c->q_max + (size_t)qi) * (size_t)c->L;
    int m32 = 0;
    PIE_CHIP(c, hipMemcpy(&m32, msg + c->u_pad + 1, 4, hipMemcpyDeviceToHost));
    if (m_out) *m_out = (size_t)m32;
    if (offsets_out) PIE_CHIP(c, hipMemcpy(offsets_out, msg, ((size_t)c->u_pad + 1) * 4, hipMemcpyDeviceToHost));
    if (idx_out) {
        if ((size_t)m32 > idx_cap) return cfail(c, PIE_E_CAPACITY, "idx_cap %zu < %d rows", idx_cap, m32);
        if (m32 > 0) PIE_CHIP(c, hipMemcpy(idx_out, msg + c->u_pad + 2, (size_t)m32 * 4, hipMemcpyDeviceToHost));
    }
    return PIE_OK;
}

} // extern "C"
