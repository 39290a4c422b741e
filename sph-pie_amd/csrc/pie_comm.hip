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
#include <cstdlib>
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
    // PIE_RCCL_LIB: another library with the same entry points (tests/stub_rccl.c: several "ranks" of ONE process on ONE GPU,
    // which RCCL itself refuses — the stand-in that lets the communicator's logic run with world > 1 on a one-GPU box)
    const char* names[] = {getenv("PIE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
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
    long long need = 0;            // row capacity the last exchange called for (the largest list / union any rank reported)
    // pipelined union exchange (pie_comm_step_*): kSets rotating buffer sets per local rank, a side stream per local rank
    static constexpr int kSets = 16;  // (twelve steps begun on four batch lanes + the exchanges queued and being read)
    std::vector<hipStream_t> xstream;                 // local index -> exchange stream
    std::vector<int*> umsg[kSets], ugath[kSets];      // [set][local index]: this rank's union message / the gathered ones [world][UL]
    std::vector<hipEvent_t> ev_ready[kSets], ev_done[kSets];
    int* h_mu[kSets] = {}; // mapped pinned: [local index][world] the Mu word of every gathered message
    int* h_mu_dev[kSets] = {};
    long long u_cap = 0, UL = 0;   // union rows per message, words per message (u_pad + 2 + mask_words' share)
    int u_words = 2;               // words per union row in a message: row + one or two mask words
    long long begun = 0, finished = 0, collected = 0;  // steps begun / exchanges issued / exchanges collected
    int step_nq[kSets] = {};
    int step_rc[kSets] = {};   // a step whose finish failed on this process: its collect reports it
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

void free_step_buffers(pie_comm* c);

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
            if (device_ids[a] == device_ids[b] && !getenv("PIE_RCCL_LIB")) // (the test stand-in runs several shards on one GPU)
                return cfail(nullptr, PIE_E_INVAL, "device %d listed twice: one shard per GPU", device_ids[a]);
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
    if (!id_128 || world < 1 || world > 64 || rank < 0 || rank >= world) return cfail(nullptr, PIE_E_INVAL, "bad id / rank %d of %d (1..64 ranks)", rank, world);
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
    for (int k = 0; k < c->n_local; ++k) {
        (void)hipSetDevice(c->device[k]);
        if (k < (int)c->xstream.size() && c->xstream[k]) (void)hipStreamSynchronize(c->xstream[k]);
    }
    free_buffers(c);
    free_step_buffers(c);
    for (int k = 0; k < c->n_local; ++k) {
        (void)hipSetDevice(c->device[k]);
        for (int s = 0; s < pie_comm::kSets; ++s) {
            if (k < (int)c->ev_ready[s].size() && c->ev_ready[s][k]) (void)hipEventDestroy(c->ev_ready[s][k]);
            if (k < (int)c->ev_done[s].size() && c->ev_done[s][k]) (void)hipEventDestroy(c->ev_done[s][k]);
        }
        if (k < (int)c->xstream.size() && c->xstream[k]) (void)hipStreamDestroy(c->xstream[k]);
    }
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

namespace {

// the Mu word of every gathered message (world of them, rank_stride words apart) into mapped host memory: no copy node behind
// the exchange (a 2-D copy call cost more host time than the whole step's launches)
__global__ void k_pick_words(const int* __restrict__ gath, long long rank_stride, long long word, int world, int* __restrict__ out)
{
    const int r = (int)threadIdx.x;
    if (r < world) __hip_atomic_store(&out[r], gath[(long long)r * rank_stride + word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the direct exchange of one fixed-length message per rank: inside one group every local rank posts one send and one
// receive per peer (each message crosses its own xGMI link once); its own message is a local copy on the same stream
int exchange(pie_comm* c, const std::vector<int*>& msg, const std::vector<int*>& gath, size_t count, size_t rank_stride, const std::vector<hipStream_t>& streams)
{
    if (c->world > 1) {
        PIE_CNCCL(c, g_rccl.GroupStart());
        for (int k = 0; k < c->n_local; ++k) {
            const int me = c->rank_of[k];
            for (int p = 0; p < c->world; ++p) {
                if (p == me) continue;
                ncclResult_t r1 = g_rccl.Send(msg[k], count, kNcclInt32, p, c->comm[k], streams[k]);
                ncclResult_t r2 = g_rccl.Recv(gath[k] + (size_t)p * rank_stride, count, kNcclInt32, p, c->comm[k], streams[k]);
                if (r1 != 0 || r2 != 0) {
                    (void)g_rccl.GroupEnd();
                    return cfail(c, PIE_E_HIP, "ncclSend/ncclRecv: %s", g_rccl.GetErrorString(r1 != 0 ? r1 : r2));
                }
            }
        }
        PIE_CNCCL(c, g_rccl.GroupEnd());
    }
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        PIE_CHIP(c, hipMemcpyAsync(gath[k] + (size_t)c->rank_of[k] * rank_stride, msg[k], count * 4, hipMemcpyDeviceToDevice, streams[k]));
    }
    return PIE_OK;
}

// u_pad every rank uses: the largest shard's user count (single process), or the value the ranks agreed on
int resolve_u_pad(pie_comm* c, int32_t u_pad_in, long long* u_pad_out)
{
    long long u_pad = u_pad_in;
    for (int k = 0; k < c->n_local; ++k) {
        pie_stats st;
        st.struct_size = sizeof(pie_stats);
        PIE_CCTX(c, k, pie_stats_get(c->ctx[k], &st));
        if (u_pad_in <= 0 && (long long)st.users > u_pad) u_pad = (long long)st.users;
        if ((long long)st.users > u_pad) return cfail(c, PIE_E_INVAL, "rank %d holds %llu users, above u_pad %lld", c->rank_of[k], (unsigned long long)st.users, u_pad);
    }
    if (c->n_local != c->world && u_pad_in <= 0) return cfail(c, PIE_E_INVAL, "process-per-GPU communicator: pass the agreed u_pad");
    *u_pad_out = u_pad;
    return PIE_OK;
}

void free_step_buffers(pie_comm* c)
{
    for (int s = 0; s < pie_comm::kSets; ++s) {
        for (int k = 0; k < c->n_local; ++k) {
            (void)hipSetDevice(c->device[k]);
            if (k < (int)c->umsg[s].size() && c->umsg[s][k]) (void)hipFree(c->umsg[s][k]);
            if (k < (int)c->ugath[s].size() && c->ugath[s][k]) (void)hipFree(c->ugath[s][k]);
        }
        c->umsg[s].assign((size_t)c->n_local, nullptr);
        c->ugath[s].assign((size_t)c->n_local, nullptr);
        if (c->h_mu[s]) (void)hipHostFree(c->h_mu[s]);
        c->h_mu[s] = nullptr;
    }
    c->u_cap = c->UL = 0;
}

// rotating buffer sets of the pipelined union exchange; only while no step is in flight
int ensure_step_buffers(pie_comm* c, int n_q, long long u_pad, long long cap)
{
    const int words = n_q > 32 ? 3 : 2;
    if (c->UL > 0 && c->u_pad == u_pad && cap <= c->u_cap && words <= c->u_words && c->umsg[0].size() == (size_t)c->n_local && c->umsg[0][0]) return PIE_OK;
    if (c->begun != c->collected) return cfail(c, PIE_E_STATE, "steps are in flight: collect them before the message size changes");
    free_step_buffers(c);
    c->u_pad = u_pad;
    c->u_cap = cap;
    c->u_words = words > c->u_words ? words : c->u_words;
    c->UL = u_pad + 2 + (long long)c->u_words * cap;
    if (c->xstream.size() != (size_t)c->n_local) c->xstream.assign((size_t)c->n_local, nullptr);
    for (int s = 0; s < pie_comm::kSets; ++s) {
        if (c->ev_ready[s].size() != (size_t)c->n_local) { c->ev_ready[s].assign((size_t)c->n_local, nullptr); c->ev_done[s].assign((size_t)c->n_local, nullptr); }
        PIE_CHIP(c, hipHostMalloc(&c->h_mu[s], (size_t)c->n_local * (size_t)c->world * 4, hipHostMallocMapped));
        PIE_CHIP(c, hipHostGetDevicePointer((void**)&c->h_mu_dev[s], c->h_mu[s], 0));
    }
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        if (!c->xstream[k]) PIE_CHIP(c, hipStreamCreateWithFlags(&c->xstream[k], hipStreamNonBlocking));
        for (int s = 0; s < pie_comm::kSets; ++s) {
            PIE_CHIP(c, hipMalloc(&c->umsg[s][k], (size_t)c->UL * 4));
            PIE_CHIP(c, hipMalloc(&c->ugath[s][k], (size_t)c->UL * 4 * (size_t)c->world));
            PIE_CHIP(c, hipMemset(c->umsg[s][k], 0, (size_t)c->UL * 4));
            if (!c->ev_ready[s][k]) PIE_CHIP(c, hipEventCreateWithFlags(&c->ev_ready[s][k], hipEventDisableTiming));
            if (!c->ev_done[s][k]) PIE_CHIP(c, hipEventCreateWithFlags(&c->ev_done[s][k], hipEventDisableTiming));
        }
    }
    return PIE_OK;
}

} // namespace

// One synchronous step, per-query lists.  The exchange ALWAYS runs once every local shard has finished its batch — the
// messages have the fixed length L and simply truncate a list that outgrew the capacity (M stays in the header) — and the
// capacity check is made on the GATHERED headers, which every rank sees alike: all ranks return PIE_E_CAPACITY together
// (pie_comm_needed_cap says what to reserve), none is left waiting inside ncclSend / ncclRecv for a peer that bailed out
// (ADVICE r02).  An error on one local shard still finishes the batches begun on the others before it is reported.
int pie_comm_scan_batch_gather(pie_comm* c, const pie_query* queries, int32_t n_q, int32_t u_pad_in, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (!queries || n_q < 1 || n_q > PIE_BATCH_MAX) return cfail(c, PIE_E_INVAL, "a batch holds 1..%d queries (got %d)", PIE_BATCH_MAX, n_q);
    if (c->begun != c->collected) return cfail(c, PIE_E_STATE, "pipelined steps are in flight (pie_comm_step_*): collect them first");
    long long u_pad = 0;
    int rc = resolve_u_pad(c, u_pad_in, &u_pad);
    if (rc) return rc;
    std::vector<size_t> m((size_t)c->n_local * (size_t)n_q, 0);
    long long cap = c->cap > 0 ? c->cap : 1024;
    std::vector<int> heads((size_t)c->world * (size_t)n_q);
    for (int attempt = 0; attempt < 3; ++attempt) {
        rc = ensure_buffers(c, n_q, u_pad, cap);
        if (rc) return rc;
        // 1. every shard scans its batch and leaves its n_q messages back to back in msg[k]
        int begun = 0, first_rc = PIE_OK;
        for (int k = 0; k < c->n_local && first_rc == PIE_OK; ++k) {
            first_rc = pie_scan_batch_begin_packed(c->ctx[k], queries, n_q, c->msg[k], (size_t)c->L, (size_t)c->u_pad, (size_t)c->cap, nullptr, 0);
            if (first_rc == PIE_OK) ++begun;
            else cfail(c, first_rc, "rank %d: %s", c->rank_of[k], pie_last_error(c->ctx[k]));
        }
        for (int k = 0; k < begun; ++k) { // every batch that was begun is finished, whatever happened elsewhere
            int ready = 0;
            const int r2 = pie_scan_batch_finish_packed(c->ctx[k], &m[(size_t)k * n_q], &ready);
            if (r2 != PIE_OK && first_rc == PIE_OK) {
                first_rc = r2;
                cfail(c, r2, "rank %d: %s", c->rank_of[k], pie_last_error(c->ctx[k]));
            }
        }
        if (first_rc != PIE_OK) return first_rc; // a local failure (bad argument, HIP error): no rank of THIS process entered the exchange
        // 2. the exchange: always
        const size_t count = (size_t)n_q * (size_t)c->L;
        rc = exchange(c, c->msg, c->gath, count, (size_t)c->q_max * (size_t)c->L, c->stream);
        if (rc) return rc;
        // 3. every rank's M words, as gathered at the first local rank: the same numbers on every rank of the communicator
        PIE_CHIP(c, hipSetDevice(c->device[0]));
        for (int p = 0; p < c->world; ++p)
            PIE_CHIP(c, hipMemcpy2DAsync(&heads[(size_t)p * n_q], 4, c->gath[0] + (size_t)p * (size_t)c->q_max * (size_t)c->L + (size_t)c->u_pad + 1,
                                         (size_t)c->L * 4, 4, (size_t)n_q, hipMemcpyDeviceToHost, c->stream[0]));
        for (int k = 0; k < c->n_local; ++k) {
            PIE_CHIP(c, hipSetDevice(c->device[k]));
            PIE_CHIP(c, hipStreamSynchronize(c->stream[k]));
        }
        long long need = 0;
        for (int v : heads) need = v > need ? v : need;
        c->need = need;
        c->last_nq = n_q;
        if (need <= c->cap) break;
        if (c->n_local != c->world || attempt == 2)
            return cfail(c, PIE_E_CAPACITY, "a row list of %lld rows exceeds the reserved capacity %lld on some rank: every rank calls pie_comm_reserve(pie_comm_needed_cap) and repeats the step", need, c->cap);
        cap = need + need / 16 + 64; // a single process owns every rank: grow and run the step again
    }
    if (m_out) memcpy(m_out, m.data(), m.size() * sizeof(size_t));
    return PIE_OK;
}

size_t pie_comm_needed_cap(const pie_comm* c) { return c ? (size_t)(c->need + c->need / 16 + 64) : 0; }

// ---- the pipelined exchange: ONE union message per step, written by the batch's own tail kernel; the exchange of a step
// runs on a side stream while the shards scan the next ones (the C-ABI counterpart of shard.py's BatchedFeeds.run_steps)
int pie_comm_step_reserve(pie_comm* c, int32_t n_q, int32_t u_pad_in, size_t union_cap)
{
    if (!c) return PIE_E_INVAL;
    if (n_q < 1 || n_q > PIE_BATCH_MAX) return cfail(c, PIE_E_INVAL, "bad reservation");
    long long u_pad = 0;
    int rc = resolve_u_pad(c, u_pad_in, &u_pad);
    if (rc) return rc;
    return ensure_step_buffers(c, n_q, u_pad, (long long)(union_cap > 0 ? union_cap : 1024));
}

int pie_comm_step_begin(pie_comm* c, const pie_query* queries, int32_t n_q)
{
    if (!c) return PIE_E_INVAL;
    if (!queries || n_q < 1 || n_q > PIE_BATCH_MAX) return cfail(c, PIE_E_INVAL, "a batch holds 1..%d queries (got %d)", PIE_BATCH_MAX, n_q);
    if (c->UL <= 0) return cfail(c, PIE_E_STATE, "pie_comm_step_reserve first");
    if ((n_q > 32 ? 3 : 2) > c->u_words) return cfail(c, PIE_E_STATE, "reserved for batches of at most 32 queries: pie_comm_step_reserve again");
    // steps begun and unfinished: what the shards' batch lanes hold (three per lane, pie_set_batch_lanes; lane 0 only on the ordered run)
    if (c->begun - c->finished >= pie_comm::kSets - 4)
        return cfail(c, PIE_E_STATE, "%d steps are already begun: pie_comm_step_finish first", pie_comm::kSets - 4);
    for (int k = 0; k < c->n_local; ++k)
        if (pie_batch_room(c->ctx[k]) <= 0)
            return cfail(c, PIE_E_STATE, "rank %d: the steps begun fill its batch slots (three per batch lane): pie_comm_step_finish first", c->rank_of[k]);
    if (c->begun - c->collected >= pie_comm::kSets) return cfail(c, PIE_E_STATE, "%d steps are uncollected: pie_comm_step_collect first", pie_comm::kSets);
    const int s = (int)(c->begun % pie_comm::kSets);
    int begun = 0, rc = PIE_OK;
    for (int k = 0; k < c->n_local && rc == PIE_OK; ++k) {
        rc = pie_scan_batch_begin_union(c->ctx[k], queries, n_q, c->umsg[s][k], (size_t)c->u_pad, (size_t)c->u_cap);
        if (rc == PIE_OK) ++begun;
        else cfail(c, rc, "rank %d: %s", c->rank_of[k], pie_last_error(c->ctx[k]));
    }
    if (rc != PIE_OK) { // the step is not counted; what the other shards began is finished and dropped
        for (int k = 0; k < begun; ++k) (void)pie_scan_batch_finish_packed(c->ctx[k], nullptr, nullptr);
        return rc;
    }
    c->step_nq[s] = n_q;
    c->begun++;
    return PIE_OK;
}

int pie_comm_step_finish(pie_comm* c, size_t* m_out)
{
    if (!c) return PIE_E_INVAL;
    if (c->finished >= c->begun) return cfail(c, PIE_E_STATE, "pie_comm_step_finish without pie_comm_step_begin");
    const int s = (int)(c->finished % pie_comm::kSets);
    const int n_q = c->step_nq[s];
    std::vector<size_t> m((size_t)c->n_local * (size_t)n_q, 0);
    int first_rc = PIE_OK;
    for (int k = 0; k < c->n_local; ++k) {
        int ready = 0;
        const int rc = pie_scan_batch_finish_packed(c->ctx[k], &m[(size_t)k * n_q], &ready);
        if (rc != PIE_OK && first_rc == PIE_OK) {
            first_rc = rc;
            cfail(c, rc, "rank %d: %s", c->rank_of[k], pie_last_error(c->ctx[k]));
        }
        // ready = 0: the message was packed on the context's stream after the batch: the exchange stream waits for exactly that
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        if (!ready) {
            PIE_CHIP(c, hipEventRecord(c->ev_ready[s][k], c->stream[k]));
            PIE_CHIP(c, hipStreamWaitEvent(c->xstream[k], c->ev_ready[s][k], 0));
        }
    }
    c->finished++;
    c->step_rc[s] = first_rc;
    if (first_rc != PIE_OK) return first_rc; // every local batch has been finished; no exchange is queued for this step (its collect says so)
    // the exchange of this step, on the side streams: nothing here waits for it
    int rc = exchange(c, c->umsg[s], c->ugath[s], (size_t)c->UL, (size_t)c->UL, c->xstream);
    if (rc) return rc;
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        hipLaunchKernelGGL(k_pick_words, dim3(1), dim3(64), 0, c->xstream[k], c->ugath[s][k], (long long)c->UL, (long long)c->u_pad + 1, c->world,
                           c->h_mu_dev[s] + (size_t)k * (size_t)c->world);
        PIE_CHIP(c, hipEventRecord(c->ev_done[s][k], c->xstream[k]));
    }
    if (m_out) memcpy(m_out, m.data(), m.size() * sizeof(size_t));
    return PIE_OK;
}

int pie_comm_step_collect(pie_comm* c, int64_t* step_out)
{
    if (!c) return PIE_E_INVAL;
    if (c->collected >= c->finished) return cfail(c, PIE_E_STATE, "no exchange to collect: pie_comm_step_finish first");
    const int s = (int)(c->collected % pie_comm::kSets);
    if (step_out) *step_out = (int64_t)c->collected;
    if (c->step_rc[s] != PIE_OK) {
        c->collected++;
        return cfail(c, PIE_E_STATE, "step %lld failed in pie_comm_step_finish (status %d): nothing was exchanged", (long long)c->collected - 1, c->step_rc[s]);
    }
    for (int k = 0; k < c->n_local; ++k) {
        PIE_CHIP(c, hipSetDevice(c->device[k]));
        PIE_CHIP(c, hipEventSynchronize(c->ev_done[s][k])); // the side stream only: the shards' scan streams run on
    }
    c->collected++;
    long long need = 0;
    bool merged_overflow = false;
    for (int p = 0; p < c->world; ++p) { // every rank holds the same world Mu words
        const int mu = c->h_mu[s][p];
        if (mu < 0) merged_overflow = true;
        if (mu > need) need = mu;
    }
    c->need = need;
    if (merged_overflow)
        return cfail(c, PIE_E_CAPACITY, "a shard's union could not be formed (Mu = -1: queries fell back and a user's merged union exceeds 32 rows, or more than 32 queries): use pie_comm_scan_batch_gather for this batch");
    if (need > c->u_cap)
        return cfail(c, PIE_E_CAPACITY, "a union of %lld rows exceeds the reserved capacity %lld on some rank: collect what is in flight, pie_comm_step_reserve(pie_comm_needed_cap), repeat the step", need, c->u_cap);
    return PIE_OK;
}

int pie_comm_step_gathered_ptr(pie_comm* c, int32_t at_rank, int64_t step, void** base_out, size_t* rank_stride_words, size_t* u_pad_out, size_t* cap_out)
{
    if (!c) return PIE_E_INVAL;
    const int k = local_index(c, at_rank);
    if (k < 0 || c->UL <= 0) return cfail(c, PIE_E_STATE, "rank %d is not local to this communicator or nothing was reserved", at_rank);
    if (step < 0 || step >= c->collected || step + pie_comm::kSets <= c->begun) return cfail(c, PIE_E_STATE, "step %lld is not collected or its buffers were reused", (long long)step);
    if (base_out) *base_out = c->ugath[step % pie_comm::kSets][k];
    if (rank_stride_words) *rank_stride_words = (size_t)c->UL;
    if (u_pad_out) *u_pad_out = (size_t)c->u_pad;
    if (cap_out) *cap_out = (size_t)c->u_cap;
    return PIE_OK;
}

int pie_comm_step_read_gathered(pie_comm* c, int32_t at_rank, int32_t src_rank, int64_t step, int32_t* uoff_out, int32_t* rows_out, uint64_t* masks_out,
                                size_t cap, size_t* mu_out)
{
    if (!c) return PIE_E_INVAL;
    void* base = nullptr;
    int rc = pie_comm_step_gathered_ptr(c, at_rank, step, &base, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (src_rank < 0 || src_rank >= c->world) return cfail(c, PIE_E_INVAL, "source rank outside the communicator");
    const int k = local_index(c, at_rank);
    PIE_CHIP(c, hipSetDevice(c->device[k]));
    const int* msg = static_cast<const int*>(base) + (size_t)src_rank * (size_t)c->UL;
    int mu32 = 0;
    PIE_CHIP(c, hipMemcpy(&mu32, msg + c->u_pad + 1, 4, hipMemcpyDeviceToHost));
    const size_t mu = mu32 > 0 ? (size_t)mu32 : 0;
    if (mu_out) *mu_out = mu;
    if (uoff_out) PIE_CHIP(c, hipMemcpy(uoff_out, msg, ((size_t)c->u_pad + 1) * 4, hipMemcpyDeviceToHost));
    const size_t have = mu < (size_t)c->u_cap ? mu : (size_t)c->u_cap;
    if ((rows_out || masks_out) && have > cap) return cfail(c, PIE_E_CAPACITY, "cap %zu < %zu union rows", cap, have);
    if (rows_out && have) PIE_CHIP(c, hipMemcpy(rows_out, msg + c->u_pad + 2, have * 4, hipMemcpyDeviceToHost));
    if (masks_out && have) {
        std::vector<uint32_t> lo(have), hi;
        PIE_CHIP(c, hipMemcpy(lo.data(), msg + c->u_pad + 2 + c->u_cap, have * 4, hipMemcpyDeviceToHost));
        if (c->u_words == 3 && c->step_nq[step % pie_comm::kSets] > 32) {
            hi.resize(have);
            PIE_CHIP(c, hipMemcpy(hi.data(), msg + c->u_pad + 2 + 2 * c->u_cap, have * 4, hipMemcpyDeviceToHost));
        }
        for (size_t i = 0; i < have; ++i) masks_out[i] = (uint64_t)lo[i] | (hi.empty() ? 0ull : ((uint64_t)hi[i] << 32));
    }
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
