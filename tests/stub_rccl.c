/* tests/stub_rccl.c — TEST INFRASTRUCTURE: a stand-in for librccl.so with the nine entry points sph-pie_amd/csrc/pie_comm.hip
 * opens, for ONE process that drives several "ranks" on ONE GPU.  RCCL itself refuses two ranks on one device, and the GPU boxes
 * have one MI355X, so this is what lets the communicator's own logic (buffer rotation, capacity negotiation from the gathered
 * headers, the pipelined step calls, error paths) run with world > 1 there.  ncclSend / ncclRecv posted inside one group are
 * matched by (source, destination) in posting order at ncclGroupEnd and become device-to-device copies on the receiver's stream,
 * ordered behind the sender's stream by an event — the stream semantics of the real thing.  Loaded through PIE_RCCL_LIB. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

typedef struct ncclComm { int rank, world; } *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;

#define MAX_OPS 4096
typedef struct { int is_send, src, dst; const void *sbuf; void *rbuf; size_t bytes; hipStream_t stream; } op_t;
static op_t g_ops[MAX_OPS];
static int g_n_ops = 0, g_depth = 0;

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return 0; }
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    (void)devs;
    for (int i = 0; i < n; ++i) {
        comms[i] = (ncclComm_t)malloc(sizeof(struct ncclComm));
        comms[i]->rank = i;
        comms[i]->world = n;
    }
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank)
{
    (void)id;
    if (world != 1) return 5; /* one process per rank needs the real library */
    *comm = (ncclComm_t)malloc(sizeof(struct ncclComm));
    (*comm)->rank = rank;
    (*comm)->world = world;
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "ok" : "stub_rccl: unsupported"; }
ncclResult_t ncclGroupStart(void) { ++g_depth; return 0; }

static size_t type_bytes(int t) { return t == 2 || t == 3 || t == 7 ? 4 : t == 0 || t == 1 ? 1 : 8; }

ncclResult_t ncclSend(const void *buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s)
{
    if (g_n_ops >= MAX_OPS || g_depth == 0) return 5;
    op_t o = {1, c->rank, peer, buf, NULL, count * type_bytes(type), s};
    g_ops[g_n_ops++] = o;
    return 0;
}
ncclResult_t ncclRecv(void *buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s)
{
    if (g_n_ops >= MAX_OPS || g_depth == 0) return 5;
    op_t o = {0, peer, c->rank, NULL, buf, count * type_bytes(type), s};
    g_ops[g_n_ops++] = o;
    return 0;
}
ncclResult_t ncclGroupEnd(void)
{
    if (--g_depth > 0) return 0;
    int rc = 0;
    for (int i = 0; i < g_n_ops && rc == 0; ++i) {
        if (g_ops[i].is_send != 1) continue;
        int j;
        for (j = 0; j < g_n_ops; ++j)
            if (g_ops[j].is_send == 0 && g_ops[j].src == g_ops[i].src && g_ops[j].dst == g_ops[i].dst) break;
        if (j == g_n_ops || g_ops[j].bytes != g_ops[i].bytes) { rc = 5; break; } /* an unmatched send: the real library would hang */
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = 1; break; }
        if (hipEventRecord(ev, g_ops[i].stream) != hipSuccess || hipStreamWaitEvent(g_ops[j].stream, ev, 0) != hipSuccess ||
            hipMemcpyAsync(g_ops[j].rbuf, g_ops[i].sbuf, g_ops[i].bytes, hipMemcpyDeviceToDevice, g_ops[j].stream) != hipSuccess) rc = 1;
        /* the sender's buffer may be rewritten once ITS stream moves on: make the sender wait for the copy */
        hipEvent_t ev2;
        if (rc == 0 && hipEventCreateWithFlags(&ev2, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(ev2, g_ops[j].stream) != hipSuccess || hipStreamWaitEvent(g_ops[i].stream, ev2, 0) != hipSuccess) rc = 1;
            (void)hipEventDestroy(ev2);
        }
        (void)hipEventDestroy(ev);
        g_ops[i].is_send = 2;
        g_ops[j].is_send = 2;
    }
    for (int i = 0; i < g_n_ops && rc == 0; ++i)
        if (g_ops[i].is_send != 2) rc = 5; /* an unmatched receive */
    g_n_ops = 0;
    return rc;
}
