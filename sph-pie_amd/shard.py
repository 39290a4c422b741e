"""User-hash sharding of the session table over the GPUs of one node and the all-gather that reassembles
cross-user feeds.  One process per GPU; torch.distributed is the transport (backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU for the world_size-2 tests).

Rows of different users are independent, so each rank scans its shard with no communication.  The one exchange
step gathers, per rank, ONE contiguous int32 message
    [ off[0..U_pad] (exclusive offsets, = M past the shard's last user) | M | rows[0..cap) ]
packed by a single kernel launch (`pie_pack_results_device`); Feed(rank r, local user u) =
rows[r, off[r,u] : off[r,u+1]] with no further arithmetic on the receiving side.  xGMI is point-to-point and these
messages are small (≈2 MB), so the gather is latency-bound: it is issued on a side stream and overlaps the next
scan; the host never waits on the scan stream for it, and the per-step host work is a handful of calls.
"""
import numpy as np
import torch
import torch.distributed as dist

from .binding import PieScan, shard_of


def partition_by_user_hash(start, end, user, disc, n_users, world):
    """Host-side load-time partition.  -> list (one per rank) of dicts with local columns,
    `rows` (local row -> global row) and `users` (local dense user -> global user, ascending)."""
    user = np.asarray(user, np.int32)
    owner_of_user = np.array([shard_of(u, world) for u in range(n_users)], np.int32)
    owner = owner_of_user[user] if user.size else np.zeros(0, np.int32)
    shards = []
    for r in range(world):
        users_r = np.nonzero(owner_of_user == r)[0].astype(np.int32)
        local_of_global = np.full(n_users, -1, np.int32)
        local_of_global[users_r] = np.arange(users_r.size, dtype=np.int32)
        rows = np.nonzero(owner == r)[0].astype(np.int64)
        shards.append({
            "start": np.asarray(start, np.int64)[rows], "end": np.asarray(end, np.int64)[rows],
            "user": local_of_global[user[rows]], "disc": np.asarray(disc, np.int32)[rows],
            "rows": rows, "users": users_r, "n_users": max(int(users_r.size), 1),
        })
    return shards


class HipShardBackend:
    """Scans the local shard with the HIP library; the scan itself writes the result message on this rank's GPU."""

    direct_message = True   # scan_begin takes the message buffer; scan_finish_packed returns (M, ready)

    def __init__(self, ctx: PieScan, device):
        self.ctx = ctx
        self.device = torch.device(device)
        # the scan runs on the library's own stream; it is wrapped here so torch events can be recorded on it
        self.result_stream = torch.cuda.ExternalStream(self.ctx.aux_stream(), device=self.device)

    def scan_begin(self, now, cutoff, dst=None, u_pad=0, cap=0):
        """Enqueue the table pass + offsets kernel; returns immediately.  With dst (int32 device tensor of
        u_pad + 2 + cap words) the scan writes its result message there as it goes."""
        if dst is None:
            self.ctx.scan_begin(now, cutoff)
        else:
            self.ctx.scan_begin_packed(now, cutoff, dst.data_ptr(), u_pad, cap)

    def scan_finish_packed(self, dst, u_pad, cap):
        """Wait for the oldest scan's summary.  -> (M, ready): ready = the message is already complete in device memory
        (written by the scan's own kernels); otherwise a pack kernel was enqueued on result_stream to write it."""
        if self.ctx.in_flight_packed():
            return self.ctx.scan_finish_packed()
        m = self.ctx.scan_finish()
        self.ctx.pack_results_device(dst.data_ptr(), u_pad, cap)
        return m, False

    # batched scans: Q queries, one table pass, Q messages back to back in dst (stride words each)
    def batch_begin(self, queries, dst=None, stride=0, u_pad=0, cap=0):
        if dst is None:
            self.ctx.scan_batch_begin(queries)
        else:
            self.ctx.scan_batch_begin_packed(queries, dst.data_ptr(), stride, u_pad, cap)

    def batch_finish(self, want_m=True):
        """-> ([M per query] (None when not wanted), ready)"""
        return self.ctx.scan_batch_finish(packed=True, want_m=want_m)

    def batch_depth(self):
        """batches the context takes in flight: three per lane (pie_set_batch_lanes)"""
        return 3 * self.ctx.batch_lanes()

    def batch_room(self):
        return self.ctx.batch_room()

    def batch_flush(self):
        self.ctx.scan_batch_flush()

    union_direct = True   # batch_begin_union: the batch's own kernels write the union message

    def batch_begin_union(self, queries, dst, u_pad, cap):
        """A batch whose tail kernel writes the ONE union message of the step into dst (an int32 device tensor, or its address)
        as it goes."""
        self.ctx.scan_batch_begin_union(queries, dst if isinstance(dst, int) else dst.data_ptr(), u_pad, cap)

    def batch_pack_union(self, dst, u_pad, cap):
        """Enqueue (on result_stream) the UNION message of the last finished batch into dst: u_pad + 2 + 2 * cap int32 words
        (3 * cap when the batch holds more than 32 queries)."""
        self.ctx.batch_pack_union_device(dst.data_ptr(), u_pad, cap)


N_SETS = 4


def _staged_all_gather(out_dev, msg_dev, h_out, h_msg, group, wait_event, device):
    """transport="host": the ranks' messages live on their GPUs but the collective runs between CPU tensors (gloo, or any
    backend without a device transport): D2H into pinned memory, all-gather on the host, H2D into the gather buffer.
    Synchronous — the rehearsal / fallback transport (two ranks on ONE GPU, where RCCL refuses to form a group; a node
    without xGMI), not the fast path; everything around it (sharding, batched scans, message formats, capacity
    negotiation, buffer rotation) is the code the RCCL run executes."""
    cur = torch.cuda.current_stream(device)
    if wait_event is not None:
        cur.wait_event(wait_event)
    h_msg.copy_(msg_dev, non_blocking=True)
    cur.synchronize()
    dist.all_gather_into_tensor(h_out, h_msg, group=group)
    out_dev.copy_(h_out, non_blocking=True)
    cur.synchronize()


class _Buffers:
    """Four message / gather buffer sets of one capacity; scan i of a pipelined run uses set i % 4 (one being written by
    the scan just begun, one by the scan in flight, one being gathered, one being collected)."""

    def __init__(self, world, u_pad, cap, device, cuda, staged=False):
        self.cap, self.u_pad = cap, u_pad
        L = self.L = u_pad + 2 + cap
        self.msg = [torch.zeros(L, dtype=torch.int32, device=device) for _ in range(N_SETS)]
        self.out = [torch.zeros(world * L, dtype=torch.int32, device=device) for _ in range(N_SETS)]
        if staged:   # transport="host": pinned staging of one message / one gather
            self.h_msg = torch.zeros(L, dtype=torch.int32, pin_memory=True)
            self.h_out = torch.zeros(world * L, dtype=torch.int32, pin_memory=True)
        self.len_host = [torch.zeros(world, dtype=torch.int32, pin_memory=cuda) for _ in range(N_SETS)]
        # strided view of the M word of every rank's message, made once
        self.len_dev = [o.view(world, L)[:, u_pad + 1] for o in self.out]
        self.busy = [False] * N_SETS   # a gather that reads msg[p] / writes out[p] has been issued and not collected
        if cuda:
            self.ev_packed = [torch.cuda.Event() for _ in range(N_SETS)]
            self.ev_done = [torch.cuda.Event() for _ in range(N_SETS)]
            # the zero fills above ran on torch's stream; the scans write these buffers from the library's own stream
            torch.cuda.current_stream(device).synchronize()


class _Ticket:
    __slots__ = ("bufs", "parity", "m", "ready", "issued", "query")


class _BatchBuffers:
    """Three buffer sets for run_steps with batch > 1: one set holds the messages of `batch` consecutive scans back to
    back (scan j of a batch writes slice j), and ONE all-gather moves them all.  xGMI collectives of this size are
    latency-bound (a ring / mesh step costs microseconds before the first byte moves), so fewer, larger collectives
    carry the same lists for a fraction of the fixed cost; the price is that a step's gathered lists arrive up to
    `batch` steps later."""

    def __init__(self, world, u_pad, cap, batch, device, cuda, staged=False):
        self.cap, self.u_pad, self.batch, self.world = cap, u_pad, batch, world
        L = self.L = u_pad + 2 + cap
        self.msg = [torch.zeros(batch * L, dtype=torch.int32, device=device) for _ in range(3)]
        self.out = [torch.zeros(world * batch * L, dtype=torch.int32, device=device) for _ in range(3)]
        if staged:
            self.h_msg = torch.zeros(batch * L, dtype=torch.int32, pin_memory=True)
            self.h_out = torch.zeros(world * batch * L, dtype=torch.int32, pin_memory=True)
        self.len_host = [torch.zeros(world, batch, dtype=torch.int32, pin_memory=cuda) for _ in range(3)]
        self.len_dev = [o.view(world, batch, L)[:, :, u_pad + 1] for o in self.out]
        if cuda:
            self.ev_packed = [torch.cuda.Event() for _ in range(3)]
            self.ev_done = [torch.cuda.Event() for _ in range(3)]
            torch.cuda.current_stream(device).synchronize()

    def slice(self, s, j):
        return self.msg[s][j * self.L:(j + 1) * self.L]


class ShardedFeeds:
    """Per-rank driver: scan the local shard, all-gather the packed messages.

    `backend.scan_begin(now, cutoff[, dst, u_pad, cap])` / `backend.scan_finish_packed(dst, u_pad, cap) -> M | (M, ready)`
    fill an int32 tensor on `backend.device` (the GPU for nccl/RCCL, the CPU for gloo).  Pipeline stages:
        begin            enqueue the scan (no host wait); the scan writes its message as it goes
        finish_and_pack  wait for the oldest begun scan's summary                                    -> ticket
        exchange         issue the all-gather of that message on the side stream (GPU-async)
        collect          wait for the side stream only; hand out the gathered views
    run_steps() keeps two scans queued on the GPU while the host issues and collects gathers, so neither the host work
    nor the gather sits between two table passes.  Four message / result buffer sets rotate."""

    def __init__(self, backend, rank, world, n_users_local, group=None, cap=None, always_collective=False, batch=1,
                 transport="device"):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group
        self.batch = max(1, int(batch))   # run_steps: scans per all-gather (see _BatchBuffers)
        self.bbufs = None
        self.collective = world > 1 or always_collective  # world 1 + always_collective: rehearsal of the exchange step
        self.n_users_local = int(n_users_local)
        self.device = torch.device(getattr(backend, "device", "cpu"))
        self.cuda = self.device.type == "cuda"
        # "device": the collective moves the GPU tensors (nccl = RCCL over xGMI; gloo when they are CPU tensors);
        # "host": GPU messages staged through pinned host memory, collective between CPU tensors (see _staged_all_gather)
        self.staged = transport == "host" and self.cuda
        self.direct = bool(getattr(backend, "direct_message", False))
        # offsets are padded to the largest shard's user count so the gather has one fixed size
        self.u_pad = self._all_max(self.n_users_local)
        self.cap = cap  # capacity of one rank's row list in the message; negotiated on first use
        self.bufs = None
        self.step = 0       # scans begun so far: picks the buffer set
        self.begun = []     # tickets of scans begun and not finished, oldest first
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(self.device)
            self.rs = getattr(backend, "result_stream", None) or torch.cuda.current_stream(self.device)

    # ---- helpers
    def _all_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if self.staged else self.device)
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    @staticmethod
    def _grow(need):
        # 6 % headroom: the message is what crosses xGMI every step, so slack is kept small; a query that outgrows
        # it is detected by every rank from the gathered lengths and costs one collective re-negotiation
        return max(1024, int(need * 1.06) + 64)

    def _buffers(self):
        if self.bufs is None or self.bufs.cap != self.cap:
            # scans begun earlier keep their own (old) set alive through their tickets
            self.bufs = _Buffers(self.world, self.u_pad, self.cap, self.device, self.cuda, self.staged)
        return self.bufs

    # ---- pipeline stages
    def begin(self, now, cutoff):
        t = _Ticket()
        t.query, t.issued, t.m, t.ready = (now, cutoff), False, 0, False
        if self.cap is None:
            t.bufs, t.parity = None, 0          # first use: a probe scan learns M
            self.backend.scan_begin(now, cutoff)
        else:
            t.bufs, t.parity = self._buffers(), self.step % N_SETS
            self.step += 1
            if t.bufs.busy[t.parity]:
                raise RuntimeError("collect() the ticket issued four steps ago before its buffers are used again")
            if self.direct:
                self.backend.scan_begin(now, cutoff, t.bufs.msg[t.parity], self.u_pad, t.bufs.cap)
            else:
                self.backend.scan_begin(now, cutoff)
        self.begun.append(t)

    def _finish(self, t, dst, cap):
        r = self.backend.scan_finish_packed(dst, self.u_pad, cap)
        return r if isinstance(r, tuple) else (r, False)

    def finish_and_pack(self):
        t = self.begun.pop(0)
        if t.bufs is None:
            # the probe: learn M, agree on a capacity, and redo this scan with real buffers
            if self.begun:
                raise RuntimeError("the first scan negotiates the message capacity: finish it before beginning another")
            probe = torch.zeros(self.u_pad + 2, dtype=torch.int32, device=self.device)
            m, _ = self._finish(t, probe, 0)
            self.cap = self._grow(self._all_max(m))
            self.begin(*t.query)
            t = self.begun.pop(0)
        p = t.parity
        t.m, t.ready = self._finish(t, t.bufs.msg[p], t.bufs.cap)
        if self.cuda and not t.ready:
            t.bufs.ev_packed[p].record(self.rs)
        return t

    def exchange(self, t):
        b, p = t.bufs, t.parity
        t.issued = True
        b.busy[p] = True
        if not self.collective:
            if self.cuda and not t.ready:     # the pack kernel runs on the scan's stream: order the copy behind it
                torch.cuda.current_stream(self.device).wait_event(b.ev_packed[p])
            b.out[p].copy_(b.msg[p])
            return
        if self.staged:
            _staged_all_gather(b.out[p], b.msg[p], b.h_out, b.h_msg, self.group, None if t.ready else b.ev_packed[p], self.device)
            b.len_host[p].copy_(b.h_out.view(self.world, b.L)[:, b.u_pad + 1])
            return
        if self.cuda:
            prev = torch.cuda.current_stream(self.device)
            torch.cuda.set_stream(self.comm_stream)
            try:
                if not t.ready:
                    self.comm_stream.wait_event(b.ev_packed[p])
                dist.all_gather_into_tensor(b.out[p], b.msg[p], group=self.group)  # stream-ordered, host-async
                b.len_host[p].copy_(b.len_dev[p], non_blocking=True)
                b.ev_done[p].record(self.comm_stream)
            finally:
                torch.cuda.set_stream(prev)
        else:
            dist.all_gather_into_tensor(b.out[p], b.msg[p], group=self.group)

    def collect(self, t):
        """-> dict(offsets [world, U_pad+1] int32, lengths [world], rows [world, cap] int32) or None when a rank's row
        list outgrew the message capacity (every rank sees the same lengths, so every rank gets None, the capacity has
        been raised, and the caller resubmits).  Feed(r, u) = rows[r, offsets[r,u] : offsets[r,u+1]]."""
        b, p = t.bufs, t.parity
        if self.staged and self.collective:
            pass                         # the staged exchange is synchronous: lengths are already on the host
        elif self.cuda and self.collective:
            b.ev_done[p].synchronize()   # waits for the side stream only, never for the scan stream
        else:
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            b.len_host[p].copy_(b.len_dev[p])
        b.busy[p] = False
        g = b.out[p].view(self.world, b.L)
        need = int(b.len_host[p].max())
        if need > b.cap:
            self.cap = max(self.cap, self._grow(need))
            return None
        return {"offsets": g[:, : b.u_pad + 1], "lengths": b.len_host[p].clone(), "rows": g[:, b.u_pad + 2:]}

    def submit(self, now, cutoff):
        """One whole step without overlap: scan, pack, issue the gather.  -> ticket for collect()."""
        self.begin(now, cutoff)
        t = self.finish_and_pack()
        self.exchange(t)
        return t

    def scan_and_gather(self, now, cutoff):
        """Synchronous form: one scan, one gather, retried with a larger message if a row list did not fit."""
        while True:
            res = self.collect(self.submit(now, cutoff))
            if res is not None:
                return res

    def run_steps(self, k, now, cutoff):
        """k steps of the same query, software-pipelined.  Per iteration: scan i+1 is queued (its launch carries the offsets
        kernel of scan i, which also writes scan i's message), then — while that runs — the host issues the gather of
        step i-1 and collects the gather of step i-2, and only then waits for scan i's summary.  Every scan begun is
        finished and every gather collected before returning.
        -> last collected result (None if a message overflowed: the capacity has been raised, call again)."""
        if k <= 0:
            return None
        last = None
        if self.cap is None:          # capacity negotiation needs a whole scan of its own
            last = self.scan_and_gather(now, cutoff)
            k -= 1
            if k == 0:
                return last
        if self.batch > 1 and self.direct:
            return self._run_steps_batched(k, now, cutoff)
        overflow = False
        finished = None               # ticket of the scan finished in the previous iteration: its gather is issued next
        flying = None                 # ticket whose gather has been issued and not collected
        self.begin(now, cutoff)
        for i in range(k):
            if i + 1 < k and not overflow:
                self.begin(now, cutoff)          # queued right behind scan i
            if finished is not None:
                self.exchange(finished)          # gather of step i-1, on the side stream
            if flying is not None:
                overflow = (self.collect(flying) is None) or overflow   # gather of step i-2: issued a whole scan ago
            flying, finished = finished, None
            finished = self.finish_and_pack()    # scan i's summary (its offsets kernel ran beside scan i+1's table pass)
            if overflow and not self.begun:
                break
        if finished is not None:
            self.exchange(finished)
        if flying is not None:
            overflow = (self.collect(flying) is None) or overflow
        last = self.collect(finished) if finished is not None else last
        return None if overflow or last is None else last


def _run_steps_batched(self, k, now, cutoff):
    """run_steps with `batch` scans per all-gather (direct-message backends).  Scan i writes slice i % batch of buffer set
    (i // batch) % 3; when a batch is complete its one gather is issued in the host-work window of the next iteration
    (while the offsets kernel of the scan in flight rides in the next table pass) and collected one batch later."""
    B = self.batch
    if self.bbufs is None or self.bbufs.cap != self.cap or self.bbufs.batch != B:
        self.bbufs = _BatchBuffers(self.world, self.u_pad, self.cap, B, self.device, self.cuda, self.staged)
    bb = self.bbufs
    L = bb.L
    overflow = False
    last = None

    def begin(i):
        s_, j = (i // B) % 3, i % B
        self.backend.scan_begin(now, cutoff, bb.slice(s_, j), self.u_pad, bb.cap)

    def issue(s_, all_ready):
        if not self.collective:
            if self.cuda and not all_ready:
                torch.cuda.current_stream(self.device).wait_event(bb.ev_packed[s_])
            bb.out[s_].copy_(bb.msg[s_])
            return
        if self.staged:
            _staged_all_gather(bb.out[s_], bb.msg[s_], bb.h_out, bb.h_msg, self.group, None if all_ready else bb.ev_packed[s_], self.device)
            bb.len_host[s_].copy_(bb.h_out.view(self.world, B, L)[:, :, bb.u_pad + 1])
            return
        if self.cuda:
            prev = torch.cuda.current_stream(self.device)
            torch.cuda.set_stream(self.comm_stream)
            try:
                if not all_ready:
                    self.comm_stream.wait_event(bb.ev_packed[s_])
                dist.all_gather_into_tensor(bb.out[s_], bb.msg[s_], group=self.group)
                bb.len_host[s_].copy_(bb.len_dev[s_], non_blocking=True)
                bb.ev_done[s_].record(self.comm_stream)
            finally:
                torch.cuda.set_stream(prev)
        else:
            dist.all_gather_into_tensor(bb.out[s_], bb.msg[s_], group=self.group)

    def collect(s_, filled):
        nonlocal overflow, last
        if self.staged and self.collective:
            pass
        elif self.cuda and self.collective:
            bb.ev_done[s_].synchronize()
        else:
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            bb.len_host[s_].copy_(bb.len_dev[s_])
        need = int(bb.len_host[s_][:, :filled].max())
        if need > bb.cap:
            self.cap = max(self.cap, self._grow(need))
            overflow = True
            return
        g = bb.out[s_].view(self.world, B, L)
        j = filled - 1
        last = {"offsets": g[:, j, : bb.u_pad + 1], "lengths": bb.len_host[s_][:, j].clone(), "rows": g[:, j, bb.u_pad + 2:]}

    to_issue = None      # (set, filled, all_ready) of a completed batch whose gather is issued in the next window
    flying = []          # (set, filled) of gathers issued and not collected
    ready_all = True
    begin(0)
    for i in range(k):
        if i + 1 < k:
            begin(i + 1)
        if to_issue is not None:
            issue(to_issue[0], to_issue[2])
            flying.append(to_issue[:2])
            to_issue = None
            if len(flying) > 1:
                collect(*flying.pop(0))
        m, ready = self.backend.scan_finish_packed(bb.slice((i // B) % 3, i % B), self.u_pad, bb.cap)
        ready_all = ready_all and ready
        if i % B == B - 1 or i == k - 1:
            s_ = (i // B) % 3
            if self.cuda and not ready_all:
                bb.ev_packed[s_].record(self.rs)
            to_issue = (s_, i % B + 1, ready_all)
            ready_all = True
    if to_issue is not None:
        issue(to_issue[0], to_issue[2])
        flying.append(to_issue[:2])
    while flying:
        collect(*flying.pop(0))
    return None if overflow else last


ShardedFeeds._run_steps_batched = _run_steps_batched


class UnionOverflow(RuntimeError):
    """A shard's union message is unusable (a user's union of row lists exceeds 32 rows): exchange the per-query lists."""


def union_feed(res, r, q, u):
    """Feed of user u (local index on shard r) for query q out of a union-mode result of BatchedFeeds.run_steps."""
    a, b = int(res["u_offsets"][r, u]), int(res["u_offsets"][r, u + 1])
    masks = res["masks_hi"] if q >= 32 else res["masks"]
    rows, masks = res["rows"][r, a:b], masks[r, a:b]
    return rows[((masks >> (q & 31)) & 1) == 1]


class BatchedFeeds:
    """Per-rank driver of the BATCHED exchange: every step is one batched scan of Q queries over the local shard (one table
    pass, pie_scan_batch_*), whose offsets kernel writes the Q result messages back to back, and ONE all-gather moves them
    all — the "fewer, larger collectives" of a point-to-point fabric come with the batch.  Same pipeline as
    ShardedFeeds.run_steps: begin(i+1) | gather(i-1) issued on the side stream | gather(i-2) collected | finish(i); four
    rotating buffer sets.

    backend.batch_begin(queries[, dst, stride, u_pad, cap]) / backend.batch_finish() -> ([M], ready).

    union=True: ONE message per step instead of Q — per user the union of the Q row lists in (start, row) order with a query
    mask per row (pie_batch_pack_union_device; backend.batch_pack_union(dst, u_pad, cap)).  The queries of a batch are
    requests of the same few seconds and select almost the same rows, so the union is little longer than one list: an eighth
    of the bytes cross the links for Q = 16.  run_steps then returns u_offsets [world, U_pad+1], lengths [world], rows
    [world, cap], masks [world, cap]; Feed(r, q, u) = rows[r, a:b][(masks[r, a:b] >> q) & 1 == 1], a, b = u_offsets[r, u : u+2].
    With a backend that has batch_begin_union (the HIP backend) the batch's own tail kernel writes the message — nothing is
    packed afterwards and no event sits between the scan stream and the gather.  More than 32 queries: a second mask word per
    row (masks_hi).  A shard whose batch had to be merged from per-query lists (queries fell back, skewed users on the ordered
    run) and holds more than 32 union rows for a user reports length -1: UnionOverflow, use the lists."""

    def __init__(self, backend, rank, world, n_users_local, q_max, group=None, cap=None, always_collective=False, union=False,
                 steps_per_gather=1, transport="device"):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group
        self.staged = transport == "host" and torch.device(getattr(backend, "device", "cpu")).type == "cuda"   # see ShardedFeeds
        self.union = bool(union)
        self.steps_per_gather = max(1, int(steps_per_gather))
        self.q_max = int(q_max)
        self.collective = world > 1 or always_collective
        self.device = torch.device(getattr(backend, "device", "cpu"))
        self.cuda = self.device.type == "cuda"
        self.n_users_local = int(n_users_local)
        self.u_pad = self._all_max(self.n_users_local)
        self.cap = cap
        self.sets = None
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(self.device)
            self.rs = getattr(backend, "result_stream", None) or torch.cuda.current_stream(self.device)

    def _all_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if self.staged else self.device)
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _sets(self):
        if self.sets is None or self.sets["cap"] != self.cap:
            # union: rows + mask words (one word per 32 queries); lists: rows
            self.mask_words = 2 if self.q_max > 32 else 1
            L = self.u_pad + 2 + ((1 + self.mask_words) if self.union else 1) * self.cap
            # messages of one step: one (union) or q_max (lists); G steps share an all-gather
            Qm, W, G = (1 if self.union else self.q_max), self.world, self.steps_per_gather
            self.sets = {
                "cap": self.cap, "L": L, "Qm": Qm,
                "msg": [torch.zeros(G * Qm * L, dtype=torch.int32, device=self.device) for _ in range(N_SETS)],
                "out": [torch.zeros(W * G * Qm * L, dtype=torch.int32, device=self.device) for _ in range(N_SETS)],
                "len_host": [torch.zeros(W, G * Qm, dtype=torch.int32, pin_memory=self.cuda) for _ in range(N_SETS)],
            }
            self.sets["len_dev"] = [o.view(W, G * Qm, L)[:, :, self.u_pad + 1] for o in self.sets["out"]]
            if self.staged:
                self.sets["h_msg"] = torch.zeros(G * Qm * L, dtype=torch.int32, pin_memory=True)
                self.sets["h_out"] = torch.zeros(W * G * Qm * L, dtype=torch.int32, pin_memory=True)
            if self.cuda:
                self.sets["ev_packed"] = [torch.cuda.Event() for _ in range(N_SETS)]
                self.sets["ev_done"] = [torch.cuda.Event() for _ in range(N_SETS)]
                torch.cuda.current_stream(self.device).synchronize()
        return self.sets

    def _negotiate(self, queries):
        """First use: one batch without messages learns the largest row list; every rank agrees on the capacity."""
        self.backend.batch_begin(queries)
        ms, _ = self.backend.batch_finish()
        if self.union:   # the union's length: a message with no room for rows still carries it
            head = torch.zeros(self.u_pad + 2, dtype=torch.int32, device=self.device)
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            self.backend.batch_pack_union(head, self.u_pad, 0)
            if self.cuda:
                self.rs.synchronize()
            mu = int(head[self.u_pad + 1])
            if self._all_max(1 if mu < 0 else 0):
                raise UnionOverflow("a user's union of row lists exceeds 32 rows on some shard")
            self.cap = ShardedFeeds._grow(self._all_max(mu))
            return
        self.cap = ShardedFeeds._grow(self._all_max(max(ms) if ms else 0))

    def _issue(self, st, p, ready):
        if not self.collective:
            if self.cuda and not ready:
                torch.cuda.current_stream(self.device).wait_event(st["ev_packed"][p])
            st["out"][p].copy_(st["msg"][p])
            return
        if self.staged:
            _staged_all_gather(st["out"][p], st["msg"][p], st["h_out"], st["h_msg"], self.group, None if ready else st["ev_packed"][p], self.device)
            st["len_host"][p].copy_(st["h_out"].view(self.world, self.steps_per_gather * st["Qm"], st["L"])[:, :, self.u_pad + 1])
            return
        if self.cuda:
            prev = torch.cuda.current_stream(self.device)
            torch.cuda.set_stream(self.comm_stream)
            try:
                if not ready:
                    self.comm_stream.wait_event(st["ev_packed"][p])
                dist.all_gather_into_tensor(st["out"][p], st["msg"][p], group=self.group)
                st["len_host"][p].copy_(st["len_dev"][p], non_blocking=True)
                st["ev_done"][p].record(self.comm_stream)
            finally:
                torch.cuda.set_stream(prev)
        else:
            dist.all_gather_into_tensor(st["out"][p], st["msg"][p], group=self.group)

    def _collect(self, st, p, nq, n_slots):
        """Wait for the all-gather of set p (n_slots steps' messages) -> the LAST of those steps as a result dict, or None
        when a message of the group outgrew the capacity (raised)."""
        if self.staged and self.collective:
            pass   # synchronous exchange: lengths are on the host already
        elif self.cuda and self.collective:
            st["ev_done"][p].synchronize()
        else:
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            st["len_host"][p].copy_(st["len_dev"][p])
        Qm, G, L, cap = st["Qm"], self.steps_per_gather, st["L"], st["cap"]
        lens_all = st["len_host"][p].view(self.world, G, Qm)[:, :n_slots]
        if self.union:
            if int(lens_all.min()) < 0:
                raise UnionOverflow("a user's union of row lists exceeds 32 rows on some shard")
            need = int(lens_all.max())
        else:
            need = int(lens_all[:, :, :nq].max())
        if need > cap:
            self.cap = max(self.cap, ShardedFeeds._grow(need))
            return None
        last = n_slots - 1
        g = st["out"][p].view(self.world, G, Qm, L)[:, last]
        lens = lens_all[:, last]
        if self.union:
            g = g[:, 0]
            res = {"u_offsets": g[:, : self.u_pad + 1], "lengths": lens[:, 0].clone(), "rows": g[:, self.u_pad + 2: self.u_pad + 2 + cap],
                   "masks": g[:, self.u_pad + 2 + cap: self.u_pad + 2 + 2 * cap]}
            if self.mask_words == 2:
                res["masks_hi"] = g[:, self.u_pad + 2 + 2 * cap: self.u_pad + 2 + 3 * cap]
            return res
        return {"offsets": g[:, :nq, : self.u_pad + 1], "lengths": lens[:, :nq].clone(), "rows": g[:, :nq, self.u_pad + 2:]}

    def run_steps(self, k, queries):
        """k steps of the same batch of queries, software-pipelined; the messages of steps_per_gather consecutive steps
        travel in ONE all-gather (a collective costs tens of microseconds of host and link latency whatever it carries).
        -> the last step's result.  lists: offsets [world, Q, U_pad+1], lengths [world, Q], rows [world, Q, cap];
        Feed(r, q, u) = rows[r, q, offsets[r,q,u] : offsets[r,q,u+1]].  union: see the class docstring.  None when a message
        outgrew the capacity (raised: call again)."""
        if k <= 0:
            return None
        nq = len(queries)
        if nq > self.q_max:
            raise ValueError("batch of %d queries exceeds q_max %d" % (nq, self.q_max))
        if self.cap is None:
            self._negotiate(queries)
        st = self._sets()
        L, cap, Qm, G = st["L"], st["cap"], st["Qm"], self.steps_per_gather
        overflow, last = False, None
        pending, flying = None, None   # a complete group whose gather is not issued yet / issued and not collected: (set, ready, steps)

        def slot_of(i):
            return st["msg"][(i // G) % N_SETS][(i % G) * Qm * L: (i % G + 1) * Qm * L]

        direct_union = self.union and bool(getattr(self.backend, "union_direct", False))
        # addresses of the message slots, once (a tensor slice per step is a microsecond of Python the step does not have)
        slot_addr = None
        if direct_union and self.cuda:
            slot_addr = [st["msg"][p].data_ptr() + 4 * g * Qm * L for p in range(N_SETS) for g in range(G)]
        slim = bool(getattr(self.backend, "union_direct", False))   # the HIP backend: finish need not build the list of M

        def begin(i):
            if slot_addr is not None:
                self.backend.batch_begin_union(queries, slot_addr[i % (N_SETS * G)], self.u_pad, cap)
            elif direct_union:   # the batch's tail kernel writes the step's ONE message itself
                self.backend.batch_begin_union(queries, slot_of(i), self.u_pad, cap)
            elif self.union:
                self.backend.batch_begin(queries)
            else:
                self.backend.batch_begin(queries, slot_of(i), L, self.u_pad, cap)

        def collect(t):
            nonlocal overflow, last
            res = self._collect(st, t[0], nq, t[2])
            overflow = overflow or res is None
            last = res if res is not None else last

        # batches in flight when the host waits for one: two on one lane's stream; a backend with batch lanes (the HIP library on a
        # shard-sized table: several batches side by side on the chip) takes what its lanes hold.  A step's message slot belongs to
        # group i // G, whose buffer set comes round again N_SETS groups later and is collected two groups after its own: D <= 2 G
        # keeps every begin off a set that is still being gathered or read.
        D = max(2, min(int(getattr(self.backend, "batch_depth", lambda: 2)()), 2 * G))
        begun = 0
        room = getattr(self.backend, "batch_room", lambda: 1)   # (a table whose batches run on the ordered run takes three, whatever the lanes)
        flush = getattr(self.backend, "batch_flush", None)
        group_ready = True
        for i in range(k):
            while begun < k and begun - i < D and (begun == i or room() > 0):
                begin(begun)
                begun += 1
                if begun == k and flush is not None:   # the run's last batch: the lanes' waiting tails go out together
                    flush()
            if pending is not None:   # the gather of the group before runs beside this group's scans
                self._issue(st, pending[0], pending[1])
                if flying is not None:
                    collect(flying)
                flying, pending = pending, None
            _, ready = self.backend.batch_finish(want_m=False) if slim else self.backend.batch_finish()
            p = (i // G) % N_SETS
            if self.union and not direct_union:   # packed from the finished batch, on the result stream
                self.backend.batch_pack_union(slot_of(i), self.u_pad, cap)
                ready = False
            group_ready = group_ready and ready
            if i % G == G - 1 or i == k - 1:
                if self.cuda and not group_ready:
                    st["ev_packed"][p].record(self.rs)
                pending = (p, group_ready, i % G + 1)
                group_ready = True
        if pending is not None:
            self._issue(st, pending[0], pending[1])
            if flying is not None:
                collect(flying)
            flying, pending = pending, None
        if flying is not None:
            collect(flying)
        return None if overflow else last


def gather_expired_queues(local_queue, local_to_global_rows, rank, world, device="cpu", group=None):
    """Multi-GPU form of the expired-session dispatch queue: every rank contributes the ordered
    queue of ITS shard (local row indices from `pie_expired_queue`), mapped to global row ids; one all-gather of the
    padded lists; every rank merges them into the single ascending global order in which the host drains the queue
    (the sequential-await order of /root/reference/server/storage/sqlProvider.js:834-861).
    local_queue: int array of local rows (ascending); local_to_global_rows: the shard's row map (ascending, so each
    rank's global list is ascending too).  -> np.int64 array, identical on every rank."""
    glob = np.asarray(local_to_global_rows, np.int64)[np.asarray(local_queue, np.int64)]
    dev = torch.device(device)
    n = torch.tensor([glob.size], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    cap = int(n.item())
    msg = torch.full((cap + 1,), -1, dtype=torch.int64, device=dev)
    msg[0] = glob.size
    if glob.size:
        msg[1:1 + glob.size] = torch.from_numpy(glob).to(dev)
    if world > 1:
        out = torch.empty(world * (cap + 1), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, msg, group=group)
    else:
        out = msg
    out = out.view(world, cap + 1).cpu().numpy()
    lists = [out[r, 1:1 + int(out[r, 0])] for r in range(world)]
    merged = np.concatenate(lists) if lists else np.zeros(0, np.int64)
    merged.sort(kind="stable")   # k-way merge of ascending lists; ids are unique
    return merged


def gather_archive_queues(local_queue, local_group_of_row, local_to_global_rows, rank, world, device="cpu", group=None):
    """Multi-GPU form of the archive dispatch queue (the reference's own chain
    /root/reference/server/storage/sqlProvider.js:758-816): groups are users, the table is sharded by user hash, so every
    group lives whole on one rank and `pie_archive_queue` on each shard already yields that shard's qualifying groups,
    each group's rows in table order, groups in order of first appearance WITHIN the shard.  The global queue orders the
    groups by first appearance in the WHOLE table (Map insertion order, :769-789): one all-gather of the per-rank lists
    (global row ids + group lengths, padded), then a merge of the groups by the global id of their first row.
    local_queue: local rows as returned by the shard's archive queue; local_group_of_row: the shard's group (user) column;
    local_to_global_rows: the shard's row map (ascending).  -> np.int64 array of global rows, identical on every rank."""
    q = np.asarray(local_queue, np.int64)
    glob = np.asarray(local_to_global_rows, np.int64)[q]
    grp = np.asarray(local_group_of_row)[q] if q.size else np.zeros(0, np.int64)
    # group boundaries inside the local queue: a group's rows are contiguous there
    starts = np.nonzero(np.r_[True, grp[1:] != grp[:-1]])[0] if q.size else np.zeros(0, np.int64)
    lens = np.diff(np.r_[starts, q.size]) if q.size else np.zeros(0, np.int64)
    dev = torch.device(device)
    dims = torch.tensor([glob.size, lens.size], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(dims, op=dist.ReduceOp.MAX, group=group)
    cap_rows, cap_groups = int(dims[0]), int(dims[1])
    msg = torch.full((2 + cap_rows + cap_groups,), -1, dtype=torch.int64, device=dev)
    msg[0], msg[1] = glob.size, lens.size
    if glob.size:
        msg[2:2 + glob.size] = torch.from_numpy(glob).to(dev)
        msg[2 + cap_rows:2 + cap_rows + lens.size] = torch.from_numpy(lens.astype(np.int64)).to(dev)
    if world > 1:
        out = torch.empty(world * msg.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, msg, group=group)
    else:
        out = msg
    out = out.view(world, msg.numel()).cpu().numpy()
    groups = []   # (global id of the group's first row, its rows)
    for r in range(world):
        n_rows, n_groups = int(out[r, 0]), int(out[r, 1])
        rows_r = out[r, 2:2 + n_rows]
        lens_r = out[r, 2 + cap_rows:2 + cap_rows + n_groups]
        at = 0
        for ln in lens_r:
            groups.append((int(rows_r[at]), rows_r[at:at + int(ln)]))
            at += int(ln)
    groups.sort(key=lambda g: g[0])
    return np.concatenate([g[1] for g in groups]) if groups else np.zeros(0, np.int64)
