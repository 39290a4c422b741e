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
    """Scans the local shard with the HIP library and packs the result message on this rank's GPU."""

    def __init__(self, ctx: PieScan, device):
        self.ctx = ctx
        self.device = torch.device(device)
        # the scan runs on the library's own stream; it is wrapped here so torch events can be recorded on it
        self.result_stream = torch.cuda.ExternalStream(self.ctx.aux_stream(), device=self.device)

    def scan_begin(self, now, cutoff):
        """Enqueue the table pass + offsets kernel; returns immediately."""
        self.ctx.scan_begin(now, cutoff)

    def scan_finish_packed(self, dst, u_pad, cap):
        """Wait for the scan's summary, enqueue its tail, then write the message into dst (int32, device) with one
        kernel launch.  -> M (host int)."""
        m = self.ctx.scan_finish()
        self.ctx.pack_results_device(dst.data_ptr(), u_pad, cap)
        return m


class _Ticket:
    __slots__ = ("parity", "m", "cap", "u_pad", "issued")


class ShardedFeeds:
    """Per-rank driver: scan the local shard, all-gather the packed messages.

    `backend.scan_begin(now, cutoff)` / `backend.scan_finish_packed(dst, u_pad, cap) -> M` fill an int32 tensor on
    `backend.device` (the GPU for nccl/RCCL, the CPU for gloo).  Pipeline stages:
        begin            enqueue the scan (no host wait)
        finish_and_pack  wait for its summary, enqueue tail + pack, mark the message ready          -> ticket
        exchange         issue the all-gather of that message on the side stream (GPU-async)
        collect          wait for the side stream only; hand out the gathered views
    run_steps() interleaves them so that the host work of exchange(i) happens while the GPU scans step i+1.
    Two message / result buffers alternate."""

    def __init__(self, backend, rank, world, n_users_local, group=None, cap=None, always_collective=False):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group
        self.collective = world > 1 or always_collective  # world 1 + always_collective: rehearsal of the exchange step
        self.n_users_local = int(n_users_local)
        self.device = torch.device(getattr(backend, "device", "cpu"))
        self.cuda = self.device.type == "cuda"
        # offsets are padded to the largest shard's user count so the gather has one fixed size
        self.u_pad = self._all_max(self.n_users_local)
        self.cap = cap  # capacity of one rank's row list in the message; negotiated on first use
        self.parity = 0
        self.msg = self.out = None
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(self.device)
            self.rs = getattr(backend, "result_stream", None) or torch.cuda.current_stream(self.device)
            self.ev_packed = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.busy = [False, False]   # a gather that reads msg[p] / writes out[p] has been issued and not collected

    # ---- helpers
    def _all_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    @staticmethod
    def _grow(need):
        # 6 % headroom: the message is what crosses xGMI every step, so slack is kept small; a query that outgrows
        # it is detected by every rank from the gathered lengths and costs one collective re-negotiation
        return max(1024, int(need * 1.06) + 64)

    def _alloc(self):
        L = self.u_pad + 2 + self.cap
        self.msg = [torch.zeros(L, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.out = [torch.zeros(self.world * L, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.len_host = [torch.zeros(self.world, dtype=torch.int32, pin_memory=self.cuda) for _ in range(2)]
        # strided view of the M word of every rank's message, made once
        self.len_dev = [o.view(self.world, L)[:, self.u_pad + 1] for o in self.out]
        self.busy = [False, False]

    # ---- pipeline stages
    def begin(self, now, cutoff):
        self._query = (now, cutoff)
        self.backend.scan_begin(now, cutoff)

    def finish_and_pack(self):
        if self.cap is None:
            # first use: learn M, agree on a capacity, and redo this scan with real buffers
            probe = torch.zeros(self.u_pad + 2, dtype=torch.int32, device=self.device)
            self.cap = self._grow(self._all_max(self.backend.scan_finish_packed(probe, self.u_pad, 0)))
            self.backend.scan_begin(*self._query)
        if self.msg is None or self.msg[0].numel() != self.u_pad + 2 + self.cap:
            self._alloc()
        p = self.parity
        self.parity ^= 1
        if self.busy[p]:
            raise RuntimeError("collect() the ticket issued two steps ago before packing into its buffers again")
        t = _Ticket()
        t.parity, t.cap, t.u_pad, t.issued = p, self.cap, self.u_pad, False
        t.m = self.backend.scan_finish_packed(self.msg[p], self.u_pad, self.cap)
        if self.cuda:
            self.ev_packed[p].record(self.rs)
        return t

    def exchange(self, t):
        p = t.parity
        t.issued = True
        self.busy[p] = True
        if not self.collective:
            self.out[p].copy_(self.msg[p])
            return
        if self.cuda:
            prev = torch.cuda.current_stream(self.device)
            torch.cuda.set_stream(self.comm_stream)
            try:
                self.comm_stream.wait_event(self.ev_packed[p])
                dist.all_gather_into_tensor(self.out[p], self.msg[p], group=self.group)  # stream-ordered, host-async
                self.len_host[p].copy_(self.len_dev[p], non_blocking=True)
                self.ev_done[p].record(self.comm_stream)
            finally:
                torch.cuda.set_stream(prev)
        else:
            dist.all_gather_into_tensor(self.out[p], self.msg[p], group=self.group)

    def collect(self, t):
        """-> dict(offsets [world, U_pad+1] int32, lengths [world], rows [world, cap] int32) or None when a rank's row
        list outgrew the message capacity (every rank sees the same lengths, so every rank gets None, the capacity has
        been raised, and the caller resubmits).  Feed(r, u) = rows[r, offsets[r,u] : offsets[r,u+1]]."""
        p = t.parity
        L = t.u_pad + 2 + t.cap
        if self.cuda and self.collective:
            self.ev_done[p].synchronize()   # waits for the side stream only, never for the scan stream
        else:
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            self.len_host[p].copy_(self.len_dev[p])
        self.busy[p] = False
        g = self.out[p].view(self.world, L)
        need = int(self.len_host[p].max())
        if need > t.cap:
            self.cap = max(self.cap, self._grow(need))
            return None
        return {"offsets": g[:, : t.u_pad + 1], "lengths": self.len_host[p].clone(), "rows": g[:, t.u_pad + 2:]}

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
        """k steps of the same query, software-pipelined: while the GPU runs the table pass of step i+1 the host
        issues the gather of step i, and collects it one step later.  Every gather is collected before returning.
        -> last collected result (None if a message overflowed: the capacity has been raised, call again)."""
        last, flying = None, None
        if k <= 0:
            return None
        if self.cap is None:          # capacity negotiation needs a whole scan of its own
            last = self.scan_and_gather(now, cutoff)
            k -= 1
            if k == 0:
                return last
        self.begin(now, cutoff)
        for i in range(k):
            t = self.finish_and_pack()       # waits for scan i's summary, queues its tail + pack
            if i + 1 < k:
                self.begin(now, cutoff)      # the next table pass is queued right behind them ...
            self.exchange(t)                 # ... and runs while the host issues this step's gather
            if flying is not None:           # gather of step i-1: issued a whole table pass ago
                last = self.collect(flying)
            flying = t
        return self.collect(flying)


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
