"""User-hash sharding of the session table over the GPUs of one node and the all-gather that reassembles
cross-user feeds.  One process per GPU; torch.distributed is the transport (backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU for the world_size-2 tests).

Rows of different users are independent, so each rank scans its shard with no communication; the only
exchange step is the gather of per-user counts (fixed size) and of the selected-row lists (variable size,
sent as a fixed-capacity buffer whose first word is the length).  xGMI is point-to-point: an all-gather of
small buffers is latency-bound, so both gathers carry one contiguous buffer each.
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
    """Scans the local shard with the HIP library and hands back torch tensors on this rank's GPU."""

    def __init__(self, ctx: PieScan, device):
        self.ctx = ctx
        self.device = torch.device(device)
        # run the scan on torch's current stream so the D2D copies and the collectives order naturally
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.counts = torch.empty(ctx.n_users, dtype=torch.int32, device=self.device)
        self.payload = None

    def scan(self, now, cutoff, cap):
        """-> (counts[U] int32, payload[1+cap] int32 with payload[0] = M)."""
        m = self.ctx.scan_device(now, cutoff)
        if self.payload is None or self.payload.numel() != cap + 1:
            self.payload = torch.empty(cap + 1, dtype=torch.int32, device=self.device)
        self.ctx.copy_results_device(self.counts.data_ptr(), None, self.payload.data_ptr() + 4, cap)
        self.payload[0] = m
        return self.counts, self.payload, m


class ShardedFeeds:
    """Per-rank driver: scan the local shard, all-gather counts and row lists, build global offsets.

    `backend.scan(now, cutoff, cap)` -> (counts[U_local] int32, payload[1+cap] int32 with payload[0] = M, M) as
    tensors on `backend.device` (the GPU for nccl/RCCL, the CPU for gloo)."""

    def __init__(self, backend, rank, world, n_users_local, group=None, cap=None):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group
        self.n_users_local = int(n_users_local)
        self.device = torch.device(getattr(backend, "device", "cpu"))
        # counts are padded to the largest shard's user count so the gather has one fixed size
        self.u_pad = self._all_max(self.n_users_local)
        self.cap = cap  # capacity of one rank's row list in the payload gather; negotiated on first use

    def _all_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    @staticmethod
    def _grow(need):
        return max(1024, int(need * 1.25) + 64)

    def scan_and_gather(self, now, cutoff):
        """-> dict(counts [world, U_pad] int32, lengths [world] int32, rows [world, cap] int32,
        offsets [world*U_pad+1] int64).  Feed of local user u of rank r = rows[r, off[u] : off[u+1]] with
        off = exclusive prefix of counts[r]."""
        if self.cap is None:
            _, _, m = self.backend.scan(now, cutoff, 0)
            self.cap = self._grow(self._all_max(m))
        while True:
            counts, payload, m = self.backend.scan(now, cutoff, self.cap)
            if counts.numel() < self.u_pad:
                counts = torch.cat([counts, counts.new_zeros(self.u_pad - counts.numel())])
            if self.world == 1:
                g_counts, g_payload = counts, payload
            else:
                g_counts = torch.empty(self.world * counts.numel(), dtype=counts.dtype, device=counts.device)
                g_payload = torch.empty(self.world * payload.numel(), dtype=payload.dtype, device=payload.device)
                dist.all_gather_into_tensor(g_counts, counts, group=self.group)
                dist.all_gather_into_tensor(g_payload, payload, group=self.group)
            g_counts = g_counts.view(self.world, -1)
            g_payload = g_payload.view(self.world, -1)
            lengths = g_payload[:, 0]
            # one tiny D2H per step (world ints); every rank sees the same gathered lengths, so every rank
            # takes the same branch
            need = int(lengths.max().item())
            if need <= self.cap:
                break
            self.cap = self._grow(need)
        offsets = torch.zeros(g_counts.numel() + 1, dtype=torch.int64, device=g_counts.device)
        torch.cumsum(g_counts.reshape(-1), 0, out=offsets[1:])
        return {"counts": g_counts, "lengths": lengths, "rows": g_payload[:, 1:], "offsets": offsets}
