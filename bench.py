#!/usr/bin/env python3
"""bench.py — headline benchmark of the session-scan -> per-user feed path on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

A step = one scan (one `now`/`cutoff`/discipline-mask query) over the resident session table, producing
counts/offsets/idx for every user; for N > 1 the ONE 10^8-row corpus is sharded by user hash over the ranks
(BASELINE.json configs[3], strong scaling) and a step also all-gathers the per-user offsets and row lists (RCCL).
Workload at N = 1: BASELINE.json config 3 — 10^8 sessions / 10^5 users / 32 disciplines, SoA int64 start/end +
int32 user/disc, resident in HBM before the timed region.

What one run measures (rank 0 prints ONE JSON line on stdout, everything else goes to stderr):
  value / ms_per_step   the headline loop: K steps, R times (median; min / max beside it), two scans in flight
  roofline              the dominant kernel of that loop, HIP-event timed in the run; frac = HBM traffic / time / peak
                        (traffic: rocprofv3 PMC of this very command, committed under profiles/; a byte model from the
                        run's own row counters beside it) — never above 1; the 24 B/row figure is `alg_equiv_gbs`
  roofline_full_read    the same query with the table pass pinned to the form that reads every byte of the four columns
                        (24 B/row = SURVEY.md 8d's algorithmic bytes): t_scan = first kernel start -> last kernel end
  value_with_d2h        the headline loop with counts + offsets delivered to pinned host memory by every scan
  cpu_baseline          oracle C port on this box's host cores: full N on 1 thread and on all threads, checked against
                        the GPU result; reference-faithful JS (Map of session objects) beside it
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SEED = 0x5EED5EED
T0_MS = 1700000000000
DAY = 86400 * 1000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spread(xs):
    return {"median": statistics.median(xs), "min": min(xs), "max": max(xs), "n": len(xs)}


def kernel_name(variant, rides, mode):
    """Name of the table-pass kernel a scan form runs (as rocprofv3 prints it, without 'void pie::')."""
    if mode == "expired":
        return "k_expired_stage<8>" if os.environ.get("PIE_EXPIRED_ON_END") else "k_expired_stage_keyed<2>"
    if variant & 0x2000:   # the ordered run (pie_ordered.h)
        if variant & 0x400:
            return "k_ord_scan_keyed<%s, 4>" % ("unsigned char" if variant & 0x800 else "unsigned short")
        return "k_ord_scan_dense"
    if variant & 0x400:
        kt = "unsigned char" if variant & 0x800 else "unsigned short"
        agg = "true" if variant & 0x40 else "false"
        if rides:
            return "k_scan_keyed_with_tail<8, true, %s, %s>" % (kt, agg)
        un = 2 if (variant & 0xA0) == 0x20 else 8 if (variant & 0xA0) == 0x80 else 4
        return "k_scan_keyed<%d, %s, %s, %s, %s>" % (un, agg, "true" if variant & 1 else "false", kt, "true" if variant & 0x10 else "false")
    if variant & 0x200:
        return "k_scan_live_first_part<8, true>"
    un = 2 if (variant & 0xA0) == 0x20 else 8 if (variant & 0xA0) == 0x80 else 4
    if variant & 4:
        return "k_scan_live_first<%d, %s, %s>" % (un, "true" if variant & 1 else "false", "true" if variant & 0x40 else "false")
    if variant == 0x43:
        return "k_scan_compact<4, true, true, false, true>"
    return "k_scan_compact<%d, %s, %s, false, false>" % (un, "true" if variant & 1 else "false", "true" if variant & 2 else "false")


def lib_srchash():
    """Digest of the sources the loaded libpie_hip.so was built from (written beside it by sph-pie_amd/build.py)."""
    try:
        import sph_pie_amd
        with open(sph_pie_amd.build.HIP_LIB + ".srchash") as f:
            return f.read().strip()
    except Exception:
        return None


def pmc_traffic(kname, default_workload, fname="traffic.json"):
    """HBM bytes per launch of `kname` from the committed rocprofv3 PMC summary of this command (profiles/traffic.json, or
    traffic_wide.json / traffic_zipf.json for those two secondary workloads; written by tools/pmc_traffic.py), or None.
    Only quoted for the workload it was measured on AND for the binary it was measured with: the file records the source
    digest of the library that ran under the profiler; when the loaded library differs (a kernel changed since the PMC
    passes) the figure is stale, and the byte model from this run's own counters is used instead."""
    path = os.path.join(REPO, "profiles", fname)
    if not default_workload or not os.path.exists(path):
        return None, None
    doc = json.load(open(path))
    if not doc.get("lib_srchash") or doc.get("lib_srchash") != lib_srchash():
        return None, "profiles/%s was measured with another build of the library (srchash %s..., loaded %s...): byte model used" % (
            fname, str(doc.get("lib_srchash"))[:12], str(lib_srchash())[:12])
    for k, v in doc.get("kernels", {}).items():
        if kname in k:
            return v, doc.get("source")
    return None, None


def start_js_baseline(args, U, D):
    """Reference-faithful JS (Map of {userId, createdAt, expiresAt} objects, one thread): started first, collected last —
    it runs on one host core while the GPU phases run, and is finished before the multi-thread CPU leg starts."""
    try:
        import shutil
        node = shutil.which("node")
        js = os.path.join(REPO, "oracle", "ref_faithful.js")
        if node and os.path.exists(js) and args.js_rows > 0:
            return subprocess.Popen([node, "--max-old-space-size=12288", js, "--bench", str(args.js_rows), str(max(int(args.js_rows * U // max(args.rows, 1)), 10)), str(D)],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except Exception as ex:  # the JS leg is optional context, never fatal
        log("js baseline skipped:", ex)
    return None


def cpu_baseline(args, U, D, now, cutoff, mask, flags, gpu_result, js_proc):
    """Baseline B2 of BASELINE.md section 3: the oracle (C port of the path) over the same SoA columns at FULL N, on one
    thread and on all host threads, outputs compared with the GPU result; B1 (JS) collected from its subprocess."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np
    import oracle_py
    cores = os.cpu_count() or 1
    N = args.rows
    n_cpu = min(N, args.cpu_rows)
    out = {"unit": "feeds/s", "kind": "port", "host_cores": cores}
    js = None
    if js_proc is not None:
        try:
            so, se = js_proc.communicate(timeout=args.js_timeout)
            if js_proc.returncode == 0:
                js = json.loads(so.strip().splitlines()[-1])
            else:
                log("js baseline failed:", se[-300:])
        except Exception as ex:
            js_proc.kill()
            log("js baseline skipped:", ex)
    t0 = time.perf_counter()
    s, e, u, d = oracle_py.gen_mt(SEED, N, 0, n_cpu, U, D, flags, threads=cores)
    t_gen = time.perf_counter() - t0
    bufs = (np.empty(U, np.int32), np.empty(U + 1, np.int64), np.empty(n_cpu, np.int32))

    def timed(threads, budget):
        reps, t1 = 0, time.perf_counter()
        while True:
            res = oracle_py.scan_mt(s, e, u, d, U, now, cutoff, mask, threads, out=bufs)
            reps += 1
            dt = time.perf_counter() - t1
            if dt >= budget or reps >= 200:
                return res, reps, dt

    res1, reps1, dt1 = timed(1, args.cpu_seconds)
    resA, repsA, dtA = timed(cores, args.cpu_seconds / 2)
    rows1, rowsA = n_cpu * reps1 / dt1, n_cpu * repsA / dtA
    verified = None
    if gpu_result is not None and n_cpu == N:
        verified = all(np.array_equal(a, b) for a, b in zip(resA, gpu_result))
        if not verified:
            log("WARNING: the GPU result differs from the CPU oracle on the full table")
    out.update({
        "value": rows1 * U / N, "cores": 1, "sessions_per_sec": rows1,
        "all_threads": {"value": rowsA * U / N, "cores": cores, "sessions_per_sec": rowsA, "reps": repsA, "seconds": dtA},
        "gpu_result_equals_cpu_result": verified,
        "sample": "%d reps x %d rows (%s) of the same corpus (U=%d, D=%d), oracle/pie_oracle.c scan on 1 thread, %.1f s; all-thread "
                  "leg: %d threads, %d reps, %.1f s; corpus generated on the host in %.1f s; feeds/s = rows/s x U/N" %
                  (reps1, n_cpu, "the FULL table" if n_cpu == N else "first rows", U, D, dt1, cores, repsA, dtA, t_gen),
    })
    if js is not None:
        out["js_reference_faithful"] = js
        out["js_reference_faithful"]["extrapolated_seconds_per_1e8_row_scan"] = 1e8 / js["sessions_per_sec"]
    return out


def main():
    # exactly ONE JSON line may reach stdout: libraries (RCCL prints a version banner) write to fd 1, so fd 1 is
    # pointed at stderr for the whole run and the result line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=5, help="timed regions of --steps steps each (median / min / max reported)")
    ap.add_argument("--rows", type=int, default=10 ** 8, help="sessions of the whole corpus (sharded over the ranks)")
    ap.add_argument("--users", type=int, default=10 ** 5, help="users of the whole corpus")
    ap.add_argument("--disc", type=int, default=32)
    ap.add_argument("--order", choices=["random", "clustered", "time"], default="random",
                    help="random (SURVEY.md 8d); clustered: rows of a user contiguous; time: rows in order of creation, as a session "
                         "store appends them — every live row sits at the end of the table")
    ap.add_argument("--users-dist", choices=["uniform", "zipf"], default="uniform", help="zipf: Zipf(1.1) over the users")
    ap.add_argument("--variant", choices=["auth", "interval"], default="auth")
    ap.add_argument("--query", choices=["spec", "wide", "future"], default="spec",
                    help="spec: now=T0-6h, cutoff=T0-61d, 16/32 disciplines (SURVEY.md §8d); wide: ~25%% selected")
    ap.add_argument("--cpu-rows", type=int, default=10 ** 8, help="rows of the CPU baseline (default: the full table)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--js-rows", type=int, default=10 ** 7)
    ap.add_argument("--js-timeout", type=float, default=240.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the full-read / D2H legs (secondary workloads, sweeps)")
    ap.add_argument("--no-mixed-leg", action="store_true",
                    help="skip the heterogeneous 64-query leg (tools/run_pmc.sh: its launches carry the headline kernel's name and would "
                         "be averaged into the headline's PMC figures)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE corpus of --rows sharded by user hash (BASELINE configs[3]); weak = --rows per rank")
    ap.add_argument("--gather-batch", type=int, default=8, help="multi-GPU: scans per all-gather (1 = one gather per scan)")
    ap.add_argument("--exchange", choices=["union", "lists"], default="union",
                    help="multi-GPU, batched scans: what a step's all-gather moves — union: per user the union of the Q row lists "
                         "+ a query mask per row (an eighth of the bytes; falls back to lists on skewed users); lists: Q messages")
    ap.add_argument("--transport", choices=["nccl", "gloo"], default=os.environ.get("PIE_BENCH_TRANSPORT", "nccl"),
                    help="multi-GPU: nccl = RCCL over xGMI (the product path); gloo = the ranks' messages staged through pinned host "
                         "memory and exchanged between CPU tensors — the rehearsal transport (several ranks on one GPU with "
                         "PIE_BENCH_DEVICE=0, where RCCL refuses to form a group); everything else is the code the RCCL run executes")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="every n-th timed step carries HIP events around the scan kernels (each pair drains the stream for a "
                         "few microseconds); 0 = min(16, steps // 3), i.e. at least three samples")
    ap.add_argument("--mixed-rows", type=int, default=1000, help="--mode mixed: sessions created and sessions touched per step")
    ap.add_argument("--mixed-clock", choices=["query", "end"], default="query",
                    help="--mode mixed: sessions are created at the query's clock (default) or after the corpus's last createdAt")
    ap.add_argument("--mode", choices=["scan", "expired", "mixed", "archive"], default="scan",
                    help="scan: the headline feed scan; expired: the 'next' row of SURVEY.md 8f-1 — newly-expired change "
                         "predicate -> ordered dispatch queue (reads only the end column: 8 B/row algorithmic); mixed: every step "
                         "creates --mixed-rows sessions (append) and touches as many (set_end) before its scan: the upkeep of the "
                         "derived key columns inside the timed region; archive: the reference's archive chain (sqlProvider.js:758-816: "
                         "group min -> threshold -> whole groups in first-appearance order) as one device chain per step")
    ap.add_argument("--queries-per-launch", type=int, default=64,
                    help="every step is ONE batched scan of Q queries (Q feed requests, each with its own `now`, answered by one table "
                         "pass: pie_scan_batch_*); value counts Q x U feeds per step; 1 = one query per step (single_query reports that "
                         "form in every run)")
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2],
                    help="scans in flight: 2 = the table pass of step i+1 overlaps the scatter/order tail of step i")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "PIE_BENCH_DEVICE" in os.environ:   # rehearsal only: several ranks on one GPU (if RCCL accepts it)
        local_rank = int(os.environ["PIE_BENCH_DEVICE"])
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    N, U, D = args.rows, args.users, args.disc
    js_proc = None
    if args.users_dist == "zipf":
        args.no_cpu_baseline = True   # the CPU leg generates the uniform-user corpus
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "scan":
        js_proc = start_js_baseline(args, U, D)

    import numpy as np
    import torch
    import torch.distributed as dist
    import sph_pie_amd as pie
    from sph_pie_amd.shard import BatchedFeeds, HipShardBackend, ShardedFeeds

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PIE_BENCH_FORCE_GATHER=1 exercises the RCCL exchange step with a single rank (rehearsal on a 1-GPU box)
    gather = world > 1 or os.environ.get("PIE_BENCH_FORCE_GATHER") == "1"
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.transport == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if args.transport == "nccl" else torch.device("cpu")   # where the small control collectives run
    transport = "device" if args.transport == "nccl" else "host"

    flags = (pie.PIE_GEN_INTERVAL if args.variant == "interval" else 0) | (pie.PIE_GEN_CLUSTERED if args.order == "clustered" else 0) | \
        (pie.PIE_GEN_TIME_ORDERED if args.order == "time" else 0)
    if args.query == "spec":
        now, cutoff, mask = T0_MS - 6 * 3600 * 1000, T0_MS - 61 * DAY, 0x5555555555555555
    elif args.query == "future":   # nothing is live: the table pass with no candidate rows (pure key streaming)
        now, cutoff, mask = T0_MS + DAY, T0_MS - 61 * DAY, 0x5555555555555555
    else:
        now, cutoff, mask = T0_MS - 100 * DAY, T0_MS - 61 * DAY, 0x5555555555555555
    if os.environ.get("PIE_BENCH_MASK"):   # experiments: another discipline mask (changes M, not the candidate rows)
        mask = int(os.environ["PIE_BENCH_MASK"], 0)
    mask &= (1 << D) - 1 if D < 64 else 2 ** 64 - 1

    pie.build_hip()
    ctx = pie.PieScan(local_rank)
    t_gen = time.perf_counter()
    strong = world > 1 and args.scaling == "strong"
    if args.users_dist == "zipf":
        cdf = pie.zipf_cdf(U)
        if world > 1 and not strong:
            ctx.gen_synthetic_cdf(SEED, N * world, N * rank, N, U, D, flags, cdf)
        else:
            ctx.gen_synthetic_cdf(SEED, N, 0, N, U, D, flags, cdf)
    elif world > 1 and not strong:
        ctx.gen_synthetic(SEED, N * world, N * rank, N, U, D, flags)   # weak: rows [rank*N, (rank+1)*N) of a world*N-row corpus
    else:
        ctx.gen_synthetic(SEED, N, 0, N, U, D, flags)
    shard_info = None
    if strong:
        # ONE corpus, sharded on the device: this rank keeps the rows of the users that hash to it (pie_shard_of), users
        # re-numbered densely; nothing leaves the GPU
        shard_info = ctx.shard_table(rank, world)
    ctx.set_disciplines(mask, D)
    n_local, u_local = ctx.n, ctx.n_users
    log("rank %d: %d rows / %d users resident after %.2f s" % (rank, n_local, u_local, time.perf_counter() - t_gen))
    info = ctx.table_info()

    backend = HipShardBackend(ctx, dev) if gather else None
    # --gather-batch scans per all-gather: the collective is latency-bound at this size, so the same lists travel in
    # fewer, larger messages
    feeds = ShardedFeeds(backend, rank, world, u_local, always_collective=gather, batch=args.gather_batch, transport=transport) if gather else None

    expired_window = (T0_MS - 30 * DAY, T0_MS - 29 * DAY)   # one day of expiries: ~0.83 % of the rows queue up
    # the archive chain: a group (user) qualifies iff its earliest session is at least an hour older than `now`; with `now` two
    # hours into the corpus ~29 % of the groups (those with a session in the corpus's first hour) and of the rows are queued
    archive_query = (T0_MS - 120 * DAY + 2 * 3600 * 1000, 3600 * 1000)
    Q = max(1, min(args.queries_per_launch, pie.PIE_BATCH_MAX))
    # the batch: Q requests that arrived within a few seconds of each other — each samples its own clock (sessionStore.js:67),
    # same day's cutoff, same role mask; query 0 is the single-query workload
    batch_queries = [(now - 977 * q, cutoff, mask) for q in range(Q)]
    bfeeds = BatchedFeeds(backend, rank, world, u_local, q_max=Q, always_collective=gather, union=args.exchange == "union",
                          steps_per_gather=args.gather_batch, transport=transport) if gather and Q > 1 else None
    exchange_state = {"format": args.exchange if bfeeds is not None else None, "fallback": None}

    def exchange_steps(k):
        """k batched steps with their all-gathers; the union form declines on skewed users (every rank sees it in the gathered
        lengths, so every rank switches together) and the per-query messages take over"""
        nonlocal bfeeds
        from sph_pie_amd.shard import UnionOverflow
        try:
            last = bfeeds.run_steps(k, batch_queries)
            if last is None:  # a row list outgrew the messages: capacity was raised, once more
                last = bfeeds.run_steps(1, batch_queries)
            return last
        except UnionOverflow as ex:
            while ctx._batches:   # nothing of the declined run stays in flight
                ctx.scan_batch_finish()
            exchange_state["format"], exchange_state["fallback"] = "lists", str(ex)
            bfeeds = BatchedFeeds(backend, rank, world, u_local, q_max=Q, always_collective=gather, union=False, steps_per_gather=args.gather_batch,
                                  transport=transport)
            return exchange_steps(k)

    def run_steps(k):
        """k steps; with the exchange step the all-gather of step i overlaps the scan of step i+1."""
        last = None
        if args.mode == "expired":
            for _ in range(k):
                last = ctx.expired_queue(expired_window[0], expired_window[1], fetch=False)
            return last
        if args.mode == "archive":
            for _ in range(k):
                last = ctx.archive_queue(archive_query[0], archive_query[1], fetch=False)
            return last
        if args.mode == "mixed":
            # a live server between two feed scans: logins (createSession -> append) and touches (touchSession -> set_end),
            # then one scan at the step's own clock; one scan at a time (table changes need an idle context)
            kk = args.mixed_rows
            for _ in range(k):
                mixed_state["step"] += 1
                t_step = now + mixed_state["step"] * 1000
                # the clock sessions are created at: the query's `now` (the spec query sits 6 h before the corpus's last
                # createdAt, so new rows fall among existing ones), or the corpus's end (a live store: created after everything)
                t_new = (T0_MS if args.mixed_clock == "end" else now) + mixed_state["step"] * 1000
                t_a = time.perf_counter()
                # created within the last second, in order of creation (the order a store appends them in)
                st_new = np.sort(np.full(kk, t_new, np.int64) - mixed_state["rng"].integers(0, 1000, kk))
                ctx.append_rows(st_new, st_new + 43200000, mixed_state["rng"].integers(0, u_local, kk).astype(np.int32),
                                mixed_state["rng"].integers(0, D, kk).astype(np.int32), u_local)
                t_b = time.perf_counter()
                rows = mixed_state["rng"].integers(0, ctx.n, kk).astype(np.int32)
                ctx.set_end(rows, np.full(kk, t_new + 43200000, np.int64))
                t_c = time.perf_counter()
                last = ctx.scan_device(t_step, cutoff)
                t_d = time.perf_counter()
                mixed_state["append_s"] += t_b - t_a
                mixed_state["touch_s"] += t_c - t_b
                mixed_state["scan_s"] += t_d - t_c
                mixed_state["n"] += 1
            return last
        if not gather:
            if Q > 1:
                return ctx.scan_batch_pipelined(k, batch_queries)
            if args.depth == 1:
                for _ in range(k):
                    last = ctx.scan_device(now, cutoff)
                return last
            return ctx.scan_pipelined(k, now, cutoff)
        if bfeeds is not None:
            return exchange_steps(k)
        last = feeds.run_steps(k, now, cutoff)
        if last is None:  # a row list outgrew the message: capacity was raised, redo synchronously once
            last = feeds.scan_and_gather(now, cutoff)
        return last

    mixed_state = {"step": 0, "append_s": 0.0, "touch_s": 0.0, "scan_s": 0.0, "n": 0, "rng": None}
    if args.mode == "mixed":
        mixed_state["rng"] = np.random.default_rng(SEED)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region(fn, k):
        """EXACTLY k steps between two fences; max over ranks.  -> seconds"""
        fence()
        t0 = time.perf_counter()
        out = fn(k)
        fence()
        dt = time.perf_counter() - t0
        if gather:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    run_steps(max(args.warmup, 1))
    fence()
    ctx.stats_reset()
    mixed_state.update({"append_s": 0.0, "touch_s": 0.0, "scan_s": 0.0, "n": 0})
    # HIP events around the scan kernels, on the stream they are launched on; every 16th step (at least three per region)
    # carries them (an event between two kernels drains the pipeline for a few microseconds)
    profile_every = args.profile_every if args.profile_every > 0 else max(1, min(16, args.steps // 3))
    ctx.set_profiling(1 if args.mode in ("expired", "archive") else profile_every)
    region_ms, kernel_ms_regions, scan_ms_regions, n_prof = [], [], [], 0
    last = None
    mixed_regions = []
    arch_acc = [0.0, 0, 0]
    for _ in range(max(args.repeat, 1)):
        mixed_state.update({"append_s": 0.0, "touch_s": 0.0, "scan_s": 0.0, "n": 0})
        dt, last = timed_region(run_steps, args.steps)
        region_ms.append(dt * 1e3 / args.steps)
        if mixed_state["n"]:
            mixed_regions.append({k: mixed_state[k] * 1e3 / mixed_state["n"] for k in ("append_s", "touch_s", "scan_s")})
        st = ctx.stats()
        if st["n_profiled"]:
            kernel_ms_regions.append(st["k1_ms_sum"] / st["n_profiled"])
            scan_ms_regions.append(st["scan_ms_sum"] / st["n_profiled"])
            n_prof += st["n_profiled"]
        if args.mode == "archive":
            a = ctx.archive_stats()
            arch_acc[0] += a[0]
            arch_acc[1] += a[1]
            arch_acc[2] = a[2]
        ctx.stats_reset()
    arch_stats = tuple(arch_acc)
    ctx.set_profiling(0)
    st = ctx.stats()
    # With batch lanes (pie_set_batch_lanes: several batches side by side on the chip) a launch's event time is its duration
    # while it SHARES the GPU with the other lanes' launches.  The roofline figure wants the kernel alone: the same steps once
    # more on one lane, profiled, after the timed regions (not part of value).
    lanes = ctx.batch_lanes()
    k1_ms_lanes = None
    if lanes > 1 and Q > 1 and not gather and args.mode == "scan":
        k1_ms_lanes = statistics.median(kernel_ms_regions) if kernel_ms_regions else None
        ctx.set_batch_lanes(1)
        ctx.scan_batch_pipelined(max(args.warmup, 1), batch_queries)
        ctx.stats_reset()
        ctx.set_profiling(profile_every)
        kernel_ms_regions, scan_ms_regions, n_prof = [], [], 0
        for _ in range(max(args.repeat, 1)):
            ctx.scan_batch_pipelined(args.steps, batch_queries)
            s1l = ctx.stats()
            if s1l["n_profiled"]:
                kernel_ms_regions.append(s1l["k1_ms_sum"] / s1l["n_profiled"])
                scan_ms_regions.append(s1l["scan_ms_sum"] / s1l["n_profiled"])
                n_prof += s1l["n_profiled"]
            ctx.stats_reset()
        ctx.set_profiling(0)
        st = ctx.stats()
        ctx.set_batch_lanes(int(os.environ.get("PIE_BATCH_LANES", "0")))
    batch_ms = None
    def union_masks64(res, r, mu):
        """the 64-bit query mask of every union row of rank r in a gathered union result"""
        mk = res["masks"][r][:mu].cpu().numpy().astype(np.uint32).astype(np.uint64)
        if "masks_hi" in res:
            mk |= res["masks_hi"][r][:mu].cpu().numpy().astype(np.uint32).astype(np.uint64) << np.uint64(32)
        return mk

    union_rows = None
    if Q > 1 and not gather and args.mode == "scan":
        batch_ms = list(last)
        last = batch_ms[0]
        union_rows = ctx.batch_union_device_ptrs()[4] or None
    elif Q > 1 and gather and args.mode == "scan":
        if "u_offsets" in last:   # union exchange: a query's rows are the union rows that carry its bit
            union_rows = int(last["lengths"][rank])
            mk = union_masks64(last, rank, union_rows)
            batch_ms = [int(((mk >> np.uint64(q)) & np.uint64(1)).sum()) for q in range(Q)]
        else:
            batch_ms = [int(x) for x in last["lengths"][rank]]
    m = batch_ms[0] if (batch_ms is not None and gather) else (last if not gather else int(last["lengths"][rank]))
    ms_per_step = statistics.median(region_ms)
    k1_ms = statistics.median(kernel_ms_regions) if kernel_ms_regions else 0.0

    scan_only_ms = None
    gather_ok = None
    if gather and args.mode == "scan":
        # the gathered lists of this rank (as every rank received them) against this rank's own result of the same query
        if bfeeds is not None:
            ok = True
            union = "u_offsets" in last
            if union:   # every query's list is a filter of the union rows (in order); the offsets follow from the masks
                mu = int(last["lengths"][rank])
                u_rows, u_masks = last["rows"][rank].cpu().numpy()[:mu], union_masks64(last, rank, mu)
                u_off = last["u_offsets"][rank].cpu().numpy()[: u_local + 1].astype(np.int64)
            for q, (qn, qc, qm) in enumerate(batch_queries):
                ctx.set_disciplines(qm, D)
                ctx.scan_device(qn, qc)
                _, own_off, own_idx = ctx.read_results()
                if union:
                    sel = ((u_masks >> np.uint64(q)) & np.uint64(1)) == 1
                    csum = np.concatenate([[0], np.cumsum(sel)])
                    ok = ok and np.array_equal(u_rows[sel], own_idx) and np.array_equal(csum[u_off], own_off)
                    continue
                ok = ok and int(last["lengths"][rank, q]) == own_idx.size and \
                    np.array_equal(last["offsets"][rank, q].cpu().numpy()[: u_local + 1], own_off.astype(np.int32)) and \
                    np.array_equal(last["rows"][rank, q].cpu().numpy()[: own_idx.size], own_idx)
            ctx.set_disciplines(mask, D)
        else:
            ctx.scan_device(now, cutoff)
            _, own_off, own_idx = ctx.read_results()
            ok = int(last["lengths"][rank]) == own_idx.size and \
                np.array_equal(last["offsets"][rank].cpu().numpy()[: u_local + 1], own_off.astype(np.int32)) and \
                np.array_equal(last["rows"][rank].cpu().numpy()[: own_idx.size], own_idx)
        t_ok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        gather_ok = bool(int(t_ok.item()))
        if not gather_ok:
            log("WARNING: rank %d: gathered lists differ from the local result" % rank)
        # scan-only throughput beside scan + gather — the same K steps without the exchange, after the
        # timed regions (not part of value); max over ranks like the headline
        scan_only = (lambda k: ctx.scan_batch_pipelined(k, batch_queries)) if Q > 1 else (lambda k: ctx.scan_pipelined(k, now, cutoff))
        so = [timed_region(scan_only, args.steps)[0] * 1e3 / args.steps for _ in range(3)]
        scan_only_ms = spread(so)

    # totals over the ranks (strong scaling: every rank holds a different number of rows / users)
    tot_rows, tot_users = n_local, u_local
    if gather:
        t = torch.tensor([n_local, u_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot_rows, tot_users = int(t[0]), int(t[1])

    line = None
    if rank == 0 and args.mode == "archive":
        a_ms, a_calls, a_alg = arch_stats
        chain_ms = a_ms / max(a_calls, 1)
        line = {
            "metric": "archive chain (SURVEY 8f-1, sqlProvider.js:758-816): sessions scanned/sec; value counts table rows per chain",
            "value": tot_rows / (ms_per_step * 1e-3), "unit": "sessions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "timing": {"timed_regions": len(region_ms), "ms_per_step": spread(region_ms)},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "archive chain on %d sessions / %d users: group = user, now = corpus start + 2 h, window 1 h" % (N, U),
                       "queued_rows": int(last), "queued_fraction": int(last) / max(n_local, 1)},
            "roofline": {"bound": "hbm", "kernel": "the whole chain: k_group_stats | k_arch_qualify | radix sort (U) | k_arch_rank | k_arch_count | "
                                                  "k_block_prefix | k_arch_write | radix sort (M)",
                         "alg_bytes_per_chain": a_alg, "alg_bytes_note": "20 B/row group statistics (start, end, user) + 12 B/row selection (end, user) + 4 B per queued row",
                         "chain_ms": chain_ms, "chains_timed": a_calls, "achieved": a_alg / (chain_ms * 1e-3) / 1e9 if chain_ms > 0 else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_alg / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if chain_ms > 0 else None,
                         "traffic": None},
        }
        os.write(result_fd, (json.dumps(line) + "\n").encode())
        ctx.close()
        return
    if rank == 0:
        variant = st["k1_variant"]
        info = ctx.table_info()   # again: an ordered run, if the scans called for one, was built during the warm-up
        rides = (args.depth == 2 or gather) and os.environ.get("PIE_K2_RIDE") != "0" and (variant & ~0x840) == 0x485
        kname = kernel_name(variant, rides, args.mode)
        if batch_ms is not None:
            kname = "k_scan_batch_with_tail<8, true, %s, %s>" % ("unsigned char" if variant & 0x800 else "unsigned short", "true" if Q > 32 else "false")
            if variant & 0x2000:
                kname = "k_ord_batch_scan_t<%s, 4>" % ("unsigned char" if variant & 0x800 else "unsigned short")
        default_workload = (N, U, D, args.order, args.variant, args.query, args.mode, args.users_dist, world) == \
            (10 ** 8, 10 ** 5, 32, "random", "auth", "spec", "scan", "uniform", 1)
        alg = (8.0 if args.mode == "expired" else 24.0) * n_local
        traffic_doc, traffic_src = pmc_traffic(kname, default_workload and (batch_ms is None or Q == 64))
        # the two secondary workloads of the ordered run that have PMC passes of their own
        base_shape = (N, U, D, args.order, args.variant, args.mode, world) == (10 ** 8, 10 ** 5, 32, "random", "auth", "scan", 1)
        if base_shape and batch_ms is None and args.query == "wide" and args.users_dist == "uniform":
            traffic_doc, traffic_src = pmc_traffic(kname, True, "traffic_wide.json")
        if base_shape and batch_ms is None and args.query == "spec" and args.users_dist == "zipf":
            traffic_doc, traffic_src = pmc_traffic(kname, True, "traffic_zipf.json")
        traffic = traffic_doc["hbm_bytes_per_launch"] if traffic_doc else None
        # byte model from the run's own counters (keyed form): key stream + one 128-B sector per candidate payload record
        # and per ambiguous `end` + one 64-B write per selected row (bucket slot) + K2's outputs
        model = None
        if variant & 0x2000:
            # the ordered run (pie_ordered.h): positions = rows + spare slots; dense form: every key, the 16-byte record of every
            # slice that may hold a live row (bounded by all of them), 4 B staged per selected row; keyed forms: the key
            # stream, one 64-B sector per candidate record (a user's candidates are neighbours), 4 / 8 B staged per selected row
            pos_n = info["ordered_positions"] or n_local
            kb = 1 if variant & 0x800 else 2
            if not variant & 0x400:
                model = pos_n * 2 + pos_n * 16 + int(m) * 4
            elif batch_ms is not None:
                model = pos_n * kb + st["candidates"] * 64 + max(batch_ms) * 8
            else:
                model = pos_n * kb + st["candidates"] * 64 + int(m) * 4
        elif batch_ms is not None:
            kb = 1 if variant & 0x800 else 2
            # per launch: the key stream; one 128-B sector per candidate payload record; one 64-B sector per union bucket store
            # (whatever Q); the tail of the batch before: histogram + bucket records read (4 B per user, a sector per bucket),
            # the union written (uoff 8 B per user, 8 or 12 B per union row), the next span zeroed
            mu_rows = union_rows if union_rows else max(batch_ms)
            model = n_local * kb + st["candidates"] * 128 + mu_rows * 64 + u_local * (4 + 64 + 8) + mu_rows * (12 if Q > 32 else 8) + u_local * 4
        elif args.mode == "scan" and variant & 0x400:
            kb = 1 if variant & 0x800 else 2
            model = n_local * kb + st["candidates"] * 128 + st["key_ambiguous"] * 128 + int(m) * 64 + u_local * 12 + int(m) * 4
        elif args.mode == "scan" and not variant & 4:
            model = n_local * (24 if not variant & 2 else 20) + int(m) * (64 + 4 + (4 if variant & 2 else 0)) + u_local * 12
        basis = traffic if traffic else model
        achieved = (basis / (k1_ms * 1e-3) / 1e9) if basis and k1_ms > 0 else None
        per_step_units = (tot_users * (Q if batch_ms is not None else 1) if args.mode in ("scan", "mixed") else tot_rows)
        line = {
            "metric": "feeds/sec + sessions scanned/sec, 10^8 synthetic sessions, 1/2/4/8 MI355X" if args.mode == "scan" else
                      "mixed workload: per step %d sessions created + %d touched + one feed scan; feeds/sec" % (args.mixed_rows, args.mixed_rows) if args.mode == "mixed" else
                      "expired-queue pass (SURVEY 8f-1): sessions scanned/sec; value counts table rows, not feeds",
            "value": per_step_units / (ms_per_step * 1e-3),
            "unit": "feeds/s" if args.mode in ("scan", "mixed") else "sessions/s",
            ("logical_sessions_per_sec" if (batch_ms is not None or variant & 0x400) else "sessions_per_sec"):
                tot_rows * (Q if batch_ms is not None else 1) / (ms_per_step * 1e-3),
            "logical_sessions_note": "rows of the table x queries answered per second: what the answers cover, NOT bytes scanned — the keyed "
                                     "pass reads a 1-byte liveness key per row and one 16-byte record per candidate row (physical_bytes_per_step); "
                                     "sessions really scanned at 24 B each per second is roofline.full_read.sessions_per_sec",
            "physical_bytes_per_step": basis,
            "queries_per_launch": Q if batch_ms is not None else 1,
            "batch_lanes": lanes if batch_ms is not None else None,
            "batch_lanes_note": "batches in flight are dealt to this many independent streams of the context (pie_set_batch_lanes; chosen by "
                                "table size) and run side by side on the chip; three batches in flight per lane",
            "union_rows_rank0": union_rows,
            "table_passes_per_sec": 1.0 / (ms_per_step * 1e-3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "timing": {"timed_regions": len(region_ms), "steps_per_region": args.steps, "ms_per_step": spread(region_ms),
                       "ms_per_step_all": region_ms,
                       "note": "every region is exactly --steps steps between barrier + synchronize fences (max over ranks); "
                               "value and ms_per_step are the median region"},
            "scan_only_ms_per_step": scan_only_ms, "gather_verified": gather_ok,
            "exchange": None if bfeeds is None else {
                "format": exchange_state["format"], "fallback": exchange_state["fallback"], "transport": args.transport,
                "bytes_per_rank_per_step": int(bfeeds.sets["L"]) * 4 * (1 if bfeeds.union else Q) if bfeeds.sets else None,
                "steps_per_gather": bfeeds.steps_per_gather,
                "note": "what every rank contributes to a step's all-gather (and receives from every other rank): union = per user "
                        "the union of the Q row lists in (start, row) order + a query mask per row; lists = Q messages of offsets + rows"},
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {
                "workload": "BASELINE config %s: %d sessions / %d users / %d disciplines%s, SoA int64 start/end + int32 "
                            "user/disc, splitmix64 seed 0x5EED5EED, %s order, %s users, %s variant, %s query" %
                            ("3" if world == 1 else "4", N if (strong or world == 1) else N * world, U, D,
                             "" if world == 1 else (" sharded by user hash over %d GPUs" % world if strong else " per GPU (weak)"),
                             args.order, args.users_dist, args.variant, args.query),
                "sessions_total": tot_rows, "users_total": tot_users, "sessions_rank0": n_local, "users_rank0": u_local,
                "disciplines": D, "selected_rows_rank0": int(m), "selected_rows_per_query": batch_ms,
                "parallelism": "user-hash shards x%d (device-side partition of one corpus), RCCL all-gather of per-user offsets + row "
                               "lists (%d queries per collective), overlapped with the next scans" % (world, Q if Q > 1 else args.gather_batch) if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "kernel_variant": hex(variant),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic, "traffic_source": traffic_src, "lib_srchash": lib_srchash(),
                # with batch lanes the launches of consecutive steps overlap: HBM bytes per second while the pipeline runs =
                # bytes per launch over the STEP time (one launch per step), beside bytes over the launch's own duration above
                "achieved_pipelined": (basis / (ms_per_step * 1e-3) / 1e9) if (basis and k1_ms_lanes is not None) else None,
                "frac_pipelined": (basis / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if (basis and k1_ms_lanes is not None) else None,
                "pipelined_note": None if k1_ms_lanes is None else "achieved / frac: bytes per launch over the launch's own time (alone on the chip); "
                                  "*_pipelined: the same bytes over ms_per_step — what the HBM delivers while %d lanes' launches overlap" % lanes,
                "traffic_model": model,
                "traffic_model_note": "bytes from this run's own counters: key stream N x key bytes + 128 B per candidate payload record + 128 B "
                                      "per ambiguous `end` + 64 B per selected row stored + K2 outputs; `achieved` uses the PMC traffic when the "
                                      "committed profile covers this kernel and workload, else this model",
                "alg_bytes_per_launch": alg,
                "alg_equiv_gbs": (alg / (k1_ms * 1e-3) / 1e9) if k1_ms > 0 else None,
                "alg_equiv_note": "24 B/row x rows over kernel time: what a scan that read every byte would need to sustain to match this "
                                  "kernel; NOT a bandwidth — the keyed pass streams a 1- or 2-byte liveness key per row and gathers one 16-byte "
                                  "payload record per candidate (DESIGN.md sections 3, 4, 6); see roofline_full_read for the every-byte form",
                "kernel_ms": k1_ms, "kernel_ms_regions": spread(kernel_ms_regions) if kernel_ms_regions else None,
                "kernel_ms_sharing_the_chip": k1_ms_lanes,
                "kernel_ms_note": None if k1_ms_lanes is None else "kernel_ms: the launch alone on the chip (the same steps on ONE lane, after the timed "
                                  "regions); kernel_ms_sharing_the_chip: the same launch's event time inside the timed regions, where %d lanes' "
                                  "launches overlap (ms_per_step < kernel_ms is that overlap)" % lanes,
                "launches_timed": n_prof,
                "note": "the timed launch is the table pass of one scan plus, in its first blocks, the offsets + order kernel of the scan "
                        "before it" if rides else "the timed launch is the table pass",
                # latency of one scan, first kernel start -> last kernel end; with two scans in flight the tail of
                # scan i is queued behind the table pass of scan i+1, so this exceeds ms_per_step by design
                "scan_latency_ms": statistics.median(scan_ms_regions) if scan_ms_regions else None,
                "scans_in_flight": args.depth if not gather else 1,
                "k1_blocks": st["k1_blocks"],
            },
            "index": {"index_build_ms": info["index_build_ms"], "derived_bytes": info["derived_bytes"],
                      "table_bytes": info["table_bytes"], "workspace_bytes": info["workspace_bytes"],
                      "ordered_run": {"rows": info["ordered_rows"], "bytes": info["ordered_bytes"], "build_ms": info["ordered_build_ms"],
                                      "builds": info["ordered_builds"], "positions": info["ordered_positions"], "respreads": info["ordered_respreads"]},
                      "note": "the keyed pass reads derived columns built at load (outside the timed region) and kept in step by every "
                              "writer of `end`; index_build_ms = one full build on this table"},
        }

    if rank == 0 and args.mode == "mixed" and mixed_regions:
        med = lambda k: statistics.median(r[k] for r in mixed_regions)
        line["mixed"] = {"append_ms": med("append_s"), "touch_ms": med("touch_s"), "scan_ms": med("scan_s"),
                         "per_region": mixed_regions, "rows_per_step": args.mixed_rows, "table_rows_now": ctx.n,
                         "note": "host wall time per step: append = pie_append_rows (H2D of the new rows + their keys / payload records), "
                                 "touch = pie_set_end (H2D of the row list + the kernel that rewrites end and both keys), scan = one "
                                 "synchronous scan; the key columns stay in step, so the scan keeps its keyed form"}
    gpu_result = None
    if world == 1 and args.mode == "scan" and not gather and not args.no_extra and Q > 1:
        # ---- one query per step (the unit SURVEY.md 8d defines: one scan = one query over all N rows), two scans in flight
        ctx.scan_pipelined(max(args.warmup, 1), now, cutoff)
        ctx.stats_reset()
        ctx.set_profiling(profile_every)
        sq_regions, sq_k1, sq_lat = [], [], []
        for _ in range(max(args.repeat, 1)):
            dt, m1 = timed_region(lambda k: ctx.scan_pipelined(k, now, cutoff), args.steps)
            sq_regions.append(dt * 1e3 / args.steps)
            s1 = ctx.stats()
            if s1["n_profiled"]:
                sq_k1.append(s1["k1_ms_sum"] / s1["n_profiled"])
                sq_lat.append(s1["scan_ms_sum"] / s1["n_profiled"])
            ctx.stats_reset()
        ctx.set_profiling(0)
        s1 = ctx.stats()
        v1 = s1["k1_variant"]
        rides1 = os.environ.get("PIE_K2_RIDE") != "0" and (v1 & ~0x840) == 0x485
        k1name = kernel_name(v1, rides1, "scan")
        t1_doc, t1_src = pmc_traffic(k1name, default_workload)
        kb1 = 1 if v1 & 0x800 else 2
        model1 = (N * kb1 + s1["candidates"] * 128 + s1["key_ambiguous"] * 128 + int(m1) * 64 + U * 12 + int(m1) * 4) if v1 & 0x400 else None
        basis1 = t1_doc["hbm_bytes_per_launch"] if t1_doc else model1
        sq_ms = statistics.median(sq_regions)
        sq_kms = statistics.median(sq_k1) if sq_k1 else 0.0
        ach1 = (basis1 / (sq_kms * 1e-3) / 1e9) if basis1 and sq_kms > 0 else None
        line["single_query"] = {
            "value": U / (sq_ms * 1e-3), "unit": "feeds/s", "ms_per_step": sq_ms, "ms_per_step_spread": spread(sq_regions),
            "selected_rows": int(m1),
            "roofline": {"bound": "hbm", "kernel": k1name, "kernel_variant": hex(v1), "achieved": ach1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (ach1 / HBM_PEAK_GBS) if ach1 else None,
                         "traffic": t1_doc["hbm_bytes_per_launch"] if t1_doc else None, "traffic_source": t1_src, "traffic_model": model1,
                         "kernel_ms": sq_kms, "kernel_ms_regions": spread(sq_k1) if sq_k1 else None,
                         "alg_equiv_gbs": (24.0 * N / (sq_kms * 1e-3) / 1e9) if sq_kms > 0 else None,
                         "scan_latency_ms": statistics.median(sq_lat) if sq_lat else None},
            "note": "one query per scan, two scans in flight (the r01 headline form); the timed launch is the keyed table pass plus, in its "
                    "first blocks, the offsets + order kernel of the scan before it",
        }
    if world == 1 and args.mode == "scan" and not gather and not args.no_extra and Q > 1:
        # ---- other batches through the same pass, same table, same run (each: warm-up, then --repeat regions of --steps steps)
        def batch_leg(queries, note):
            ctx.scan_batch_pipelined(max(args.warmup, 1), queries)
            regs = []
            ms_l = None
            for _ in range(max(args.repeat, 1)):
                dt, ms_l = timed_region(lambda k: ctx.scan_batch_pipelined(k, queries), args.steps)
                regs.append(dt * 1e3 / args.steps)
            stb = ctx.stats()
            med = statistics.median(regs)
            return {"queries": len(queries), "ms_per_step": med, "ms_per_step_spread": spread(regs), "value": U * len(queries) / (med * 1e-3),
                    "unit": "feeds/s", "us_per_query": med * 1e3 / len(queries), "union_rows": ctx.batch_union_device_ptrs()[4] or None,
                    "selected_rows_per_query": {"min": int(min(ms_l)), "max": int(max(ms_l)), "distinct": len(set(int(x) for x in ms_l))},
                    "distinct_now_cutoff_mask": [len(set(q[k] for q in queries)) for k in range(3)],
                    "candidates": stb["candidates"], "kernel_variant": hex(stb["k1_variant"]), "note": note}

        lim = (1 << D) - 1 if D < 64 else 2 ** 64 - 1
        role_masks = [mask, 0xAAAAAAAAAAAAAAAA & lim, lim, 0x0F0F0F0F0F0F0F0F & lim]
        line["batch_q16"] = batch_leg(batch_queries[:16], "the round-2 headline batch: 16 requests 977 ms apart, same cutoff, same discipline mask")
        if not args.no_mixed_leg:
            line["batch_mixed"] = batch_leg(
                [(now - 977 * q, cutoff - (q % 3) * DAY, role_masks[q % 4]) for q in range(Q)],
                "a heterogeneous batch: every request its own clock, three different cutoffs (requests either side of a day change), four "
                "role masks (16 of 32 disciplines, the other 16, all 32, nibbles): the union is the rows ANY of them selects")
        line["batch_mixed_q16"] = batch_leg(
            [(now - 977 * q, cutoff - (q % 3) * DAY, (mask, 0xAAAAAAAAAAAAAAAA & lim, lim)[q % 3]) for q in range(16)],
            "the 16 heterogeneous queries of tests/test_gpu_parity.py::test_full_size_properties_1e8 (three cutoffs, three masks), timed")
    if world == 1 and args.mode == "scan" and not gather and not args.no_extra:
        # ---- the every-byte form of the same query, same table, same run (SURVEY.md 8d's 24 B/row really read)
        ctx.set_scan_form(0x01)
        for _ in range(3):
            ctx.scan_device(now, cutoff)
        ctx.stats_reset()
        ctx.set_profiling(1)
        n_sync = max(5, min(args.steps, 20))
        for _ in range(n_sync):   # one scan at a time: e0 | K1 | K2 [| K3 K4] | e2
            ctx.scan_device(now, cutoff)
        sf = ctx.stats()
        ctx.set_profiling(0)
        ctx.stats_reset()
        fr_regions = [timed_region(lambda k: ctx.scan_pipelined(k, now, cutoff), args.steps)[0] * 1e3 / args.steps
                      for _ in range(max(args.repeat, 1))]
        ctx.set_scan_form(-1)
        fr_k1 = sf["k1_ms_sum"] / max(sf["n_profiled"], 1)
        fr_scan = sf["scan_ms_sum"] / max(sf["n_profiled"], 1)
        fr_name = kernel_name(0x01, False, "scan")
        fr_doc, fr_src = pmc_traffic(fr_name, default_workload)
        fr_ms = statistics.median(fr_regions)
        line["roofline_full_read"] = {
            "bound": "hbm", "kernel": fr_name, "kernel_variant": hex(sf["k1_variant"]), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "alg_bytes_per_scan": 24.0 * N,
            "t_scan_ms": fr_scan, "achieved": 24.0 * N / (fr_scan * 1e-3) / 1e9, "frac": 24.0 * N / (fr_scan * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "t_scan_note": "first kernel start -> last kernel end of ONE scan (HIP events, scan stream, %d scans one at a time): table pass "
                           "+ offsets/order kernel; frac = 24 B x N / t_scan / peak" % sf["n_profiled"],
            "kernel_ms": fr_k1, "kernel_achieved": 24.0 * N / (fr_k1 * 1e-3) / 1e9, "kernel_frac": 24.0 * N / (fr_k1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "ms_per_step": fr_ms, "ms_per_step_spread": spread(fr_regions),
            "step_frac": 24.0 * N / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_note": "steady state with two scans in flight, same fences and --steps as the headline",
            "feeds_per_sec": U / (fr_ms * 1e-3),
            "traffic": fr_doc["hbm_bytes_per_launch"] if fr_doc else None, "traffic_source": fr_src,
        }
        # ---- what a kernel that ONLY reads reaches on this very GPU, now (libpie_ubench.so: 16-byte nontemporal loads over a
        # buffer of the table's size, eight grid / unroll / split forms, median of 7 launches each): boxes of one pool differ by
        # 5 - 8 % in HBM throughput, so the every-byte scan is also quoted against this box's own read ceiling
        ceiling = None
        try:
            import ctypes
            ub = ctypes.CDLL(pie.build_ubench())
            ub.pie_ubench_read_bw.restype = ctypes.c_int
            ub.pie_ubench_read_bw.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
            ms8 = (ctypes.c_double * 12)()
            ctx.synchronize()
            rc_ub = ub.pie_ubench_read_bw(local_rank, 24 * N, 7, ms8)
            if rc_ub == 0:
                forms = ["grid-stride, 16 blocks/CU, unroll 4", "grid-stride, 16 blocks/CU, unroll 8", "grid-stride, 48 blocks/CU, unroll 4",
                         "grid-stride, 48 blocks/CU, unroll 8", "contiguous ranges, 16 blocks/CU, unroll 4", "contiguous ranges, 16 blocks/CU, unroll 8",
                         "contiguous ranges, 48 blocks/CU, unroll 4", "contiguous ranges, 48 blocks/CU, unroll 8",
                         "four columns (8+8+4+4 B/row), 16 blocks/CU, unroll 2", "four columns, 16 blocks/CU, unroll 4",
                         "four columns, 48 blocks/CU, unroll 2", "four columns, 48 blocks/CU, unroll 4"]
                best = min(range(8), key=lambda i: ms8[i])
                best4 = min(range(8, 12), key=lambda i: ms8[i])
                ceiling = {"bytes": 24 * N, "best_ms": ms8[best], "best_form": forms[best], "best_gbs": 24.0 * N / (ms8[best] * 1e-3) / 1e9,
                           "four_columns_best_ms": ms8[best4], "four_columns_best_form": forms[best4],
                           "four_columns_best_gbs": 24.0 * N / (ms8[best4] * 1e-3) / 1e9,
                           "all_ms": {forms[i]: ms8[i] for i in range(12)},
                           "note": "a kernel that only reads (folds every 16-byte load into a word nobody stores): the streaming-read "
                                   "ceiling of THIS GPU in THIS run; the scan's table pass reads the same bytes and also evaluates, "
                                   "counts and scatters"}
            else:
                log("read-ceiling probe failed: rc %d" % rc_ub)
        except Exception as ex:   # a measurement aid: its absence must not cost the bench line
            log("read-ceiling probe unavailable: %r" % (ex,))
        if ceiling is not None:
            fr = line["roofline_full_read"]
            fr["read_ceiling"] = ceiling
            fr["kernel_frac_of_four_column_read"] = ceiling["four_columns_best_ms"] / fr["kernel_ms"]
            fr["kernel_frac_of_read_ceiling"] = ceiling["best_ms"] / fr["kernel_ms"]
            fr["t_scan_frac_of_read_ceiling"] = ceiling["best_ms"] / fr["t_scan_ms"]
        # ---- the headline loop with counts + offsets delivered to pinned host memory by every scan
        sets = []
        for _ in range(3):
            off_h, off_d, off_addr = ctx.host_alloc(U + 2)
            cnt_h, cnt_d, cnt_addr = ctx.host_alloc(U)
            sets.append((off_h, off_d, off_addr, cnt_h, cnt_d, cnt_addr))

        def run_d2h(k):
            mm = 0
            ctx.scan_begin_packed2(now, cutoff, sets[0][1], U, 0, sets[0][4])
            for i in range(k):
                if i + 1 < k:
                    sx = sets[(i + 1) % 3]
                    ctx.scan_begin_packed2(now, cutoff, sx[1], U, 0, sx[4])
                mm, ready = ctx.scan_finish_packed()
                if not ready:
                    ctx.synchronize()
            return mm

        run_d2h(5)
        d2h_regions = [timed_region(run_d2h, args.steps)[0] * 1e3 / args.steps for _ in range(max(args.repeat, 1))]
        # the host copies of the last scan against a plain read-back of the same result
        k_last = (args.steps - 1) % 3
        counts_dev, offsets_dev, idx_dev = ctx.read_results()
        gpu_result = (counts_dev, offsets_dev, idx_dev)
        host_ok = bool(np.array_equal(sets[k_last][3][:U], counts_dev) and
                       np.array_equal(sets[k_last][0][: U + 1].astype(np.int64), offsets_dev) and int(sets[k_last][0][U + 1]) == idx_dev.size)
        d2h_ms = statistics.median(d2h_regions)
        line["value_with_d2h"] = {
            "value": U / (d2h_ms * 1e-3), "unit": "feeds/s", "ms_per_step": d2h_ms, "ms_per_step_spread": spread(d2h_regions),
            "host_copy_verified": host_ok, "bytes_per_scan_to_host": (2 * U + 2) * 4,
            "note": "every scan's offsets[U+1] + M (int32) and counts[U] land in mapped pinned host memory, written by the scan's own "
                    "offsets kernel (no copy node, no event): complete when the scan's summary is; idx stays in HBM (SURVEY.md 8d "
                    "reports its D2H separately)",
        }
        h_idx = torch.empty(max(int(m), 1), dtype=torch.int32).pin_memory()
        ctx.read_results_into(None, None, h_idx.data_ptr(), h_idx.numel())
        t1 = time.perf_counter()
        for _ in range(20):
            ctx.read_results_into(None, None, h_idx.data_ptr(), h_idx.numel())
        line["value_with_d2h"]["idx_to_host_ms"] = (time.perf_counter() - t1) * 1e3 / 20
        line["value_with_d2h"]["idx_bytes"] = int(m) * 4
        # the figure SURVEY.md 8(d) defines, inside `roofline` (the record's first place to look): algorithmic 24 B x N over the
        # every-byte kernel's time and over t_scan
        fr = line["roofline_full_read"]
        line["roofline"]["full_read"] = {
            "kernel": fr["kernel"], "alg_bytes_per_scan": fr["alg_bytes_per_scan"], "kernel_ms": fr["kernel_ms"], "t_scan_ms": fr["t_scan_ms"],
            "frac_kernel": fr["kernel_frac"], "frac_t_scan": fr["frac"], "ms_per_step": fr["ms_per_step"], "frac_step": fr["step_frac"],
            "sessions_per_sec": N / (fr["t_scan_ms"] * 1e-3), "feeds_per_sec": U / (fr["t_scan_ms"] * 1e-3), "traffic": fr["traffic"],
            "read_ceiling_gbs": fr["read_ceiling"]["best_gbs"] if "read_ceiling" in fr else None,
            "frac_kernel_of_read_ceiling": fr.get("kernel_frac_of_read_ceiling"), "frac_t_scan_of_read_ceiling": fr.get("t_scan_frac_of_read_ceiling"),
            "four_column_read_gbs": fr["read_ceiling"]["four_columns_best_gbs"] if "read_ceiling" in fr else None,
            "frac_kernel_of_four_column_read": fr.get("kernel_frac_of_four_column_read"),
            "read_ceiling_note": "read_ceiling_gbs: what a kernel that only reads reaches on this GPU in this run (libpie_ubench.so, best of "
                                 "eight forms); frac_*_of_read_ceiling = that kernel's time / the scan's",
            "note": "the same query on the same table with the table pass pinned to the form that reads every byte of the four columns "
                    "(24 B/row = SURVEY 8d's algorithmic bytes; PMC traffic = algorithmic): frac_* = 24 B x N / time / 8 TB/s; t_scan = first "
                    "kernel start -> last kernel end of ONE scan; sessions_per_sec = rows really read per second",
        }
        for sx in sets:
            ctx.host_free(sx[2])
            ctx.host_free(sx[5])
        nonempty = int((counts_dev > 0).sum())
        line["config"]["nonempty_feeds_rank0"] = nonempty
        line["nonempty_feeds_per_sec"] = nonempty / (ms_per_step * 1e-3)
    elif world == 1 and args.mode == "scan" and not gather and not args.no_cpu_baseline:
        ctx.scan_device(now, cutoff)
        gpu_result = ctx.read_results()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.mode == "scan":
            line["cpu_baseline"] = cpu_baseline(args, U, D, now, cutoff, mask, flags, gpu_result, js_proc)
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if gather:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
