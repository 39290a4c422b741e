#!/usr/bin/env python3
"""bench.py — headline benchmark of the session-scan -> per-user feed path on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

A step = one scan (one `now`/`cutoff`/discipline-mask query) over the rank's resident session table,
producing counts/offsets/idx for every user of the shard; for N > 1 followed by the all-gather of per-user
counts and row lists (RCCL).  Workload at N = 1: BASELINE.json config 3 — 10^8 sessions / 10^5 users /
32 disciplines, SoA int64 start/end + int32 user/disc, resident in HBM before the timed region.
Scaling is weak: every rank holds a 10^8-row shard of its own users (user-hash sharded table of N x 10^8).

One JSON line on stdout (rank 0).  Everything else goes to stderr.
"""
import argparse
import json
import os
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


def cpu_baseline(args, U, D, now, cutoff, mask, flags):
    """Oracle (CPU port, 1 thread) timed on a bounded sample of the same workload, on this box's host cores."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py
    sample = args.cpu_sample_rows
    s, e, u, d = oracle_py.gen(SEED, args.rows, 0, sample, U, D, flags)
    t0 = time.perf_counter()
    reps = 0
    while True:
        oracle_py.scan(s, e, u, d, U, now, cutoff, mask)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or reps >= 1000:
            break
    rows_per_s = sample * reps / dt
    out = {
        "value": rows_per_s * U / args.rows, "unit": "feeds/s", "cores": 1, "kind": "port",
        "sessions_per_sec": rows_per_s,
        "sample": "%d reps x first %d rows of the same corpus (U=%d, D=%d), oracle/pie_oracle.c scan, %.1f s; "
                  "feeds/s scaled by U/N of the full workload; host has %d cores" % (reps, sample, U, D, dt, os.cpu_count()),
    }
    # reference-faithful JS (Map of {userId, createdAt, expiresAt} objects, single thread) when node exists
    try:
        import shutil
        import subprocess
        node = shutil.which("node")
        js = os.path.join(REPO, "oracle", "ref_faithful.js")
        if node and os.path.exists(js):
            res = subprocess.run([node, "--max-old-space-size=8192", js, "--bench", str(args.js_rows), str(max(U // 100, 10)), str(D)],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
            if res.returncode == 0:
                out["js_reference_faithful"] = json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as ex:  # the JS leg is optional context, never fatal
        log("js baseline skipped:", ex)
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
    ap.add_argument("--rows", type=int, default=10 ** 8, help="sessions per GPU (weak scaling)")
    ap.add_argument("--users", type=int, default=10 ** 5, help="users per GPU shard")
    ap.add_argument("--disc", type=int, default=32)
    ap.add_argument("--order", choices=["random", "clustered"], default="random")
    ap.add_argument("--users-dist", choices=["uniform", "zipf"], default="uniform", help="zipf: Zipf(1.1) over the shard's users")
    ap.add_argument("--variant", choices=["auth", "interval"], default="auth")
    ap.add_argument("--query", choices=["spec", "wide", "future"], default="spec",
                    help="spec: now=T0-6h, cutoff=T0-61d, 16/32 disciplines (SURVEY.md §8d); wide: ~25%% selected")
    ap.add_argument("--cpu-sample-rows", type=int, default=2 * 10 ** 7)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--js-rows", type=int, default=10 ** 6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-batch", type=int, default=8, help="multi-GPU: scans per all-gather (1 = one gather per scan)")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="every n-th timed step carries HIP events around the scan kernels (each pair drains the stream for a "
                         "few microseconds); 0 = min(16, steps // 3), i.e. at least three samples")
    ap.add_argument("--mode", choices=["scan", "expired"], default="scan",
                    help="scan: the headline feed scan; expired: the 'next' row of SURVEY.md 8f-1 — newly-expired change "
                         "predicate -> ordered dispatch queue (reads only the end column: 8 B/row algorithmic)")
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

    import torch
    import torch.distributed as dist
    import sph_pie_amd as pie
    from sph_pie_amd.shard import HipShardBackend, ShardedFeeds

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PIE_BENCH_FORCE_GATHER=1 exercises the RCCL exchange step with a single rank (rehearsal on a 1-GPU box)
    gather = world > 1 or os.environ.get("PIE_BENCH_FORCE_GATHER") == "1"
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    N, U, D = args.rows, args.users, args.disc
    flags = (pie.PIE_GEN_INTERVAL if args.variant == "interval" else 0) | (pie.PIE_GEN_CLUSTERED if args.order == "clustered" else 0)
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
    # every rank owns the users that hash to it; its shard is rows [rank*N, (rank+1)*N) of the world*N-row corpus
    if args.users_dist == "zipf":
        ctx.gen_synthetic_cdf(SEED, N * world, N * rank, N, U, D, flags, pie.zipf_cdf(U))
    else:
        ctx.gen_synthetic(SEED, N * world, N * rank, N, U, D, flags)
    ctx.set_disciplines(mask, D)
    log("rank %d: generated %d rows in %.2f s" % (rank, N, time.perf_counter() - t_gen))

    backend = HipShardBackend(ctx, dev) if gather else None
    # --gather-batch scans per all-gather: the collective is latency-bound at this size (1.7 MB per rank), so the same
    # lists travel in fewer, larger messages
    feeds = ShardedFeeds(backend, rank, world, U, always_collective=gather, batch=args.gather_batch) if gather else None

    expired_window = (T0_MS - 30 * DAY, T0_MS - 29 * DAY)   # one day of expiries: ~0.83 % of the rows queue up

    def run_steps(k):
        """k steps; with the exchange step the all-gather of step i overlaps the scan of step i+1 (depth-1 pipeline,
        every gather is collected inside the same call)."""
        last = None
        if args.mode == "expired":
            for _ in range(k):
                last = ctx.expired_queue(expired_window[0], expired_window[1], fetch=False)
            return last
        if not gather:
            if args.depth == 1:
                for _ in range(k):
                    last = ctx.scan_device(now, cutoff)
                return last
            return ctx.scan_pipelined(k, now, cutoff)
        last = feeds.run_steps(k, now, cutoff)
        if last is None:  # a row list outgrew the message: capacity was raised, redo synchronously once
            last = feeds.scan_and_gather(now, cutoff)
        return last

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(max(args.warmup, 1))
    fence()
    ctx.stats_reset()
    # HIP events around the scan kernels, on the stream they are launched on; every 16th step (at least three per run) carries them (an event
    # between two kernels drains the pipeline for a few microseconds, which would inflate ms_per_step)
    profile_every = args.profile_every if args.profile_every > 0 else max(1, min(16, args.steps // 3))
    ctx.set_profiling(1 if args.mode == "expired" else profile_every)
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    ctx.set_profiling(0)
    if gather:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = ctx.stats()
    m = last if not gather else int(last["lengths"][rank])
    scan_only_ms = None
    gather_ok = None
    if gather and args.mode == "scan":
        # the gathered lists of this rank (as every rank received them) against this rank's own result of the same query
        import numpy as np
        ctx.scan_device(now, cutoff)
        _, own_off, own_idx = ctx.read_results()
        ok = int(last["lengths"][rank]) == own_idx.size and \
            np.array_equal(last["offsets"][rank].cpu().numpy()[: U + 1], own_off.astype(np.int32)) and \
            np.array_equal(last["rows"][rank].cpu().numpy()[: own_idx.size], own_idx)
        t_ok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        gather_ok = bool(int(t_ok.item()))
        if not gather_ok:
            log("WARNING: rank %d: gathered lists differ from the local result" % rank)
    if gather and args.mode == "scan":
        # scan-only throughput beside scan + gather — the same K steps without the exchange, after the
        # timed region (not part of value); max over ranks like the headline
        fence()
        t1 = time.perf_counter()
        ctx.scan_pipelined(args.steps, now, cutoff)
        fence()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        scan_only_ms = float(t.item()) * 1e3 / args.steps

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        k1_ms = st["k1_ms_sum"] / max(st["n_profiled"], 1)
        scan_ms = st["scan_ms_sum"] / max(st["n_profiled"], 1)
        alg = (8.0 if args.mode == "expired" else 24.0) * N
        achieved = alg / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        variant = st["k1_variant"]
        kname = "k_scan_live_first" if variant & 4 else "k_scan_compact"
        if variant & 0x200:
            kname = "k_scan_live_first_part"
        if variant & 0x400:
            # with two scans in flight the launch also carries the offsets + order kernel of the scan before
            rides = (args.depth == 2 or gather) and os.environ.get("PIE_K2_RIDE") != "0" and (variant & ~0x840) == 0x485
            kname = "k_scan_keyed_with_tail" if rides else "k_scan_keyed"
        if args.mode == "expired":
            kname = "k_expired_stage" if os.environ.get("PIE_EXPIRED_ON_END") else "k_expired_stage_keyed"
        traffic = None
        tpath = os.path.join(REPO, "profiles", "k1_traffic.json")
        default_workload = (N, U, D, args.order, args.variant, args.query, args.mode, args.users_dist) == \
            (10 ** 8, 10 ** 5, 32, "random", "auth", "spec", "scan", "uniform")
        if default_workload and os.path.exists(tpath):
            tdoc = json.load(open(tpath))
            if tdoc.get("kernel", "").endswith(kname) or kname in tdoc.get("kernel", ""):
                traffic = tdoc["hbm_bytes_per_launch"]   # PMC-measured for this kernel form and this workload
        line = {
            "metric": "feeds/sec + sessions scanned/sec, 10^8 synthetic sessions, 1/2/4/8 MI355X" if args.mode == "scan" else
                      "expired-queue pass (SURVEY 8f-1): sessions scanned/sec; value counts table rows, not feeds",
            "value": (U if args.mode == "scan" else N) * world / (ms_per_step * 1e-3),
            "unit": "feeds/s" if args.mode == "scan" else "sessions/s",
            "sessions_per_sec": N * world / (ms_per_step * 1e-3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "scan_only_ms_per_step": scan_only_ms, "gather_verified": gather_ok,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {
                "workload": "BASELINE config 3: %d sessions / %d users / %d disciplines per GPU, SoA int64 start/end + int32 "
                            "user/disc, splitmix64 seed 0x5EED5EED, %s order, %s users, %s variant, %s query" % (N, U, D, args.order, args.users_dist, args.variant, args.query),
                "sessions_per_gpu": N, "users_per_gpu": U, "disciplines": D, "selected_rows_rank0": int(m),
                "parallelism": "user-hash shards x%d, RCCL all-gather of per-user offsets + row lists (%d scans per collective), overlapped with the next scans" % (world, args.gather_batch) if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "kernel_variant": hex(variant), "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/k1_traffic.json (rocprofv3 PMC, separate passes, gfx950 FETCH_SIZE x2 correction)" if traffic else None,
                "hbm_gbs_from_traffic": (traffic / (k1_ms * 1e-3) / 1e9) if traffic and k1_ms > 0 else None,
                "hbm_frac_of_peak_from_traffic": (traffic / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and k1_ms > 0 else None,
                "note": "the timed launch is the table pass of one scan plus, in its first blocks, the offsets + order kernel of the "
                        "scan before it; achieved = algorithmic 24 B/row over kernel time, as the metric defines it; the keyed table pass streams a "
                        "1- or 2-byte liveness key per row and gathers one 16-byte payload record per candidate row, so the HBM "
                        "bytes it moves (traffic, PMC-measured) are far below the algorithmic bytes: judge the kernel by "
                        "hbm_frac_of_peak_from_traffic (DESIGN.md sections 3, 4, 6)",
                "alg_bytes_per_launch": alg, "kernel_ms": k1_ms, "launches_timed": st["n_profiled"],
                # whole step (table pass + offsets + scatter + per-bucket order [+ exchange]) against the same peak
                "whole_step_frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # latency of one scan, first kernel start -> last kernel end; with two scans in flight the tail of
                # scan i is queued behind the table pass of scan i+1, so this exceeds ms_per_step by design
                "scan_latency_ms": scan_ms, "scans_in_flight": args.depth if not gather else 1,
                "k1_blocks": st["k1_blocks"],
            },
        }
        if world == 1 and args.mode == "scan" and not gather:
            # D2H of counts/offsets and of idx, reported beside the headline (results normally stay in HBM
            # for the exchange step / the host serialiser's fetch); pinned host buffers, synchronous calls, 20 reps
            h_counts = torch.empty(U, dtype=torch.int32).pin_memory()
            h_offsets = torch.empty(U + 1, dtype=torch.int64).pin_memory()
            h_idx = torch.empty(max(int(m), 1), dtype=torch.int32).pin_memory()
            reps = 20
            ctx.scan_device(now, cutoff)
            ctx.read_results_into(h_counts.data_ptr(), h_offsets.data_ptr(), h_idx.data_ptr(), h_idx.numel())  # first use of the buffers
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.scan_device(now, cutoff)
            t2 = time.perf_counter()
            for _ in range(reps):
                ctx.scan_device(now, cutoff)
                ctx.read_results_into(h_counts.data_ptr(), h_offsets.data_ptr())
            t3 = time.perf_counter()
            for _ in range(reps):
                ctx.read_results_into(None, None, h_idx.data_ptr(), h_idx.numel())
            t4 = time.perf_counter()
            line["d2h"] = {
                "one_scan_synchronous_ms": (t2 - t1) * 1e3 / reps,
                "one_scan_plus_counts_offsets_to_host_ms": (t3 - t2) * 1e3 / reps,
                "idx_to_host_ms": (t4 - t3) * 1e3 / reps, "idx_bytes": int(m) * 4, "counts_offsets_bytes": U * 4 + (U + 1) * 8,
                "note": "one scan at a time (no run-ahead), host-synchronous, pinned buffers; not part of value",
            }
            nonempty = int((h_counts > 0).sum())
            line["config"]["nonempty_feeds_rank0"] = nonempty
            line["nonempty_feeds_per_sec"] = nonempty / (ms_per_step * 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, U, D, now, cutoff, mask, flags)
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if gather:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
