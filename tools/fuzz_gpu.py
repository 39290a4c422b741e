#!/usr/bin/env python3
"""Differential fuzz on the GPU box: random tables, mutations, queries and scan forms against the CPU oracle, for a time
budget.  usage: python tools/fuzz_gpu.py [seconds] [seed]   (prints the failing case's seed and stops at the first mismatch)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import oracle_py as oracle  # noqa: E402  (the checker)
import sph_pie_amd as pie  # noqa: E402
import torch  # noqa: E402

INT64_MIN = -(2 ** 63)
T0, DAY = oracle.T0_MS, 86400 * 1000
FORMS = [None, None, None, 0x03, 0x01, 0x85, 0xC5, 0x485, 0x4C5, 0xC85, 0xCC5, 0xC95, 0x425, 0xC05]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
t_end = time.time() + budget
cases = scans = 0
dev = torch.device("cuda", 0)


def same(got, want, what):
    for name, a, b in zip(("counts", "offsets", "idx"), got, want):
        if a.dtype != b.dtype or not np.array_equal(a, b):
            raise AssertionError("%s: %s differs" % (what, name))


case_seed = seed0
while time.time() < t_end:
    case_seed += 1
    rng = np.random.default_rng(case_seed)
    form = FORMS[int(rng.integers(len(FORMS)))]
    if form is None:
        os.environ.pop("PIE_K1_VARIANT", None)
    else:
        os.environ["PIE_K1_VARIANT"] = hex(form)
    n = int(rng.choice([1, 7, 300, 5000, 70001, 300000, 1 << 20]))
    U = int(rng.choice([1, 3, 50, 1000, 40000, 600000]))
    D = int(rng.choice([1, 7, 32, 64]))
    flags = int(rng.integers(8))   # interval / user-clustered / creation-ordered, in any combination
    ordered = int(rng.choice([0, 1, 1, 2, 2]))   # the ordered run: never / adaptive / always (pinned forms never use it)
    what = "seed %d form %s n %d U %d D %d flags %d ordered %d" % (case_seed, form, n, U, D, flags, ordered)
    try:
        s, e, u, d = [c.copy() for c in oracle.gen(int(rng.integers(1, 2 ** 60)), n, 0, n, U, D, flags)]
        if rng.random() < 0.3:   # heavy head user
            u = np.where(rng.random(n) < 0.5, int(rng.integers(U)), u).astype(np.int32)
        if rng.random() < 0.3:   # many equal ends / starts
            e = (e // (3600 * 1000)) * (3600 * 1000)
            s = (s // (3600 * 1000)) * (3600 * 1000)
        mask = int(rng.integers(0, 2 ** 63)) | (int(rng.integers(0, 2)) << 63)
        if rng.random() < 0.3:
            mask = 2 ** 64 - 1
        m_eff = mask if D >= 64 else mask & ((1 << D) - 1)
        with pie.PieScan(0) as ctx:
            ctx.set_ordered_run(ordered)
            ctx.load_columns(s, e, u, d, U)
            ctx.set_disciplines(mask, D)
            for rnd in range(int(rng.integers(2, 7))):
                # a mutation now and then
                r = rng.random()
                if r < 0.2 and n > 1:
                    rows = rng.choice(n, min(n, 200), replace=False).astype(np.int32)
                    ne = rng.integers(T0 - 200 * DAY, T0 + 50 * DAY, rows.size).astype(np.int64)
                    ne[: rows.size // 8] = INT64_MIN
                    ctx.set_end(rows, ne)
                    e[rows] = ne
                elif r < 0.3:
                    gone = ctx.delete_user(int(rng.integers(U)))
                    e[gone] = INT64_MIN
                elif r < 0.4 and n < 400000:
                    k = int(rng.integers(1, 3000))
                    mode = rng.random()
                    if mode < 0.4:   # a session store's appends: created now, i.e. not before anything the table holds
                        s2 = (int(s.max()) + np.sort(rng.integers(0, 4000, k))).astype(np.int64)
                    elif mode < 0.7:   # ... a little late and not sorted: inserted a few rows before the end of their users' segments
                        s2 = (int(s.max()) - rng.integers(0, 20000, k)).astype(np.int64)
                    else:
                        s2 = rng.integers(T0 - 150 * DAY, T0 + 300 * DAY, k).astype(np.int64)
                    e2 = s2 + rng.integers(-50 * DAY, 50 * DAY, k)
                    u2, d2 = rng.integers(0, U, k).astype(np.int32), rng.integers(0, D, k).astype(np.int32)
                    ctx.append_rows(s2, e2, u2, d2, U)
                    s, e, u, d = np.concatenate([s, s2]), np.concatenate([e, e2]), np.concatenate([u, u2]), np.concatenate([d, d2])
                    n = s.size
                # a query: recent, mid, extreme, exactly on a value
                q = rng.random()
                now = int(T0 - rng.integers(0, 20 * 3600 * 1000)) if q < 0.5 else int(T0 - rng.integers(0, 130 * DAY)) if q < 0.8 else \
                    int(rng.choice([INT64_MIN, 2 ** 62, int(e[int(rng.integers(n))]), int(e[int(rng.integers(n))]) - 1]))
                cutoff = int(rng.choice([INT64_MIN, T0 - 61 * DAY, int(s[int(rng.integers(n))])]))
                want = oracle.scan(s, e, u, d, U, now, cutoff, m_eff)
                reps = int(rng.integers(1, 4))   # repeats let the adaptive forms engage
                for _ in range(reps):
                    same(ctx.scan(now, cutoff), want, what + " scan now %d cutoff %d" % (now, cutoff))
                    scans += 1
                if rng.random() < 0.5:   # a chain of scans with two in flight (K2 of one rides in the launch of the next)
                    chain = []
                    for _ in range(int(rng.integers(2, 6))):
                        n2 = now if rng.random() < 0.6 else int(T0 - rng.integers(0, 30 * 3600 * 1000))
                        chain.append((n2, oracle.scan(s, e, u, d, U, n2, cutoff, m_eff)))
                    ctx.scan_begin(chain[0][0], cutoff)
                    for k in range(len(chain)):
                        if k + 1 < len(chain):
                            ctx.scan_begin(chain[k + 1][0], cutoff)
                        if ctx.scan_finish() != chain[k][1][2].size:
                            raise AssertionError(what + " chained scan %d: M differs" % k)
                        same(ctx.read_results(), chain[k][1], what + " chained scan %d now %d" % (k, chain[k][0]))
                        scans += 1
                if rng.random() < 0.4:   # the scan-written exchange message
                    m = want[2].size
                    u_pad, cap = U + int(rng.integers(0, 5)), int(rng.choice([m, m + 3, max(m // 2, 1)]))
                    msg = torch.full((u_pad + 2 + cap,), -7, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()   # the fill ran on torch's stream, the scan writes from the library's
                    ctx.scan_begin_packed(now, cutoff, msg.data_ptr(), u_pad, cap)
                    got_m, _ = ctx.scan_finish_packed()
                    ctx.synchronize()
                    a = msg.cpu().numpy()
                    k = min(m, cap)
                    ok = got_m == m and np.array_equal(a[: U + 1], want[1].astype(np.int32)) and np.all(a[U + 1: u_pad + 2] == m) and \
                        np.array_equal(a[u_pad + 2: u_pad + 2 + k], want[2][:k]) and np.all(a[u_pad + 2 + k:] == -7)
                    if not ok:
                        bad = np.nonzero(a[: U + 1] != want[1].astype(np.int32))[0][:5]
                        raise AssertionError(what + " message now %d cutoff %d u_pad %d cap %d: got_m %d m %d, first bad offsets %s, pad %s, rows %s, variant %s"
                                             % (now, cutoff, u_pad, cap, got_m, m, bad.tolist(), a[U + 1: u_pad + 2].tolist(), a[u_pad + 2:].tolist()[:8], hex(ctx.stats()["k1_variant"])))
                    scans += 1
                if rng.random() < 0.5:   # batched scans: random queries with their own now / cutoff / mask, 1 or 2 batches in flight
                    def rand_query():
                        qq = rng.random()
                        nw = int(T0 - rng.integers(0, 20 * 3600 * 1000)) if qq < 0.6 else int(T0 - rng.integers(0, 130 * DAY)) if qq < 0.85 else \
                            int(rng.choice([INT64_MIN, 2 ** 62, int(e[int(rng.integers(n))]), int(e[int(rng.integers(n))]) - 1]))
                        ct = int(rng.choice([INT64_MIN, T0 - 61 * DAY, int(s[int(rng.integers(n))])]))
                        mk = int(rng.integers(0, 2 ** 63)) | (int(rng.integers(0, 2)) << 63) if rng.random() < 0.7 else 2 ** 64 - 1
                        return nw, ct, mk
                    lim = 2 ** 64 - 1 if D >= 64 else (1 << D) - 1
                    def batch_size():   # 1 .. 64 queries; more than 16 and more than 32 are shapes of their own
                        return int(rng.choice([int(rng.integers(1, 17)), int(rng.integers(17, 33)), int(rng.integers(33, 65))], p=[0.6, 0.2, 0.2]))
                    batches = [[rand_query() for _ in range(batch_size())] for _ in range(int(rng.integers(1, 10)))]
                    if rng.random() < 0.3:   # the requests of a few seconds: clocks a second apart, two cutoffs, three masks
                        base_now = int(T0 - rng.integers(0, 20 * 3600 * 1000))
                        mk3 = [int(rng.integers(0, 2 ** 63)), 2 ** 64 - 1, int(rng.integers(0, 2 ** 63))]
                        batches[0] = [(base_now - 977 * q, T0 - (61 + q % 2) * DAY, mk3[q % 3]) for q in range(len(batches[0]))]
                    # now and then the batch writes its union exchange message itself (pie_scan_batch_begin_union)
                    direct = [rng.random() < 0.3 for _ in batches]
                    dmsg = {}

                    def begin(k):
                        if direct[k]:
                            up, cp = U + int(rng.integers(0, 3)), int(rng.choice([1 << 16, 40, 3]))
                            t = torch.full((up + 2 + 3 * cp,), -7, dtype=torch.int32, device=dev)
                            torch.cuda.synchronize()
                            dmsg[k] = (t, up, cp)
                            ctx.scan_batch_begin_union(batches[k], t.data_ptr(), up, cp)
                        else:
                            ctx.scan_batch_begin(batches[k])

                    ctx.set_batch_lanes(int(rng.integers(0, 5)))   # 0: by table size; batches of different lanes run side by side
                    if os.environ.get("PIE_FUZZ_LANES"):   # (debugging a case: same random stream, lanes pinned)
                        ctx.set_batch_lanes(int(os.environ["PIE_FUZZ_LANES"]))
                    depth = int(rng.integers(1, 3 * ctx.batch_lanes() + 1))   # up to three batches in flight per lane
                    begun_b = 0
                    for k in range(len(batches)):
                        while begun_b < len(batches) and begun_b - k < depth and ctx.batch_room() > 0:
                            begin(begun_b)
                            begun_b += 1
                            fl_mid, fl_end = rng.random() < 0.15, begun_b == len(batches)
                            if os.environ.get("PIE_FUZZ_DEBUG"):
                                print("begun", begun_b - 1, "flush mid", fl_mid, "end", fl_end, flush=True)
                            if os.environ.get("PIE_FUZZ_FLUSH") == "end":
                                fl_mid = False
                            if os.environ.get("PIE_FUZZ_FLUSH") == "mid":
                                fl_end = False
                            if (fl_mid or fl_end) and not os.environ.get("PIE_FUZZ_NO_FLUSH"):   # a burst ends (or seems to): waiting tails go out together
                                ctx.scan_batch_flush()
                        ms, ready = ctx.scan_batch_finish(packed=True)
                        wants_k = [oracle.scan(s, e, u, d, U, nw, ct, mk & lim) for nw, ct, mk in batches[k]]
                        nqk = len(batches[k])
                        if os.environ.get("PIE_FUZZ_DEBUG"):
                            print("batch", k, "of", len(batches), "nq", nqk, "depth", depth, "lanes", ctx.batch_lanes(), "direct", direct[k], "ready", ready,
                                  "got", list(ms), "want", [int(w[2].size) for w in wants_k], "stats", ctx.stats(), flush=True)
                        if os.environ.get("PIE_FUZZ_DEBUG"):
                            und = ctx.batch_read_union()
                            bad = [qi for qi, w in enumerate(wants_k) if ms[qi] != w[2].size]
                            print("  union", None if und is None else und[1].size, "bad queries", bad, flush=True)
                            for qi in bad:
                                got_q = ctx.batch_read_results(qi)
                                print("  query", qi, batches[k][qi], "lists: M", got_q[2].size, "want", wants_k[qi][2].size,
                                      "union count", None if und is None else int((((und[2] >> np.uint64(qi)) & np.uint64(1)) == 1).sum()), flush=True)
                        for qi, w in enumerate(wants_k):
                            if ms[qi] != w[2].size:
                                raise AssertionError(what + " batch %d query %d: M differs" % (k, qi))
                        un = ctx.batch_read_union()
                        if un is not None:   # the primary result: every query's counts / offsets / rows are a filter of it, in order
                            uoff_d, rows_d, masks_d = un
                            for qi, w in enumerate(wants_k):
                                sel = ((masks_d >> np.uint64(qi)) & np.uint64(1)) == 1
                                csum = np.concatenate([[0], np.cumsum(sel)])
                                if not (np.array_equal(rows_d[sel], w[2]) and np.array_equal(csum[uoff_d], w[1])):
                                    raise AssertionError(what + " batch %d query %d: union result differs" % (k, qi))
                            if np.any(masks_d == 0) or (nqk < 64 and np.any(masks_d >> np.uint64(nqk))):
                                raise AssertionError(what + " batch %d: a union row without a query / with a query bit beyond the batch" % k)
                        if k in dmsg:   # the message the batch wrote (or, without a union, the one packed from the lists)
                            t, up, cp = dmsg.pop(k)
                            ctx.synchronize()
                            a = t.cpu().numpy()
                            mu = int(a[up + 1])
                            if un is not None and not ready and not (ctx.stats()["k1_variant"] & 0x2000):   # (on the ordered run it is copied from the union)
                                raise AssertionError(what + " batch %d: has a union but its message was not written by the batch" % k)
                            if mu >= 0:
                                kk = min(mu, cp)
                                lo_w = a[up + 2 + cp: up + 2 + cp + kk].astype(np.uint32).astype(np.uint64)
                                hi_w = a[up + 2 + 2 * cp: up + 2 + 2 * cp + kk].astype(np.uint32).astype(np.uint64) if nqk > 32 else np.zeros(kk, np.uint64)
                                m64 = lo_w | (hi_w << np.uint64(32))
                                uo, rw = a[: U + 1].astype(np.int64), a[up + 2: up + 2 + kk]
                                if not np.all(a[U + 1: up + 1] == mu):
                                    raise AssertionError(what + " batch %d: message padding offsets" % k)
                                if un is not None and not (mu == un[1].size and np.array_equal(uo, un[0]) and np.array_equal(rw, un[1][:kk]) and np.array_equal(m64, un[2][:kk])):
                                    raise AssertionError(what + " batch %d: union message differs from the union result (cap %d)" % (k, cp))
                                if mu <= cp:
                                    for qi, w in enumerate(wants_k):
                                        sel = ((m64 >> np.uint64(qi)) & np.uint64(1)) == 1
                                        if not np.array_equal(rw[sel], w[2]):
                                            raise AssertionError(what + " batch %d query %d: union message rows differ" % (k, qi))
                            elif un is not None:
                                raise AssertionError(what + " batch %d: message says -1 although the batch has a union" % k)
                        if rng.random() < 0.5:   # a handful of requests fetched together
                            nreq = int(rng.integers(1, 40))
                            rq, ru = rng.integers(0, nqk, nreq).astype(np.int32), rng.integers(-1, U + 1, nreq).astype(np.int32)
                            off_r, idx_r, st_r, en_r, di_r = ctx.batch_fetch_requests(rq, ru, cap_rows=nreq * max(int(w[0].max()) if w[0].size else 0 for w in wants_k) + 8)
                            for i in range(nreq):
                                got_rows = idx_r[off_r[i]:off_r[i + 1]]
                                uu = int(ru[i])
                                exp = wants_k[int(rq[i])]
                                exp_rows = exp[2][exp[1][uu]:exp[1][uu + 1]] if 0 <= uu < U else np.zeros(0, np.int32)
                                if not np.array_equal(got_rows, exp_rows):
                                    raise AssertionError(what + " batch %d request %d (query %d user %d): fetched rows differ" % (k, i, int(rq[i]), uu))
                            if not (np.array_equal(st_r, s[idx_r]) and np.array_equal(en_r, e[idx_r]) and np.array_equal(di_r, d[idx_r])):
                                raise AssertionError(what + " batch %d: fetched columns differ" % k)
                        order_q = rng.permutation(nqk)[: int(rng.integers(1, min(nqk, 6) + 1))] if nqk > 16 else range(nqk)
                        for qi in order_q:   # per-query results, materialised on request (all of them for small batches)
                            qi = int(qi)
                            nw, ct, mk = batches[k][qi]
                            same(ctx.batch_read_results(qi), wants_k[qi], what + " batch %d query %d now %d cutoff %d mask %x" % (k, qi, nw, ct, mk))
                            scans += 1
                        uq = int(rng.integers(nqk))
                        uu = int(rng.integers(U))
                        if not np.array_equal(ctx.batch_read_user_feed(uq, uu), wants_k[uq][2][wants_k[uq][1][uu]:wants_k[uq][1][uu + 1]]):
                            raise AssertionError(what + " batch %d: feed of user %d, query %d" % (k, uu, uq))
                        if rng.random() < 0.4 and nqk <= 32:   # the union exchange message packed after the fact: every query's list is a filter of it
                            wants = wants_k
                            per_user_max = max(int(np.max(sum((w[0] for w in wants), np.zeros(U, np.int64)))), 0)
                            u_pad, cap = U + int(rng.integers(0, 3)), int(sum(w[2].size for w in wants)) + 5
                            msg = torch.full((u_pad + 2 + 2 * cap,), -7, dtype=torch.int32, device=dev)
                            torch.cuda.synchronize()
                            ctx.batch_pack_union_device(msg.data_ptr(), u_pad, cap)
                            ctx.synchronize()
                            a = msg.cpu().numpy()
                            mu = int(a[u_pad + 1])
                            if mu < 0:
                                if per_user_max <= 32:   # the sum over the queries bounds a user's union from above
                                    raise AssertionError(what + " batch %d: union message declined although no user has more than 32 rows" % k)
                            else:
                                uoff, rows_u, masks_u = a[: U + 1].astype(np.int64), a[u_pad + 2: u_pad + 2 + mu], a[u_pad + 2 + cap: u_pad + 2 + cap + mu]
                                for qi, w in enumerate(wants):
                                    sel = ((masks_u >> qi) & 1) == 1
                                    csum = np.concatenate([[0], np.cumsum(sel)])
                                    if not (np.array_equal(rows_u[sel], w[2]) and np.array_equal(csum[uoff], w[1])):
                                        raise AssertionError(what + " batch %d query %d: union message differs" % (k, qi))
                    ctx.set_disciplines(mask, D)
                if rng.random() < 0.3:
                    prev = int(now - rng.integers(0, 3 * DAY)) if now > INT64_MIN + 4 * DAY else INT64_MIN
                    if not np.array_equal(ctx.expired_queue(prev, now), oracle.expired_queue(e, prev, now)):
                        raise AssertionError(what + " expired (%d, %d]" % (prev, now))
                if rng.random() < 0.2 and n <= 300000:   # the archive chain (group min -> threshold -> whole groups in first-appearance order)
                    a_now = int(rng.choice([T0, T0 - 100 * DAY, T0 - 119 * DAY, int(s[int(rng.integers(n))]), 2 ** 62, INT64_MIN + 5]))
                    a_win = int(rng.choice([0, 3600 * 1000, 43200000, 30 * DAY, 2 ** 62]))
                    if not np.array_equal(ctx.archive_queue(a_now, a_win), oracle.archive_queue_numpy(s, e, u, U, a_now, a_win)):
                        raise AssertionError(what + " archive queue now %d window %d" % (a_now, a_win))
        cases += 1
    except Exception as exc:  # noqa: BLE001
        print("FAIL %s: %r" % (what, exc), flush=True)
        sys.exit(1)
    if cases % 20 == 0:
        print("ok: %d cases, %d scans, %.0f s left" % (cases, scans, t_end - time.time()), flush=True)
print("fuzz ok: %d cases, %d scans in %.0f s (first seed %d)" % (cases, scans, budget, seed0 + 1))
