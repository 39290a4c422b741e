"""Replay of tests/golden/sessionstore_g5_trace.json — a recorded sequence of calls into the real reference
`server/sessionStore.js` and its answers (oracle/gen_trace_golden.js) — against a column table.

The table side is a small backend (CPU: numpy columns + the oracle's predicates; GPU: the HIP library through the C
ABI); the bookkeeping around it mirrors what sph-pie_amd/host/sessionStore.js does with the same calls:
    create   -> append one row (start = now, end = now + TTL)
    get      -> live iff end > now (sessionStore.js:30); a dead session is dropped as a side effect (:31-32)
    touch    -> live ? end = now + TTL, start kept (:37-45) : null
    del      -> tombstone (:47-53);  delUser -> tombstone every row of that user, falsy / unknown id = no-op (:55-64)
    purge    -> tombstone every row with end <= now, one `now` for the whole pass (:66-73)
    census   -> get on every token ever issued, in issue order
At every census the batched scan (no window, every discipline) must also give, per user, exactly the live rows in
row order."""
import json
import os

import numpy as np

INT64_MIN = -(2 ** 63)


def load_trace(golden_dir):
    return json.load(open(os.path.join(golden_dir, "sessionstore_g5_trace.json")))


def replay(trace, backend):
    """backend: append(start, end, user) / end_of(row) / start_of(row) / set_end(row, value) / delete_user(user_index) /
    expired(now) -> rows with end <= now that are not tombstoned / feeds(now) -> (counts, offsets, idx).
    Returns the number of answers compared."""
    ttl = trace["ttl_ms"]
    users = {name: i for i, name in enumerate(trace["users"])}
    user_of_row = []
    checked = 0

    def get(row, now):
        e = backend.end_of(row)
        if e == INT64_MIN:
            return None
        if e <= now:                      # dead: reported as missing and dropped
            backend.set_end(row, INT64_MIN)
            return None
        return {"userId": trace["users"][user_of_row[row]], "createdAt": backend.start_of(row), "expiresAt": e}

    for op in trace["ops"]:
        now, kind = op["now"], op["op"]
        if kind == "create":
            backend.append(now, now + ttl, users[op["user"]])
            user_of_row.append(users[op["user"]])
            assert now + ttl == op["expiresAt"]
        elif kind == "get":
            assert get(op["tok"], now) == op["result"], op
        elif kind == "touch":
            cur = get(op["tok"], now)
            if cur is None:
                assert op["result"] is None, op
            else:
                backend.set_end(op["tok"], now + ttl)
                assert {"userId": cur["userId"], "expiresAt": now + ttl} == op["result"], op
        elif kind == "del":
            backend.set_end(op["tok"], INT64_MIN)
        elif kind == "delUser":
            if op["user"] in users:       # '' and unknown ids match nothing
                backend.delete_user(users[op["user"]])
        elif kind == "purge":
            for row in backend.expired(now):
                backend.set_end(int(row), INT64_MIN)
        elif kind == "census":
            live = [row for row in range(len(user_of_row)) if get(row, now) is not None]
            assert live == op["live"], (now, live, op["live"])
            counts, offsets, idx = backend.feeds(now)
            want = [[r for r in live if user_of_row[r] == u] for u in range(len(users))]
            assert counts.tolist() == [len(w) for w in want]
            for u, w in enumerate(want):
                assert idx[offsets[u]:offsets[u + 1]].tolist() == w, (now, u)
        else:
            raise AssertionError("unknown op " + kind)
        checked += 1
    return checked


class OracleTable:
    """CPU backend: numpy columns, every predicate through the oracle library."""

    def __init__(self, oracle, n_users):
        self.o, self.U = oracle, n_users
        self.s, self.e, self.u = [], [], []

    def _cols(self):
        n = len(self.s)
        return np.array(self.s, np.int64), np.array(self.e, np.int64), np.array(self.u, np.int32), np.zeros(n, np.int32)

    def append(self, start, end, user):
        self.s.append(start); self.e.append(end); self.u.append(user)

    def end_of(self, row):
        return self.e[row]

    def start_of(self, row):
        return self.s[row]

    def set_end(self, row, value):
        self.e[row] = value

    def delete_user(self, user):
        for r in range(len(self.u)):
            if self.u[r] == user:
                self.e[r] = INT64_MIN

    def expired(self, now):
        q = self.o.expired_queue(np.array(self.e, np.int64), INT64_MIN, now)
        return [int(r) for r in q]

    def feeds(self, now):
        s, e, u, d = self._cols()
        if s.size == 0:
            return np.zeros(self.U, np.int32), np.zeros(self.U + 1, np.int64), np.zeros(0, np.int32)
        return self.o.scan(s, e, u, d, self.U, now, INT64_MIN, 1)


class DeviceTable:
    """GPU backend: the HIP library through the ctypes binding of the C ABI."""

    def __init__(self, ctx, n_users):
        self.ctx, self.U, self.n = ctx, n_users, 0
        z64, z32 = np.zeros(0, np.int64), np.zeros(0, np.int32)
        ctx.load_columns(z64, z64, z32, z32, n_users)
        ctx.set_disciplines(1, 1)

    def append(self, start, end, user):
        self.ctx.append_rows(np.array([start], np.int64), np.array([end], np.int64), np.array([user], np.int32),
                             np.zeros(1, np.int32), self.U)
        self.n += 1

    def _row(self, row):
        return self.ctx.fetch_rows(np.array([row], np.int32))

    def end_of(self, row):
        return int(self._row(row)[1][0])

    def start_of(self, row):
        return int(self._row(row)[0][0])

    def set_end(self, row, value):
        self.ctx.set_end(np.array([row], np.int32), np.array([value], np.int64))

    def delete_user(self, user):
        self.ctx.delete_user(user)

    def expired(self, now):
        return [int(r) for r in self.ctx.expired_queue(INT64_MIN, now)]

    def feeds(self, now):
        if self.n == 0:
            return np.zeros(self.U, np.int32), np.zeros(self.U + 1, np.int64), np.zeros(0, np.int32)
        return self.ctx.scan(now, INT64_MIN)
