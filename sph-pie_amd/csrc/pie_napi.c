/*
 * pie_napi.c — raw N-API (no node-addon-api) shim over the C ABI of include/pie_scan.h.
 *
 * The addon holds no logic: it dlopen()s libpie_hip.so and forwards typed-array pointers (zero copy:
 * BigInt64Array for the int64 columns, Int32Array for ids) to the pie_* entry points.  A non-zero status
 * becomes a thrown JS Error carrying `.code` (the PIE_E_* value) and the library's error text — the same
 * shape the route's 500 handler reads (/root/reference/server/index.js:526-536).  All calls are made from
 * the JS main thread; scanAsync() runs the scan on the libuv pool (napi_create_async_work) so a long scan
 * does not block the event loop, one in-flight scan per context: the context handle is marked busy until the worker
 * is done and every other entry point refuses a busy (or destroyed) handle with code PIE_E_STATE.  Output arrays are
 * checked against the lengths the ABI will write (users, users + 1, rows) before any pointer is handed over.
 *
 * Build: gcc -shared -fPIC -I/usr/include/node -Iinclude pie_napi.c -ldl -o host/pie_napi.node
 */
#define NAPI_VERSION 6
#include <node_api.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pie_scan.h"

/* ---- the C ABI, resolved at open() ------------------------------------------------------------------ */
#define PIE_SYMBOLS(X)                                                                                              \
    X(int, pie_abi_version, (void))                                                                                 \
    X(int, pie_device_count, (void))                                                                                \
    X(int, pie_ctx_create, (int, pie_ctx **))                                                                       \
    X(int, pie_ctx_destroy, (pie_ctx *))                                                                            \
    X(const char *, pie_last_error, (const pie_ctx *))                                                              \
    X(int, pie_load_columns, (pie_ctx *, const int64_t *, const int64_t *, const int32_t *, const int32_t *, size_t, int32_t)) \
    X(int, pie_append_rows, (pie_ctx *, const int64_t *, const int64_t *, const int32_t *, const int32_t *, size_t, int32_t)) \
    X(int, pie_gen_synthetic, (pie_ctx *, uint64_t, int64_t, int64_t, int64_t, int32_t, int32_t, uint32_t))         \
    X(int, pie_read_columns, (pie_ctx *, int64_t *, int64_t *, int32_t *, int32_t *, size_t))                       \
    X(int, pie_set_end, (pie_ctx *, const int32_t *, const int64_t *, size_t))                                      \
    X(int, pie_delete_user, (pie_ctx *, int32_t, int32_t *, size_t, size_t *))                                      \
    X(int, pie_batch_fetch_requests, (pie_ctx *, const int32_t *, const int32_t *, size_t, size_t, int64_t *, int32_t *, int64_t *, int64_t *, int32_t *, size_t *)) \
    X(int, pie_retention_purge_tz, (pie_ctx *, int64_t, int32_t, const int64_t *, const int64_t *, int32_t, int32_t *, size_t, size_t *)) \
    X(int, pie_set_disciplines, (pie_ctx *, uint64_t, int32_t))                                                     \
    X(int, pie_scan, (pie_ctx *, int64_t, int64_t, int32_t *, int64_t *, int32_t *, size_t, size_t *))              \
    X(int, pie_fetch_rows, (pie_ctx *, const int32_t *, size_t, int64_t *, int64_t *, int32_t *, int32_t *))        \
    X(int, pie_save_columns, (pie_ctx *, const char *))                                                             \
    X(int, pie_load_columns_dir, (pie_ctx *, const char *))                                                         \
    X(int, pie_scan_device, (pie_ctx *, int64_t, int64_t, size_t *))                                                \
    X(int, pie_read_user_feed, (pie_ctx *, int32_t, int32_t *, size_t, size_t *))                                   \
    X(int, pie_expired_queue, (pie_ctx *, int64_t, int64_t, int32_t *, size_t, size_t *))                           \
    X(int, pie_archive_queue, (pie_ctx *, int64_t, int64_t, int32_t *, size_t, size_t *))                           \
    X(int, pie_set_profiling, (pie_ctx *, int))                                                                     \
    X(int, pie_set_ordered_run, (pie_ctx *, int))                                                                   \
    X(int, pie_set_batch_lanes, (pie_ctx *, int))                                                                   \
    X(int, pie_batch_lanes, (pie_ctx *))                                                                            \
    X(int, pie_stats_get, (pie_ctx *, pie_stats *))                                                                 \
    X(int, pie_stats_reset, (pie_ctx *))                                                                            \
    X(int, pie_scan_batch, (pie_ctx *, const pie_query *, int, size_t *))                                           \
    X(int, pie_batch_read_user_feed, (pie_ctx *, int, int32_t, int32_t *, size_t, size_t *))                        \
    X(int, pie_comm_create, (const int32_t *, int32_t, pie_comm **))                                                \
    X(int, pie_comm_destroy, (pie_comm *))                                                                          \
    X(const char *, pie_comm_last_error, (const pie_comm *))                                                        \
    X(int32_t, pie_comm_world, (const pie_comm *))                                                                  \
    X(pie_ctx *, pie_comm_ctx, (pie_comm *, int32_t))                                                               \
    X(int, pie_comm_gen_synthetic_sharded, (pie_comm *, uint64_t, int64_t, int32_t, int32_t, uint32_t))             \
    X(int, pie_comm_scan_batch_gather, (pie_comm *, const pie_query *, int32_t, int32_t, size_t *))                 \
    X(int, pie_comm_gathered_device_ptr, (pie_comm *, int32_t, void **, size_t *, size_t *, size_t *))              \
    X(int, pie_comm_read_gathered, (pie_comm *, int32_t, int32_t, int32_t, int32_t *, int32_t *, size_t, size_t *))            \
    X(size_t, pie_comm_needed_cap, (const pie_comm *))                                                              \
    X(int, pie_comm_step_reserve, (pie_comm *, int32_t, int32_t, size_t))                                           \
    X(int, pie_comm_step_begin, (pie_comm *, const pie_query *, int32_t))                                           \
    X(int, pie_comm_step_finish, (pie_comm *, size_t *))                                                            \
    X(int, pie_comm_step_collect, (pie_comm *, int64_t *))                                                          \
    X(int, pie_comm_step_gathered_ptr, (pie_comm *, int32_t, int64_t, void **, size_t *, size_t *, size_t *))       \
    X(int, pie_comm_step_read_gathered, (pie_comm *, int32_t, int32_t, int64_t, int32_t *, int32_t *, uint64_t *, size_t, size_t *))

#define X(ret, name, args) static ret(*p_##name) args;
PIE_SYMBOLS(X)
#undef X
static void *g_lib;

#define CHECK(env, call)                                                 \
    do {                                                                 \
        if ((call) != napi_ok) {                                         \
            napi_throw_error((env), NULL, "N-API call failed: " #call); \
            return NULL;                                                 \
        }                                                                \
    } while (0)

static napi_value throw_pie(napi_env env, pie_ctx *ctx, int rc)
{
    napi_value msg, err, code;
    char buf[640];
    snprintf(buf, sizeof buf, "pie_scan error %d: %s", rc, p_pie_last_error ? p_pie_last_error(ctx) : "library not open");
    napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_create_int32(env, rc, &code);
    napi_set_named_property(env, err, "code", code);
    napi_throw(env, err);
    return NULL;
}

static int need_lib(napi_env env)
{
    if (g_lib) return 1;
    napi_throw_error(env, NULL, "libpie_hip.so is not open: call open(path) first (there is no CPU fallback)");
    return 0;
}

/* Number (integer-valued double, exact to 2^53 — what Date.now() returns) or BigInt -> int64 */
static int get_i64(napi_env env, napi_value v, int64_t *out)
{
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_bigint) {
        bool lossless;
        return napi_get_value_bigint_int64(env, v, out, &lossless) == napi_ok;
    }
    if (t == napi_number) {
        double d;
        if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
        if (!(d == d) || d > 9.2e18 || d < -9.2e18) return 0; /* NaN / out of range: Number.isFinite guard */
        *out = (int64_t)d;
        return 1;
    }
    return 0;
}

static void *typed(napi_env env, napi_value v, napi_typedarray_type want, size_t *len)
{
    bool is = false;
    napi_typedarray_type t;
    void *data = NULL;
    size_t n = 0;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return NULL;
    if (napi_get_typedarray_info(env, v, &t, &n, &data, NULL, NULL) != napi_ok || t != want) return NULL;
    if (len) *len = n;
    return data ? data : (void *)(uintptr_t)16; /* zero-length arrays may report NULL */
}

/* What a JS context handle points at.  The header asks for one thread per context: while scanAsync() runs a scan on the
 * libuv pool the box is `busy` and every other entry point refuses it (PIE_E_STATE) instead of racing the worker on the
 * context's state; after ctxDestroy the box stays (the external keeps pointing at it) with ctx = NULL, so a stale handle
 * throws instead of touching freed memory.  `owned` = 0: the context belongs to a communicator (commCtx). */
struct comm_box_s;
typedef struct ctx_box_s {
    pie_ctx *ctx;
    int busy;
    int owned;
    /* a context handed out by commCtx: it lives inside its communicator.  The box holds a reference on the communicator's
     * JS handle (the communicator cannot be collected while a context handle is alive) and sits on the communicator's list,
     * so commDestroy can clear box->ctx: a stale handle then throws PIE_E_STATE like a destroyed context. */
    struct comm_box_s *parent;
    struct ctx_box_s *next;
    napi_ref comm_ref;
} ctx_box;

static void comm_unlink_ctx(ctx_box *b);

static void box_finalize(napi_env env, void *data, void *hint)
{
    (void)hint;
    ctx_box *b = (ctx_box *)data;
    if (b && b->ctx && b->owned && !b->busy && p_pie_ctx_destroy) p_pie_ctx_destroy(b->ctx);
    if (b) {
        comm_unlink_ctx(b);
        if (b->comm_ref) napi_delete_reference(env, b->comm_ref);
    }
    free(b);
}

static napi_value throw_state(napi_env env, const char *text)
{
    napi_value msg, err, code;
    napi_create_string_utf8(env, text, NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_create_int32(env, PIE_E_STATE, &code);
    napi_set_named_property(env, err, "code", code);
    napi_throw(env, err);
    return NULL;
}

static ctx_box *get_box(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a pie context");
        return NULL;
    }
    return (ctx_box *)p;
}

static pie_ctx *get_ctx(napi_env env, napi_value v)
{
    ctx_box *b = get_box(env, v);
    if (!b) return NULL;
    if (!b->ctx) {
        throw_state(env, "pie_scan error -6: this context was destroyed");
        return NULL;
    }
    if (b->busy) {
        throw_state(env, "pie_scan error -6: an asynchronous scan is in flight on this context (one thread per context)");
        return NULL;
    }
    return b->ctx;
}

/* rows / users of the resident table, for the length checks of caller-supplied output arrays */
static int table_shape(pie_ctx *ctx, size_t *rows, size_t *users)
{
    pie_stats st;
    memset(&st, 0, sizeof st);
    st.struct_size = sizeof st;
    int rc = p_pie_stats_get(ctx, &st);
    if (rc) return rc;
    *rows = (size_t)st.rows;
    *users = (size_t)st.users;
    return 0;
}

#define ARGS(n)                                                          \
    size_t argc = (n);                                                   \
    napi_value argv[(n)];                                                \
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));    \
    if (argc < (n)) {                                                    \
        napi_throw_type_error(env, NULL, "too few arguments");           \
        return NULL;                                                     \
    }

static napi_value js_int(napi_env env, int64_t v)
{
    napi_value out;
    napi_create_int64(env, v, &out);
    return out;
}

/* ---- open(path) -------------------------------------------------------------------------------------- */
static napi_value fn_open(napi_env env, napi_callback_info info)
{
    ARGS(1)
    char path[4096];
    size_t n = 0;
    CHECK(env, napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n));
    if (!g_lib) {
        void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
            char buf[4400];
            snprintf(buf, sizeof buf, "cannot load %s: %s (build it first; there is no CPU fallback)", path, dlerror());
            napi_throw_error(env, NULL, buf);
            return NULL;
        }
#define X(ret, name, args)                                                        \
    p_##name = (ret(*) args)dlsym(lib, #name);                                    \
    if (!p_##name) {                                                              \
        napi_throw_error(env, NULL, "libpie_hip.so lacks symbol " #name);         \
        dlclose(lib);                                                             \
        return NULL;                                                              \
    }
        PIE_SYMBOLS(X)
#undef X
        g_lib = lib;
    }
    return js_int(env, p_pie_abi_version());
}

static napi_value fn_device_count(napi_env env, napi_callback_info info)
{
    (void)info;
    if (!need_lib(env)) return NULL;
    return js_int(env, p_pie_device_count());
}

static napi_value fn_ctx_create(napi_env env, napi_callback_info info)
{
    ARGS(1)
    if (!need_lib(env)) return NULL;
    int32_t dev = 0;
    CHECK(env, napi_get_value_int32(env, argv[0], &dev));
    pie_ctx *ctx = NULL;
    int rc = p_pie_ctx_create(dev, &ctx);
    if (rc) return throw_pie(env, NULL, rc);
    ctx_box *b = (ctx_box *)calloc(1, sizeof *b);
    if (!b) {
        p_pie_ctx_destroy(ctx);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    b->ctx = ctx;
    b->owned = 1;
    napi_value ext;
    if (napi_create_external(env, b, box_finalize, NULL, &ext) != napi_ok) {
        p_pie_ctx_destroy(ctx);
        free(b);
        napi_throw_error(env, NULL, "N-API call failed: napi_create_external");
        return NULL;
    }
    return ext;
}

static napi_value fn_ctx_destroy(napi_env env, napi_callback_info info)
{
    ARGS(1)
    ctx_box *b = get_box(env, argv[0]);
    if (!b) return NULL;
    if (b->busy) return throw_state(env, "pie_scan error -6: an asynchronous scan is in flight on this context");
    if (b->ctx && b->owned) p_pie_ctx_destroy(b->ctx);
    b->ctx = NULL; /* the handle stays valid as an object and throws on any further use */
    return js_int(env, 0);
}

/* loadColumns(ctx, start, end, user, disc, nUsers) / appendRows(...) */
static napi_value load_or_append(napi_env env, napi_callback_info info, int append)
{
    ARGS(6)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    size_t n0, n1, n2, n3;
    int64_t *s = typed(env, argv[1], napi_bigint64_array, &n0), *e = typed(env, argv[2], napi_bigint64_array, &n1);
    int32_t *u = typed(env, argv[3], napi_int32_array, &n2), *d = typed(env, argv[4], napi_int32_array, &n3);
    int32_t n_users = 0;
    if (!s || !e || !u || !d || n0 != n1 || n0 != n2 || n0 != n3) {
        napi_throw_type_error(env, NULL, "columns must be BigInt64Array, BigInt64Array, Int32Array, Int32Array of equal length");
        return NULL;
    }
    CHECK(env, napi_get_value_int32(env, argv[5], &n_users));
    int rc = append ? p_pie_append_rows(ctx, s, e, u, d, n0, n_users) : p_pie_load_columns(ctx, s, e, u, d, n0, n_users);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)n0);
}
static napi_value fn_load_columns(napi_env env, napi_callback_info info) { return load_or_append(env, info, 0); }
static napi_value fn_append_rows(napi_env env, napi_callback_info info) { return load_or_append(env, info, 1); }

/* genSynthetic(ctx, seed(BigInt|Number), nTotal, row0, n, nUsers, nDisc, flags) */
static napi_value fn_gen(napi_env env, napi_callback_info info)
{
    ARGS(8)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t seed, n_total, row0, n;
    int32_t n_users, n_disc, flags;
    if (!get_i64(env, argv[1], &seed) || !get_i64(env, argv[2], &n_total) || !get_i64(env, argv[3], &row0) ||
        !get_i64(env, argv[4], &n)) {
        napi_throw_type_error(env, NULL, "bad integer argument");
        return NULL;
    }
    CHECK(env, napi_get_value_int32(env, argv[5], &n_users));
    CHECK(env, napi_get_value_int32(env, argv[6], &n_disc));
    CHECK(env, napi_get_value_int32(env, argv[7], &flags));
    int rc = p_pie_gen_synthetic(ctx, (uint64_t)seed, n_total, row0, n, n_users, n_disc, (uint32_t)flags);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, n);
}

/* readColumns(ctx, start, end, user, disc) */
static napi_value fn_read_columns(napi_env env, napi_callback_info info)
{
    ARGS(5)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    size_t n = 0, n1 = 0, n2 = 0, n3 = 0;
    int64_t *s = typed(env, argv[1], napi_bigint64_array, &n), *e = typed(env, argv[2], napi_bigint64_array, &n1);
    int32_t *u = typed(env, argv[3], napi_int32_array, &n2), *d = typed(env, argv[4], napi_int32_array, &n3);
    if (!s || !e || !u || !d || n1 < n || n2 < n || n3 < n) {
        napi_throw_type_error(env, NULL, "readColumns(ctx, BigInt64Array, BigInt64Array, Int32Array, Int32Array): every array at least as long as the first");
        return NULL;
    }
    int rc = p_pie_read_columns(ctx, s, e, u, d, n);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)n);
}

/* saveColumns(ctx, dir) / loadColumnsDir(ctx, dir) -> {rows, users}: the flat column files of pie_save_columns */
static napi_value dir_call(napi_env env, napi_callback_info info, int load)
{
    ARGS(2)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    char dir[4096];
    size_t len = 0;
    if (napi_get_value_string_utf8(env, argv[1], dir, sizeof dir, &len) != napi_ok || len == 0 || len >= sizeof dir - 1) {
        napi_throw_type_error(env, NULL, "directory path expected");
        return NULL;
    }
    int rc = load ? p_pie_load_columns_dir(ctx, dir) : p_pie_save_columns(ctx, dir);
    if (rc) return throw_pie(env, ctx, rc);
    pie_stats st;
    memset(&st, 0, sizeof st);
    st.struct_size = sizeof st;
    rc = p_pie_stats_get(ctx, &st);
    if (rc) return throw_pie(env, ctx, rc);
    napi_value out;
    CHECK(env, napi_create_object(env, &out));
    napi_set_named_property(env, out, "rows", js_int(env, (int64_t)st.rows));
    napi_set_named_property(env, out, "users", js_int(env, (int64_t)st.users));
    return out;
}
static napi_value fn_save_columns(napi_env env, napi_callback_info info) { return dir_call(env, info, 0); }
static napi_value fn_load_columns_dir(napi_env env, napi_callback_info info) { return dir_call(env, info, 1); }

/* setEnd(ctx, rows Int32Array, newEnd BigInt64Array) */
static napi_value fn_set_end(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    size_t k = 0, k2 = 0;
    int32_t *rows = typed(env, argv[1], napi_int32_array, &k);
    int64_t *ne = typed(env, argv[2], napi_bigint64_array, &k2);
    if (!rows || !ne || k != k2) {
        napi_throw_type_error(env, NULL, "setEnd(ctx, Int32Array rows, BigInt64Array newEnd)");
        return NULL;
    }
    int rc = p_pie_set_end(ctx, rows, ne, k);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)k);
}

/* deleteUser(ctx, user, rowsOut Int32Array) -> number of rows tombstoned (rowsOut holds them, ascending) */
static napi_value fn_delete_user(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t user = -1;
    size_t cap = 0, k = 0;
    CHECK(env, napi_get_value_int32(env, argv[1], &user));
    int32_t *rows = typed(env, argv[2], napi_int32_array, &cap);
    if (!rows) {
        napi_throw_type_error(env, NULL, "deleteUser(ctx, user, Int32Array rowsOut)");
        return NULL;
    }
    int rc = p_pie_delete_user(ctx, user, rows, cap, &k);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)k);
}

/* retentionPurgeTz(ctx, now, months, transitions BigInt64Array[n], offsets BigInt64Array[n + 1], rowsOut Int32Array) -> rows
 * tombstoned (rowsOut holds them, ascending): now >= addMonths(start, months) with `setMonth` on a LOCAL Date under the zone the
 * table describes (host/tzTable.js builds it from this process's zone rules); sqlProvider.js:863-890,991-1009 */
static napi_value fn_retention_purge_tz(napi_env env, napi_callback_info info)
{
    ARGS(6)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t now = 0;
    int32_t months = 0;
    size_t nt = 0, no = 0, cap = 0, k = 0;
    CHECK(env, napi_get_value_int64(env, argv[1], &now));
    CHECK(env, napi_get_value_int32(env, argv[2], &months));
    int64_t *tr = typed(env, argv[3], napi_bigint64_array, &nt), *off = typed(env, argv[4], napi_bigint64_array, &no);
    int32_t *rows = typed(env, argv[5], napi_int32_array, &cap);
    if (!tr || !off || !rows || no != nt + 1) {
        napi_throw_type_error(env, NULL, "retentionPurgeTz(ctx, now, months, BigInt64Array transitions[n], BigInt64Array offsets[n + 1], Int32Array rowsOut)");
        return NULL;
    }
    int rc = p_pie_retention_purge_tz(ctx, now, months, tr, off, (int32_t)nt, rows, cap, &k);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)k);
}

/* batchFetchRequests(ctx, qi Int32Array[n], users Int32Array[n], off BigInt64Array[n + 1], idx Int32Array[cap], start BigInt64Array[cap],
 * end BigInt64Array[cap], disc Int32Array[cap]) -> total rows: the feeds of n (query, user) requests of the last batch in one call;
 * request i's rows are idx[off[i] .. off[i + 1]) with their columns.  Throws code -5 when cap is too small. */
static napi_value fn_batch_fetch_requests(napi_env env, napi_callback_info info)
{
    ARGS(8)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    size_t nq = 0, nu = 0, no = 0, ci = 0, cs = 0, ce = 0, cd = 0, total = 0;
    int32_t *qi = typed(env, argv[1], napi_int32_array, &nq), *us = typed(env, argv[2], napi_int32_array, &nu);
    int64_t *off = typed(env, argv[3], napi_bigint64_array, &no);
    int32_t *idx = typed(env, argv[4], napi_int32_array, &ci);
    int64_t *st = typed(env, argv[5], napi_bigint64_array, &cs), *en = typed(env, argv[6], napi_bigint64_array, &ce);
    int32_t *di = typed(env, argv[7], napi_int32_array, &cd);
    if (!qi || !us || !off || !idx || !st || !en || !di || nq != nu || no < nq + 1 || cs < ci || ce < ci || cd < ci) {
        napi_throw_type_error(env, NULL, "batchFetchRequests(ctx, Int32Array qi[n], Int32Array users[n], BigInt64Array off[n + 1], Int32Array idx[cap], BigInt64Array start[cap], BigInt64Array end[cap], Int32Array disc[cap])");
        return NULL;
    }
    int rc = p_pie_batch_fetch_requests(ctx, qi, us, nq, ci, off, idx, st, en, di, &total);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)total);
}

/* setDisciplines(ctx, mask BigInt|Number, nDisc) */
static napi_value fn_set_disc(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    uint64_t mask = 0;
    napi_valuetype t;
    napi_typeof(env, argv[1], &t);
    if (t == napi_bigint) {
        bool lossless;
        CHECK(env, napi_get_value_bigint_uint64(env, argv[1], &mask, &lossless));
    } else {
        int64_t v;
        if (!get_i64(env, argv[1], &v)) {
            napi_throw_type_error(env, NULL, "mask must be a BigInt or an integer Number");
            return NULL;
        }
        mask = (uint64_t)v;
    }
    int32_t n_disc;
    CHECK(env, napi_get_value_int32(env, argv[2], &n_disc));
    int rc = p_pie_set_disciplines(ctx, mask, n_disc);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, 0);
}

/* scan(ctx, now, cutoff, counts Int32Array[U], offsets BigInt64Array[U+1], idx Int32Array[cap]) -> M */
static napi_value fn_scan(napi_env env, napi_callback_info info)
{
    ARGS(6)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t now, cutoff;
    if (!get_i64(env, argv[1], &now) || !get_i64(env, argv[2], &cutoff)) {
        napi_throw_type_error(env, NULL, "now / cutoff must be finite integers (Number or BigInt)");
        return NULL;
    }
    size_t cap = 0, nc = 0, no = 0, rows = 0, users = 0;
    int32_t *counts = typed(env, argv[3], napi_int32_array, &nc);
    int64_t *offsets = typed(env, argv[4], napi_bigint64_array, &no);
    int32_t *idx = typed(env, argv[5], napi_int32_array, &cap);
    int rc = table_shape(ctx, &rows, &users);
    if (rc) return throw_pie(env, ctx, rc);
    if (!counts || !offsets || !idx || nc < users || no < users + 1) { /* the ABI writes counts[U] and offsets[U + 1] */
        napi_throw_type_error(env, NULL, "scan(ctx, now, cutoff, Int32Array[>= users], BigInt64Array[>= users + 1], Int32Array)");
        return NULL;
    }
    size_t m = 0;
    rc = p_pie_scan(ctx, now, cutoff, counts, offsets, idx, cap, &m);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)m);
}

/* scanDevice(ctx, now, cutoff) -> M: the scan with its result left in HBM (read a user's slice with userFeed) */
static napi_value fn_scan_device(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t now, cutoff;
    if (!get_i64(env, argv[1], &now) || !get_i64(env, argv[2], &cutoff)) {
        napi_throw_type_error(env, NULL, "now / cutoff must be finite integers (Number or BigInt)");
        return NULL;
    }
    size_t m = 0;
    int rc = p_pie_scan_device(ctx, now, cutoff, &m);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)m);
}

/* userFeed(ctx, user, idx Int32Array) -> k: rows of that user's feed in the last scan, written to idx[0..k) */
static napi_value fn_user_feed(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t user = 0;
    size_t cap = 0;
    int32_t *idx = typed(env, argv[2], napi_int32_array, &cap);
    if (!get_i64(env, argv[1], &user) || !idx || user < INT32_MIN || user > INT32_MAX) {
        napi_throw_type_error(env, NULL, "userFeed(ctx, user, Int32Array)");
        return NULL;
    }
    size_t k = 0;
    int rc = p_pie_read_user_feed(ctx, (int32_t)user, idx, cap, &k);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)k);
}

/* scanAsync(ctx, now, cutoff, counts, offsets, idx, callback(err, m)) — same scan on the libuv pool */
typedef struct {
    napi_async_work work;
    napi_ref cb, keep[3];
    pie_ctx *ctx;
    ctx_box *box;
    int64_t now, cutoff;
    int32_t *counts, *idx;
    int64_t *offsets;
    size_t cap, m;
    int rc;
    char err[600];
} scan_job;

static void scan_exec(napi_env env, void *data)
{
    (void)env;
    scan_job *j = (scan_job *)data;
    j->rc = p_pie_scan(j->ctx, j->now, j->cutoff, j->counts, j->offsets, j->idx, j->cap, &j->m);
    if (j->rc) snprintf(j->err, sizeof j->err, "pie_scan error %d: %s", j->rc, p_pie_last_error(j->ctx));
}

static void scan_done(napi_env env, napi_status status, void *data)
{
    scan_job *j = (scan_job *)data;
    napi_value cb, global, args[2], res;
    j->box->busy = 0; /* the worker is done with the context: the main thread may use it again */
    napi_get_reference_value(env, j->cb, &cb);
    napi_get_global(env, &global);
    if (status != napi_ok || j->rc) {
        napi_value msg, code;
        napi_create_string_utf8(env, j->rc ? j->err : "scan cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &args[0]);
        napi_create_int32(env, j->rc, &code);
        napi_set_named_property(env, args[0], "code", code);
        napi_get_undefined(env, &args[1]);
    } else {
        napi_get_null(env, &args[0]);
        napi_create_int64(env, (int64_t)j->m, &args[1]);
    }
    napi_call_function(env, global, cb, 2, args, &res);
    napi_delete_reference(env, j->cb);
    for (int i = 0; i < 3; ++i) napi_delete_reference(env, j->keep[i]);
    napi_delete_async_work(env, j->work);
    free(j);
}

static napi_value fn_scan_async(napi_env env, napi_callback_info info)
{
    ARGS(7)
    pie_ctx *ctx = get_ctx(env, argv[0]); /* refuses a context that is already busy */
    if (!ctx) return NULL;
    size_t nc = 0, no = 0, rows = 0, users = 0;
    int rc0 = table_shape(ctx, &rows, &users);
    if (rc0) return throw_pie(env, ctx, rc0);
    scan_job *j = (scan_job *)calloc(1, sizeof *j);
    if (!j) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    j->ctx = ctx;
    j->box = get_box(env, argv[0]);
    j->counts = typed(env, argv[3], napi_int32_array, &nc);
    j->offsets = typed(env, argv[4], napi_bigint64_array, &no);
    j->idx = typed(env, argv[5], napi_int32_array, &j->cap);
    if (!get_i64(env, argv[1], &j->now) || !get_i64(env, argv[2], &j->cutoff) || !j->counts || !j->offsets || !j->idx ||
        nc < users || no < users + 1) {
        free(j);
        napi_throw_type_error(env, NULL, "scanAsync(ctx, now, cutoff, Int32Array[>= users], BigInt64Array[>= users + 1], Int32Array, cb)");
        return NULL;
    }
    napi_value name;
    napi_create_string_utf8(env, "pie_scan", NAPI_AUTO_LENGTH, &name);
    napi_create_reference(env, argv[6], 1, &j->cb);
    for (int i = 0; i < 3; ++i) napi_create_reference(env, argv[3 + i], 1, &j->keep[i]); /* keep buffers alive */
    j->box->busy = 1;
    if (napi_create_async_work(env, NULL, name, scan_exec, scan_done, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        j->box->busy = 0;
        free(j);
        napi_throw_error(env, NULL, "cannot queue async work");
        return NULL;
    }
    return js_int(env, 0);
}

/* fetchRows(ctx, idx Int32Array, m, start, end, user, disc) */
static napi_value fn_fetch_rows(napi_env env, napi_callback_info info)
{
    ARGS(7)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    size_t cap = 0, c1 = 0;
    int32_t *idx = typed(env, argv[1], napi_int32_array, &cap);
    int64_t m = 0;
    if (!idx || !get_i64(env, argv[2], &m) || m < 0 || (size_t)m > cap) {
        napi_throw_type_error(env, NULL, "fetchRows(ctx, Int32Array idx, m <= idx.length, ...)");
        return NULL;
    }
    size_t c2 = 0, c3 = 0, c4 = 0;
    int64_t *s = typed(env, argv[3], napi_bigint64_array, &c1), *e = typed(env, argv[4], napi_bigint64_array, &c2);
    int32_t *u = typed(env, argv[5], napi_int32_array, &c3), *d = typed(env, argv[6], napi_int32_array, &c4);
    if (!s || !e || !u || !d || c1 < (size_t)m || c2 < (size_t)m || c3 < (size_t)m || c4 < (size_t)m) {
        napi_throw_type_error(env, NULL, "bad output arrays");
        return NULL;
    }
    int rc = p_pie_fetch_rows(ctx, idx, (size_t)m, s, e, u, d);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, m);
}

/* expiredQueue(ctx, prevNow, now, queue Int32Array) -> q */
static napi_value fn_expired_queue(napi_env env, napi_callback_info info)
{
    ARGS(4)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t prev, now;
    size_t cap = 0, q = 0;
    int32_t *queue = typed(env, argv[3], napi_int32_array, &cap);
    if (!get_i64(env, argv[1], &prev) || !get_i64(env, argv[2], &now) || !queue) {
        napi_throw_type_error(env, NULL, "expiredQueue(ctx, prevNow, now, Int32Array)");
        return NULL;
    }
    int rc = p_pie_expired_queue(ctx, prev, now, queue, cap, &q);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)q);
}

/* archiveQueue(ctx, now, windowMs, queue Int32Array) -> q */
static napi_value fn_archive_queue(napi_env env, napi_callback_info info)
{
    ARGS(4)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int64_t now, window;
    size_t cap = 0, q = 0;
    int32_t *queue = typed(env, argv[3], napi_int32_array, &cap);
    if (!get_i64(env, argv[1], &now) || !get_i64(env, argv[2], &window) || !queue) {
        napi_throw_type_error(env, NULL, "archiveQueue(ctx, now, windowMs, Int32Array)");
        return NULL;
    }
    int rc = p_pie_archive_queue(ctx, now, window, queue, cap, &q);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)q);
}

/* ---- native feed serialiser ---------------------------------------------------------------
 * serializeEvents(idx Int32Array, m, start BigInt64Array, end BigInt64Array, disc Int32Array, perDisc Array) -> Buffer
 * holding exactly the bytes of JSON.stringify({events: rows.map(eventFromRow)}) for the event shape of
 * /root/reference/server/calendarFeed.js:66-79 as host/calendarFeed.js builds it from session rows.  perDisc[d] =
 * [titlePrefixJson (no quotes, already escaped), eventNameJson, colorJson] — the per-discipline constants are computed
 * once in JS (parseCalendarMetadata), the per-row work (numbers, ISO dates, allDay) happens here.  Returns null when a
 * row needs something this fast path does not cover (year outside 0000..9999, discipline outside the table): the
 * caller then uses the JS path. */
typedef struct { char *p; size_t len, cap; } sbuf;
static int sb_need(sbuf *b, size_t extra)
{
    if (b->len + extra <= b->cap) return 1;
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->len + extra) nc *= 2;
    char *np = (char *)realloc(b->p, nc);
    if (!np) return 0;
    b->p = np;
    b->cap = nc;
    return 1;
}
static int sb_put(sbuf *b, const char *s, size_t n)
{
    if (!sb_need(b, n)) return 0;
    memcpy(b->p + b->len, s, n);
    b->len += n;
    return 1;
}
#define SB_LIT(b, lit) sb_put((b), (lit), sizeof(lit) - 1)
static int sb_i64(sbuf *b, int64_t v)
{
    char tmp[24];
    int pos = 24;
    uint64_t u = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    do { tmp[--pos] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) tmp[--pos] = '-';
    return sb_put(b, tmp + pos, (size_t)(24 - pos));
}
/* ms since epoch -> "YYYY-MM-DDTHH:mm:ss.sssZ"; returns 0 when the year is outside 0000..9999; *h / *mi = UTC hour, minute */
static int iso_utc(int64_t ms, char out[25], int *h, int *mi)
{
    int64_t days = ms / 86400000, rem = ms % 86400000;
    if (rem < 0) { rem += 86400000; days -= 1; }
    int64_t z = days + 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const int64_t doe = z - era * 146097;
    const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const int64_t mp = (5 * doy + 2) / 153;
    const int64_t d = doy - (153 * mp + 2) / 5 + 1;
    const int64_t m = mp < 10 ? mp + 3 : mp - 9;
    const int64_t y = yoe + era * 400 + (m <= 2 ? 1 : 0);
    if (y < 0 || y > 9999) return 0;
    *h = (int)(rem / 3600000);
    *mi = (int)(rem / 60000 % 60);
    const int sec = (int)(rem / 1000 % 60), msec = (int)(rem % 1000), yy = (int)y;
#define D2(pos, v) (out[pos] = (char)('0' + (v) / 10), out[(pos) + 1] = (char)('0' + (v) % 10))
    D2(0, yy / 100); D2(2, yy % 100); out[4] = '-'; D2(5, (int)m); out[7] = '-'; D2(8, (int)d); out[10] = 'T';
    D2(11, *h); out[13] = ':'; D2(14, *mi); out[16] = ':'; D2(17, sec); out[19] = '.';
    out[20] = (char)('0' + msec / 100); D2(21, msec % 100); out[23] = 'Z'; out[24] = 0;
#undef D2
    return 1;
}

static napi_value fn_serialize_events(napi_env env, napi_callback_info info)
{
    ARGS(6)
    size_t cap = 0, n1 = 0, n2 = 0, n3 = 0;
    int32_t *idx = typed(env, argv[0], napi_int32_array, &cap);
    int64_t m = 0;
    int64_t *start = typed(env, argv[2], napi_bigint64_array, &n1), *end = typed(env, argv[3], napi_bigint64_array, &n2);
    int32_t *disc = typed(env, argv[4], napi_int32_array, &n3);
    uint32_t n_disc = 0;
    if (!idx || !get_i64(env, argv[1], &m) || m < 0 || (size_t)m > cap || !start || !end || !disc || n1 < (size_t)m ||
        n2 < (size_t)m || n3 < (size_t)m || napi_get_array_length(env, argv[5], &n_disc) != napi_ok || n_disc > 64) {
        napi_throw_type_error(env, NULL, "serializeEvents(Int32Array idx, m, BigInt64Array start, BigInt64Array end, Int32Array disc, Array perDisc)");
        return NULL;
    }
    /* per-discipline constants */
    char *parts[64][3];
    size_t plen[64][3];
    memset(parts, 0, sizeof parts);
    int ok = 1;
    for (uint32_t dix = 0; dix < n_disc && ok; ++dix) {
        napi_value row;
        if (napi_get_element(env, argv[5], dix, &row) != napi_ok) { ok = 0; break; }
        for (uint32_t k = 0; k < 3 && ok; ++k) {
            napi_value sv;
            size_t len = 0;
            if (napi_get_element(env, row, k, &sv) != napi_ok || napi_get_value_string_utf8(env, sv, NULL, 0, &len) != napi_ok) { ok = 0; break; }
            parts[dix][k] = (char *)malloc(len + 1);
            if (!parts[dix][k] || napi_get_value_string_utf8(env, sv, parts[dix][k], len + 1, &plen[dix][k]) != napi_ok) ok = 0;
        }
    }
    sbuf b = {NULL, 0, 0};
    int covered = ok;
    if (ok) ok = SB_LIT(&b, "{\"events\":[");
    for (int64_t i = 0; i < m && ok && covered; ++i) {
        const int32_t dv = disc[i];
        if (dv < 0 || (uint32_t)dv >= n_disc) { covered = 0; break; }
        const int has_end = end[i] != INT64_MIN;
        char iso_s[25], iso_e[25];
        int hs = 0, ms_ = 0, he = 0, me = 0;
        if (!iso_utc(start[i], iso_s, &hs, &ms_) || (has_end && !iso_utc(end[i], iso_e, &he, &me))) { covered = 0; break; }
        /* allDay heuristic of calendarFeed.js:64 on the UTC clock fields */
        const int all_day = hs == 0 && ms_ == 0 && (!has_end || he == 0);
        ok = (i == 0 || SB_LIT(&b, ",")) && SB_LIT(&b, "{\"id\":\"session-") && sb_i64(&b, idx[i]) && SB_LIT(&b, "\",\"title\":\"") &&
             sb_put(&b, parts[dv][0], plen[dv][0]) && SB_LIT(&b, " session #") && sb_i64(&b, idx[i]) &&
             SB_LIT(&b, "\",\"description\":\"\",\"location\":\"\",\"start\":\"") && sb_put(&b, iso_s, 24) && SB_LIT(&b, "\",\"end\":\"") &&
             (has_end ? sb_put(&b, iso_e, 24) : 1) && SB_LIT(&b, "\",\"startTs\":") && sb_i64(&b, start[i]) && SB_LIT(&b, ",\"endTs\":") &&
             (has_end ? sb_i64(&b, end[i]) : SB_LIT(&b, "null")) && SB_LIT(&b, ",\"allDay\":") &&
             (all_day ? SB_LIT(&b, "true") : SB_LIT(&b, "false")) && SB_LIT(&b, ",\"eventName\":") &&
             sb_put(&b, parts[dv][1], plen[dv][1]) && SB_LIT(&b, ",\"showNumber\":") && sb_i64(&b, idx[i]) && SB_LIT(&b, ",\"color\":") &&
             sb_put(&b, parts[dv][2], plen[dv][2]) && SB_LIT(&b, "}");
    }
    if (ok && covered) ok = SB_LIT(&b, "]}");
    for (uint32_t dix = 0; dix < 64; ++dix)
        for (int k = 0; k < 3; ++k) free(parts[dix][k]);
    napi_value out = NULL;
    if (!ok) {
        free(b.p);
        napi_throw_error(env, NULL, "serializeEvents: out of memory or bad perDisc table");
        return NULL;
    }
    if (!covered) {
        free(b.p);
        napi_get_null(env, &out);
        return out;
    }
    void *copy = NULL;
    if (napi_create_buffer_copy(env, b.len, b.p, &copy, &out) != napi_ok) {
        free(b.p);
        napi_throw_error(env, NULL, "cannot create Buffer");
        return NULL;
    }
    free(b.p);
    return out;
}

/* serializeICal(idx Int32Array, m, start BigInt64Array, end BigInt64Array, disc Int32Array, summaries Array<string>,
 *               dtstampMs) -> Buffer with exactly the bytes of host/calendarFeed.js toICalendar(rows.map(eventFromRow)):
 * the iCalendar (RFC 5545) form of the feed — new functionality, the reference only consumes .ics.  summaries[d] = the
 * escaped SUMMARY prefix of discipline d ("<name> session #" follows the row number).  Returns null when a row is
 * outside what this fast path covers (year outside 0000..9999, unknown discipline, a SUMMARY line that would need
 * folding): the caller then uses the JS path. */
static int ics_utc(int64_t ms, char out[17])
{
    char iso[25];
    int h, mi;
    int64_t sec_ms = ms - (((ms % 1000) + 1000) % 1000); /* floor to the second, also for negative ms */
    if (!iso_utc(sec_ms, iso, &h, &mi)) return 0;
    /* YYYY-MM-DDTHH:mm:ss.sssZ -> YYYYMMDDTHHMMSSZ */
    memcpy(out, iso, 4); memcpy(out + 4, iso + 5, 2); memcpy(out + 6, iso + 8, 2);
    out[8] = 'T';
    memcpy(out + 9, iso + 11, 2); memcpy(out + 11, iso + 14, 2); memcpy(out + 13, iso + 17, 2);
    out[15] = 'Z';
    out[16] = 0;
    return 1;
}

static napi_value fn_serialize_ical(napi_env env, napi_callback_info info)
{
    ARGS(7)
    size_t cap = 0, n1 = 0, n2 = 0, n3 = 0;
    int32_t *idx = typed(env, argv[0], napi_int32_array, &cap);
    int64_t m = 0, stamp_ms = 0;
    int64_t *start = typed(env, argv[2], napi_bigint64_array, &n1), *end = typed(env, argv[3], napi_bigint64_array, &n2);
    int32_t *disc = typed(env, argv[4], napi_int32_array, &n3);
    uint32_t n_disc = 0;
    if (!idx || !get_i64(env, argv[1], &m) || m < 0 || (size_t)m > cap || !start || !end || !disc || n1 < (size_t)m ||
        n2 < (size_t)m || n3 < (size_t)m || napi_get_array_length(env, argv[5], &n_disc) != napi_ok || n_disc > 64 ||
        !get_i64(env, argv[6], &stamp_ms)) {
        napi_throw_type_error(env, NULL, "serializeICal(Int32Array idx, m, BigInt64Array start, BigInt64Array end, Int32Array disc, Array summaries, dtstampMs)");
        return NULL;
    }
    char *sum[64];
    size_t sum_len[64];
    memset(sum, 0, sizeof sum);
    int ok = 1, covered = 1;
    for (uint32_t d = 0; d < n_disc && ok; ++d) {
        napi_value sv;
        size_t len = 0;
        if (napi_get_element(env, argv[5], d, &sv) != napi_ok || napi_get_value_string_utf8(env, sv, NULL, 0, &len) != napi_ok) { ok = 0; break; }
        sum[d] = (char *)malloc(len + 1);
        if (!sum[d] || napi_get_value_string_utf8(env, sv, sum[d], len + 1, &sum_len[d]) != napi_ok) ok = 0;
        /* "SUMMARY:" + prefix + up to 10 digits must stay within 75 octets, or the line would need folding */
        if (ok && 8 + sum_len[d] + 10 > 75) covered = 0;
    }
    char stamp[17];
    if (ok && !ics_utc(stamp_ms, stamp)) covered = 0;
    sbuf b = {NULL, 0, 0};
    if (ok && covered) ok = SB_LIT(&b, "BEGIN:VCALENDAR\r\nVERSION:2.0\r\nPRODID:-//sph-pie_amd//session feed//EN\r\nCALSCALE:GREGORIAN\r\n");
    for (int64_t i = 0; i < m && ok && covered; ++i) {
        const int32_t dv = disc[i];
        char ds[17], de[17];
        const int has_end = end[i] != INT64_MIN;
        if (dv < 0 || (uint32_t)dv >= n_disc || !ics_utc(start[i], ds) || (has_end && !ics_utc(end[i], de))) { covered = 0; break; }
        ok = SB_LIT(&b, "BEGIN:VEVENT\r\nUID:session-") && sb_i64(&b, idx[i]) && SB_LIT(&b, "\r\nDTSTAMP:") && sb_put(&b, stamp, 16) &&
             SB_LIT(&b, "\r\nDTSTART:") && sb_put(&b, ds, 16) && (has_end ? (SB_LIT(&b, "\r\nDTEND:") && sb_put(&b, de, 16)) : 1) &&
             SB_LIT(&b, "\r\nSUMMARY:") && sb_put(&b, sum[dv], sum_len[dv]) && sb_i64(&b, idx[i]) && SB_LIT(&b, "\r\nEND:VEVENT\r\n");
    }
    if (ok && covered) ok = SB_LIT(&b, "END:VCALENDAR\r\n");
    for (uint32_t d = 0; d < 64; ++d) free(sum[d]);
    napi_value out = NULL;
    if (!ok) {
        free(b.p);
        napi_throw_error(env, NULL, "serializeICal: out of memory or bad summaries table");
        return NULL;
    }
    if (!covered) {
        free(b.p);
        napi_get_null(env, &out);
        return out;
    }
    void *copy = NULL;
    if (napi_create_buffer_copy(env, b.len, b.p, &copy, &out) != napi_ok) {
        free(b.p);
        napi_throw_error(env, NULL, "cannot create Buffer");
        return NULL;
    }
    free(b.p);
    return out;
}

/* serializeCsv(idx Int32Array, m, start, end BigInt64Array, user, disc Int32Array, userIds Array<string>, discIds Array<string>,
 *              headerLine string) -> Buffer: header + one CSV line per row, '\n' between lines — exactly the bytes of
 * host/dispatchQueue.js [header].concat(rows.map(buildCsvRow)).join('\n') with the columns sessionRow, userId, discipline,
 * createdAt, expiredAt.  csvEscape is the reference's rule (/root/reference/server/webhookDispatcher.js:332-338): a field is
 * quoted iff it holds '"', ',', '\n' or '\r', inner quotes doubled.  Returns null for rows this fast path does not cover
 * (a year outside 0000..9999): the caller then uses the JS path. */
static int sb_csv_field(sbuf *b, const char *s, size_t n)
{
    int quote = 0;
    for (size_t i = 0; i < n; ++i)
        if (s[i] == '"' || s[i] == ',' || s[i] == '\n' || s[i] == '\r') { quote = 1; break; }
    if (!quote) return sb_put(b, s, n);
    if (!SB_LIT(b, "\"")) return 0;
    for (size_t i = 0; i < n; ++i) {
        if (s[i] == '"' && !SB_LIT(b, "\"")) return 0;
        if (!sb_put(b, s + i, 1)) return 0;
    }
    return SB_LIT(b, "\"");
}

static napi_value fn_serialize_csv(napi_env env, napi_callback_info info)
{
    ARGS(9)
    size_t cap = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;
    int32_t *idx = typed(env, argv[0], napi_int32_array, &cap);
    int64_t m = 0;
    int64_t *start = typed(env, argv[2], napi_bigint64_array, &n1), *end = typed(env, argv[3], napi_bigint64_array, &n2);
    int32_t *user = typed(env, argv[4], napi_int32_array, &n3), *disc = typed(env, argv[5], napi_int32_array, &n4);
    uint32_t n_users = 0, n_disc = 0;
    size_t hlen = 0;
    if (!idx || !get_i64(env, argv[1], &m) || m < 0 || (size_t)m > cap || !start || !end || !user || !disc || n1 < (size_t)m ||
        n2 < (size_t)m || n3 < (size_t)m || n4 < (size_t)m || napi_get_array_length(env, argv[6], &n_users) != napi_ok ||
        napi_get_array_length(env, argv[7], &n_disc) != napi_ok || napi_get_value_string_utf8(env, argv[8], NULL, 0, &hlen) != napi_ok) {
        napi_throw_type_error(env, NULL, "serializeCsv(Int32Array idx, m, BigInt64Array start, BigInt64Array end, Int32Array user, Int32Array disc, Array userIds, Array discIds, header)");
        return NULL;
    }
    sbuf b = {NULL, 0, 0};
    int ok = sb_need(&b, hlen + 1), covered = 1;
    if (ok) {
        napi_get_value_string_utf8(env, argv[8], b.p, hlen + 1, &hlen);
        b.len = hlen;
    }
    char *tmp = NULL;
    size_t tmp_cap = 0;
    for (int64_t i = 0; i < m && ok && covered; ++i) {
        char iso_s[25], iso_e[25];
        int h, mi;
        const int has_end = end[i] != INT64_MIN;
        if (!iso_utc(start[i], iso_s, &h, &mi) || (has_end && !iso_utc(end[i], iso_e, &h, &mi))) { covered = 0; break; }
        ok = SB_LIT(&b, "\n") && sb_i64(&b, idx[i]) && SB_LIT(&b, ",");
        /* the two string columns: '' when the index is outside the table (userIds[u] === undefined / no discipline) */
        for (int col = 0; col < 2 && ok; ++col) {
            const int32_t k = col == 0 ? user[i] : disc[i];
            const uint32_t lim = col == 0 ? n_users : n_disc;
            if (k >= 0 && (uint32_t)k < lim) {
                napi_value sv;
                size_t len = 0;
                napi_valuetype t;
                if (napi_get_element(env, argv[6 + col], (uint32_t)k, &sv) != napi_ok || napi_typeof(env, sv, &t) != napi_ok) { ok = 0; break; }
                if (t == napi_string) {
                    if (napi_get_value_string_utf8(env, sv, NULL, 0, &len) != napi_ok) { ok = 0; break; }
                    if (len + 1 > tmp_cap) {
                        char *np = (char *)realloc(tmp, len + 1);
                        if (!np) { ok = 0; break; }
                        tmp = np;
                        tmp_cap = len + 1;
                    }
                    if (napi_get_value_string_utf8(env, sv, tmp, len + 1, &len) != napi_ok) { ok = 0; break; }
                    ok = sb_csv_field(&b, tmp, len);
                } else if (t != napi_undefined && t != napi_null) {
                    covered = 0; /* a non-string id: String(value) is the JS path's business */
                }
            }
            if (ok) ok = SB_LIT(&b, ",");
        }
        if (ok && covered) ok = sb_put(&b, iso_s, 24) && SB_LIT(&b, ",") && (has_end ? sb_put(&b, iso_e, 24) : 1);
    }
    free(tmp);
    napi_value out = NULL;
    if (!ok) {
        free(b.p);
        napi_throw_error(env, NULL, "serializeCsv: out of memory or bad id table");
        return NULL;
    }
    if (!covered) {
        free(b.p);
        napi_get_null(env, &out);
        return out;
    }
    void *copy = NULL;
    if (napi_create_buffer_copy(env, b.len, b.p, &copy, &out) != napi_ok) {
        free(b.p);
        napi_throw_error(env, NULL, "cannot create Buffer");
        return NULL;
    }
    free(b.p);
    return out;
}

/* ---- batched scan: the requests of one event-loop turn, one table pass ------------------------------------------------
 * queries travel as three typed arrays of equal length (<= PIE_BATCH_MAX): now, cutoff (BigInt64Array), mask (BigUint64Array) */
static int read_queries(napi_env env, napi_value nows, napi_value cutoffs, napi_value masks, pie_query *out, int *n_q)
{
    size_t a = 0, b = 0, c = 0;
    int64_t *pn = typed(env, nows, napi_bigint64_array, &a), *pc = typed(env, cutoffs, napi_bigint64_array, &b);
    uint64_t *pm = typed(env, masks, napi_biguint64_array, &c);
    if (!pn || !pc || !pm || a != b || a != c || a < 1 || a > PIE_BATCH_MAX) return 0;
    for (size_t k = 0; k < a; ++k) {
        out[k].now = pn[k];
        out[k].cutoff = pc[k];
        out[k].mask = pm[k];
    }
    *n_q = (int)a;
    return 1;
}

/* scanBatch(ctx, nows, cutoffs, masks) -> Array of M per query; results stay in HBM (batchUserFeed reads a slice) */
static napi_value fn_scan_batch(napi_env env, napi_callback_info info)
{
    ARGS(4)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    pie_query q[PIE_BATCH_MAX];
    int n_q = 0;
    if (!read_queries(env, argv[1], argv[2], argv[3], q, &n_q)) {
        napi_throw_type_error(env, NULL, "scanBatch(ctx, BigInt64Array nows, BigInt64Array cutoffs, BigUint64Array masks): 1..64 queries, equal lengths");
        return NULL;
    }
    size_t m[PIE_BATCH_MAX];
    int rc = p_pie_scan_batch(ctx, q, n_q, m);
    if (rc) return throw_pie(env, ctx, rc);
    napi_value out;
    CHECK(env, napi_create_array_with_length(env, (size_t)n_q, &out));
    for (int k = 0; k < n_q; ++k) napi_set_element(env, out, (uint32_t)k, js_int(env, (int64_t)m[k]));
    return out;
}

/* batchUserFeed(ctx, qi, user, idx Int32Array) -> k */
static napi_value fn_batch_user_feed(napi_env env, napi_callback_info info)
{
    ARGS(4)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t qi = 0;
    int64_t user = 0;
    size_t cap = 0, k = 0;
    int32_t *idx = typed(env, argv[3], napi_int32_array, &cap);
    if (napi_get_value_int32(env, argv[1], &qi) != napi_ok || !get_i64(env, argv[2], &user) || !idx || user < INT32_MIN || user > INT32_MAX) {
        napi_throw_type_error(env, NULL, "batchUserFeed(ctx, qi, user, Int32Array)");
        return NULL;
    }
    int rc = p_pie_batch_read_user_feed(ctx, qi, (int32_t)user, idx, cap, &k);
    if (rc) return throw_pie(env, ctx, rc);
    return js_int(env, (int64_t)k);
}

/* ---- communicator: the sharded table over several GPUs (pie_comm_*, RCCL behind the C ABI) -------------------------- */
static napi_value throw_comm(napi_env env, pie_comm *cm, int rc)
{
    napi_value msg, err, code;
    char buf[640];
    snprintf(buf, sizeof buf, "pie_comm error %d: %s", rc, p_pie_comm_last_error ? p_pie_comm_last_error(cm) : "library not open");
    napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_create_int32(env, rc, &code);
    napi_set_named_property(env, err, "code", code);
    napi_throw(env, err);
    return NULL;
}

typedef struct comm_box_s {
    pie_comm *comm;
    ctx_box *boxes; /* context handles handed out by commCtx and still alive */
} comm_box;

static void comm_unlink_ctx(ctx_box *b)
{
    if (!b->parent) return;
    for (ctx_box **pp = &b->parent->boxes; *pp; pp = &(*pp)->next)
        if (*pp == b) {
            *pp = b->next;
            break;
        }
    b->parent = NULL;
    b->next = NULL;
}

/* the communicator's contexts are about to die (or just died): every handle to them becomes a destroyed context */
static void comm_orphan_ctxs(comm_box *cb)
{
    for (ctx_box *b = cb->boxes; b;) {
        ctx_box *nx = b->next;
        b->ctx = NULL;
        b->parent = NULL;
        b->next = NULL;
        b = nx;
    }
    cb->boxes = NULL;
}

static int comm_any_busy(const comm_box *cb)
{
    for (const ctx_box *b = cb->boxes; b; b = b->next)
        if (b->busy) return 1;
    return 0;
}

static void comm_finalize(napi_env env, void *data, void *hint)
{
    (void)env;
    (void)hint;
    comm_box *b = (comm_box *)data;
    if (b) comm_orphan_ctxs(b);
    if (b && b->comm && p_pie_comm_destroy) p_pie_comm_destroy(b->comm);
    free(b);
}

static comm_box *get_comm_box(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((comm_box *)p)->comm) {
        napi_throw_type_error(env, NULL, "expected a live pie communicator");
        return NULL;
    }
    if (comm_any_busy((comm_box *)p)) {
        throw_state(env, "pie_comm error -6: an asynchronous scan is in flight on one of this communicator's contexts");
        return NULL;
    }
    return (comm_box *)p;
}

static pie_comm *get_comm(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((comm_box *)p)->comm) {
        napi_throw_type_error(env, NULL, "expected a live pie communicator");
        return NULL;
    }
    return ((comm_box *)p)->comm;
}

/* commCreate(Int32Array deviceIds) -> communicator: one shard per listed GPU, driven by this process */
static napi_value fn_comm_create(napi_env env, napi_callback_info info)
{
    ARGS(1)
    if (!need_lib(env)) return NULL;
    size_t n = 0;
    int32_t *ids = typed(env, argv[0], napi_int32_array, &n);
    if (!ids || n < 1) {
        napi_throw_type_error(env, NULL, "commCreate(Int32Array deviceIds)");
        return NULL;
    }
    pie_comm *cm = NULL;
    int rc = p_pie_comm_create(ids, (int32_t)n, &cm);
    if (rc) return throw_comm(env, NULL, rc);
    comm_box *b = (comm_box *)calloc(1, sizeof *b);
    napi_value ext;
    if (!b || napi_create_external(env, b, comm_finalize, NULL, &ext) != napi_ok) {
        p_pie_comm_destroy(cm);
        free(b);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    b->comm = cm;
    return ext;
}

static napi_value fn_comm_destroy(napi_env env, napi_callback_info info)
{
    ARGS(1)
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a pie communicator");
        return NULL;
    }
    comm_box *b = (comm_box *)p;
    if (comm_any_busy(b)) return throw_state(env, "pie_comm error -6: an asynchronous scan is in flight on one of this communicator's contexts");
    comm_orphan_ctxs(b); /* handles handed out by commCtx now read as destroyed contexts */
    if (b->comm) p_pie_comm_destroy(b->comm);
    b->comm = NULL;
    return js_int(env, 0);
}

static napi_value fn_comm_world(napi_env env, napi_callback_info info)
{
    ARGS(1)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    return js_int(env, p_pie_comm_world(cm));
}

/* commCtx(comm, rank) -> context handle of that shard (owned by the communicator) */
static napi_value fn_comm_ctx(napi_env env, napi_callback_info info)
{
    ARGS(2)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int32_t rank = 0;
    CHECK(env, napi_get_value_int32(env, argv[1], &rank));
    pie_ctx *ctx = p_pie_comm_ctx(cm, rank);
    if (!ctx) return throw_state(env, "pie_comm error -6: that rank is not local to this communicator");
    void *pcb = NULL;
    CHECK(env, napi_get_value_external(env, argv[0], &pcb));
    comm_box *cb = (comm_box *)pcb;
    ctx_box *b = (ctx_box *)calloc(1, sizeof *b);
    napi_value ext;
    if (!b || napi_create_reference(env, argv[0], 1, &b->comm_ref) != napi_ok) {
        free(b);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    if (napi_create_external(env, b, box_finalize, NULL, &ext) != napi_ok) {
        napi_delete_reference(env, b->comm_ref);
        free(b);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    b->ctx = ctx;
    b->owned = 0;
    b->parent = cb;
    b->next = cb->boxes;
    cb->boxes = b;
    return ext;
}

/* commGenSyntheticSharded(comm, seed, nTotal, nUsers, nDisc, flags) */
static napi_value fn_comm_gen(napi_env env, napi_callback_info info)
{
    ARGS(6)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int64_t seed, n_total;
    int32_t n_users, n_disc, flags;
    if (!get_i64(env, argv[1], &seed) || !get_i64(env, argv[2], &n_total)) {
        napi_throw_type_error(env, NULL, "bad integer argument");
        return NULL;
    }
    CHECK(env, napi_get_value_int32(env, argv[3], &n_users));
    CHECK(env, napi_get_value_int32(env, argv[4], &n_disc));
    CHECK(env, napi_get_value_int32(env, argv[5], &flags));
    int rc = p_pie_comm_gen_synthetic_sharded(cm, (uint64_t)seed, n_total, n_users, n_disc, (uint32_t)flags);
    if (rc) return throw_comm(env, cm, rc);
    return js_int(env, n_total);
}

/* commScanBatchGather(comm, nows, cutoffs, masks) -> Array [rank][query] of M: every shard scans the batch, the messages
 * are exchanged over RCCL; commReadGathered then reads one (source rank, query) message as a given rank holds it */
static napi_value fn_comm_scan_gather(napi_env env, napi_callback_info info)
{
    ARGS(4)
    comm_box *cbx = get_comm_box(env, argv[0]); /* refuses while scanAsync runs on one of the shard contexts */
    if (!cbx) return NULL;
    pie_comm *cm = cbx->comm;
    pie_query q[PIE_BATCH_MAX];
    int n_q = 0;
    if (!read_queries(env, argv[1], argv[2], argv[3], q, &n_q)) {
        napi_throw_type_error(env, NULL, "commScanBatchGather(comm, BigInt64Array nows, BigInt64Array cutoffs, BigUint64Array masks)");
        return NULL;
    }
    const int world = p_pie_comm_world(cm);
    size_t *m = (size_t *)calloc((size_t)world * (size_t)n_q, sizeof *m);
    if (!m) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    int rc = p_pie_comm_scan_batch_gather(cm, q, n_q, 0, m);
    if (rc) {
        free(m);
        return throw_comm(env, cm, rc);
    }
    napi_value out;
    napi_create_array_with_length(env, (size_t)world, &out);
    for (int r = 0; r < world; ++r) {
        napi_value row;
        napi_create_array_with_length(env, (size_t)n_q, &row);
        for (int k = 0; k < n_q; ++k) napi_set_element(env, row, (uint32_t)k, js_int(env, (int64_t)m[(size_t)r * n_q + k]));
        napi_set_element(env, out, (uint32_t)r, row);
    }
    free(m);
    return out;
}

/* commReadGathered(comm, atRank, srcRank, qi, offsets Int32Array[>= uPad + 1], idx Int32Array) -> M */
static napi_value fn_comm_read(napi_env env, napi_callback_info info)
{
    ARGS(6)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int32_t at = 0, src = 0, qi = 0;
    size_t no = 0, cap = 0, m = 0, u_pad = 0, rs = 0, qs = 0;
    void *base = NULL;
    CHECK(env, napi_get_value_int32(env, argv[1], &at));
    CHECK(env, napi_get_value_int32(env, argv[2], &src));
    CHECK(env, napi_get_value_int32(env, argv[3], &qi));
    int32_t *off = typed(env, argv[4], napi_int32_array, &no), *idx = typed(env, argv[5], napi_int32_array, &cap);
    int rc = p_pie_comm_gathered_device_ptr(cm, at, &base, &rs, &qs, &u_pad);
    if (rc) return throw_comm(env, cm, rc);
    if (!off || !idx || no < u_pad + 1) {
        napi_throw_type_error(env, NULL, "commReadGathered(comm, at, src, qi, Int32Array[>= uPad + 1], Int32Array)");
        return NULL;
    }
    rc = p_pie_comm_read_gathered(cm, at, src, qi, off, idx, cap, &m);
    if (rc) return throw_comm(env, cm, rc);
    return js_int(env, (int64_t)m);
}

/* commUPad(comm, atRank) -> users per message (the largest shard's user count) of the last exchange */
static napi_value fn_comm_upad(napi_env env, napi_callback_info info)
{
    ARGS(2)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int32_t at = 0;
    size_t u_pad = 0, rs = 0, qs = 0;
    void *base = NULL;
    CHECK(env, napi_get_value_int32(env, argv[1], &at));
    int rc = p_pie_comm_gathered_device_ptr(cm, at, &base, &rs, &qs, &u_pad);
    if (rc) return throw_comm(env, cm, rc);
    return js_int(env, (int64_t)u_pad);
}

/* ---- the pipelined union exchange (pie_comm_step_*): commStepReserve(comm, nQ, unionCap); commStepBegin(comm, nows, cutoffs,
 * masks); commStepFinish(comm) -> [rank][query] M; commStepCollect(comm) -> step number (throws code -5 on every rank alike when a
 * union outgrew the reservation: commNeededCap says what to reserve); commStepReadGathered(comm, at, src, step, uoff, rows, masks) */
static napi_value fn_comm_needed_cap(napi_env env, napi_callback_info info)
{
    ARGS(1)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    return js_int(env, (int64_t)p_pie_comm_needed_cap(cm));
}

static napi_value fn_comm_step_reserve(napi_env env, napi_callback_info info)
{
    ARGS(3)
    comm_box *cbx = get_comm_box(env, argv[0]);
    if (!cbx) return NULL;
    int32_t n_q = 0;
    int64_t cap = 0;
    CHECK(env, napi_get_value_int32(env, argv[1], &n_q));
    CHECK(env, napi_get_value_int64(env, argv[2], &cap));
    int rc = p_pie_comm_step_reserve(cbx->comm, n_q, 0, (size_t)(cap > 0 ? cap : 0));
    if (rc) return throw_comm(env, cbx->comm, rc);
    return js_int(env, 0);
}

static int step_nq_ring[16];
static unsigned step_nq_head = 0, step_nq_tail = 0; /* queries of the steps begun and not finished (three per batch lane of the shards: at most twelve) */

static napi_value fn_comm_step_begin(napi_env env, napi_callback_info info)
{
    ARGS(4)
    comm_box *cbx = get_comm_box(env, argv[0]);
    if (!cbx) return NULL;
    pie_query q[PIE_BATCH_MAX];
    int n_q = 0;
    if (!read_queries(env, argv[1], argv[2], argv[3], q, &n_q)) {
        napi_throw_type_error(env, NULL, "commStepBegin(comm, BigInt64Array nows, BigInt64Array cutoffs, BigUint64Array masks)");
        return NULL;
    }
    int rc = p_pie_comm_step_begin(cbx->comm, q, n_q);
    if (rc) return throw_comm(env, cbx->comm, rc);
    step_nq_ring[step_nq_head++ & 15] = n_q;
    return js_int(env, 0);
}

static napi_value fn_comm_step_finish(napi_env env, napi_callback_info info)
{
    ARGS(1)
    comm_box *cbx = get_comm_box(env, argv[0]);
    if (!cbx) return NULL;
    if (step_nq_tail == step_nq_head) return throw_state(env, "pie_comm error -6: commStepFinish without commStepBegin");
    const int n_q = step_nq_ring[step_nq_tail++ & 15];
    const int world = p_pie_comm_world(cbx->comm);
    size_t *m = (size_t *)calloc((size_t)world * (size_t)n_q, sizeof *m);
    if (!m) {
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    int rc = p_pie_comm_step_finish(cbx->comm, m);
    if (rc) {
        free(m);
        return throw_comm(env, cbx->comm, rc);
    }
    napi_value out;
    napi_create_array_with_length(env, (size_t)world, &out);
    for (int r = 0; r < world; ++r) {
        napi_value row;
        napi_create_array_with_length(env, (size_t)n_q, &row);
        for (int k = 0; k < n_q; ++k) napi_set_element(env, row, (uint32_t)k, js_int(env, (int64_t)m[(size_t)r * n_q + k]));
        napi_set_element(env, out, (uint32_t)r, row);
    }
    free(m);
    return out;
}

static napi_value fn_comm_step_collect(napi_env env, napi_callback_info info)
{
    ARGS(1)
    comm_box *cbx = get_comm_box(env, argv[0]);
    if (!cbx) return NULL;
    int64_t step = -1;
    int rc = p_pie_comm_step_collect(cbx->comm, &step);
    if (rc) return throw_comm(env, cbx->comm, rc);
    return js_int(env, step);
}

/* commStepReadGathered(comm, atRank, srcRank, step, uoff Int32Array[>= uPad + 1], rows Int32Array, masks BigUint64Array) -> Mu */
static napi_value fn_comm_step_read(napi_env env, napi_callback_info info)
{
    ARGS(7)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int32_t at = 0, src = 0;
    int64_t step = 0;
    size_t no = 0, nr = 0, nm = 0, mu = 0, u_pad = 0, rs = 0, cap = 0;
    void *base = NULL;
    CHECK(env, napi_get_value_int32(env, argv[1], &at));
    CHECK(env, napi_get_value_int32(env, argv[2], &src));
    CHECK(env, napi_get_value_int64(env, argv[3], &step));
    int32_t *off = typed(env, argv[4], napi_int32_array, &no), *rows = typed(env, argv[5], napi_int32_array, &nr);
    uint64_t *masks = typed(env, argv[6], napi_biguint64_array, &nm);
    int rc = p_pie_comm_step_gathered_ptr(cm, at, step, &base, &rs, &u_pad, &cap);
    if (rc) return throw_comm(env, cm, rc);
    if (!off || !rows || !masks || no < u_pad + 1) {
        napi_throw_type_error(env, NULL, "commStepReadGathered(comm, at, src, step, Int32Array[>= uPad + 1], Int32Array rows, BigUint64Array masks)");
        return NULL;
    }
    rc = p_pie_comm_step_read_gathered(cm, at, src, step, off, rows, masks, nr < nm ? nr : nm, &mu);
    if (rc) return throw_comm(env, cm, rc);
    return js_int(env, (int64_t)mu);
}

/* commStepUPad(comm, atRank, step) -> users per union message */
static napi_value fn_comm_step_upad(napi_env env, napi_callback_info info)
{
    ARGS(3)
    pie_comm *cm = get_comm(env, argv[0]);
    if (!cm) return NULL;
    int32_t at = 0;
    int64_t step = 0;
    size_t u_pad = 0, rs = 0, cap = 0;
    void *base = NULL;
    CHECK(env, napi_get_value_int32(env, argv[1], &at));
    CHECK(env, napi_get_value_int64(env, argv[2], &step));
    int rc = p_pie_comm_step_gathered_ptr(cm, at, step, &base, &rs, &u_pad, &cap);
    if (rc) return throw_comm(env, cm, rc);
    return js_int(env, (int64_t)u_pad);
}

/* stats(ctx) -> {rows, users, selected, algBytes, k1MsSum, scanMsSum, nProfiled, maxBucket} */
static napi_value fn_stats(napi_env env, napi_callback_info info)
{
    ARGS(1)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    pie_stats st;
    memset(&st, 0, sizeof st);
    st.struct_size = sizeof st;
    int rc = p_pie_stats_get(ctx, &st);
    if (rc) return throw_pie(env, ctx, rc);
    napi_value o, v;
    napi_create_object(env, &o);
#define PUT(name, val)                      \
    napi_create_double(env, (double)(val), &v); \
    napi_set_named_property(env, o, name, v);
    PUT("rows", st.rows) PUT("users", st.users) PUT("selected", st.selected) PUT("algBytes", st.alg_bytes)
    PUT("k1MsSum", st.k1_ms_sum) PUT("scanMsSum", st.scan_ms_sum) PUT("nProfiled", st.n_profiled)
    PUT("maxBucket", st.max_bucket) PUT("k1Variant", st.k1_variant)
#undef PUT
    return o;
}

static napi_value fn_set_profiling(napi_env env, napi_callback_info info)
{
    ARGS(2)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    bool on = false;
    CHECK(env, napi_get_value_bool(env, argv[1], &on));
    p_pie_set_profiling(ctx, on ? 1 : 0);
    if (!on) p_pie_stats_reset(ctx);
    return js_int(env, 0);
}

/* setOrderedRun(ctx, mode): 0 never, 1 where the general path is weak (default), 2 always (pie_set_ordered_run) */
static napi_value fn_set_ordered_run(napi_env env, napi_callback_info info)
{
    ARGS(2)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t mode = 1;
    CHECK(env, napi_get_value_int32(env, argv[1], &mode));
    const int rc = p_pie_set_ordered_run(ctx, mode);
    if (rc != 0) return throw_pie(env, ctx, rc);
    return js_int(env, 0);
}

/* setBatchLanes(ctx, n): lanes of the batched scan — independent streams whose batches run side by side — 1..4, 0 = by table
 * size (pie_set_batch_lanes) -> lanes in use now */
static napi_value fn_set_batch_lanes(napi_env env, napi_callback_info info)
{
    ARGS(2)
    pie_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t n = 0;
    CHECK(env, napi_get_value_int32(env, argv[1], &n));
    const int rc = p_pie_set_batch_lanes(ctx, n);
    if (rc != 0) return throw_pie(env, ctx, rc);
    return js_int(env, p_pie_batch_lanes(ctx));
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct {
        const char *name;
        napi_callback fn;
    } table[] = {
        {"open", fn_open}, {"deviceCount", fn_device_count}, {"ctxCreate", fn_ctx_create}, {"ctxDestroy", fn_ctx_destroy},
        {"loadColumns", fn_load_columns}, {"appendRows", fn_append_rows}, {"genSynthetic", fn_gen},
        {"readColumns", fn_read_columns}, {"saveColumns", fn_save_columns}, {"loadColumnsDir", fn_load_columns_dir}, {"setEnd", fn_set_end}, {"deleteUser", fn_delete_user}, {"retentionPurgeTz", fn_retention_purge_tz},
        {"setDisciplines", fn_set_disc}, {"scan", fn_scan}, {"scanDevice", fn_scan_device}, {"userFeed", fn_user_feed}, {"scanAsync", fn_scan_async}, {"fetchRows", fn_fetch_rows},
        {"expiredQueue", fn_expired_queue}, {"archiveQueue", fn_archive_queue}, {"serializeEvents", fn_serialize_events}, {"serializeICal", fn_serialize_ical}, {"stats", fn_stats}, {"setProfiling", fn_set_profiling},
        {"setOrderedRun", fn_set_ordered_run}, {"setBatchLanes", fn_set_batch_lanes},
        {"serializeCsv", fn_serialize_csv}, {"scanBatch", fn_scan_batch}, {"batchUserFeed", fn_batch_user_feed}, {"batchFetchRequests", fn_batch_fetch_requests},
        {"commCreate", fn_comm_create}, {"commDestroy", fn_comm_destroy}, {"commWorld", fn_comm_world}, {"commCtx", fn_comm_ctx},
        {"commGenSyntheticSharded", fn_comm_gen}, {"commScanBatchGather", fn_comm_scan_gather}, {"commReadGathered", fn_comm_read}, {"commUPad", fn_comm_upad},
        {"commNeededCap", fn_comm_needed_cap}, {"commStepReserve", fn_comm_step_reserve}, {"commStepBegin", fn_comm_step_begin},
        {"commStepFinish", fn_comm_step_finish}, {"commStepCollect", fn_comm_step_collect}, {"commStepReadGathered", fn_comm_step_read},
        {"commStepUPad", fn_comm_step_upad},
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i) {
        napi_value fn;
        if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &fn) != napi_ok) return NULL;
        napi_set_named_property(env, exports, table[i].name, fn);
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
