/*
 * TEST INFRASTRUCTURE — sanitizer self-test of the CPU oracle (GPU AddressSanitizer is not available on the pool, so
 * the native CPU code is what ASan/UBSan can cover).  Built by `make -C oracle selftest_asan` with
 * -fsanitize=address,undefined and run by tests/test_oracle_golden.py::test_oracle_under_sanitizers.
 */
#include "pie_oracle.h"

#include <stdio.h>
#include <stdlib.h>

static int check(int cond, const char *what)
{
    if (!cond) fprintf(stderr, "selftest FAILED: %s\n", what);
    return cond ? 0 : 1;
}

int main(void)
{
    int bad = 0;
    const int sizes[] = {0, 1, 7, 8, 9, 63, 64, 65, 1000, 4097, 50003};
    for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; ++si) {
        const int n = sizes[si];
        const int U = n < 10 ? 1 : n / 7 + 1, D = 32;
        int64_t *s = malloc((size_t)(n + 1) * 8), *e = malloc((size_t)(n + 1) * 8), *off = malloc((size_t)(U + 1) * 8);
        int32_t *u = malloc((size_t)(n + 1) * 4), *d = malloc((size_t)(n + 1) * 4), *cnt = malloc((size_t)U * 4);
        int32_t *idx = malloc((size_t)(n + 1) * 4), *q = malloc((size_t)(n + 1) * 4);
        for (unsigned flags = 0; flags < 4; ++flags) {
            pie_oracle_gen(0x5EED5EEDULL, n, 0, n, U, D, flags, s, e, u, d);
            const int64_t nows[] = {INT64_MIN, PIE_ORACLE_T0_MS - 6 * 3600 * 1000LL, INT64_MAX};
            for (int k = 0; k < 3; ++k) {
                size_t m = 0, qn = 0;
                int rc = pie_oracle_scan(s, e, u, d, (size_t)n, U, nows[k], PIE_ORACLE_T0_MS - 61LL * 86400000LL,
                                         0x5555555555555555ULL, cnt, off, idx, (size_t)n, &m);
                bad += check(rc == 0 && off[U] == (int64_t)m, "scan rc / offsets");
                for (size_t j = 1; j < m; ++j) {
                    const int a = idx[j - 1], b = idx[j];
                    if (u[a] == u[b]) bad += check(s[a] < s[b] || (s[a] == s[b] && a < b), "bucket order");
                    else bad += check(u[a] < u[b], "user order");
                }
                rc = pie_oracle_expired_queue(e, (size_t)n, INT64_MIN, nows[k], q, (size_t)n, &qn);
                bad += check(rc == 0 && qn <= (size_t)n, "expired queue");
                if (n > 4) { /* capacity error path */
                    rc = pie_oracle_scan(s, e, u, d, (size_t)n, U, INT64_MIN, INT64_MIN, ~0ULL, cnt, off, idx, 2, &m);
                    bad += check(rc == -1 && m == (size_t)n, "capacity path");
                }
            }
        }
        if (n > 0) { /* out-of-range user id is reported, not dereferenced */
            u[n / 2] = U;
            size_t m = 0;
            bad += check(pie_oracle_scan(s, e, u, d, (size_t)n, U, INT64_MIN, INT64_MIN, ~0ULL, cnt, off, idx, (size_t)n, &m) == -2,
                         "bad user id");
        }
        free(s); free(e); free(off); free(u); free(d); free(cnt); free(idx); free(q);
    }
    for (int g = 1; g <= 8; ++g)
        for (int user = 0; user < 1000; ++user) bad += check(pie_oracle_shard_of(user, g) >= 0 && pie_oracle_shard_of(user, g) < g, "shard range");
    if (bad) return 1;
    printf("oracle selftest ok\n");
    return 0;
}
