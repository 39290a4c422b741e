import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import oracle_py as oracle
import torch  # noqa
import sph_pie_amd as pie
os.environ["PIE_K1_VARIANT"] = sys.argv[1]
n, U, D, flags = int(sys.argv[2]), int(sys.argv[3]), 1, 3
cols = oracle.gen(12345, n, 0, n, U, D, flags)
ctx = pie.PieScan(0)
try:
    ctx.load_columns(*cols, U)
    ctx.set_disciplines(1, D)
    T0 = oracle.T0_MS
    for now, cutoff in [(T0 - 6 * 3600 * 1000, -(2 ** 63)), (T0 - 87 * 86400000, T0 - 70 * 86400000), (-(2 ** 63), -(2 ** 63)), (T0 - 6 * 3600 * 1000, -(2 ** 63))]:
        t0 = time.time()
        got = ctx.scan(now, cutoff)
        want = oracle.scan(*cols, U, now, cutoff, 1)
        print(sys.argv[1], 'now', now, 'M', got[2].size, 'ok', all(np.array_equal(a, b) for a, b in zip(got, want)), '%.3f s' % (time.time() - t0), ctx.stats()['max_bucket'], flush=True)
except BaseException as ex:
    print('EXC', repr(ex), flush=True)
    os._exit(1)
os._exit(0)
