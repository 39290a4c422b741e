#!/usr/bin/env python3
"""The streaming-read ceiling of this GPU (libpie_ubench.so; what bench.py reports as roofline.full_read.read_ceiling_gbs): a kernel
that only reads 2.4 GB, as one stream (eight grid / split / unroll forms) and as the table's four columns (four forms); median of
`reps` launches each.  usage: read_ceiling.py [bytes] [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sph_pie_amd as pie

nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ub = ctypes.CDLL(pie.build_ubench())
ub.pie_ubench_read_bw.restype = ctypes.c_int
ub.pie_ubench_read_bw.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
ms = (ctypes.c_double * 12)()
rc = ub.pie_ubench_read_bw(0, nbytes, reps, ms)
if rc:
    raise SystemExit("pie_ubench_read_bw failed: %d" % rc)
names = ["one stream, grid-stride, 16 blocks/CU, unroll 4", "one stream, grid-stride, 16 blocks/CU, unroll 8", "one stream, grid-stride, 48 blocks/CU, unroll 4",
         "one stream, grid-stride, 48 blocks/CU, unroll 8", "one stream, contiguous ranges, 16 blocks/CU, unroll 4", "one stream, contiguous ranges, 16 blocks/CU, unroll 8",
         "one stream, contiguous ranges, 48 blocks/CU, unroll 4", "one stream, contiguous ranges, 48 blocks/CU, unroll 8",
         "four columns (8+8+4+4 B/row), 16 blocks/CU, unroll 2", "four columns, 16 blocks/CU, unroll 4", "four columns, 48 blocks/CU, unroll 2",
         "four columns, 48 blocks/CU, unroll 4"]
for n, m in zip(names, ms):
    print("%-60s %.4f ms  %7.0f GB/s  %.3f of 8 TB/s" % (n, m, nbytes / (m * 1e-3) / 1e9, nbytes / (m * 1e-3) / 1e9 / 8000))

ub.pie_ubench_like_scan.restype = ctypes.c_int
ub.pie_ubench_like_scan.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
ms6 = (ctypes.c_double * 6)()
rows = nbytes // 24
rc = ub.pie_ubench_like_scan(0, rows, reps, ms6)
if rc:
    raise SystemExit("pie_ubench_like_scan failed: %d" % rc)
print("the four-column read dressed as the scan's streaming form (blocks of contiguous rows, four waves taking 512-row tiles in turn):")
for n, m in zip(["split only, 12 288 blocks", "+ block barriers and 10 KB LDS", "+ predicate, ballots, neighbour statistic (no atomics)", "... 6 144 blocks", "... 24 576 blocks",
                 "... 12 288 blocks, tiles of 256 rows (unroll 2)"], ms6):
    print("%-60s %.4f ms  %7.0f GB/s  %.3f of 8 TB/s" % (n, m, rows * 24 / (m * 1e-3) / 1e9, rows * 24 / (m * 1e-3) / 1e9 / 8000))
