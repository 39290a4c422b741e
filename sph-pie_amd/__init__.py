"""sph-pie_amd — MI355X-native session-scan -> per-user feed path (drop-in for the path named in
BASELINE.json's north_star; nothing else of sph-pie is rebuilt here).

  csrc/      HIP kernels for gfx950 + the C ABI of include/pie_scan.h   -> libpie_hip.so
  host/      Node.js host mirror of the reference modules over a raw N-API addon
  binding.py ctypes binding of the same C ABI (used by tests, bench.py and the multi-GPU driver)
  shard.py   user-hash sharding + all-gather of per-user feeds (torch.distributed; nccl == RCCL on ROCm)
"""
from .binding import (ABI_SYMBOLS, PIE_BATCH_MAX, PIE_END_NONE, PIE_GEN_CLUSTERED, PIE_GEN_INTERVAL, PIE_GEN_TIME_ORDERED, PieComm, PieError, PieScan,
                      load_library, shard_of, tz_table)
from .build import UBENCH_LIB, build_all, build_hip, build_napi, build_oracle, build_ubench


def zipf_cdf(n_users, exponent=1.1):
    """uint64 thresholds floor(CDF_k * 2^64) of Zipf(exponent) over n_users ranks (for gen_synthetic_cdf)."""
    import numpy as np
    w = 1.0 / np.arange(1, n_users + 1, dtype=np.float64) ** exponent
    c = np.cumsum(w) / w.sum()
    thr = np.minimum(np.floor(c * 2.0 ** 64), 2.0 ** 64 - 2048).astype(np.uint64)
    thr[-1] = np.uint64(2 ** 64 - 1)
    return np.maximum.accumulate(thr)

__all__ = ["ABI_SYMBOLS", "PIE_BATCH_MAX", "PIE_END_NONE", "PIE_GEN_CLUSTERED", "PIE_GEN_INTERVAL", "PIE_GEN_TIME_ORDERED", "PieComm", "PieError", "PieScan",
           "load_library", "shard_of", "tz_table", "zipf_cdf", "build_all", "build_hip", "build_napi", "build_oracle", "build_ubench", "UBENCH_LIB"]
