"""sph-pie_amd — MI355X-native session-scan -> per-user feed path (drop-in for the path named in
BASELINE.json's north_star; nothing else of sph-pie is rebuilt here).

  csrc/      HIP kernels for gfx950 + the C ABI of include/pie_scan.h   -> libpie_hip.so
  host/      Node.js host mirror of the reference modules over a raw N-API addon
  binding.py ctypes binding of the same C ABI (used by tests, bench.py and the multi-GPU driver)
  shard.py   user-hash sharding + all-gather of per-user feeds (torch.distributed; nccl == RCCL on ROCm)
"""
from .binding import (ABI_SYMBOLS, PIE_END_NONE, PIE_GEN_CLUSTERED, PIE_GEN_INTERVAL, PieError, PieScan,
                      load_library, shard_of)
from .build import build_all, build_hip, build_napi, build_oracle

__all__ = ["ABI_SYMBOLS", "PIE_END_NONE", "PIE_GEN_CLUSTERED", "PIE_GEN_INTERVAL", "PieError", "PieScan",
           "load_library", "shard_of", "build_all", "build_hip", "build_napi", "build_oracle"]
