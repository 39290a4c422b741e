"""Build recipes for the native pieces (gfx950 only).  Everything is built in-tree so the .so files travel
to the GPU box with the repo snapshot."""
import fcntl
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
HIP_LIB = os.path.join(PKG_DIR, "libpie_hip.so")
UBENCH_LIB = os.path.join(PKG_DIR, "libpie_ubench.so")
NAPI_ADDON = os.path.join(PKG_DIR, "host", "pie_napi.node")
ORACLE_LIB = os.path.join(REPO, "oracle", "libpie_oracle.so")


def _digest(sources):
    h = hashlib.sha256()
    for path in sources:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _newer(target, sources):
    """Is `target` built from exactly these sources?  Judged by a digest of the sources stored beside the target (copies
    of the tree — the GPU box gets one — do not keep modification times in order); by modification time without one."""
    if not os.path.exists(target):
        return False
    side = target + ".srchash"
    if os.path.exists(side):
        with open(side) as f:
            return f.read().strip() == _digest(sources)
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _build(target, sources, make_cmd, force):
    """Build `target` unless it is fresh.  Safe when several processes ask at once (one rank per GPU does): an exclusive
    lock serialises them, whoever comes second finds the target fresh, and the output appears by an atomic rename."""
    if not force and _newer(target, sources):
        return target
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _newer(target, sources):
            return target
        tmp = "%s.tmp.%d" % (target, os.getpid())
        try:
            _run(make_cmd(tmp))
            os.replace(tmp, target)
            with open(target + ".srchash", "w") as f:
                f.write(_digest(sources))
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return target


def _run(cmd, cwd=None):
    res = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), res.stdout))
    return res.stdout


def build_hip(force=False):
    """hipcc --offload-arch=gfx950 -> sph-pie_amd/libpie_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "pie_scan.hip"), os.path.join(CSRC, "pie_comm.hip"), os.path.join(CSRC, "pie_kernels.h"),
            os.path.join(CSRC, "pie_ordered.h"), os.path.join(REPO, "include", "pie_scan.h")]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # pie_comm.hip opens RCCL with dlopen at run time (no link-time dependency: the scan library loads without RCCL)
    return _build(HIP_LIB, srcs, lambda out: [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
                                              "-o", out, os.path.join(CSRC, "pie_scan.hip"), os.path.join(CSRC, "pie_comm.hip"),
                                              "-ldl"], force)


def build_ubench(force=False):
    """hipcc --offload-arch=gfx950 -> sph-pie_amd/libpie_ubench.so: the streaming-read ceiling bench.py measures beside the scan
    (a measurement aid with its own shared object: the scan library's source digest does not move with it)."""
    src = os.path.join(CSRC, "pie_ubench.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return _build(UBENCH_LIB, [src], lambda out: [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, src], force)


def build_oracle(force=False):
    """gcc -> oracle/libpie_oracle.so (test infrastructure only)."""
    odir = os.path.join(REPO, "oracle")
    srcs = [os.path.join(odir, "pie_oracle.c"), os.path.join(odir, "pie_oracle.h")]
    return _build(ORACLE_LIB, srcs, lambda out: ["make", "-C", odir, "-B", "libpie_oracle.so", "OUT=" + out], force)


def build_napi(force=False):
    """g++ -> sph-pie_amd/host/pie_napi.node (raw N-API shim over the C ABI).  Needs node's headers."""
    src = os.path.join(CSRC, "pie_napi.c")
    if not os.path.exists(src):
        return None
    inc = None
    for cand in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(cand, "node_api.h")):
            inc = cand
            break
    if inc is None:
        return None
    return _build(NAPI_ADDON, [src, os.path.join(REPO, "include", "pie_scan.h")],
                  lambda out: ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", "-I", inc, "-I", os.path.join(REPO, "include"),
                               "-o", out, src, "-ldl"], force)


def build_all(force=False):
    out = {"hip": build_hip(force), "ubench": build_ubench(force), "oracle": build_oracle(force)}
    out["napi"] = build_napi(force)
    return out
