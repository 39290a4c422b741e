"""Build recipes for the native pieces (gfx950 only).  Everything is built in-tree so the .so files travel
to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
HIP_LIB = os.path.join(PKG_DIR, "libpie_hip.so")
NAPI_ADDON = os.path.join(PKG_DIR, "host", "pie_napi.node")
ORACLE_LIB = os.path.join(REPO, "oracle", "libpie_oracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd, cwd=None):
    res = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), res.stdout))
    return res.stdout


def build_hip(force=False):
    """hipcc --offload-arch=gfx950 -> sph-pie_amd/libpie_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, "pie_scan.hip"), os.path.join(CSRC, "pie_kernels.h"),
            os.path.join(REPO, "include", "pie_scan.h")]
    if not force and _newer(HIP_LIB, srcs):
        return HIP_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
          "-o", HIP_LIB, os.path.join(CSRC, "pie_scan.hip")])
    return HIP_LIB


def build_oracle(force=False):
    """gcc -> oracle/libpie_oracle.so (test infrastructure only)."""
    odir = os.path.join(REPO, "oracle")
    srcs = [os.path.join(odir, "pie_oracle.c"), os.path.join(odir, "pie_oracle.h")]
    if not force and _newer(ORACLE_LIB, srcs):
        return ORACLE_LIB
    _run(["make", "-C", odir, "-B", "libpie_oracle.so"])
    return ORACLE_LIB


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
    if not force and _newer(NAPI_ADDON, [src, os.path.join(REPO, "include", "pie_scan.h")]):
        return NAPI_ADDON
    _run(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", "-I", inc, "-I", os.path.join(REPO, "include"),
          "-o", NAPI_ADDON, src, "-ldl"])
    return NAPI_ADDON


def build_all(force=False):
    out = {"hip": build_hip(force), "oracle": build_oracle(force)}
    out["napi"] = build_napi(force)
    return out
