"""CPU: the C-ABI library builds for gfx950, loads without a GPU, and exports every symbol that
include/pie_scan.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def header_symbols():
    text = open(os.path.join(REPO, "include", "pie_scan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pie_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(pie):
    assert header_symbols() == sorted(pie.ABI_SYMBOLS)


def test_library_exports_every_symbol(pie):
    lib = pie.load_library()
    for name in header_symbols():
        assert getattr(lib, name) is not None
    assert lib.pie_abi_version() == 1


def test_fails_loudly_without_gpu(pie):
    lib = pie.load_library()
    if lib.pie_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pie.PieError) as ei:
        pie.PieScan(0)
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)


def test_missing_library_is_an_error_not_a_fallback(pie, tmp_path):
    from sph_pie_amd import binding
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        binding.load_library(str(tmp_path / "libpie_hip.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "sph-pie_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".js", ".hip", ".h", ".c", ".cpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_py" not in text and "pie_oracle_" not in text, f
                if f != "build.py":  # build.py may BUILD the checker (make -C oracle); nothing may load it
                    assert "libpie_oracle" not in text, f


def test_shard_rule_matches_oracle(pie, oracle):
    for g in (1, 2, 4, 8):
        for u in list(range(50)) + [99999, 2 ** 31 - 1]:
            assert pie.shard_of(u, g) == oracle.shard_of(u, g)
    hist = [0] * 8
    for u in range(8000):
        hist[pie.shard_of(u, 8)] += 1
    assert min(hist) > 800  # roughly even
