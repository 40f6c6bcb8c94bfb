"""CPU-only: the C-ABI library builds, loads and exports every symbol include/fmk.h declares,
and refuses to work (loudly) without a GPU instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="fmk.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fmk_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for s in ["fmk_time_bar_indexer_dev", "fmk_tick_bar_indexer_dev", "fmk_volume_bar_indexer_dev",
              "fmk_dollar_bar_indexer_dev", "fmk_comp_bar_ohlcv_dev", "fmk_comp_bar_directional_dev",
              "fmk_comp_bar_footprints_size_dev", "fmk_comp_bar_footprints_fill_dev",
              "fmk_comp_lagged_returns_dev", "fmk_ewmst_dev"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    from finmlkit_amd import _ffi
    lib = _ffi.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/fmk.h but not exported: {missing}"
    assert lib.fmk_abi_version() == 1


def test_diagnostics_are_not_in_the_drop_in_abi():
    """include/fmk.h declares no probe / counter; include/fmk_diag.h declares them all, the probes are exported by libfmk_diag.so
    (and NOT by the product library), the counters of the last call by libfmk_hip.so."""
    from finmlkit_amd import _ffi
    assert not [s for s in declared_symbols() if s.startswith("fmk_diag_")]
    diag = [s for s in declared_symbols("fmk_diag.h") if s.startswith("fmk_diag_")]
    assert len(diag) >= 9
    for s in diag:
        if s in _ffi.DIAG_PROBES:
            assert hasattr(_ffi.diag_lib(), s), s
            assert not hasattr(_ffi.lib(), s), f"{s} is still exported by the product library"
        else:
            assert hasattr(_ffi.lib(), s), s


def test_no_cpu_fallback_without_device():
    from finmlkit_amd import _ffi
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.FmkError):
        _ffi.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no module of the product package may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "finmlkit_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "libfmk_oracle" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
