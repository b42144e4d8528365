"""The C-ABI library loads and exports exactly what include/flowcon_hip.h declares (CPU-only)."""
import ctypes
import os
import re

from flowconductor_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "flowcon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\bint\s+(fc_[a-z0-9_]+)\s*\(", text))


def test_library_is_built():
    assert _hip.is_built(), "run `python -c 'import __graft_entry__ as g; g.build()'` first"


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared()
    assert declared, "no declarations parsed from include/flowcon_hip.h"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libflowcon_hip.so does not export %s" % name
    assert declared == set(_hip.SIGNATURES), (
        "header vs ctypes binding mismatch: %s" % sorted(declared ^ set(_hip.SIGNATURES)))


def test_abi_version():
    lib = _hip.load()
    assert lib.fc_abi_version() == _hip.ABI_VERSION


def test_rq_config_layout_matches_header():
    # 4 int32 + 4 float + 3 double + 4 float, naturally aligned
    assert ctypes.sizeof(_hip.RQConfig) == 16 + 16 + 24 + 16
    assert _hip.RQConfig.min_bin_width.offset == 32
    assert _hip.RQConfig.wh_divisor.offset == 56
