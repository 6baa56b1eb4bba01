"""CPU-side checks of the boundary: the shared library builds, loads and exports exactly what
include/jchemo_hip.h declares; the host mirror fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "jchemo_hip.h")).read()
    return sorted(set(re.findall(r"JCH_API[^;]*?\b(jch_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import jchemo_hip as J
    if not os.path.exists(J.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = J.load()
    names = _declared()
    assert len(names) >= 14
    assert sorted(J.SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.jch_version() == 108


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import jchemo_hip as J
    with pytest.raises(J.JchError) as ei:
        J.plskern(np.zeros((10, 3)), np.zeros((10, 1)), nlv=1)
    assert ei.value.code == J._lib.JCH_ENODEV


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "jchemo.jl_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|oracle\.|libplsr_oracle|plsr_oracle", src), os.path.join(d, f)
