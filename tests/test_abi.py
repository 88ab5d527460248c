"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports
exactly what include/pgd_amd.h declares.  No compute call is made (no GPU here)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "pgd_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgd_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from pgdrome_amd import build, _lib
    build.build(verbose=False)
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in pgd_amd.h but not exported"


def test_binding_covers_the_header_exactly():
    from pgdrome_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse to create a context."""
    from pgdrome_amd import _lib
    lib = _lib.load()
    if lib.pgd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def test_invalid_handles_return_error_codes():
    from pgdrome_amd import _lib
    lib = _lib.load()
    assert lib.pgd_sync(12345) == -1          # PGD_ERR_INVALID, no crash
    assert lib.pgd_ctx_destroy(0) == -1
    assert lib.pgd_version() == 100
