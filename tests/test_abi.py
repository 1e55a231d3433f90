"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and exports every
symbol include/aether_hip.h declares (no compute call is made: there is no GPU in the build container)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "aether_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text.split("/* internal helpers")[0] if "/* internal helpers" in text else text
    text = text.split("#ifdef __cplusplus\nextern \"C\" int aether_set_error")[0]
    return sorted(set(re.findall(r"\b(aether_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hip_lib):
    from aether_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/aether_hip.h but not exported"
    assert set(names) - {"aether_set_error", "aether_check_launch"} <= set(_lib.SIGNATURES), "ctypes table out of date"


def test_error_reporting_without_gpu(hip_lib):
    """Argument validation happens on the host before any launch, so it is testable without a device."""
    assert hip_lib.aether_version() >= 100
    rc = hip_lib.aether_gemm_bf16(4096, 8, 4096, 8, 4096, 8, 16, 32, 100, None, 0, None, 0, None, None, 0, 0, 0, None, 0, 0, None)
    assert rc == -2 and b"multiple of 64" in hip_lib.aether_last_error()
    rc = hip_lib.aether_layernorm_modulate(16, 8, 16, 8, 4, 300, 1e-5, None, None, None, None, None, None, 0, 0, 0, None)
    assert rc == -2
    rc = hip_lib.aether_flash_attn_fwd(4096, 4096, 4096, 4096, 1, 2, 100, 100, 0, None)      # Spad not a multiple of 64
    assert rc == -2 and b"Spad" in hip_lib.aether_last_error()
    rc = hip_lib.aether_conv_gemm_bf16(4096, 1, 3, 4, 4, 64, 1, 2, 2, 3, 4096, 27, 4096, 64, 4096, 64, None, None, 0, None, 0, 0, None)
    assert rc == -2                                                                                   # stride must be 1 or 2
    assert hip_lib.aether_dit_create(None) is None


def test_product_path_never_imports_oracle():
    bad = []
    for sub in ("aether_amd", "aether", "scripts"):
        for f in (ROOT / sub).rglob("*.py") if (ROOT / sub).exists() else []:
            src = f.read_text()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                bad.append(str(f))
    assert not bad, f"product files import the oracle: {bad}"
