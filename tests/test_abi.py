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
    # undefined flag bits (e.g. the round-2/3 kernel-variant flags 4 | 512 that no longer exist) are refused, not ignored
    import ctypes
    from aether_amd import _lib
    cfg = _lib.AetherDitConfig(num_layers=1, num_heads=8, head_dim=64, in_channels=16, out_channels=8, patch_size=2, text_dim=128, time_embed_dim=64,
                               ff_mult=4, max_text_len=20, norm_eps=1e-5, qk_norm_eps=1e-6, use_pos_embedding=0, flags=1 | 4 | 512)
    assert hip_lib.aether_dit_create(ctypes.byref(cfg)) is None and b"undefined flag bits" in hip_lib.aether_last_error()
    cfg.flags = _lib.AETHER_GEMM_WIDE_STORE
    h = hip_lib.aether_dit_create(ctypes.byref(cfg))
    assert h is not None
    assert hip_lib.aether_dit_set_flags(h, 2) == -1 and b"undefined flag bits" in hip_lib.aether_last_error()
    assert hip_lib.aether_dit_set_flags(h, _lib.AETHER_GEMM_WIDE_STORE | _lib.AETHER_ATTN_EXACT_MAX) == 0
    hip_lib.aether_dit_destroy(h)
    vcfg = _lib.AetherVaeConfig(flags=_lib.AETHER_GEMM_WIDE_STORE | _lib.AETHER_CONV_TAP_REUSE, num_levels=4, latent_channels=16, layers_per_block=3)
    assert hip_lib.aether_vae_create(ctypes.byref(vcfg)) is None and b"undefined flag bits" in hip_lib.aether_last_error()


def test_product_path_never_imports_oracle():
    bad = []
    for sub in ("aether_amd", "aether", "scripts"):
        for f in (ROOT / sub).rglob("*.py") if (ROOT / sub).exists() else []:
            src = f.read_text()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                bad.append(str(f))
    assert not bad, f"product files import the oracle: {bad}"
