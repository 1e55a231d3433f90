import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle of the parity tests is mostly small fp32 operators: on the GPU box's 256 logical cores torch's default (one thread per
    # physical core) spends its time in thread hand-offs — the toy-geometry fixture run took 124 s there against 0.2 s on 8 cores.
    import torch
    if (os.cpu_count() or 1) > 32 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(32)


@pytest.fixture(scope="session")
def hip_lib():
    """libaether_hip.so built in-tree (fails loudly if absent: the GPU tests must exercise native code)."""
    from aether_amd import _lib
    from aether_amd.build import build_native

    if not _lib.lib_path().exists():
        build_native()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from aether_amd import _lib

    _lib.check(_lib.load().aether_check_device(), "aether_check_device")
    return torch.device("cuda:0")


# ---- the seeded full-size modules (42-block DiT, real-width VAE) shared by tests/test_fullsize_*_gpu.py: built once per session (64 s) ----
@pytest.fixture(scope="session")
def fullsize_modules(cuda, hip_lib):
    import gc
    import time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fullsize_cases as fc
    from aether_amd.transformer import AetherTransformer3D
    from aether_amd.vae import AetherVAE
    t0 = time.perf_counter()
    oracle, cfg = fc.build_oracle_dit()
    sd = fc.bf16_state_dict(oracle)
    del oracle
    gc.collect()
    dit = AetherTransformer3D({k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, device=cuda).load_state_dict(sd)
    del sd
    gc.collect()
    vae = AetherVAE(dict(fc.VAE_KW), device=cuda).load_state_dict(fc.bf16_state_dict(fc.build_oracle_vae()))
    vae.enable_tiling()
    vae.enable_slicing()
    # what a pipeline call does first (AetherV1PipelineCogVideoX.__call__ -> AetherVAE.reserve_workspace): one workspace for every task at this geometry,
    # so that no test — whichever runs first, pipeline call or bare vae.encode / decode — re-allocates it or drops a captured hipGraph
    vae.reserve_workspace(fc.FRAMES, fc.HEIGHT, fc.WIDTH)
    print(f"\n[fullsize] {cfg.num_layers}-block seeded weights built on the host and packed on the device in {time.perf_counter() - t0:.0f} s")
    return dit, vae
