"""The C++ VAE launch plan (csrc/vae_plan.hip) without a GPU: its sizing pass walks the whole encoder / decoder graph (every
registered convolution and norm is looked up, every tile batch, frame chunk and cache is laid out in the arena) and launches
nothing, so the walk itself, the output geometry and the error paths are testable on CPU."""
import ctypes as C

import pytest
import torch


@pytest.fixture(scope="module")
def vae():
    from aether_amd.vae import AetherVAE
    return AetherVAE(device="cpu").init_random_weights(0)        # host tensors: pointers are registered, never dereferenced here


def _shape(vae, decode, T, H, W):
    shp = [C.c_int() for _ in range(4)]
    assert vae._lib.aether_vae_output_shape(vae._handle, decode, T, H, W, *[C.byref(v) for v in shp]) == 0
    return tuple(v.value for v in shp)


def test_baseline_geometry_plan(vae):
    L = vae._lib
    enc = L.aether_vae_workspace_bytes(vae._handle, 0, 41, 480, 720, 1)
    dec = L.aether_vae_workspace_bytes(vae._handle, 1, 11, 60, 90, 1)
    assert 4 << 30 < enc < 12 << 30 and 8 << 30 < dec < 20 << 30, (enc, dec)         # 9.0 / 15.1 GiB (round 3: 13.6 / 28.9 with the volume pool)
    assert _shape(vae, 0, 41, 480, 720) == (32, 11, 60, 90) and _shape(vae, 1, 11, 60, 90) == (3, 41, 480, 720)
    assert _shape(vae, 0, 1, 480, 720) == (32, 1, 60, 90)                             # a single conditioning image (P:554-569)


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 7, 9, 11, 13])
def test_frame_counts_follow_the_chunk_rules(vae, T):
    """Decoder: chunks of 2 latent frames, the remainder joins chunk 0, every temporal up-sampler doubles a chunk's frames except
    the first frame of an odd chunk (so 4 latent frames -> 16 frames but 5 -> 17); encoder likewise with 8-frame chunks."""
    from oracle.vae import OracleVAE

    def frames(decode, n):
        tot = 0
        for s, e in OracleVAE._chunks(n, 2 if decode else 8):
            t = min(e, n) - s
            for _ in range(2):
                t = (t if t == 1 else (2 * t - 1 if t % 2 else 2 * t)) if decode else (t // 2 + 1 if t % 2 else t // 2)
            tot += t
        return tot
    assert _shape(vae, 1, T, 60, 90)[1] == frames(True, T)
    assert _shape(vae, 0, 4 * (T - 1) + 1, 480, 720)[1] == frames(False, 4 * (T - 1) + 1) == T


def test_missing_weight_is_reported():
    from aether_amd import _lib
    L = _lib.load()
    cfg = _lib.AetherVaeConfig(in_channels=3, out_channels=3, latent_channels=16, layers_per_block=3, num_levels=4, norm_num_groups=32,
                               temporal_compression_ratio=4, sample_height=480, sample_width=720, norm_eps=1e-6, tap_reuse_max_waste=1.06, flags=1)
    h = L.aether_vae_create(C.byref(cfg))
    assert L.aether_vae_workspace_bytes(h, 0, 41, 480, 720, 1) == 0
    assert b"convolution not registered: encoder.conv_" in L.aether_last_error()
    L.aether_vae_destroy(h)
