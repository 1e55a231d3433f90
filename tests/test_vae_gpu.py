"""VAE parity on MI355X: each HIP kernel against an fp32 PyTorch evaluation of the same operator, then the whole native
encoder / decoder (tiled + frame-batched, as the reference runs it, D:229-230) against the fp32 CPU oracle with a
tolerance calibrated by the bf16 run of the oracle itself."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _vae(cuda):
    from aether_amd.vae import AetherVAE
    return AetherVAE({"block_out_channels": (64, 128, 128, 128), "layers_per_block": 1, "sample_height": 96, "sample_width": 240}, device=cuda)


@pytest.mark.parametrize("cin,cout,kt", [(64, 256, 3), (128, 128, 3), (64, 32, 3), (64, 64, 1)])
def test_conv_gemm_vs_conv3d(cuda, hip_lib, cin, cout, kt):
    """3x3x3 causal conv (kt=3) and 3x3 conv2d (kt=1) on a zero-bordered channels-last volume."""
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(cin + cout)
    NB, T, H, W = 2, 3, 10, 13
    x = torch.randn(NB, cin, T + kt - 1, H, W, generator=g).to(torch.bfloat16)         # already causally padded in time
    w = (torch.randn(cout, cin, kt, 3, 3, generator=g) / (cin * 9 * kt) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, generator=g)
    res = torch.randn(NB, T, H, W, cout, generator=g).to(torch.bfloat16)
    ref = F.conv3d(x.float(), w.float(), b, padding=(0, 1, 1)).permute(0, 2, 3, 4, 1) + res.float()
    vae = _vae(cuda)
    conv = _Conv(w if kt == 3 else w[:, :, 0], b, cuda)
    vol = torch.zeros(NB, T + kt - 1, H + 2, W + 2, cin, dtype=torch.bfloat16, device=cuda)
    vol[:, :, 1:-1, 1:-1] = x.permute(0, 2, 3, 4, 1).to(cuda)
    out = vae._conv(vol, conv, (T, H, W), 1, res.to(cuda))
    torch.cuda.synchronize()
    assert _rel(out[..., :cout].cpu(), ref) < 6e-3


@pytest.mark.parametrize("cin,cout,NB,T,H,W", [(512, 512, 1, 2, 10, 18), (512, 512, 2, 3, 30, 45), (256, 128, 1, 2, 20, 36)])
def test_conv_gemm_split_k(cuda, hip_lib, cin, cout, NB, T, H, W):
    """Deep low-resolution layers (K = 27*512 = 216 K tiles, a handful of output tiles): the split-K path (fp32 partial
    tiles summed in slice order + finalize) against conv3d, against the single-pass kernel, and run-to-run bit-identical."""
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(NB, cin, T + 2, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, generator=g)
    res = torch.randn(NB, T, H, W, cout, generator=g).to(torch.bfloat16)
    ref = F.conv3d(x.float(), w.float(), b, padding=(0, 1, 1)).permute(0, 2, 3, 4, 1) + res.float()
    conv = _Conv(w, b, cuda)
    vol = torch.zeros(NB, T + 2, H + 2, W + 2, cin, dtype=torch.bfloat16, device=cuda)
    vol[:, :, 1:-1, 1:-1] = x.permute(0, 2, 3, 4, 1).to(cuda)
    vae = _vae(cuda)
    out_split = vae._conv(vol, conv, (T, H, W), 1, res.to(cuda))
    out_split2 = vae._conv(vol, conv, (T, H, W), 1, res.to(cuda))
    vae_single = _vae(cuda)
    vae_single.splitk_ws_bytes = 0                      # no workspace -> single-pass kernel
    out_single = vae_single._conv(vol, conv, (T, H, W), 1, res.to(cuda))
    torch.cuda.synchronize()
    assert _rel(out_split.cpu(), ref) < 6e-3 and _rel(out_single.cpu(), ref) < 6e-3
    assert torch.equal(out_split, out_split2)           # deterministic reduction order
    assert _rel(out_split.cpu(), out_single.cpu()) < 4e-3   # same math, different fp32 summation order + one rounding


def test_conv_stride2_vs_conv2d(cuda, hip_lib):
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(9)
    NB, T, H, W, Cc = 1, 2, 12, 16, 64
    x = torch.randn(NB * T, Cc, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) / (Cc * 9) ** 0.5).to(torch.bfloat16)
    b = torch.randn(Cc, generator=g)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=2).reshape(NB, T, Cc, H // 2, W // 2).permute(0, 1, 3, 4, 2)
    vae = _vae(cuda)
    xs = x.reshape(NB, T, Cc, H, W).permute(0, 1, 3, 4, 2).contiguous().to(cuda)
    vol = vae._resample(xs, 0, (NB, T, H + 1, W + 1, Cc), (0, 0, 0))
    out = vae._conv(vol, _Conv(w, b, cuda), (T, H // 2, W // 2), stride=2)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < 6e-3


@pytest.mark.parametrize("T", [1, 2, 5, 8])
def test_resamplers_match_oracle(cuda, hip_lib, T):
    from oracle.vae import Downsample3D, Upsample3D
    g = torch.Generator().manual_seed(T)
    NB, H, W, Cc = 2, 6, 8, 64
    x = torch.randn(NB, Cc, T, H, W, generator=g).to(torch.bfloat16)
    xs = x.permute(0, 2, 3, 4, 1).contiguous().to(cuda)
    vae = _vae(cuda)
    # temporal average pool (the part of CogVideoXDownsample3D before the conv)
    ds = Downsample3D(Cc, True)
    xp = x.float().permute(0, 3, 4, 1, 2).reshape(NB * H * W, Cc, T)
    if T % 2 == 1:
        rest = F.avg_pool1d(xp[..., 1:], 2, 2) if T > 1 else xp[..., 1:]
        ref = torch.cat([xp[..., :1], rest], -1)
    else:
        ref = F.avg_pool1d(xp, 2, 2)
    Tn = ref.shape[-1]
    ref = ref.reshape(NB, H, W, Cc, Tn).permute(0, 4, 1, 2, 3)
    vol = vae._resample(xs, 1, (NB, Tn, H + 1, W + 1, Cc), (0, 0, 0))
    torch.cuda.synchronize()
    assert _rel(vol[:, :, :H, :W].cpu(), ref) < 3e-3 and float(vol[:, :, H].abs().max()) == 0 and float(vol[:, :, :, W].abs().max()) == 0
    # nearest up-sampling with the first-frame rule (the part of CogVideoXUpsample3D before the conv)
    up = Upsample3D(Cc, True)
    up.conv = torch.nn.Identity()
    ref_up = up(x.float()).permute(0, 2, 3, 4, 1)
    Tu = ref_up.shape[1]
    vol = vae._resample(xs, 3, (NB, Tu, 2 * H + 2, 2 * W + 2, Cc), (0, 1, 1))
    torch.cuda.synchronize()
    assert torch.equal(vol[:, :, 1:-1, 1:-1].cpu().float(), ref_up)
    _ = ds


@pytest.mark.parametrize("Cc,T", [(64, 3), (128, 2), (512, 5)])
def test_groupnorm_silu_and_spatial_norm(cuda, hip_lib, Cc, T):
    from aether_amd.vae import _Norm
    from oracle.vae import SpatialNorm3D
    g = torch.Generator().manual_seed(Cc)
    NB, H, W = 2, 8, 12
    x = (torch.randn(NB, Cc, T, H, W, generator=g) * 2 + 0.3).to(torch.bfloat16)
    xs = x.permute(0, 2, 3, 4, 1).contiguous().to(cuda)
    vae = _vae(cuda)
    gn = torch.nn.GroupNorm(32, Cc, eps=1e-6)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.1 * torch.randn(Cc, generator=g)); gn.bias.copy_(0.1 * torch.randn(Cc, generator=g))
    ref = F.silu(gn(x.float())).permute(0, 2, 3, 4, 1)
    n = _Norm({"w.weight": gn.weight, "w.bias": gn.bias}, "w.", cuda, False)
    vol = vae._norm_to_padded(xs, n, 2, 1, True)
    torch.cuda.synchronize()
    assert _rel(vol[:, 2:, 1:-1, 1:-1].cpu(), ref) < 5e-3
    assert float(vol[:, :, 0].abs().max()) == 0 and float(vol[:, :, :, -1].abs().max()) == 0      # borders stay zero
    # SpatialNorm3D with a nearest-resized latent (odd T > 1 exercises the first-frame rule)
    zT = {3: 3, 2: 2, 5: 3}[T]
    zq = torch.randn(NB, 16, zT, H // 4, W // 4, generator=g).to(torch.bfloat16)
    sn = SpatialNorm3D(Cc, 16, 32)
    with torch.no_grad():
        for name, p in sn.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
        sn.norm_layer.weight.add_(1.0)
    ref = F.silu(sn(x.float(), zq.float())).permute(0, 2, 3, 4, 1)
    sd = {"s." + k: v for k, v in sn.state_dict().items()}
    n = _Norm(sd, "s.", cuda, True)
    vol = vae._norm_to_padded(xs, n, 2, 1, True, zq.permute(0, 2, 3, 4, 1).contiguous().to(cuda))
    torch.cuda.synchronize()
    assert _rel(vol[:, 2:, 1:-1, 1:-1].cpu(), ref) < 5e-3


def test_im2col_first_matches_unfold(cuda, hip_lib):
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(4)
    Cc, Tall, Hall, Wall = 3, 7, 20, 24
    x = torch.randn(Cc, Tall, Hall, Wall, generator=g).to(torch.bfloat16)
    w = torch.randn(32, Cc, 3, 3, 3, generator=g).to(torch.bfloat16)
    conv = _Conv(w, torch.zeros(32), cuda, pad_k_to=128)
    vae = _vae(cuda)
    for (t0, T, first) in [(0, 3, True), (3, 4, False)]:
        y0, x0, H, W = 4, 8, 10, 12
        A = vae._im2col(x.to(cuda), conv, [(y0, x0)], t0, T, H, W, first)
        crop = x[:, :, y0:y0 + H, x0:x0 + W].float()
        front = crop[:, :1].repeat(1, 2, 1, 1) if first else crop[:, t0 - 2:t0]
        vol = torch.cat([front, crop[:, t0:t0 + T]], 1)[None]
        ref = F.conv3d(vol, w.float(), padding=(0, 1, 1))[0].permute(1, 2, 3, 0).reshape(T * H * W, 32)
        got = A[0].float().cpu() @ conv.w.float().cpu().t()
        torch.cuda.synchronize()
        assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2)
        assert float(A[0, :, 81:].abs().max()) == 0


def test_im2col_first_wide_untiled_row(cuda, hip_lib):
    """An untiled 1 400-pixel-wide first layer (a 1080p-class encode with tiling off): 9 * 3 * 1 402 source elements = 76 KiB of LDS, above the 64-KiB default
    of a dynamic allocation (round 5 refused it at run time; the entry point now raises the kernel's limit to the CU's 160 KiB) — and one that exceeds even that
    is refused when the plan's workspace is SIZED, not in the middle of a run."""
    from aether_amd import _lib
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(5)
    Cc, T, H, W = 3, 2, 3, 1400
    x = torch.randn(Cc, T, H, W, generator=g).to(torch.bfloat16)
    w = torch.randn(32, Cc, 3, 3, 3, generator=g).to(torch.bfloat16)
    conv = _Conv(w, torch.zeros(32), cuda, pad_k_to=128)
    vae = _vae(cuda)
    A = vae._im2col(x.to(cuda), conv, [(0, 0)], 0, T, H, W, True)
    vol = torch.cat([x[:, :1].float().repeat(1, 2, 1, 1), x.float()], 1)[None]
    ref = F.conv3d(vol, w.float(), padding=(0, 1, 1))[0].permute(1, 2, 3, 0).reshape(T * H * W, 32)
    got = A[0].float().cpu() @ conv.w.float().cpu().t()
    assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2)
    # wider than 160 KiB of LDS allows: the plan says so at workspace-query time
    big = _vae(cuda).init_random_weights(0)
    big.use_tiling = False
    lib = _lib.load()
    assert lib.aether_vae_workspace_bytes(big._handle, 0, 9, 16, 3200, 0) == 0 and b"160 KiB" in lib.aether_last_error()
    assert lib.aether_vae_workspace_bytes(big._handle, 0, 9, 16, 1400, 0) > 0


def _oracle_pair(cfg_kw, seed=0):
    from oracle.vae import OracleVAE, VaeConfig, init_random_
    cfg = VaeConfig(**cfg_kw)
    o32 = init_random_(OracleVAE(cfg), seed)
    sd = {k: v.to(torch.bfloat16) for k, v in o32.state_dict().items()}
    o32.load_state_dict({k: v.float() for k, v in sd.items()})
    o16 = OracleVAE(cfg).to(torch.bfloat16)
    o16.load_state_dict(sd)
    for o in (o32, o16):
        o.enable_tiling()
    return cfg, sd, o32, o16


def _smooth_video(T, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    frames = [torch.stack([torch.sin(0.09 * xx + 0.3 * t + c) * torch.cos(0.06 * yy - 0.1 * t) for c in range(3)]) for t in range(T)]
    v = torch.stack(frames, 1) * 0.8 + 0.1 * torch.randn(3, T, H, W, generator=g)
    return v.clamp(-1, 1)[None].to(torch.bfloat16)


@pytest.mark.parametrize("T", [1, 17])
def test_vae_encode_small_tiled(cuda, hip_lib, T):
    from aether_amd.vae import AetherVAE
    kw = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240)
    cfg, sd, o32, o16 = _oracle_pair(kw)
    x = _smooth_video(T, 96, 240)
    ref = o32.encode(x.float()).latent_dist.parameters
    ref16 = o16.encode(x).latent_dist.parameters
    vae = AetherVAE(kw, device=cuda).load_state_dict(sd)
    vae.enable_tiling(); vae.enable_slicing()
    got = vae.encode(x.to(cuda)).latent_dist.parameters
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (1, 32, (T - 1) // 4 + 1, 12, 30)
    e_n, e_16 = _rel(got.cpu(), ref), _rel(ref16, ref)
    print(f"encode T={T}: native {e_n:.3e}  bf16-oracle {e_16:.3e}")
    assert e_n < 1.5 * e_16 + 3e-3


def test_vae_decode_small_tiled_and_posterior_rng(cuda, hip_lib):
    from aether_amd.vae import AetherVAE
    kw = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240)
    cfg, sd, o32, o16 = _oracle_pair(kw, seed=1)
    z = (torch.randn(1, 16, 5, 12, 30, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
    ref = o32.decode(z.float()).sample
    ref16 = o16.decode(z).sample
    vae = AetherVAE(kw, device=cuda).load_state_dict(sd)
    vae.enable_tiling()
    got = vae.decode(z.to(cuda)).sample
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (1, 3, 17, 96, 240)
    e_n, e_16 = _rel(got.cpu(), ref), _rel(ref16, ref)
    print(f"decode: native {e_n:.3e}  bf16-oracle {e_16:.3e}")
    assert e_n < 1.5 * e_16 + 3e-3
    # untiled path too
    vae.disable_tiling(); o32.use_tiling = False
    assert _rel(vae.decode(z.to(cuda)).sample.cpu(), o32.decode(z.float()).sample) < 1.5 * e_16 + 3e-3
    # posterior sample draws randn(mean.shape) in bf16 from the caller's generator
    vae.enable_tiling()
    x = _smooth_video(5, 96, 240)
    post = vae.encode(x.to(cuda)).latent_dist
    gen = torch.Generator(device=cuda).manual_seed(5)
    s = post.sample(gen)
    noise = torch.randn(post.mean.shape, generator=torch.Generator(device=cuda).manual_seed(5), device=cuda, dtype=torch.bfloat16)
    assert torch.equal(s, post.mean + post.std * noise)


@pytest.mark.parametrize("kw,T,h,w", [
    (dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240), 17, 96, 240),   # 9 tiles, 2 chunks
    (dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240), 5, 48, 120),    # no tiling, 1 chunk
    (dict(sample_height=96, sample_width=240), 9, 96, 240),                                                                # real widths
])
def test_c_plan_matches_python_walk(cuda, hip_lib, kw, T, h, w):
    """aether_vae_encode / aether_vae_decode (launch plan in C++: csrc/vae_plan.hip, incl. the one-pass tile assembly) against
    the Python walk over the per-kernel entry points: same kernels, same order -> bit-identical; and the second call (zero-
    bordered volumes persisted in the workspace) equals the first."""
    from aether_amd.vae import AetherVAE
    vae = AetherVAE(kw, device=cuda).init_random_weights(3)
    vae.enable_tiling(); vae.enable_slicing()
    x = _smooth_video(T, h, w, seed=4).to(cuda)
    z = torch.randn(1, 16, (T - 1) // 4 + 1, h // 8, w // 8, generator=torch.Generator().manual_seed(6)).to(torch.bfloat16).to(cuda)
    vae.use_c_plan = False
    enc_py, dec_py = vae.encode(x).latent_dist.parameters, vae.decode(z).sample
    vae.use_c_plan = True
    enc_c, dec_c = vae.encode(x).latent_dist.parameters, vae.decode(z).sample
    enc_c2, dec_c2 = vae.encode(x).latent_dist.parameters, vae.decode(z).sample
    torch.cuda.synchronize()
    assert enc_c.shape == enc_py.shape and dec_c.shape == dec_py.shape
    assert torch.equal(enc_c, enc_py), (enc_c.float() - enc_py.float()).abs().max()
    assert torch.equal(dec_c, dec_py), (dec_c.float() - dec_py.float()).abs().max()
    assert torch.equal(enc_c2, enc_c) and torch.equal(dec_c2, dec_c)


def test_c_plan_is_graph_capturable(cuda, hip_lib):
    """The whole decode enqueues without allocation, synchronisation or host<->device copies: it can be captured into a hipGraph
    and replayed."""
    from aether_amd.vae import AetherVAE
    kw = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240)
    vae = AetherVAE(kw, device=cuda).init_random_weights(1)
    vae.enable_tiling()
    z = torch.randn(1, 16, 5, 12, 30, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(cuda)
    eager = vae.decode(z).sample.clone()                      # also sizes the workspace and fills the pool
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        vae.decode(z)                                          # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = vae.decode(z).sample
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_vae_full_width_decode_chunk(cuda, hip_lib):
    """Real channel widths (128,256,256,512), 3 resnets per block, on a reduced frame (96x240, 9 frames):
    every production kernel configuration (256-, 128- and 32-wide GEMM tiles, K up to 13 824)."""
    from aether_amd.vae import AetherVAE
    kw = dict(sample_height=96, sample_width=240)
    cfg, sd, o32, o16 = _oracle_pair(kw, seed=2)
    z = torch.randn(1, 16, 3, 12, 30, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16)
    ref = o32.decode(z.float()).sample
    ref16 = o16.decode(z).sample
    vae = AetherVAE(kw, device=cuda).load_state_dict(sd)
    vae.enable_tiling()
    got = vae.decode(z.to(cuda)).sample
    torch.cuda.synchronize()
    e_n, e_16 = _rel(got.cpu(), ref), _rel(ref16, ref)
    print(f"full-width decode: native {e_n:.3e}  bf16-oracle {e_16:.3e}")
    assert e_n < 1.5 * e_16 + 3e-3
    x = _smooth_video(9, 96, 240, seed=2)
    ref = o32.encode(x.float()).latent_dist.parameters
    ref16 = o16.encode(x).latent_dist.parameters
    got = vae.encode(x.to(cuda)).latent_dist.parameters
    e_n, e_16 = _rel(got.cpu(), ref), _rel(ref16, ref)
    print(f"full-width encode: native {e_n:.3e}  bf16-oracle {e_16:.3e}")
    assert e_n < 1.5 * e_16 + 3e-3
    _ = C


@pytest.mark.parametrize("Cc,W,zW,nblk", [(128, 16, 2, 1), (64, 6, 3, 3), (256, 5, 5, 2), (128, 32, 2, 7), (512, 40, 0, 2), (64, 7, 0, 5)])
def test_groupnorm_run_lengths_and_long_blocks(cuda, hip_lib, Cc, W, zW, nblk):
    """Every run length of the apply kernel (W / zW = 8, 2, 1, 16; no conditioning with W % 8 == 0 and odd W) and partial-sum
    workgroups long enough to take the four-loads-per-iteration path (few blocks over many voxels)."""
    import ctypes as C
    from aether_amd import _lib
    from aether_amd.vae import _nearest_time_map
    g = torch.Generator().manual_seed(Cc + W)
    NB, T, H, G = 2, 4, 36, 32
    x = (torch.randn(NB, T, H, W, Cc, generator=g) * 1.5 - 0.4).to(torch.bfloat16)
    xs = x.to(cuda)
    gamma = (1 + 0.1 * torch.randn(Cc, generator=g)).to(cuda)
    beta = (0.1 * torch.randn(Cc, generator=g)).to(cuda)
    V = T * H * W
    part = torch.empty(NB * nblk * 2 * Cc, dtype=torch.float32, device=cuda)
    stats = torch.empty(NB, G, 2, dtype=torch.float32, device=cuda)
    affine = torch.empty(NB, 2, Cc, dtype=torch.float32, device=cuda)
    st = _lib.current_stream()
    _lib.check(hip_lib.aether_groupnorm_stats(xs.data_ptr(), NB, V, Cc, G, 1e-6, gamma.data_ptr(), beta.data_ptr(), part.data_ptr(), nblk,
                                              stats.data_ptr(), affine.data_ptr(), st), "stats")
    xg = x.double().reshape(NB, V, G, Cc // G)
    mu = xg.mean(dim=(1, 3))
    var = xg.var(dim=(1, 3), unbiased=False)
    torch.cuda.synchronize()
    assert torch.allclose(stats[..., 0].cpu().double(), mu, atol=2e-5, rtol=1e-5)
    assert torch.allclose(stats[..., 1].cpu().double(), 1 / torch.sqrt(var + 1e-6), rtol=2e-5)
    # apply (+ optional SpatialNorm3D pair gathered at the nearest latent voxel) against the same formula in fp64
    vol = torch.zeros(NB, T + 2, H + 2, W + 2, Cc, dtype=torch.bfloat16, device=cuda)
    aff = affine.cpu()
    ref = (x.float() * aff[:, 0].view(NB, 1, 1, 1, Cc) + aff[:, 1].view(NB, 1, 1, 1, Cc)).double()
    if zW:
        zT, zH = 2, H // 4
        cond = torch.randn(NB, zT, zH, zW, 2, Cc, generator=g)
        tmap = _nearest_time_map(T, zT)
        idx_t = torch.tensor(tmap)
        idx_h = torch.arange(H) // (H // zH)
        idx_w = torch.arange(W) // (W // zW)
        cg = cond[:, idx_t][:, :, idx_h][:, :, :, idx_w].double()              # [NB, T, H, W, 2, C]
        ref = ref * cg[..., 0, :] + cg[..., 1, :]
        condd = cond.to(cuda)
        rc = hip_lib.aether_groupnorm_apply(xs.data_ptr(), NB, T, H, W, Cc, affine.data_ptr(), 1, vol.data_ptr(), T + 2, H + 2, W + 2, 2, 1, 1,
                                            condd.data_ptr(), zT, zH, zW, (C.c_int * T)(*tmap), st)
    else:
        rc = hip_lib.aether_groupnorm_apply(xs.data_ptr(), NB, T, H, W, Cc, affine.data_ptr(), 1, vol.data_ptr(), T + 2, H + 2, W + 2, 2, 1, 1,
                                            None, 0, 0, 0, None, st)
    _lib.check(rc, "apply")
    ref = ref * torch.sigmoid(ref)
    torch.cuda.synchronize()
    assert _rel(vol[:, 2:, 1:-1, 1:-1].cpu(), ref.float()) < 4e-3
    assert float(vol[:, :2].abs().max()) == 0 and float(vol[:, :, 0].abs().max()) == 0 and float(vol[:, :, :, -1].abs().max()) == 0


@pytest.mark.parametrize("T", [1, 2, 5])
def test_causal_front_single_launch(cuda, hip_lib, T):
    """Front frames from the previous chunk's cache (or copies of the first frame) and the cache for the next chunk, against
    the three tensor copies they replace; T = 1 makes the new cache overlap the front frames."""
    vae = _vae(cuda)
    g = torch.Generator().manual_seed(T)
    NB, H, W, Cc = 2, 5, 7, 64
    for with_prev in (False, True):
        vol = torch.randn(NB, T + 2, H + 2, W + 2, Cc, generator=g).to(torch.bfloat16).to(cuda)
        prev = torch.randn(NB, 2, H + 2, W + 2, Cc, generator=g).to(torch.bfloat16).to(cuda) if with_prev else None
        ref = vol.clone()
        if prev is None:
            ref[:, 0].copy_(ref[:, 2]); ref[:, 1].copy_(ref[:, 2])
        else:
            ref[:, :2].copy_(prev)
        cache = {} if prev is None else {"k": prev}
        vae._causal_front(vol, cache, "k")
        torch.cuda.synchronize()
        assert torch.equal(vol, ref) and torch.equal(cache["k"], ref[:, -2:])


@pytest.mark.parametrize("Cc,T,spatial", [(64, 1, False), (128, 2, False), (64, 5, True), (256, 1, True), (512, 3, False)])
def test_groupnorm_apply_with_fused_causal_front(cuda, hip_lib, Cc, T, spatial):
    """aether_groupnorm_apply_causal (norm + front frames + cache for the next chunk in ONE launch, what the C launch plan issues)
    against aether_groupnorm_apply followed by aether_causal_front, over two consecutive chunks (no cache, then with a cache):
    the padded volumes are bit-identical and the caches agree on their interiors (the fused form never touches cache borders)."""
    from aether_amd.vae import _Norm
    from oracle.vae import SpatialNorm3D
    g = torch.Generator().manual_seed(100 * Cc + T)
    NB, H, W = 2, 8, 12
    vae = _vae(cuda)
    if spatial:
        sn = SpatialNorm3D(Cc, 16, 32)
        with torch.no_grad():
            for name, p in sn.named_parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
            sn.norm_layer.weight.add_(1.0)
        n = _Norm({"s." + k: v for k, v in sn.state_dict().items()}, "s.", cuda, True)
    else:
        n = _Norm({"w.weight": 1 + 0.1 * torch.randn(Cc, generator=g), "w.bias": 0.1 * torch.randn(Cc, generator=g)}, "w.", cuda, False)
    cache_a, cache_b = {}, {}
    for chunk in range(2):
        x = (torch.randn(NB, T, H, W, Cc, generator=g) * 2 + 0.3).to(torch.bfloat16).to(cuda)
        zq = torch.randn(NB, max(1, (T + 1) // 2), H // 4, W // 4, 16, generator=g).to(torch.bfloat16).to(cuda) if spatial else None
        ref = vae._norm_to_padded(x, n, 2, 1, True, zq).clone()
        vae._causal_front(ref, cache_a, "k")
        out = vae._norm_to_padded(x, n, 2, 1, True, zq, causal=(cache_b, "k")).clone()
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (chunk, float((out.float() - ref.float()).abs().max()))
        assert torch.equal(cache_b["k"][:, :, 1:-1, 1:-1], cache_a["k"][:, :, 1:-1, 1:-1])
        assert torch.equal(cache_a["k"], ref[:, -2:])


@pytest.mark.parametrize("cin,cout,NB,T,H,W,kt,res", [(128, 128, 1, 2, 12, 20, 3, False), (64, 256, 2, 3, 9, 33, 3, True),
                                                      (256, 128, 1, 1, 30, 45, 3, True), (128, 256, 2, 2, 16, 24, 1, False),
                                                      (64, 128, 3, 1, 5, 7, 1, True), (512, 512, 1, 2, 40, 61, 3, False),
                                                      (128, 3, 2, 3, 24, 36, 3, False), (64, 3, 1, 2, 30, 45, 3, False)])      # conv_out: 3 channels in a 512 x 32 tile
def test_conv_tap_reuse_is_bit_identical(cuda, hip_lib, cin, cout, NB, T, H, W, kt, res):
    """conv3_kernel.hpp (one staged input tile for the three dw taps, rows enumerated over the padded plane) against the
    plain gathered kernel: same K order inside every output -> the same bits; tiles that straddle frames, batch items and
    the dropped border rows, with and without the residual epilogue, causal 3x3x3 and per-frame 3x3."""
    from aether_amd import _lib
    from aether_amd.vae import _Conv
    g = torch.Generator().manual_seed(cin + cout + H)
    shape = (cout, cin, 3, 3, 3) if kt == 3 else (cout, cin, 3, 3)
    conv = _Conv(torch.randn(shape, generator=g) * 0.05, torch.randn(cout, generator=g) * 0.1, cuda)
    assert conv.blocked
    vae = _vae(cuda)
    vae.splitk_ws_bytes = 0                      # the small plain launches would otherwise split K (another summation order)
    vol = torch.zeros(NB, T + (2 if kt == 3 else 0), H + 2, W + 2, cin, dtype=torch.bfloat16, device=cuda)
    vol[:, :, 1:-1, 1:-1] = torch.randn(NB, vol.shape[1], H, W, cin, generator=g).to(torch.bfloat16).to(cuda)
    R = torch.randn(NB, T, H, W, cout, generator=g).to(torch.bfloat16).to(cuda) if res else None
    outs = []
    for waste, wide in ((0.0, True), (100.0, True), (100.0, False)):
        vae.tap_reuse_max_waste = waste
        vae._flags = _lib.AETHER_GEMM_WIDE_STORE if wide else 0
        outs.append(vae._conv(vol, conv, (T, H, W), 1, R))
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0.1
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("lanes", [2, 1])
def test_decode_pair_is_bit_identical(cuda, hip_lib, lanes):
    """AetherVAE.decode_pair: the two final decodes of a pipeline call (rgb and disparity latents, P:931,936).  Two-lane launch plan (default): two
    calls in a row, each with its tile batches on two streams.  One lane: the two decodes enqueued on two HIP streams over one set of packed weights
    (second C handle + workspace).  Same kernels in the same order within each decode: bit-identical to the sequential calls — eagerly, on the
    captured hipGraphs, and again after the roles of the inputs are swapped."""
    from aether_amd import _lib
    from aether_amd.vae import AetherVAE
    kw = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, sample_height=96, sample_width=240)
    vae = AetherVAE(kw, device=cuda, flags=_lib.AETHER_GEMM_WIDE_STORE | (_lib.AETHER_VAE_TWO_LANES if lanes == 2 else 0)).init_random_weights(3)
    vae.enable_tiling(); vae.enable_slicing()
    g = torch.Generator(device=cuda).manual_seed(0)
    za = torch.randn(1, 16, 5, 12, 30, generator=g, device=cuda).to(torch.bfloat16)
    zb = torch.randn(1, 16, 5, 12, 30, generator=g, device=cuda).to(torch.bfloat16)
    ref_a, ref_b = vae.decode(za).sample.clone(), vae.decode(zb).sample.clone()
    for _ in range(3):                                     # eager first call, graph capture, replay
        a, b = vae.decode_pair(za, zb)
        torch.cuda.synchronize()
        assert torch.equal(a, ref_a) and torch.equal(b, ref_b)
    b2, a2 = vae.decode_pair(zb, za)
    torch.cuda.synchronize()
    assert torch.equal(a2, ref_a) and torch.equal(b2, ref_b)
