"""Checkpoint loading in the diffusers folder layout the reference reads at scripts/demo.py:206-228
(`<root>/transformer/config.json + diffusion_pytorch_model*.safetensors`, `<root>/vae/...`, `<root>/scheduler/
scheduler_config.json`).  Host-side repacking only — runs without a GPU (weights stay on the CPU, nothing is launched)."""
import json
import os

import torch
from safetensors.torch import save_file


def _write_transformer(root, sd, cfg, shards=2):
    os.makedirs(os.path.join(root, "transformer"), exist_ok=True)
    with open(os.path.join(root, "transformer", "config.json"), "w") as f:
        json.dump(dict(cfg, _class_name="CogVideoXTransformer3DModel", _diffusers_version="0.32.2"), f)
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    weight_map = {}
    for i in range(shards):
        fn = f"diffusion_pytorch_model-{i + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: sd[k].contiguous() for k in keys[i * per:(i + 1) * per]}
        save_file(part, os.path.join(root, "transformer", fn))
        weight_map.update({k: fn for k in part})
    with open(os.path.join(root, "transformer", "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)


def test_transformer_from_pretrained_sharded(tmp_path):
    from aether_amd.transformer import AetherTransformer3D
    from oracle.dit import DitConfig, OracleTransformer3D, init_random_
    kw = dict(num_attention_heads=8, num_layers=2, text_embed_dim=128, time_embed_dim=64, max_text_seq_length=20, sample_width=12,
              sample_height=8, sample_frames=9, use_learned_positional_embeddings=True)
    sd = {k: v.to(torch.bfloat16) for k, v in init_random_(OracleTransformer3D(DitConfig(**kw)), 0).state_dict().items()}
    _write_transformer(str(tmp_path), sd, kw)
    m = AetherTransformer3D.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=torch.bfloat16, device="cpu")
    assert m.config.num_layers == 2 and m.config.use_learned_positional_embeddings and m.config.patch_size_t is None
    w = m._weights
    D = 512
    assert w["qkv_w"].shape == (2, 3 * D, D) and w["qkv_w"].dtype == torch.bfloat16
    assert torch.equal(w["qkv_w"][1, D:2 * D], sd["transformer_blocks.1.attn1.to_k.weight"])
    assert torch.equal(w["qkv_b"][0, 2 * D:], sd["transformer_blocks.0.attn1.to_v.bias"].float())
    assert torch.equal(w["patch_w"], sd["patch_embed.proj.weight"].reshape(D, -1))
    # all AdaLN linears concatenated: [norm1(0), norm2(0), norm1(1), norm2(1), norm_out]
    assert w["adaln_w"].shape == (2 * 12 * D + 2 * D, 64)
    assert torch.equal(w["adaln_w"][6 * D:12 * D], sd["transformer_blocks.0.norm2.linear.weight"])
    assert torch.equal(w["adaln_w"][-2 * D:], sd["norm_out.linear.weight"])
    assert torch.equal(w["pos_emb"], sd["patch_embed.pos_embedding"].reshape(-1, D))
    assert m.num_parameters() == sum(v.numel() for v in sd.values())


def test_vae_and_scheduler_from_pretrained(tmp_path):
    from aether_amd.scheduler import CogVideoXDPMScheduler
    from aether_amd.vae import AetherVAE
    from oracle.vae import OracleVAE, VaeConfig, init_random_
    kw = dict(block_out_channels=[64, 128, 128, 128], layers_per_block=1, sample_height=96, sample_width=240)
    sd = {k: v.to(torch.bfloat16) for k, v in init_random_(OracleVAE(VaeConfig(**kw)), 0).state_dict().items()}
    os.makedirs(tmp_path / "vae")
    json.dump(dict(kw, _class_name="AutoencoderKLCogVideoX", scaling_factor=0.7, latent_channels=16), open(tmp_path / "vae" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    vae = AetherVAE.from_pretrained(str(tmp_path), subfolder="vae", torch_dtype=torch.bfloat16, device="cpu")
    assert vae.config.block_out_channels == (64, 128, 128, 128) and vae.config.scaling_factor == 0.7
    c = vae.enc.down[1].resnets[0].conv1                      # 64 -> 128, K order (dt, dh, dw, cin)
    ref = sd["encoder.down_blocks.1.resnets.0.conv1.conv.weight"].float().permute(0, 2, 3, 4, 1).reshape(128, -1)
    assert torch.equal(c.w.float(), ref.to(torch.bfloat16).float()) and c.w.shape == (128, 27 * 64)
    assert vae.enc.conv_in.w.shape == (64, 128) and float(vae.enc.conv_in.w[:, 81:].abs().max()) == 0      # K 81 zero-padded to 128
    assert vae.dec.conv_out.w.shape[0] == 32 and float(vae.dec.conv_out.w[3:].abs().max()) == 0            # 3 output channels padded to 32
    assert vae.dec.mid[0].norm1.wy.shape == (128, 16)
    # scheduler: the shipped scheduler_config.json names a DDIM class; the reference loads it into the DPM class (D:220-222)
    os.makedirs(tmp_path / "scheduler")
    json.dump({"_class_name": "CogVideoXDDIMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
               "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "v_prediction", "rescale_betas_zero_snr": True,
               "set_alpha_to_one": True, "snr_shift_scale": 1.0, "steps_offset": 0, "timestep_spacing": "trailing",
               "clip_sample_range": 1.0, "sample_max_value": 1.0, "trained_betas": None},
              open(tmp_path / "scheduler" / "scheduler_config.json", "w"))
    s = CogVideoXDPMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    s.set_timesteps(50)
    assert s.timesteps[0].item() == 999 and s.config.prediction_type == "v_prediction" and s.config.snr_shift_scale == 1.0
