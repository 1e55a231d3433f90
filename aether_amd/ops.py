"""torch-tensor wrappers over the per-kernel C-ABI entry points (include/aether_hip.h).

These are the calls the parity tests exercise one by one; the production transformer goes through the single
`aether_dit_forward` entry instead.  Every function enqueues on torch's current stream and raises on error.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import AETHER_EPI_BIAS, AETHER_EPI_BIAS_GATE_RES, AETHER_EPI_BIAS_GELU  # noqa: F401


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("aether_amd ops need device tensors (no CPU fallback)")


def gemm_bf16(A, W, bias=None, epilogue=AETHER_EPI_BIAS, R=None, gate_vid=None, gate_txt=None, rows_per_batch=0, n_text=0,
              out=None, flags=0, splitk_ws=None):
    """out[M,N] = epi(A[M,K] @ W[N,K]^T); A,W,R bf16; bias/gates fp32. gate_*: [B, N] (row stride = stride(0)).
    splitk_ws: optional fp32 scratch (>= 64 MiB) enabling the split-K tail launch."""
    _need_cuda(A, W, bias, R, gate_vid, gate_txt)
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and A.dim() == 2 and W.dim() == 2
    assert A.stride(1) == 1 and W.stride(1) == 1
    M, K = A.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=A.device)
    gstride = gate_vid.stride(0) if gate_vid is not None else 0
    lib = _lib.load()
    rc = lib.aether_gemm_bf16(_lib.ptr(A), A.stride(0), _lib.ptr(W), W.stride(0), _lib.ptr(out), out.stride(0), M, N, K,
                              _lib.ptr(bias), epilogue, _lib.ptr(R), R.stride(0) if R is not None else 0,
                              _lib.ptr(gate_vid), _lib.ptr(gate_txt), gstride, rows_per_batch, n_text, _lib.ptr(splitk_ws),
                              splitk_ws.numel() * 4 if splitk_ws is not None else 0, flags, _lib.current_stream())
    _lib.check(rc, "aether_gemm_bf16")
    return out


def layernorm_modulate(x, w=None, b=None, eps=1e-5, shift_vid=None, scale_vid=None, shift_txt=None, scale_txt=None,
                       rows_per_batch=0, n_text=0, out=None):
    _need_cuda(x, w, b, shift_vid)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    rows, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    mstride = shift_vid.stride(0) if shift_vid is not None else 0
    lib = _lib.load()
    rc = lib.aether_layernorm_modulate(_lib.ptr(x), x.stride(0), _lib.ptr(out), out.stride(0), rows, D, float(eps), _lib.ptr(w),
                                       _lib.ptr(b), _lib.ptr(shift_vid), _lib.ptr(scale_vid), _lib.ptr(shift_txt),
                                       _lib.ptr(scale_txt), mstride, rows_per_batch, n_text, _lib.current_stream())
    _lib.check(rc, "aether_layernorm_modulate")
    return out


def gemv_rows(x, W, bias=None, act_in=0, act_out=0):
    """out[B,N] = act_out(bias + act_in(x[B,K]) @ W[N,K]^T); x fp32, W bf16, out fp32."""
    _need_cuda(x, W, bias)
    assert x.dtype == torch.float32 and W.dtype == torch.bfloat16 and x.is_contiguous() and W.is_contiguous()
    B, K = x.shape
    N = W.shape[0]
    out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    rc = lib.aether_gemv_rows(_lib.ptr(x), B, K, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), N, act_in, act_out,
                              _lib.current_stream())
    _lib.check(rc, "aether_gemv_rows")
    return out


def timestep_sinusoid(t, dim):
    _need_cuda(t)
    t = t.to(torch.float32).contiguous()
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    rc = _lib.load().aether_timestep_sinusoid(_lib.ptr(t), t.shape[0], dim, _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "aether_timestep_sinusoid")
    return out


def patchify(x, p):
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 5
    B, F, Cc, H, W = x.shape
    out = torch.empty(B * F * (H // p) * (W // p), Cc * p * p, dtype=torch.bfloat16, device=x.device)
    rc = _lib.load().aether_patchify(_lib.ptr(x), _lib.ptr(out), B, F, Cc, H, W, p, _lib.current_stream())
    _lib.check(rc, "aether_patchify")
    return out


def unpatchify(Y, B, F, Cout, H, W, p):
    _need_cuda(Y)
    assert Y.dtype == torch.bfloat16 and Y.stride(1) == 1
    out = torch.empty(B, F, Cout, H, W, dtype=torch.bfloat16, device=Y.device)
    rc = _lib.load().aether_unpatchify(_lib.ptr(Y), Y.stride(0), _lib.ptr(out), B, F, Cout, H, W, p, _lib.current_stream())
    _lib.check(rc, "aether_unpatchify")
    return out


def qk_norm_rope(qkv, H, n_text, qn_w, qn_b, kn_w, kn_b, eps, cos, sin, q_scale):
    """qkv bf16 [B,S,3*H*64] -> (Qh [B,H,S,64], Kh [B,H,S,64], Vt [B,H,64,Spad])."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    B, S, _ = qkv.shape
    Spad = (S + 63) // 64 * 64
    Qh = torch.empty(B, H, S, 64, dtype=torch.bfloat16, device=qkv.device)
    Kh = torch.empty_like(Qh)
    Vt = torch.empty(B, H, 64, Spad, dtype=torch.bfloat16, device=qkv.device)
    rc = _lib.load().aether_qk_norm_rope(_lib.ptr(qkv), B, S, H, n_text, _lib.ptr(qn_w), _lib.ptr(qn_b), _lib.ptr(kn_w),
                                         _lib.ptr(kn_b), float(eps), _lib.ptr(cos), _lib.ptr(sin), float(q_scale),
                                         _lib.ptr(Qh), _lib.ptr(Kh), _lib.ptr(Vt), Spad, _lib.current_stream())
    _lib.check(rc, "aether_qk_norm_rope")
    return Qh, Kh, Vt


def flash_attn_fwd(Qh, Kh, Vt, flags=0):
    """Qh,Kh [B,H,S,64] (softmax scale x log2(e) folded into Qh: _lib.ATTN_Q_SCALE), Vt [B,H,64,Spad] -> O [B,S,H*64].
    flags: AETHER_GEMM_WIDE_STORE | AETHER_ATTN_EXACT_MAX (the conservative path alone)."""
    _need_cuda(Qh, Kh, Vt)
    B, H, S, d = Qh.shape
    assert d == 64 and Qh.is_contiguous() and Kh.is_contiguous() and Vt.is_contiguous()
    Spad = Vt.shape[-1]
    O = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=Qh.device)
    rc = _lib.load().aether_flash_attn_fwd(_lib.ptr(Qh), _lib.ptr(Kh), _lib.ptr(Vt), _lib.ptr(O), B, H, S, Spad, flags, _lib.current_stream())
    _lib.check(rc, "aether_flash_attn_fwd")
    return O
