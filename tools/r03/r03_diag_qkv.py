"""Diagnostic: fused qkv-prep epilogue vs the two-pass path at the full width, each against an fp32 reference of the preparation."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aether_amd import ops
from aether_amd._lib import ATTN_Q_SCALE
import test_kernels_gpu as T
cuda = torch.device("cuda:0")
B, H, S, n_text, K = 1, 48, 4276, 226, 3072
g = torch.Generator().manual_seed(1)
A = torch.randn(B * S, K, generator=g).to(torch.bfloat16)
W = (torch.randn(3 * H * 64, K, generator=g) * 1.5 / math.sqrt(K)).to(torch.bfloat16)
bias = 0.1 * torch.randn(3 * H * 64, generator=g)
_, qn_w, qn_b, kn_w, kn_b, cos, sin = T._attn_inputs(B, H, S, n_text, 3)
c = lambda t: t.to(cuda)
for flags in (5,):
    ws = torch.empty(16 << 20, dtype=torch.float32, device=cuda)
    for use_ws in (None, ws):
        qkv = ops.gemm_bf16(c(A), c(W), c(bias), ops.AETHER_EPI_BIAS, flags=flags, splitk_ws=use_ws).view(B, S, 3 * H * 64)
        Q0, K0, V0 = ops.qk_norm_rope(qkv, H, n_text, c(qn_w), c(qn_b), c(kn_w), c(kn_b), 1e-6, c(cos), c(sin), ATTN_Q_SCALE)
        Q1, K1, V1 = ops.gemm_qkv_prep(c(A), c(W), c(bias), H, S, n_text, c(qn_w), c(qn_b), c(kn_w), c(kn_b), 1e-6, c(cos), c(sin), ATTN_Q_SCALE, flags=flags)
        torch.cuda.synchronize()
        print("ws", use_ws is not None, "V equal", torch.equal(V0, V1), "V mismatches", (V0 != V1).sum().item())
        for name, a, b in (("q", Q0, Q1), ("k", K0, K1)):
            a, b = a.float(), b.float()
            d = (a != b)
            print(name, "differ frac", d.float().mean().item(), "max abs", (a - b).abs().max().item(), "rel-L2", ((a - b).norm() / a.norm()).item())
            if d.any():
                idx = d.nonzero()
                print("  first diffs", idx[:5].tolist(), "rows with diffs", idx[:, 2].unique().numel(), "heads", idx[:, 1].unique().numel())
        # fp32 reference of the whole thing from the bf16 projection
        rq, rk, rv = T._qk_ref(qkv.cpu(), H, n_text, qn_w, qn_b, kn_w, kn_b, cos, sin)
        for name, r, a, b in (("q", rq * ATTN_Q_SCALE, Q0, Q1), ("k", rk, K0, K1)):
            ea = ((a.cpu().float() - r).norm() / r.norm()).item(); eb = ((b.cpu().float() - r).norm() / r.norm()).item()
            print(f"  {name}: two-pass vs fp32 ref {ea:.3e}   fused vs fp32 ref {eb:.3e}")
