// Non-causal flash attention forward, head_dim 64, for the joint text+video self-attention of the DiT
// (S = 226 + 14 850 tokens at the BASELINE shape).  Replaces F.scaled_dot_product_attention inside diffusers'
// CogVideoXAttnProcessor2_0, reached from aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875.
//
// gfx950 design
//   * workgroup = 8 wavefronts = 256 query rows (32 per wave), KV tile = 64 keys, one barrier per KV tile;
//     K [64 keys][64 d] and Vᵀ [64 d][64 keys] tiles arrive by 16-byte LDS-DMA, double buffered (32 KiB LDS),
//     128-byte rows XOR-swizzled on the DMA source address so every ds_read_b128 is conflict free.
//   * "swapped" QKᵀ: S^T = K·Qᵀ on v_mfma_f32_32x32x16_bf16, so one lane holds 32 of the 64 scores of ONE
//     query row; the row maximum costs 31 in-lane max + one half-wave exchange and the soft-max runs
//     entirely in registers.  K rows enter the MFMA through a fixed permutation (pi below) chosen so the
//     exponentiated scores of a lane are, in register order, exactly the B-operand fragment of the P·V MFMA
//     (8 consecutive keys per 16-key slab) — no cross-lane movement of P.
//   * V is consumed transposed (Vᵀ is produced once per layer by aether_qk_norm_rope), so its A-operand
//     fragment is a plain 16-byte row read.
//   * exp2 domain: p = exp2(s·log2e − m), one fma + one v_exp_f32 per score; the O/l rescale is skipped by a
//     wave-uniform branch whenever no row maximum grew (exact, no threshold).
//   * workgroups are remapped so that one XCD walks the query blocks of one (batch, head) consecutively:
//     its K/V (3.9 MB at S = 15 076) stays in that XCD's 4 MiB L2.
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

constexpr int FA_QBLK = 256, FA_KVBLK = 64, FA_D = 64;
constexpr int FA_TILE = FA_KVBLK * FA_D * 2;  // 8 KiB (K tile) == 8 KiB (Vᵀ tile)
constexpr int FA_BUF = 2 * FA_TILE;

struct FlashArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    int H, S, Spad, nqb, nwg;
};

template <bool WIDE_STORE>
__global__ __launch_bounds__(512) void flash_attn_fwd_kernel(FlashArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * FA_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l32 = lane & 31;

    const int wgid = xcd_remap(blockIdx.x, p.nwg);
    const int bh = wgid / p.nqb;
    const int qb = wgid - bh * p.nqb;
    const int S = p.S;

    const bf16_t* Qg = p.Q + (size_t)bh * S * FA_D;
    const bf16_t* Kg = p.K + (size_t)bh * S * FA_D;
    const bf16_t* Vg = p.Vt + (size_t)bh * FA_D * p.Spad;

    // ---- Q fragment (B operand): lane (q = l32, hi) holds Q[q][16ks + 8hi .. +7] -------------------
    const int qrow = qb * FA_QBLK + wave * 32 + l32;
    const int qrow_c = min(qrow, S - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + (size_t)qrow_c * FA_D + 16 * ks + 8 * hi);

    // ---- staging (one 16-byte piece of K and one of Vᵀ per thread per KV tile) ---------------------
    const int srow = tid >> 3;                          // K: key row in tile; Vᵀ: d row
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);   // logical chunk fetched into physical chunk tid&7
    char* const lds_stage = smem + wave * 1024;
    const bf16_t* const vsrc = Vg + (size_t)srow * p.Spad + schunk * 8;
    auto stage = [&](int j, int buf) {
        const int krow = min(j * FA_KVBLK + srow, S - 1);
        glds16(Kg + (size_t)krow * FA_D + schunk * 8, lds_stage + buf * FA_BUF);
        glds16(vsrc + j * FA_KVBLK, lds_stage + buf * FA_BUF + FA_TILE);
    };

    // ---- fragment read offsets ---------------------------------------------------------------------
    // K row fed to MFMA row i (= l32):  pi(i) = 16*(a>>1) + 8*h + 4*(a&1) + c,  i = 8a + 4h + c
    const int a_ = l32 >> 3, h_ = (l32 >> 2) & 1, c_ = l32 & 3;
    const int pi = 16 * (a_ >> 1) + 8 * h_ + 4 * (a_ & 1) + c_;
    const int kswz = (pi >> 1) & 7;
    int koff[4];  // + t*32*128
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = pi * 128 + (((2 * ks + hi) ^ kswz) << 4);
    const int vswz = (l32 >> 1) & 7;
    int voff[2][2];  // [t][s], + dt*32*128; chunk = 4t + 2s + hi
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) voff[t][s] = FA_TILE + l32 * 128 + (((4 * t + 2 * s + hi) ^ vswz) << 4);

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -INFINITY;  // running row maximum, log2 domain (shared by the two lanes of a row)
    float l_run = 0.f;        // this lane's partial row sum

    const float LOG2E = 1.4426950408889634f;
    const int nkv = (S + FA_KVBLK - 1) / FA_KVBLK;
    stage(0, 0);
    drain_and_barrier();

    for (int j = 0; j < nkv; ++j) {
        const int cur = j & 1;
        if (j + 1 < nkv) stage(j + 1, cur ^ 1);
        const char* base = smem + cur * FA_BUF;

        // S^T tiles: sc[t][r] = score(q = l32, key = 64j + 32t + 16(r>>3) + 8hi + (r&7))
        f32x16 sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[t][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(base + t * 4096 + koff[ks]);
                sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc[t], 0, 0, 0);
            }
        }
        if (j == nkv - 1 && (S & (FA_KVBLK - 1))) {  // ragged last tile: keys >= S score -inf
            const int kb = j * FA_KVBLK + 8 * hi;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + 32 * t + 16 * (r >> 3) + (r & 7) >= S) sc[t][r] = -INFINITY;
        }

        // ---- online softmax ----
        float mx = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * LOG2E);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the first tile (m_run = -inf)
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
            l_run *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][8 * s + e], LOG2E, -m_run));
                    psum += pv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[t][s][e] = (__bf16)pv[e];
            }
        l_run += psum;

        // ---- O^T += Vᵀ · Pᵀ ----
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf = *(const bf16x8*)(base + dt * 4096 + voff[t][s]);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[t][s], o[dt], 0, 0, 0);
                }
        drain_and_barrier();
    }

    // ---- epilogue: O[q][h*64 + d], d = 32dt + 8(r>>2) + 4hi + (r&3) ---------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int b = bh / p.H, h = bh - b * p.H;
    const bool q_ok = qrow < S;
    bf16_t* orow = p.O + ((size_t)b * S + qrow_c) * (size_t)(p.H * FA_D) + h * FA_D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        unsigned pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pk[g][0] = pack_bf16x2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
            pk[g][1] = pack_bf16x2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        }
        if (WIDE_STORE) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                if (q_ok) *(uint4*)(orow + 32 * dt + 8 * g + 8 * hi) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (q_ok) *(uint2*)(orow + 32 * dt + 8 * g + 4 * hi) = make_uint2(pk[g][0], pk[g][1]);
        }
    }
}

}  // namespace aether

using namespace aether;

extern "C" int aether_flash_attn_fwd(const void* Qh, const void* Kh, const void* Vt, void* O, int B, int H, int S,
                                     int Spad, int flags, void* stream) {
    if (!Qh || !Kh || !Vt || !O) return aether_set_error(AETHER_ERR_ARG, "flash_attn: null pointer");
    if (B <= 0 || H <= 0 || S <= 0) return aether_set_error(AETHER_ERR_SHAPE, "flash_attn: empty problem");
    if (Spad % FA_KVBLK != 0 || Spad < S) return aether_set_error(AETHER_ERR_SHAPE, "flash_attn: Spad must be roundup(S,64)");
    if (((uintptr_t)Qh | (uintptr_t)Kh | (uintptr_t)Vt | (uintptr_t)O) & 15)
        return aether_set_error(AETHER_ERR_ALIGN, "flash_attn: pointers must be 16-byte aligned");
    FlashArgs p;
    p.Q = (const bf16_t*)Qh; p.K = (const bf16_t*)Kh; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
    p.H = H; p.S = S; p.Spad = Spad;
    p.nqb = (S + FA_QBLK - 1) / FA_QBLK;
    p.nwg = p.nqb * B * H;
    hipStream_t s = (hipStream_t)stream;
    if (flags & AETHER_GEMM_WIDE_STORE)
        hipLaunchKernelGGL((flash_attn_fwd_kernel<true>), dim3(p.nwg), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL((flash_attn_fwd_kernel<false>), dim3(p.nwg), dim3(512), 0, s, p);
    return aether_check_launch("flash_attn_fwd");
}
