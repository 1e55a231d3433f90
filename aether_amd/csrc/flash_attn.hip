// Non-causal flash attention forward, head_dim 64, for the joint text+video self-attention of the DiT
// (S = 226 + 14 850 tokens at the BASELINE shape).  Replaces F.scaled_dot_product_attention inside diffusers'
// CogVideoXAttnProcessor2_0, reached from aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875.
//
// Contract: Qh carries softmax_scale·log2(e) (aether_qk_norm_rope folds it into the fp32 value before the single
// rounding to bf16), so a score s = q·k is already in the log2 domain and p = exp2(s − m).
//
// gfx950 design (both kernels)
//   * workgroup = 8 wavefronts = 256 query rows (32 per wave), KV tile = 64 keys;
//     K [64 keys][64 d] and Vᵀ [64 d][64 keys] tiles arrive by 16-byte LDS-DMA; 128-byte rows XOR-swizzled on the
//     DMA source address so every ds_read_b128 is conflict free.
//   * "swapped" QKᵀ: S^T = K·Qᵀ on v_mfma_f32_32x32x16_bf16, so one lane holds 32 of the 64 scores of ONE
//     query row; the soft-max runs entirely in registers.  K rows enter the MFMA through a fixed permutation (pi
//     below) chosen so the exponentiated scores of a lane are, in register order, exactly the B-operand fragment of
//     the P·V MFMA (8 consecutive keys per 16-key slab) — no cross-lane movement of P.
//   * V is consumed transposed (Vᵀ is produced once per layer by aether_qk_norm_rope), so its A-operand
//     fragment is a plain 16-byte row read.
//   * bounded-score fast path: at head_dim 64 the kernel is VALU-issue bound (≈5 VALU slots per score against
//     16 MFMAs per 2048 scores), so the biggest lever is fewer VALU ops per score.  q and k are LayerNorm outputs,
//     so |q·k| ≤ ‖q‖·max‖k‖ (Cauchy–Schwarz); aether_qk_norm_rope emits max‖k‖² per (batch, head, 64-key tile).  When that bound
//     is ≤ 96 for every row of a wave, exp2(s) can neither overflow nor underflow in fp32/bf16 and soft-max is
//     shift invariant, so the wave runs p = exp2(s) with NO running maximum, NO subtraction and NO rescale
//     (1 v_exp + 1 v_add + ½ v_cvt_pk per score).  Otherwise (or when no bound is supplied) it runs the exact
//     online soft-max with the conditional rescale.  The decision is per wave and wave-uniform.
//   * workgroups are remapped so that one XCD walks the query blocks of one (batch, head) consecutively:
//     its K/V (3.9 MB at S = 15 076) stays in that XCD's 4 MiB L2.
//
// flash_attn_fwd_kernel: one barrier per KV tile, all waves in lock step, 128 VGPRs -> 16 waves per CU (TLP hides latency).
//   Default: one launch of 8-wave / 256-row workgroups (two per CU).  AETHER_ATTN_TAIL_SPLIT launches the first
//   floor(nwg/512)*512 workgroups that way and the rest — a partly filled last round — as twice as many 4-wave / 128-row
//   workgroups (four per CU); the "rounds" model promises 8 % for 2832 workgroups on 512 slots, the measurement gives < 1 %.
// flash_attn_swp_kernel: software-pipelined variant (AETHER_ATTN_PIPELINED), see below.
#include <type_traits>
#include <utility>
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FA_QBLK = 256, FA_KVBLK = 64, FA_D = 64;
constexpr int FA_TILE = FA_KVBLK * FA_D * 2;  // 8 KiB (K tile) == 8 KiB (Vᵀ tile)
constexpr int FA_BUF = 2 * FA_TILE;
constexpr float FA_FAST_BOUND2 = 96.f * 96.f;  // (‖q‖·max‖k‖)² limit of the bounded-score path: |s| <= 96 keeps every p = exp2(s) a normal
                                               // fp32 / bf16 number (2^-96 .. 2^96) and every sum below 2^96 · S · max|v| << 2^127
constexpr float FA_BOUND_SLACK = 1.02f;        // covers the bf16 rounding of k after its norm was taken (2^-8 rel.)

struct FlashArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    const float* kmax2;   // [B*H][Spad/64] upper bound of ‖k‖² per (batch, head, 64-key tile), or null
    int H, S, Spad, nqb, nwg;   // nqb: query blocks per head at THIS launch's block size; nwg: workgroups of this launch
    int wg_first;               // index of this launch's first workgroup in (head, query block) order
};

// ---- pieces shared by the two kernels ---------------------------------------------------------------------------
struct FaLane {
    int lane, wave, hi, l32;
    int koff[4];      // K fragment byte offsets inside a tile (+ t*4096)
    int voff[2][2];   // Vᵀ fragment byte offsets inside a buffer (+ dt*4096), includes FA_TILE
};

AE_DEV FaLane fa_lane_setup() {
    FaLane L;
    const int tid = threadIdx.x;
    L.lane = tid & 63; L.wave = tid >> 6; L.hi = L.lane >> 5; L.l32 = L.lane & 31;
    // K row fed to MFMA row i (= l32):  pi(i) = 16*(a>>1) + 8*h + 4*(a&1) + c,  i = 8a + 4h + c
    const int a_ = L.l32 >> 3, h_ = (L.l32 >> 2) & 1, c_ = L.l32 & 3;
    const int pi = 16 * (a_ >> 1) + 8 * h_ + 4 * (a_ & 1) + c_;
    const int kswz = (pi >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) L.koff[ks] = pi * 128 + (((2 * ks + L.hi) ^ kswz) << 4);
    const int vswz = (L.l32 >> 1) & 7;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) L.voff[t][s] = FA_TILE + L.l32 * 128 + (((4 * t + 2 * s + L.hi) ^ vswz) << 4);  // chunk = 4t+2s+hi
    return L;
}

// keys >= S of the ragged last tile score -inf.  sc[t][r] = score(q = l32, key = 64j + 32t + 16(r>>3) + 8hi + (r&7))
AE_DEV void fa_mask_tail(f32x16 (&sc)[2], int j, int hi, int S) {
    const int kb = j * FA_KVBLK + 8 * hi;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kb + 32 * t + 16 * (r >> 3) + (r & 7) >= S) sc[t][r] = -INFINITY;
}

// soft-max of one 32 x 64 score tile held as sc[2] -> bf16 P fragments pf[t][s] (B operand of P·V).
// FAST: p = exp2(s), no maximum.  Otherwise exact online soft-max: m_run/l_run/o are rescaled when a row maximum grew.
template <bool FAST>
AE_DEV void fa_softmax(const f32x16 (&sc)[2], bf16x8 (&pf)[2][2], f32x16 (&o)[2], float& m_run, float& l_run) {
    float shift = 0.f;
    if (!FAST) {
        float mx = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the first tile (m_run = -inf)
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
            l_run *= alpha;
            m_run = m_new;
        }
        shift = m_run;
    }
    float psum[4] = {0.f, 0.f, 0.f, 0.f};   // four independent add chains
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pv[e] = __builtin_amdgcn_exp2f(FAST ? sc[t][8 * s + e] : sc[t][8 * s + e] - shift);
                psum[e & 3] += pv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[t][s][e] = (__bf16)pv[e];
        }
    l_run += (psum[0] + psum[1]) + (psum[2] + psum[3]);
}

// per-wave decision: every row of this wave has (‖q‖·max‖k‖)² within the bounded-score limit
AE_DEV bool fa_fast_ok(const bf16x8 (&qf)[4], const float* kmax2, int bh, int ntiles) {
    if (kmax2 == nullptr) return false;
    float km = 0.f;
    for (int i = threadIdx.x & 63; i < ntiles; i += 64) km = fmaxf(km, kmax2[(size_t)bh * ntiles + i]);
    km = wave_max(km);
    float qn2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float q = (float)qf[ks][e]; qn2 += q * q; }
    qn2 += __shfl_xor(qn2, 32, 64);
    return __all(qn2 * km * FA_BOUND_SLACK <= FA_FAST_BOUND2) != 0;
}

// epilogue: O[q][h*64 + d], d = 32dt + 8(r>>2) + 4hi + (r&3)
template <bool WIDE_STORE>
AE_DEV void fa_store(const f32x16 (&o)[2], float l_run, const FlashArgs& p, int bh, int qrow, int hi) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int b = bh / p.H, h = bh - b * p.H;
    const bool q_ok = qrow < p.S;
    const int qrow_c = min(qrow, p.S - 1);
    bf16_t* orow = p.O + ((size_t)b * p.S + qrow_c) * (size_t)(p.H * FA_D) + h * FA_D;
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

template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... Is> AE_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ic<Is>{}), ...); }
template <int N, class F> AE_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- guarded static shift (the lock-step kernel's soft-max; exact, shift-invariant) -----------------------------------
// Soft-max is invariant under ANY per-row shift c:  o = Σ exp2(s−c)·v / Σ exp2(s−c).  The online algorithm uses c = the
// running maximum only to keep exp2 in range.  Here the shift of a row is a value m that is a TRUE score maximum of the tiles
// on which it was last refreshed (so the row's largest term is ≥ 1: no underflow of the sums), and a tile may be
// exponentiated against the un-refreshed m — no maximum, no subtraction (m enters through the C operand of the first QKᵀ
// MFMA: sc = K·Qᵀ − m for free), no rescale — whenever a cheap bound proves exp2 cannot overflow on it:
//     |s| ≤ ‖q‖·max_tile‖k‖  (Cauchy–Schwarz; max‖k‖² per 64-key tile comes from aether_qk_norm_rope)
//     ‖q‖²·max‖k‖² ≤ (m + FA_SHIFT_SPAN)²  with  m + FA_SHIFT_SPAN > 0   ⇒   s − m ≤ FA_SHIFT_SPAN  for every key of the tile.
// p ≤ 2^100 is a normal fp32 / bf16 number and the sums stay below 2^100 · S · max|v| < 2^127 for S·max|v| < 2^27.  A tile that fails the test (the
// first tile of every row, and any tile whose keys could exceed the span) takes the refresh path: tile maximum, m ← max,
// conditional rescale of o and l — the classic online step.  The choice is per wave and per tile, wave-uniform, and changes
// only speed: results are those of an exact soft-max in fp32 either way.  AETHER_ATTN_EXACT_MAX (no bound table) refreshes on
// every tile.
constexpr float FA_SHIFT_SPAN = 100.f;
constexpr int FA_KMAX_SLOTS = 1024;   // per-tile bounds of one (batch, head) staged in LDS: S <= 65 536 (longer rows refresh every tile)

AE_DEV float fa_row_norm2(const bf16x8 (&qf)[4]) {
    float qn2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float q = (float)qf[ks][e]; qn2 += q * q; }
    return (qn2 + __shfl_xor(qn2, 32, 64)) * FA_BOUND_SLACK;
}

AE_DEV float fa_tile_max(const f32x16 (&sc)[2]) {
    float a = fmaxf(sc[0][0], sc[1][0]), b = fmaxf(sc[0][1], sc[1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
        a = fmaxf(a, fmaxf(sc[0][r], sc[1][r]));          // v_max3_f32
        b = fmaxf(b, fmaxf(sc[0][r + 1], sc[1][r + 1]));
    }
    a = fmaxf(a, b);
    return fmaxf(a, __shfl_xor(a, 32, 64));
}

// exponentiate a score tile that already carries its shift; P fragments + partial row sum
AE_DEV void fa_exp_tile(const f32x16 (&sc)[2], bf16x8 (&pf)[2][2], float& l_run) {
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pv[e] = __builtin_amdgcn_exp2f(sc[t][8 * s + e]);
                psum[e & 3] += pv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[t][s][e] = (__bf16)pv[e];
        }
    l_run += (psum[0] + psum[1]) + (psum[2] + psum[3]);
}

// =================================================================================================================
// lock-step kernel: one barrier per KV tile, double-buffered LDS (32 KiB), 2 workgroups per CU
// =================================================================================================================
// NW waves = NW*32 query rows per workgroup (8 or 4).  PRIO 1 = s_setprio 1 around the two MFMA clusters (a wave finishes its
// MFMA burst instead of interleaving with the other waves' soft-max VALU, which does not overlap with it anyway): +1.5 %
// (profiles/r01_attn_variants_v3.json; around the soft-max instead: +0.8 %).
// ILV = 1: in a tile that passes the guard the soft-max VALU work is interleaved, inside the wave, with that wave's own MFMAs
// (QK^T of the second 32-key half under the exponentials of the first, P·V of the first half under the exponentials of the second):
// VALU issued between a wave's own MFMAs hides under them, VALU of the OTHER waves of the SIMD mostly does not
// (profiles/r01_valu_probe.jsonl: 680 vs 1027 cycles for 16 MFMAs + one tile's soft-max).
template <bool WIDE_STORE, int NW, int PRIO = 1, int ILV = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))   // <= 128 VGPRs: 16 waves per CU
void flash_attn_fwd_kernel(FlashArgs p) {
    // 2 x (K tile + V^T tile) + this workgroup's Q fragments (4 KiB per wave, lane-linear: conflict-free ds_read_b128).  Q lives in
    // LDS, not in 16 registers per lane: the registers hold the soft-max shift vector instead (see below) and the kernel stays
    // within the 128-register budget of 4 waves per SIMD without spilling.
    __shared__ __attribute__((aligned(16))) char smem[2 * FA_BUF + NW * 4096 + FA_KMAX_SLOTS * 4];
    const FaLane L = fa_lane_setup();
    const int tid = threadIdx.x, hi = L.hi;

    const int wgid = xcd_remap(blockIdx.x, p.nwg) + p.wg_first;
    const int bh = wgid / p.nqb;
    const int qb = wgid - bh * p.nqb;
    const int S = p.S;

    const bf16_t* Qg = p.Q + (size_t)bh * S * FA_D;
    const bf16_t* Kg = p.K + (size_t)bh * S * FA_D;
    const bf16_t* Vg = p.Vt + (size_t)bh * FA_D * p.Spad;

    // ---- Q fragment (B operand): lane (q = l32, hi) holds Q[q][16ks + 8hi .. +7] -------------------
    if (qb * (NW * 32) >= S) return;   // second half of a ragged last 256-row block may be empty (whole workgroup exits)
    const int qrow = qb * (NW * 32) + L.wave * 32 + L.l32;
    const int qrow_c = min(qrow, S - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + (size_t)qrow_c * FA_D + 16 * ks + 8 * hi);

    // ---- staging (512 16-byte pieces of K and of Vᵀ per KV tile: one each per thread at 8 waves, two at 4) ----
    // buffer-descriptor DMA: the per-lane offsets are loop invariant, the tile advance is a scalar offset; K rows >= S
    // of the ragged last tile are out of range of the descriptor (not fetched; their scores are masked below)
    const int srow = tid >> 3;                          // K: key row in tile; Vᵀ: d row
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);   // logical chunk fetched into physical chunk tid&7
    const int wave_s = __builtin_amdgcn_readfirstlane(L.wave);
    char* const lds_stage = smem + wave_s * 1024;
    const buf_rsrc_t k_rsrc = make_buf_rsrc(Kg, (unsigned)S * FA_D * 2);
    const buf_rsrc_t v_rsrc = make_buf_rsrc(Vg, (unsigned)p.Spad * FA_D * 2);
    const unsigned k_voff = srow * (FA_D * 2) + schunk * 16;
    const unsigned v_voff = (unsigned)srow * p.Spad * 2 + schunk * 16;
    constexpr int PASSES = 8 / NW;      // row r and r + 32*pass share the swizzle phase ((r >> 1) & 7)
    auto stage = [&](int j, int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            bglds16(k_rsrc, k_voff + ps * (NW * 8) * (FA_D * 2), j * (FA_KVBLK * FA_D * 2), lds_stage + buf * FA_BUF + ps * (NW * 1024));
            bglds16(v_rsrc, v_voff + ps * (NW * 8) * (unsigned)p.Spad * 2, j * (FA_KVBLK * 2), lds_stage + buf * FA_BUF + FA_TILE + ps * (NW * 1024));
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = 0.f;        // the row's current shift (log2 domain; both lanes of a row hold the same value)
    float l_run = 0.f;        // this lane's partial row sum
    float thr2 = -1.f;        // (m_run + FA_SHIFT_SPAN)^2 when that base is positive, else -1: a tile passes iff qn2*kmax2 <= thr2
    f32x16 negm;              // -m_run in all 16 elements: the C operand of the first QK^T MFMA of every tile
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;

    const int nkv = (S + FA_KVBLK - 1) / FA_KVBLK;
    const bool ragged = (S & (FA_KVBLK - 1)) != 0;
    stage(0, 0);
    const float qn2 = fa_row_norm2(qf);
    char* const qs = smem + 2 * FA_BUF + L.wave * 4096 + L.lane * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) *(bf16x8*)(qs + ks * 1024) = qf[ks];
    // max||k||^2 of every KV tile of this (batch, head) -> LDS (the per-tile guard reads it with one broadcast ds_read_b32: a
    // global load inside the loop would share vmcnt with the K/V DMA and serialise it)
    float* const kms = (float*)(smem + 2 * FA_BUF + NW * 4096);
    const bool bounded = p.kmax2 != nullptr && nkv <= FA_KMAX_SLOTS;
    if (bounded)
        for (int i = tid; i < nkv; i += NW * 64) kms[i] = p.kmax2[(size_t)bh * (p.Spad / FA_KVBLK) + i];
    drain_and_barrier();

    auto tile = [&](int j, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;     // the last tile: nothing to stage, ragged tail masked
        const int cur = j & 1;
        if (!LAST) stage(j + 1, cur ^ 1);
        const char* base = smem + cur * FA_BUF;
        const float km2 = bounded ? kms[j] : INFINITY;

        f32x16 sc[2];
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 qv = *(const bf16x8*)(qs + ks * 1024);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 kf = *(const bf16x8*)(base + t * 4096 + L.koff[ks]);
                sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qv, ks == 0 ? negm : sc[t], 0, 0, 0);
            }
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (LAST && ragged) fa_mask_tail(sc, j, hi, S);

        if (__builtin_expect(!__all(qn2 * km2 <= thr2), 0)) {
            // refresh (tile 0 of every row; otherwise rare): sc holds s - m_run; bring the shift up to this tile's maximum, in place
            const float rel = fa_tile_max(sc);
            const float up = (j == 0) ? rel : fmaxf(rel, 0.f);   // first tile: the shift becomes the tile's true maximum
            if (__any(up != 0.f)) {
                if (j > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
                    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
                    l_run *= alpha;
                }
                m_run += up;
                const float nb = m_run + FA_SHIFT_SPAN;
                thr2 = nb > 0.f ? nb * nb : -1.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sc[0][i] -= up; sc[1][i] -= up; negm[i] = -m_run; }
            }
        }
        bf16x8 pf[2][2];
        fa_exp_tile(sc, pf, l_run);                               // sc = s - m_run <= FA_SHIFT_SPAN (<= 0 after a refresh)

        // ---- O^T += V^T . P^T ----
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf = *(const bf16x8*)(base + dt * 4096 + L.voff[t][s]);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[t][s], o[dt], 0, 0, 0);
                }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (!LAST) drain_and_barrier();
    };
    // ---- interleaved steady-state tile (guard already checked, not the last tile) ---------------------------------------
    [[maybe_unused]] auto tile_ilv = [&](int j) {
        const int cur = j & 1;
        stage(j + 1, cur ^ 1);
        const char* base = smem + cur * FA_BUF;
        auto Kf = [&](int t, int ks) { return *(const bf16x8*)(base + t * 4096 + L.koff[ks]); };
        auto Qf = [&](int ks) { return *(const bf16x8*)(qs + ks * 1024); };
        auto Vf = [&](int dt, int t, int s2) { return *(const bf16x8*)(base + dt * 4096 + L.voff[t][s2]); };
        f32x16 s0, s1;
        bf16x8 pf[2][2];
        // (row sums as v_pk_fma_f32 with a register of ones — 6 cycles per two elements in isolation — measured 3 % SLOWER here than
        // plain v_add_f32, as in round 1's lock-step kernel: profiles/r02_attn_variants.txt)
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        // one quarter (4 scores) of a 32-key half: exp2, row sum, bf16 P fragment elements
        auto quarter = [&](const f32x16& sc, int t, int q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(sc[4 * q + e]);
                ps[e] += pv;
                asm volatile("" : "+v"(ps[e]));      // keep the row-sum add inside this group (IR passes re-associate and sink it otherwise)
                pf[t][q >> 1][4 * (q & 1) + e] = (__bf16)pv;
            }
        };
        bf16x8 fa = Kf(0, 0), fq = Qf(0);
        // segment 1: QK^T of keys 0..31 (fragments fetched one MFMA ahead)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 na = ks < 3 ? Kf(0, ks + 1) : Kf(1, 0), nq = Qf((ks + 1) & 3);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fq, ks == 0 ? negm : s0, 0, 0, 0);
            fa = na; fq = nq;
        }
        __builtin_amdgcn_sched_barrier(0);
        // segment 2: QK^T of keys 32..63 || soft-max of keys 0..31
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 na = ks < 3 ? Kf(1, ks + 1) : Vf(0, 0, 0), nq = Qf((ks + 1) & 3);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fq, ks == 0 ? negm : s1, 0, 0, 0);
            fa = na; fq = nq;
            __builtin_amdgcn_sched_barrier(0);
            quarter(s0, 0, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        // segment 3: P·V of keys 0..31 || soft-max of keys 32..63
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s2 = g >> 1, dt = g & 1;
            const bf16x8 na = g < 3 ? Vf((g + 1) & 1, 0, (g + 1) >> 1) : Vf(0, 1, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[0][s2], o[dt], 0, 0, 0);
            fa = na;
            __builtin_amdgcn_sched_barrier(0);
            quarter(s1, 1, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        // segment 4: P·V of keys 32..63
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s2 = g >> 1, dt = g & 1;
            const bf16x8 na = g < 3 ? Vf((g + 1) & 1, 1, (g + 1) >> 1) : fa;
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[1][s2], o[dt], 0, 0, 0);
            fa = na;
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        drain_and_barrier();
    };
    // ---- ILV = 2: two tiles per iteration, soft-max of each 32-key half spread over the MFMAs of its neighbours -----------------------
    // Per pair of tiles (halves h0,h1 of tile j, h2,h3 of tile j+1) the wave issues, in this order (MFMA group ∥ VALU it hides):
    //   QK h0 | QK h1 ∥ sm h0 a | QK h2 ∥ sm h0 b | PV h0 ∥ sm h1 a | QK h3 ∥ sm h1 b | PV h1 ∥ sm h2 a | X | sm h2 b | PV h2 ∥ sm h3 a | sm h3 b | PV h3 | Y
    // so 24 of the 32 MFMAs run with ~40 cycles of the wave's own soft-max VALU behind each of them (the one-tile variant pairs
    // 8 of 16); ≈ 730 issue cycles per tile against ≈ 860.  Three score halves are live at the peak, which leaves no room for the
    // shift vector: the loop runs only when EVERY wave of the workgroup is bounded outright (||q||·max||k|| <= 100 over the whole
    // head: exp2(s) cannot overflow with shift 0) — its barriers (X: tile j and K(j+1) consumed -> DMA of K,V(j+2), K(j+3);
    // Y: V(j+1) consumed -> DMA of V(j+3)) differ from the one-per-tile pattern, so the choice must be workgroup-uniform.  Any other
    // workgroup takes the guarded one-tile paths above.
    [[maybe_unused]] auto stage_k = [&](int j, int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
            bglds16(k_rsrc, k_voff + ps * (NW * 8) * (FA_D * 2), j * (FA_KVBLK * FA_D * 2), lds_stage + buf * FA_BUF + ps * (NW * 1024));
    };
    [[maybe_unused]] auto stage_v = [&](int j, int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
            bglds16(v_rsrc, v_voff + ps * (NW * 8) * (unsigned)p.Spad * 2, j * (FA_KVBLK * 2), lds_stage + buf * FA_BUF + FA_TILE + ps * (NW * 1024));
    };
    [[maybe_unused]] auto pair_loop = [&](int npairs) {
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        // K(1) -> buffer 1 now, V(1) after the barrier that publishes it (K(0), V(0) were staged and published by the prologue)
        stage_k(1, 1);
        drain_and_barrier();
        stage_v(1, 1);
        for (int pi = 0; pi < npairs; ++pi) {
            const int j = 2 * pi;
            f32x16 S[4];
            bf16x8 pf[4][2];
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // A-operand fragment of MFMA step i (0..31): QK h0 0-3, QK h1 4-7, QK h2 8-11, PV h0 12-15, QK h3 16-19, PV h1 20-23, PV h2 24-27, PV h3 28-31
            auto afrag = [&](auto I) -> bf16x8 {
                constexpr int i = decltype(I)::value;
                constexpr int grp = i >> 2, m = i & 3;
                constexpr int h = (grp == 0) ? 0 : (grp == 1) ? 1 : (grp == 2) ? 2 : (grp == 3) ? 0 : (grp == 4) ? 3 : (grp == 5) ? 1 : (grp == 6) ? 2 : 3;
                constexpr bool qk = (grp == 0 || grp == 1 || grp == 2 || grp == 4);
                const char* base = smem + (h >> 1) * FA_BUF;              // tile j in buffer 0, tile j+1 in buffer 1 (j is even)
                if constexpr (qk) return *(const bf16x8*)(base + (h & 1) * 4096 + L.koff[m]);
                else return *(const bf16x8*)(base + (m & 1) * 4096 + L.voff[h & 1][m >> 1]);      // dt = m&1, s2 = m>>1
            };
            // one eighth (2 scores) of a half's soft-max
            auto eighth = [&](auto H, auto E) {
                constexpr int h = decltype(H)::value, e = decltype(E)::value;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float pv = __builtin_amdgcn_exp2f(S[h][2 * e + u]);
                    ps[2 * (e & 1) + u] += pv;
                    asm volatile("" : "+v"(ps[2 * (e & 1) + u]));
                    pf[h][e >> 2][2 * (e & 3) + u] = (__bf16)pv;
                }
            };
            // A-operand fragments are fetched TWO MFMAs ahead (fa: this step, fb: next step, loaded now: the step after), Q fragments one
            bf16x8 fa = afrag(ic<0>{}), fb = afrag(ic<1>{}), fq = *(const bf16x8*)(qs);
            static_for<24>([&](auto I) {                                   // steps 0..23 (up to barrier X)
                constexpr int i = decltype(I)::value;
                constexpr int grp = i >> 2, m = i & 3;
                constexpr bool qk = (grp == 0 || grp == 1 || grp == 2 || grp == 4);
                constexpr int h = (grp == 0) ? 0 : (grp == 1) ? 1 : (grp == 2) ? 2 : (grp == 3) ? 0 : (grp == 4) ? 3 : 1;
                bf16x8 nb = fb, nq = fq;
                if constexpr (i + 2 < 24) nb = afrag(ic<i + 2>{});
                if constexpr (i + 1 < 24) {
                    constexpr int g2 = (i + 1) >> 2;
                    if constexpr (g2 == 0 || g2 == 1 || g2 == 2 || g2 == 4) nq = *(const bf16x8*)(qs + ((i + 1) & 3) * 1024);
                }
                if constexpr (qk) S[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fq, m == 0 ? zero : S[h], 0, 0, 0);
                else o[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[h][m >> 1], o[m & 1], 0, 0, 0);
                fa = fb; fb = nb; fq = nq;
                __builtin_amdgcn_sched_barrier(0);
                // VALU partner: groups 1,2 -> sm h0 (eighths 0-3, 4-7); 3,4 -> sm h1; 5 -> sm h2 first half
                if constexpr (grp == 1) eighth(ic<0>{}, ic<m>{});
                if constexpr (grp == 2) eighth(ic<0>{}, ic<4 + m>{});
                if constexpr (grp == 3) eighth(ic<1>{}, ic<m>{});
                if constexpr (grp == 4) eighth(ic<1>{}, ic<4 + m>{});
                if constexpr (grp == 5) eighth(ic<2>{}, ic<m>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            // X: tile j and K(j+1) are consumed by every wave; V(j+1) (issued at the previous Y) has landed
            drain_and_barrier();
            if (j + 2 < nkv) { stage_k(j + 2, 0); stage_v(j + 2, 0); }
            if (j + 3 < nkv) stage_k(j + 3, 1);
            fa = afrag(ic<24>{}); fb = afrag(ic<25>{});
            static_for<4>([&](auto E) { eighth(ic<2>{}, ic<4 + decltype(E)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto M) {                                    // PV h2 ∥ sm h3 a
                constexpr int m = decltype(M)::value;
                const bf16x8 nb = afrag(ic<26 + m>{});                     // step 24+m+2 (26..29)
                o[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[2][m >> 1], o[m & 1], 0, 0, 0);
                fa = fb; fb = nb;
                __builtin_amdgcn_sched_barrier(0);
                eighth(ic<3>{}, ic<m>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<4>([&](auto E) { eighth(ic<3>{}, ic<4 + decltype(E)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto M) {                                    // PV h3 (fa = step 28, fb = step 29 already in flight)
                constexpr int m = decltype(M)::value;
                bf16x8 nb = fb;
                if constexpr (m < 2) nb = afrag(ic<30 + m>{});
                o[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[3][m >> 1], o[m & 1], 0, 0, 0);
                fa = fb; fb = nb;
            });
            // Y: V(j+1) consumed; K,V(j+2), K(j+3) have landed
            drain_and_barrier();
            if (j + 3 < nkv) stage_v(j + 3, 1);
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if (ILV == 2) {
        // workgroup vote: every row of every wave bounded outright?  (max over the head's tiles of max||k||^2, then ||q||^2 of each row)
        float* const votes = kms + FA_KMAX_SLOTS - 8;                      // the last 8 slots of the bound table (nkv <= FA_KMAX_SLOTS - 8 checked)
        float kall = 0.f;
        const bool can_vote = bounded && nkv <= FA_KMAX_SLOTS - 8 && nkv >= 4;
        if (can_vote) {
            for (int i = L.lane; i < nkv; i += 64) kall = fmaxf(kall, kms[i]);
            kall = wave_max(kall);
        }
        const bool mine = can_vote && __all(qn2 * kall <= FA_SHIFT_SPAN * FA_SHIFT_SPAN) != 0;
        block_barrier();                                                    // every wave has read its slice of kms
        if (L.lane == 0) votes[L.wave] = mine ? 1.f : 0.f;
        block_barrier();
        bool all_ok = true;
#pragma unroll
        for (int w = 0; w < NW; ++w) all_ok = all_ok && (votes[w] != 0.f);
        int j = 0;
        if (all_ok) {
            const int npairs = (nkv - 1) / 2;                               // the last tile (ragged tail, no staging) stays with the generic tile
            pair_loop(npairs);
            j = 2 * npairs;
            thr2 = FA_SHIFT_SPAN * FA_SHIFT_SPAN;                           // shift 0 stays valid for the one or two tiles left
            if (j < nkv - 1) {                                              // K,V(j) are in place and published; the generic tile stages j+1 itself
                tile(j, std::false_type{});
                ++j;
            }
        } else {
            // not bounded outright: the guarded one-tile paths (tile 0 refreshes, passing tiles interleaved, generic after a failure)
            if (nkv > 1) { tile(0, std::false_type{}); j = 1; }
            float km = (bounded && j < nkv) ? kms[j] : INFINITY;
            for (; j < nkv - 1; ++j) {
                if (!__all(qn2 * km <= thr2)) break;
                km = kms[j + 1];
                asm volatile("" : "+v"(km));
                tile_ilv(j);
            }
            for (; j < nkv - 1; ++j) tile(j, std::false_type{});
        }
    } else if (ILV == 1) {
        // Tile 0 always refreshes.  After it, passing tiles run in the interleaved loop; the first tile that fails the guard sends
        // the wave to the generic loop for the rest of its sweep (exact as well, just not interleaved).
        int j = 0;
        if (nkv > 1) { tile(0, std::false_type{}); j = 1; }
        float km = (bounded && j < nkv) ? kms[j] : INFINITY;          // bound of the tile about to run, fetched one tile ahead
        for (; j < nkv - 1; ++j) {
            if (!__all(qn2 * km <= thr2)) break;
            km = kms[j + 1];
            asm volatile("" : "+v"(km));                                // keep the LDS read here, a whole tile ahead of its use
            tile_ilv(j);
        }
        for (; j < nkv - 1; ++j) tile(j, std::false_type{});
    } else {
        for (int j = 0; j < nkv - 1; ++j) tile(j, std::false_type{});
    }
    tile(nkv - 1, std::true_type{});

    fa_store<WIDE_STORE>(o, l_run, p, bh, qrow, hi);
}

// =================================================================================================================
// software-pipelined kernel: one workgroup per CU (two waves per SIMD), soft-max of tile j interleaved, inside each
// wave, with the MFMAs of P·V(j-1) and K(j+1)·Qᵀ
// =================================================================================================================
// Measured on MI355X (tools/probes/valu_probe.hip, profiles/r01_valu_probe.jsonl): VALU work issued by the SAME wave
// between its MFMAs hides under them (16 MFMA + 32 v_exp + 32 v_add + 16 v_cvt_pk interleaved: 680 cycles against 512
// for the bare MFMAs), whereas the same VALU work issued by the OTHER wave of the SIMD serialises with the MFMAs
// (1027 cycles) — so the overlap has to be built inside each wave: the loop body carries two score tiles and two P
// fragments (sc/pf of tile j being soft-maxed, sc of tile j+1 being produced, pf of tile j-1 being consumed).
constexpr int FA_NB = 4;  // K/V ring depth (tile t lives in slot t & 3)


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


// Timing ablations of this kernel (profiles/r01_attn_swp_ablation.json, shader cycles per KV tile for the two waves of
// a SIMD): full 1721, without the LDS-DMA 1623, without the soft-max VALU 1203 (16 MFMAs = 2 x 512); fragment reads 2 / 4 /
// 6 / 8 MFMAs ahead: 1720 / 1671 / 1703 / 1760 (profiles/r01_attn_swp_prefetch.json).
constexpr int FA_AHEAD = 4;   // fragment reads are issued this many MFMAs ahead of their use

template <bool WIDE_STORE>
__global__ __launch_bounds__(512) void flash_attn_swp_kernel(FlashArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[FA_NB * FA_BUF];
    const FaLane L = fa_lane_setup();
    const int tid = threadIdx.x, hi = L.hi;
    const int wave_s = __builtin_amdgcn_readfirstlane(L.wave);

    const int wgid = xcd_remap(blockIdx.x, p.nwg);
    const int bh = wgid / p.nqb;
    const int qb = wgid - bh * p.nqb;
    const int S = p.S;

    const bf16_t* Qg = p.Q + (size_t)bh * S * FA_D;
    const bf16_t* Kg = p.K + (size_t)bh * S * FA_D;
    const bf16_t* Vg = p.Vt + (size_t)bh * FA_D * p.Spad;

    const int qrow = qb * FA_QBLK + L.wave * 32 + L.l32;
    const int qrow_c = min(qrow, S - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + (size_t)qrow_c * FA_D + 16 * ks + 8 * hi);

    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    char* const lds_stage = smem + wave_s * 1024;
    const int nkv = (S + FA_KVBLK - 1) / FA_KVBLK;
    // buffer-descriptor DMA (scalar tile advance, no per-tile VALU).  Tiles past the end re-fetch the last tile (finite
    // data) into a slot nobody needs any more: the loop stays branch free with uniform vmcnt accounting; the scores of
    // such a tile — like those of the keys >= S of the ragged last tile, which the K descriptor does not cover — are
    // masked to -inf, so they contribute exactly 0.
    const buf_rsrc_t k_rsrc = make_buf_rsrc(Kg, (unsigned)S * FA_D * 2);
    const buf_rsrc_t v_rsrc = make_buf_rsrc(Vg, (unsigned)p.Spad * FA_D * 2);
    const unsigned k_voff = srow * (FA_D * 2) + schunk * 16;
    const unsigned v_voff = (unsigned)srow * p.Spad * 2 + schunk * 16;
    auto stage_k = [&](int j, int slot) { bglds16(k_rsrc, k_voff, min(j, nkv - 1) * (FA_KVBLK * FA_D * 2), lds_stage + slot * FA_BUF); };
    auto stage_v = [&](int j, int slot) { bglds16(v_rsrc, v_voff, min(j, nkv - 1) * (FA_KVBLK * 2), lds_stage + slot * FA_BUF + FA_TILE); };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    stage_k(0, 0); stage_v(0, 0); stage_k(1, 1); stage_v(1, 1); stage_k(2, 2);
    const bool fast = fa_fast_ok(qf, p.kmax2, bh, p.Spad / FA_KVBLK);
    drain_and_barrier();

    auto sweep = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        f32x16 sc_a[2], sc_b[2];
        u32x4 pf_a[2][2], pf_b[2][2];   // P fragments as packed bf16 pairs
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) pf_b[t][s][e] = 0u;
        // prologue: scores of tile 0
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc_a[t][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                sc_a[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + t * 4096 + L.koff[ks]), qf[ks], sc_a[t], 0, 0, 0);
        }
        f32x2 lsum[2] = {{0.f, 0.f}, {0.f, 0.f}};   // FAST: packed partial row sums (v_pk_fma_f32 with 1.0)
        f32x2 ones = {1.f, 1.f};
        asm volatile("" : "+v"(ones));   // opaque: keeps v_pk_fma_f32 (6 cycles / wave) from being folded to v_pk_add_f32 (11.6)

        // One iteration = tile j (ring slots are compile-time: the loop is unrolled by 4):
        //   MFMA  o += Vᵀ(j-1)·pf_prev (8) then sc_nxt = K(j+1)·Qᵀ (8);   VALU  pf_cur = softmax(sc_cur)
        // LDS ring at iteration j: Vᵀ(j-1), [Vᵀ(j)], K(j+1) are read/live; K(j+3) -> K half of slot (j-1)&3 (last read at
        // iteration j-2) and Vᵀ(j+2) -> V half of slot (j-2)&3 (last read at iteration j-1) are issued now and waited for
        // one iteration later (vmcnt(2)): K lands two iterations ahead of its use, Vᵀ three.
        auto iter = [&](int j, auto slot_tag, f32x16 (&sc_cur)[2], f32x16 (&sc_nxt)[2], u32x4 (&pf_prev)[2][2], u32x4 (&pf_cur)[2][2]) {
            constexpr int SL = decltype(slot_tag)::value;          // j & 3
            constexpr int KSLOT = (SL + 1) & 3, VSLOT = (SL + 3) & 3;   // slots of tile j+1 and tile j-1
            if constexpr (!FAST) {
                stage_k(j + 3, VSLOT);                              // (j+3)&3 == (j-1)&3
                stage_v(j + 2, (SL + 2) & 3);
            }
            if ((j + 1) * FA_KVBLK > S) fa_mask_tail(sc_cur, j, hi, S);
            // j = 0: pf_prev is 0 and slot 3 is still uninitialised LDS (0 x NaN): read tile 0's Vᵀ instead
            const char* kbase = smem + KSLOT * FA_BUF;
            const char* vbase = smem + ((SL == 0 && j == 0) ? 0 : VSLOT) * FA_BUF;
            // fragment i of the 16 MFMAs: 0-7 Vᵀ(dt = i&1, t = i>>2, s = (i>>1)&1), 8-15 K(t = i&1, ks = (i-8)>>1)
            auto frag = [&](auto I) -> bf16x8 {
                constexpr int i = decltype(I)::value;
                if constexpr (i < 8) return *(const bf16x8*)(vbase + (i & 1) * 4096 + L.voff[i >> 2][(i >> 1) & 1]);
                else return *(const bf16x8*)(kbase + (i & 1) * 4096 + L.koff[(i - 8) >> 1]);
            };
            if constexpr (FAST) {
                // hand interleave, pinned by sched_barrier: per MFMA one fragment read (two ahead) and the soft-max of two
                // scores (2 v_exp, 1 v_pk_fma row sum, 1 v_cvt_pk)
                bf16x8 fr[16];
                static_for<FA_AHEAD>([&](auto I) { fr[decltype(I)::value] = frag(I); });
                static_for<16>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i + FA_AHEAD < 16) fr[i + FA_AHEAD] = frag(ic<i + FA_AHEAD>{});
                    // the two DMA pieces of this iteration go out in the shadow of the MFMAs, not at the barrier
                    if constexpr (i == 2) stage_k(j + 3, VSLOT);            // (j+3)&3 == (j-1)&3
                    if constexpr (i == 9) stage_v(j + 2, (SL + 2) & 3);
                    if constexpr (i < 8) {
                        o[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], __builtin_bit_cast(bf16x8, pf_prev[i >> 2][(i >> 1) & 1]), o[i & 1], 0, 0, 0);
                    } else {
                        constexpr int t = i & 1, ks = (i - 8) >> 1;
                        if constexpr (ks == 0) {
                            f32x16 z;
#pragma unroll
                            for (int e = 0; e < 16; ++e) z[e] = 0.f;
                            sc_nxt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[ks], z, 0, 0, 0);
                        } else {
                            sc_nxt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[ks], sc_nxt[t], 0, 0, 0);
                        }
                    }
                    // scores 2i, 2i+1 of this lane: tile t = i>>3, slab s = (i>>2)&1, elements e = 2(i&3), +1
                    constexpr int t = i >> 3, s = (i >> 2) & 1, e = 2 * (i & 3);
                    f32x2 pp;
                    pp[0] = __builtin_amdgcn_exp2f(sc_cur[t][8 * s + e]);
                    pp[1] = __builtin_amdgcn_exp2f(sc_cur[t][8 * s + e + 1]);
                    lsum[i & 1] = __builtin_elementwise_fma(pp, ones, lsum[i & 1]);
                    unsigned w = pack_bf16x2(pp[0], pp[1]);
                    // the results are first USED one iteration later: without these anchors the compiler sinks the whole
                    // soft-max past the barrier, next to its use, and the interleave is gone
                    asm volatile("" : "+v"(w), "+v"(lsum[i & 1]));
                    pf_cur[t][s][e / 2] = w;
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                // exact online soft-max: compiler-scheduled (the MFMAs below do not depend on this tile's soft-max)
                static_for<16>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const bf16x8 f = frag(I);
                    if constexpr (i < 8) {
                        o[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, __builtin_bit_cast(bf16x8, pf_prev[i >> 2][(i >> 1) & 1]), o[i & 1], 0, 0, 0);
                    } else {
                        constexpr int t = i & 1, ks = (i - 8) >> 1;
                        if constexpr (ks == 0) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) sc_nxt[t][e] = 0.f;
                        }
                        sc_nxt[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[ks], sc_nxt[t], 0, 0, 0);
                    }
                });
                bf16x8 pfx[2][2];
                fa_softmax<false>(sc_cur, pfx, o, m_run, l_run);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        pf_cur[t][s] = __builtin_bit_cast(u32x4, pfx[t][s]);
                        asm volatile("" : "+v"(pf_cur[t][s]));   // keep the soft-max in this iteration (see above)
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            block_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };

        const int n4 = (nkv + 3) & ~3;   // the tile count is padded to a multiple of 4 with fully masked tiles
        for (int j = 0; j < n4; j += 4) {
            iter(j, ic<0>{}, sc_a, sc_b, pf_b, pf_a);
            iter(j + 1, ic<1>{}, sc_b, sc_a, pf_a, pf_b);
            iter(j + 2, ic<2>{}, sc_a, sc_b, pf_b, pf_a);
            iter(j + 3, ic<3>{}, sc_b, sc_a, pf_a, pf_b);
        }
        // epilogue: P·V of the last tile (pf_b), Vᵀ(n4-1) lives in slot 3
        {
            const char* vbase = smem + 3 * FA_BUF;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(vbase + dt * 4096 + L.voff[t][s]), __builtin_bit_cast(bf16x8, pf_b[t][s]), o[dt], 0, 0, 0);
        }
        if constexpr (FAST) l_run += (lsum[0][0] + lsum[0][1]) + (lsum[1][0] + lsum[1][1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail DMA must not outlive the workgroup's LDS
    };
    if (fast) sweep(std::true_type{});
    else sweep(std::false_type{});

    fa_store<WIDE_STORE>(o, l_run, p, bh, qrow, hi);
}

}  // namespace aether

using namespace aether;

extern "C" int aether_flash_attn_fwd(const void* Qh, const void* Kh, const void* Vt, void* O, int B, int H, int S,
                                     int Spad, const float* kmax2, int flags, void* stream) {
    if (!Qh || !Kh || !Vt || !O) return aether_set_error(AETHER_ERR_ARG, "flash_attn: null pointer");
    if (B <= 0 || H <= 0 || S <= 0) return aether_set_error(AETHER_ERR_SHAPE, "flash_attn: empty problem");
    if (Spad % FA_KVBLK != 0 || Spad < S) return aether_set_error(AETHER_ERR_SHAPE, "flash_attn: Spad must be roundup(S,64)");
    if (((uintptr_t)Qh | (uintptr_t)Kh | (uintptr_t)Vt | (uintptr_t)O) & 15)
        return aether_set_error(AETHER_ERR_ALIGN, "flash_attn: pointers must be 16-byte aligned");
    FlashArgs p;
    p.Q = (const bf16_t*)Qh; p.K = (const bf16_t*)Kh; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
    p.kmax2 = (flags & AETHER_ATTN_EXACT_MAX) ? nullptr : kmax2;
    p.H = H; p.S = S; p.Spad = Spad;
    p.nqb = (S + FA_QBLK - 1) / FA_QBLK;
    p.nwg = p.nqb * B * H;
    p.wg_first = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (flags & AETHER_GEMM_WIDE_STORE) != 0;
    dim3 grid(p.nwg), block(512);
    if (flags & AETHER_ATTN_PIPELINED) {
        if (wide) hipLaunchKernelGGL((flash_attn_swp_kernel<true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_attn_swp_kernel<false>), grid, block, 0, s, p);
    } else {
        // lock-step kernel: 512 resident workgroup slots (2 x 8 waves per CU at 128 VGPRs).  With AETHER_ATTN_TAIL_SPLIT whole
        // rounds run as 256-row workgroups and the remainder as twice as many 128-row workgroups (4 per CU); measured gain
        // <= 1 % (profiles/r01_attn_variants_v2.json), so the default is ONE launch of 256-row workgroups.
        const int slots = 512;
        const int full = (flags & AETHER_ATTN_TAIL_SPLIT) ? p.nwg / slots * slots : p.nwg;
        const int rest = p.nwg - full;
        if (full > 0) {
            p.nwg = full; p.wg_first = 0;
            if (flags & AETHER_ATTN_PAIR_PIPELINE) {
                if (wide) hipLaunchKernelGGL((flash_attn_fwd_kernel<true, 8, 1, 2>), dim3(full), dim3(512), 0, s, p);
                else hipLaunchKernelGGL((flash_attn_fwd_kernel<false, 8, 1, 2>), dim3(full), dim3(512), 0, s, p);
            } else if (flags & AETHER_ATTN_INTERLEAVE) {
                if (wide) hipLaunchKernelGGL((flash_attn_fwd_kernel<true, 8, 1, 1>), dim3(full), dim3(512), 0, s, p);
                else hipLaunchKernelGGL((flash_attn_fwd_kernel<false, 8, 1, 1>), dim3(full), dim3(512), 0, s, p);
            } else if (wide) hipLaunchKernelGGL((flash_attn_fwd_kernel<true, 8>), dim3(full), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((flash_attn_fwd_kernel<false, 8>), dim3(full), dim3(512), 0, s, p);
        }
        if (rest > 0) {
            p.nqb *= 2; p.nwg = 2 * rest; p.wg_first = 2 * full;
            if (wide) hipLaunchKernelGGL((flash_attn_fwd_kernel<true, 4>), dim3(2 * rest), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((flash_attn_fwd_kernel<false, 4>), dim3(2 * rest), dim3(256), 0, s, p);
        }
    }
    return aether_check_launch("flash_attn_fwd");
}
