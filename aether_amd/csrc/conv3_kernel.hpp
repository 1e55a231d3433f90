// Implicit-GEMM 3x3(x3) convolution with tap reuse along the width (the VAE's full- and half-resolution layers).
//
// CogVideoXCausalConv3d / the 3x3 convolutions of the up-samplers, reached through vae.encode / vae.decode
// (aether/pipelines/aetherv1_pipeline_cogvideox.py:557-618, 931, 936).  Same MFMA tile machinery as gemm_kernel.hpp
// (v_mfma_f32_32x32x16_bf16, swapped operands, XOR-swizzled 128-byte LDS rows, LDS-DMA through buffer descriptors,
// ping-pong main loop), with a different A operand:
//
//   * output rows enumerate the PADDED plane of every output frame:  m = (f*iH + hp)*iW + wp,  f = nb*oT + t,
//     hp < iH = oH+2, wp < iW = oW+2 (rows with hp >= oH or wp >= oW are computed and dropped: 1.4 % at 240x360).
//     In that enumeration the input voxel of tap (dt,dh,dw) for row m is  frame(nb, t+dt) + (m mod iH*iW) + dh*iW + dw:
//     the A tile of tap dw is the A tile of tap 0 shifted by dw rows.
//   * K order (dt, dh, channel block, dw), dw fastest (the order vae.py packs the weights in): the three K tiles of one
//     (dt, dh, channel block) share ONE staged A tile of BM+64 rows, read at row offsets 0, 1, 2.  LDS-DMA pieces per K tile
//     drop from BM/64 + BN/64 to (BM/64+1)/3 + BN/64 — issuing those pieces is what limits the plain gathered kernel
//     (profiles/r01_gemm_ablation.json).
//
// LDS: two A buffers (one per (dt,dh,cb) step) + two W buffers (one per K tile).  Instantiations:
//     <2,4,4,2>  256x256   (Cout % 256 == 0)     2*40 + 2*32 = 144 KiB
//     <4,2,4,2,W1>  512x128  (Cout % 128 == 0; round 6)      2*72 + 1*16 = 160 KiB — every byte of a CU's LDS
//     <8,1,2,1>  512x32    (conv_out: 3 output channels padded to 32; round 5)   2*72 + 2*4 = 152 KiB
//
// W1 (single W buffer): the 128-wide layers are 36 % of the VAE's time and the 384x128 tile gives a wave only 6 MFMAs per fragment
// set (96 x 64 per wave) where the 256-wide tile has 8 (128 x 64) — but 512 x 128 with the tap-reuse A tile (512 + 64 rows, twice) leaves
// 16 KiB for W: ONE tile.  A wave therefore pulls ALL the W fragments of a K tile into registers in its first two load slots
// (2 x NT x 2 = 8 reads, 32 VGPRs); once the later wave group has done so (end of global slot 3 of the tile) the W buffer is dead
// and the DMA of the next K tile's W goes into the same buffer in the third load slots; it has landed when the tile-end vmcnt(0) + barrier pass.
#pragma once
#include "gemm_kernel.hpp"

namespace aether {

template <int WM, int WN, int MT, int NT, int EPI, bool WIDE_STORE, bool W1 = false>
__global__ __launch_bounds__(512) void conv3_gemm_kernel(GemmArgs p) {
    static_assert(WM * WN == 8, "8 wavefronts per workgroup");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int A_ROUNDS = BM / 64 + 1;                 // 64 extra rows behind the tile: rows m0+BM .. m0+BM+1 are read by taps 1, 2
    constexpr int W_ROUNDS = (BN + 63) / 64;
    constexpr int A_BUF = A_ROUNDS * 8192, W_TILE = BN * 128;
    static_assert(BM % 64 == 0 && (BN % 64 == 0 || BN == 32), "tile shape");
    constexpr int W_BUFS = W1 ? 1 : 2;
    static_assert(2 * A_BUF + W_BUFS * W_TILE <= 160 * 1024, "LDS budget");
    static_assert(!W1 || (BN >= 64 && A_ROUNDS % 3 == 0), "single W buffer: whole DMA rounds, three A rounds per K tile");
    __shared__ __attribute__((aligned(16))) char smem[2 * A_BUF + W_BUFS * W_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int hi = lane >> 5;
    const int l32 = lane & 31;

    const int nwg = p.ntile_launch;
    const int wgid = xcd_remap((int)blockIdx.x, nwg);
    int tile_m, tile_n;
    gemm_tile_coords(wgid, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int plane = p.iH * p.iW;                        // rows per output frame in the padded enumeration

    // ---- staging addresses (bytes): one wave-instruction moves 8 rows x 128 B; round r covers rows r*64 + wave*8 + lane/8 ----
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    unsigned a_off[A_ROUNDS], w_off[W_ROUNDS];
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) {
        const int am = min(m0 + r * 64 + srow, p.M - 1);
        const int f = am / plane;
        int rr = am - f * plane;
        // rows of the two padding lines at the bottom (hp >= oH) are never a shift source of a kept row; staged as they are
        // they would reach up to two lines past their frame (past the allocation for the last frame): fetch two lines higher
        if (rr >= p.oH * p.iW) rr -= 2 * p.iW;
        const int nb = f / p.oT, t = f - nb * p.oT;
        a_off[r] = 2u * ((unsigned)(((size_t)(nb * p.iT + t) * plane + rr) * (size_t)p.iC) + schunk * 8);
    }
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) {
        const int wr = min(n0 + r * 64 + srow, p.N - 1);
        w_off[r] = 2u * ((unsigned)wr * (unsigned)p.ldw + schunk * 8);
    }
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    char* const lds_stage = smem + wave_s * 1024;
    const bool w_active = (BN >= 64) || (wave_s * 8 < BN);   // BN = 32 (conv_out): waves 0-3 carry the 32 weight rows (wave-uniform)
    const buf_rsrc_t a_rsrc = make_buf_rsrc(p.A, p.a_bytes), w_rsrc = make_buf_rsrc(p.W, p.w_bytes);

    const int nk = p.K / GEMM_BK;                         // multiple of 3
    const int ng = nk / 3;                                // (dt, dh, channel block) steps
    const int late_wave = __builtin_amdgcn_readfirstlane(wave >> 2);   // waves 4-7 share SIMDs with waves 0-3

    // ---- fragment read addresses: A rows shifted by the tap, each row with its own swizzle ---------------------------------
    const int w_row_base = (wn * NT * 32 + l32) * 128;
    const int swz_w = (lane >> 1) & 7;
    int chunk_w[4], x_row_base[3], chunk_x[3][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) chunk_w[ks] = (((2 * ks + hi) ^ swz_w) << 4);
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int row = wm * MT * 32 + l32 + dw;
        x_row_base[dw] = row * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) chunk_x[dw][ks] = (((2 * ks + hi) ^ ((row >> 1) & 7)) << 4);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // ---- prologue: A tile of step 0 (tap dw = 0 of (dt,dh,cb) = 0) and the W tile of K tile 0 -----------------------------
    {
        const unsigned a_soff = 2u * (unsigned)__builtin_amdgcn_readfirstlane(p.tap_off[0]);
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) bglds16(a_rsrc, a_off[r], a_soff, lds_stage + r * 8192);
        if (w_active) {
#pragma unroll
            for (int r = 0; r < W_ROUNDS; ++r) bglds16(w_rsrc, w_off[r], 0u, lds_stage + 2 * A_BUF + r * 8192);
        }
    }
    drain_and_barrier();

    // W1: wf[ks][nt] holds the W fragments of ALL four k-steps of the current K tile (read in load slots 0 and 1); otherwise only wf[0] is used
    bf16x8 wf[W1 ? 4 : 1][NT], xf[MT];
    auto load_frags = [&](const char* abase, const char* wbase, int dw, int ks) {
        if (W1) {
            if (ks < 2) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wf[W1 ? 2 * ks + k2 : 0][nt] = *(const bf16x8*)(wbase + w_row_base + nt * 4096 + chunk_w[2 * ks + k2]);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[0][nt] = *(const bf16x8*)(wbase + w_row_base + nt * 4096 + chunk_w[ks]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const bf16x8*)(abase + x_row_base[dw] + mt * 4096 + chunk_x[dw][ks]);
    };
    // tap offset of the A tile being staged (step g+1), fetched one step ahead and made scalar (see gemm_kernel.hpp)
    int tap_stage = __builtin_amdgcn_readfirstlane(p.tap_off[3 * min(1, ng - 1)]);
    int tap_ahead = 0;
    // DMA pieces issued during K tile (g, dw): the W tile of the next K tile, and every third A round of step g+1; split
    // over the three early load slots.  Past the end the last tiles are re-fetched into the buffers nobody reads any more
    // (W1: the last W tile is re-fetched over itself — same bytes — after every wave has its fragments in registers).
    auto stage_part = [&](int g, int dw, int part) {
        const int kt_next = min(3 * g + dw + 1, nk - 1);
        const unsigned w_soff = 2u * (unsigned)(kt_next * GEMM_BK), a_soff = 2u * (unsigned)tap_stage;
        char* adst = lds_stage + ((g + 1) & 1) * A_BUF;
        char* wdst = lds_stage + 2 * A_BUF + (W1 ? 0 : (kt_next & 1) * W_TILE);
        if (!W1 && 3 * g + dw + 1 > nk - 1) wdst = lds_stage + 2 * A_BUF + (nk & 1) * W_TILE;     // clamped re-fetch: the idle buffer
        if (W1) {
            // the three A rounds r = dw, dw+3, dw+6 of this K tile in load slots 0 and 1; the W rounds in load slot 2 only: the W buffer is
            // read by the early group in global slots 0, 2 and by the late group in slots 1, 3 of the tile; slot 2 of either group comes after
            constexpr int NA = A_ROUNDS / 3;
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if ((j * 2) / NA == part) bglds16(a_rsrc, a_off[dw + 3 * j], a_soff, adst + (dw + 3 * j) * 8192);
            if (part == 2) {
#pragma unroll
                for (int r = 0; r < W_ROUNDS; ++r) bglds16(w_rsrc, w_off[r], w_soff, wdst + r * 8192);
            }
            return;
        }
        constexpr int NA = (A_ROUNDS + 2) / 3;             // upper bound of A rounds in one K tile
        int piece = 0;
        const int n_a = (A_ROUNDS - dw + 2) / 3;           // rounds r = dw, dw+3, ...
        const int n_pieces = n_a + W_ROUNDS;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int r = dw + 3 * j;
            if (r < A_ROUNDS) {
                if (piece * 3 / n_pieces == part) bglds16(a_rsrc, a_off[r], a_soff, adst + r * 8192);
                ++piece;
            }
        }
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) {
            if (w_active && piece * 3 / n_pieces == part) bglds16(w_rsrc, w_off[r], w_soff, wdst + r * 8192);
            ++piece;
        }
    };
    auto mma = [&](int ks) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[W1 ? ks : 0][nt], xf[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_end = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // Ping-pong exactly as gemm_kernel.hpp: per K tile four (load, compute) slot pairs; waves 4-7 run one slot behind.
    // A buffer (g+1)&1 was last read in step g-1 and is written during step g; W buffer (kt+1)&1 was last read in K tile kt-1.
    if (late_wave == 0) {
        for (int g = 0; g < ng; ++g) {
            const char* abase = smem + (g & 1) * A_BUF;
            tap_ahead = p.tap_off[3 * min(g + 2, ng - 1)];
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const char* wbase = smem + 2 * A_BUF + (W1 ? 0 : ((3 * g + dw) & 1) * W_TILE);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    load_frags(abase, wbase, dw, sl);
                    if (sl < 3) stage_part(g, dw, sl);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    slot_end();
                    mma(sl);
                    if (sl == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    slot_end();
                }
            }
            tap_stage = __builtin_amdgcn_readfirstlane(tap_ahead);
        }
    } else {
        for (int g = 0; g < ng; ++g) {
            const char* abase = smem + (g & 1) * A_BUF;
            tap_ahead = p.tap_off[3 * min(g + 2, ng - 1)];
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const char* wbase = smem + 2 * A_BUF + (W1 ? 0 : ((3 * g + dw) & 1) * W_TILE);
                if (g > 0 || dw > 0) mma(3);                       // last compute slot of the previous K tile
                slot_end();
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    load_frags(abase, wbase, dw, sl);
                    if (sl < 3) stage_part(g, dw, sl);
                    if (sl == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    slot_end();
                    if (sl < 3) {
                        mma(sl);
                        slot_end();
                    }
                }
            }
            tap_stage = __builtin_amdgcn_readfirstlane(tap_ahead);
        }
        mma(3);                                                    // last compute slot of the last K tile
    }

    // ---- epilogue: rows of the padded enumeration -> compact [NB, oT, oH, oW, Cout]; padding rows are dropped ----------------
    if constexpr (NT == 2) {
        // LDS-staged (gemm_kernel.hpp: staged_epilogue_wave): whole cache lines for the residual and the output
        static_assert(8 * MT * 32 * 128 <= 2 * A_BUF + W_BUFS * W_TILE, "staging regions fit the (dead) operand buffers");
        const int M = p.M, iW = p.iW, oH = p.oH, oW = p.oW;
        staged_epilogue_wave<MT, EPI>(p, acc, m0 + wm * MT * 32, n0 + wn * NT * 32, smem + wave_s * (MT * 32 * 128), lane,
                                      [M, plane, iW, oH, oW](int m) -> int {
                                          const int f = m / plane, rr = m - f * plane;
                                          const int hp = rr / iW, wp = rr - hp * iW;
                                          return (m < M && hp < oH && wp < oW) ? (f * oH + hp) * oW + wp : -1;
                                      });
    } else {
    // register-path epilogue (NT = 1: conv_out's 32-column tile)
    // acc[mt][nt][r] = C[m][n], m = m0 + (wm*MT + mt)*32 + l32,  n = n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3)
    // operands fetched in batches before use (see gemm_kernel.hpp): the bias of the wave's column groups once, the residual rows of a
    // 32-row block together
    const bool has_bias = p.bias != nullptr;
    const bool has_res = (EPI == EPI_BIAS_GATE_RES) && p.R != nullptr;
    int ncol[NT];
    bool n_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nbase = n0 + (wn * NT + nt) * 32;
        n_ok[nt] = nbase < p.N;
        ncol[nt] = n_ok[nt] ? nbase : 0;
    }
    f32x4 bv[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[nt][g] = *(const f32x4*)(p.bias + ncol[nt] + 8 * g + 4 * hi);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + (wm * MT + mt) * 32 + l32;
        const int f = m / plane, rr = m - f * plane;
        const int hp = rr / p.iW, wp = rr - hp * p.iW;
        const bool m_ok = (m < p.M) && (hp < p.oH) && (wp < p.oW);
        const size_t orow = ((size_t)f * p.oH + hp) * p.oW + wp;
        u16x4 rv[NT][4];
        if (has_res) {
            const size_t rrow = m_ok ? orow : 0;          // rows dropped by the padded enumeration read row 0 and are not stored
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[nt][g] = *(const u16x4*)(p.R + rrow * p.ldr + ncol[nt] + 8 * g + 4 * hi);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!n_ok[nt]) continue;                      // wave-uniform
            const int nbase = ncol[nt];
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[mt][nt][4 * g + c] + bv[nt][g][c];
                if (has_res) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] += bf16_bits_to_f32(rv[nt][g][c]);
                }
                pk[g][0] = pack_bf16x2(v[0], v[1]);
                pk[g][1] = pack_bf16x2(v[2], v[3]);
            }
            if (WIDE_STORE) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    if (m_ok) {
                        uint4 o = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        *(uint4*)(p.C + orow * p.ldc + nbase + 8 * g + 8 * hi) = o;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (m_ok) {
                        uint2 o = make_uint2(pk[g][0], pk[g][1]);
                        *(uint2*)(p.C + orow * p.ldc + nbase + 8 * g + 4 * hi) = o;
                    }
                }
            }
        }
    }
    }
}

}  // namespace aether
