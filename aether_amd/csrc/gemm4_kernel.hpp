// Second main loop for the DiT linears:  C[M,N] = epi( A[M,K] · W[N,K]^T ), same 256x256 workgroup tile, tile walk and epilogues
// as gemm_kernel.hpp — but FOUR wavefronts, one per SIMD, each owning a 128x128 register tile (16 accumulator tiles = 256 of the
// 512 registers a lone wave has), everything else of the K loop issued BETWEEN that wave's own MFMAs, and the operands streamed
// through a FOUR-stage ring of 32-wide K slabs (4 x 32 KiB of LDS) instead of two 64-wide buffers.
//
// Status: correct (same parity tests as the ping-pong loop), NOT the default: measured within -8 ... +2 % of the ping-pong loop at
// the DiT shapes (profiles/r02_gemm4_ablation.json).  What the ablation of this loop shows (tools/gpu_gemm4_ablate.py, which builds
// the ABL variants below into tools/probes/gemm4_ablate.so — the product library only instantiates ABL = 0):
//   * the chip is power-capped: a bare stream of these MFMAs on random operands runs at 1.78-1.9 GHz = 1.85 PF/s
//     (profiles/r02_mfma_power_probe.json; 2.48 PF/s only on all-zero operands); this loop runs at 1.55-1.8 GHz;
//   * LDS-DMA alone needs 1.2-1.45 us per 64-wide K tile (64 KiB per CU, 80 % L2 hits) at 2.39 GHz, however many requests are in
//     flight (two 64-KiB buffers or this ring: the same) — the same time the MFMAs alone need (1.2 us at 1.9 GHz);
//   * per 64-wide K tile (timers included, 8192^3): MFMAs alone 1.24 us = 2 480 cycles at 2.00 GHz; + fragment reads 1.40 us =
//     2 550 cycles at 1.82 GHz; + LDS-DMA 1.68 us = 2 670 cycles at 1.59 GHz, of which 5 % in s_waitcnt / s_barrier: switching the
//     data movement on costs 7 % in cycles and 20 % in clock — the loop is bound by the power cap, not by its schedule.  The
//     vendor GEMM (same 256x256x64 tile, 4 waves, direct-to-LDS, same L2 hit rate and miss bytes) is within 2-5 % of our
//     ping-pong loop over the four DiT shapes (profiles/r02_vendor_gemm.txt).
// Design notes: with two 64-wide buffers a slab can only be requested one tile-time before it is needed; the ring requests slab
// s+4 while slab s is computed and only ever waits for the OLDEST of three requests in flight (s_waitcnt vmcnt(16)), and the
// LDS-DMA instructions are spread one behind every fourth MFMA.
//
// LDS slab (32 KiB): A rows 0..255 then W rows 0..255, 64 B per row (32 bf16), 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 2) & 3): the 16 rows a quarter-wave reads with ds_read_b128 hit 16 distinct 4-bank groups.
// Stage s (slab s & 3), fragment sets alternate per k-step:
//   ks0: 16 MFMAs(set 0) ∥ 8 fragment reads (s, ks1) -> set 1 ∥ LDS-DMA pieces 4..7 of slab s+3 (buffer (s-1) & 3)
//        s_waitcnt vmcnt(16) lgkmcnt(0); s_barrier     (slab s+1 has landed for every wave; nobody reads slab s any more)
//   ks1: 16 MFMAs(set 1) ∥ 8 fragment reads (s+1, ks0) -> set 0 ∥ LDS-DMA pieces 0..3 of slab s+4 (buffer s & 3)
// i.e. one LDS-DMA instruction behind every fourth MFMA, all the time.
#pragma once
#include "gemm_kernel.hpp"

namespace aether {

template <int EPI, bool WIDE_STORE, int ABL = 0>
__global__ __launch_bounds__(256) void gemm4_bf16_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, MT = 4, NT = 4, BK = 32, NSLAB = 4;
    constexpr int A_SLAB = BM * BK * 2, W_SLAB = BN * BK * 2, SLAB_BYTES = A_SLAB + W_SLAB;
    __shared__ __attribute__((aligned(16))) char smem[NSLAB * SLAB_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned long long abl_c0 = ABL ? __builtin_readcyclecounter() : 0, abl_r0 = ABL ? wall_clock64() : 0;
    unsigned long long abl_wait = 0, abl_bar = 0;
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5;
    const int l32 = lane & 31;

    const int wgid = xcd_remap((int)blockIdx.x, p.ntile_launch) + p.tile_base;
    int tile_m, tile_n;
    gemm_tile_coords(wgid, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging: piece r of this wave = slab rows r*64 + wave*16 + lane/4 (16 rows x 64 B per wave-instruction, 1 KiB of LDS) ----------
    const int srow = wave * 16 + (lane >> 2);
    const int schunk = (lane & 3) ^ ((lane >> 4) & 3);          // (row >> 2) & 3 with row = r*64 + wave*16 + lane/4
    unsigned a_off[4], w_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lm0 = (ABL & 32) ? 0 : m0, ln0 = (ABL & 32) ? 0 : n0;      // measurement: every workgroup streams tile (0,0): ~100 % L2 hits
        a_off[r] = 2u * ((unsigned)min(lm0 + r * 64 + srow, p.M - 1) * (unsigned)p.lda + schunk * 8);
        w_off[r] = 2u * ((unsigned)min(ln0 + r * 64 + srow, p.N - 1) * (unsigned)p.ldw + schunk * 8);
    }
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    char* const lds_stage = smem + wave_s * 1024;
    const buf_rsrc_t a_rsrc = make_buf_rsrc(p.A, p.a_bytes), w_rsrc = make_buf_rsrc(p.W, p.w_bytes);
    const int nslab = p.K / BK;
    // piece i (0..3: A rows, 4..7: W rows) of K slab s into ring buffer buf; slabs past the end re-fetch the last one (uniform vmcnt accounting)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t stg[2][8];
    auto dma_piece = [&](int s, int buf, int i) {
        const unsigned soff = 2u * (unsigned)(min(s, nslab - 1) * BK);
        if (ABL & 8) {                       // measurement only: the same bytes as plain loads into registers (two slabs in flight)
            if (i == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(stg[0][e]));
#pragma unroll
                for (int e = 0; e < 8; ++e) stg[0][e] = stg[1][e];
            }
            stg[1][i] = __builtin_amdgcn_raw_buffer_load_b128(i < 4 ? a_rsrc : w_rsrc, i < 4 ? a_off[i] : w_off[i - 4], soff, 0);
            return;
        }
        char* dst = lds_stage + buf * SLAB_BYTES;
        if (i < 4) bglds16(a_rsrc, a_off[i], soff, dst + i * 4096);
        else bglds16(w_rsrc, w_off[i - 4], soff, dst + A_SLAB + (i - 4) * 4096);
    };

    // ---- fragment read addresses ------------------------------------------------------------------------------------------------------
    const int swz = (l32 >> 2) & 3;
    const int x_row_base = (wm * 128 + l32) * 64;
    const int w_row_base = A_SLAB + (wn * 128 + l32) * 64;
    int chunk_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) chunk_off[ks] = (((2 * ks + hi) ^ swz) << 4);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    bf16x8 wf[2][NT], xf[2][MT];
    // fragment read number j of a k-step, in the order the MFMAs want them: X rows 0, W columns 0..3, X rows 1..3 (32 rows = 2 KiB apart)
    auto read_frag = [&](int set, const char* base, int ks, int j) {
        if (j == 0) xf[set][0] = *(const bf16x8*)(base + x_row_base + chunk_off[ks]);
        else if (j <= NT) wf[set][j - 1] = *(const bf16x8*)(base + w_row_base + (j - 1) * 2048 + chunk_off[ks]);
        else xf[set][j - NT] = *(const bf16x8*)(base + x_row_base + (j - NT) * 2048 + chunk_off[ks]);
    };

    // ---- prologue: slabs 0..2 and the first half of slab 3 requested, slab 0 landed, its first fragments in set 0 ------------------------
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int i = 0; i < (s == NSLAB - 1 ? 4 : 8); ++i) dma_piece(s, s, i);
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) read_frag(0, smem, 0, j);

    // one k-step: 16 MFMAs on fragment set `cur`; behind every MFMA at most one auxiliary instruction: LDS-DMA pieces d0..d0+3 of
    // slab `ds` behind MFMAs 1, 5, 9, 13 (one per 128 matrix-pipe cycles: the requests never arrive in a burst that fills the
    // texture-addresser queue and blocks the issuing wave), the 8 fragment reads of (rbase, rks) into the other set behind the rest
    auto kstep = [&](auto cur_tag, const char* rbase, int rks, auto d0_tag, int ds, int dbuf) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        constexpr int d0 = decltype(d0_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int q = mt * NT + nt;                         // compile-time after unrolling
                if (!(ABL & 1)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nt], xf[cur][mt], acc[mt][nt], 0, 0, 0);
                else if (q < 8) { if (q < 4) asm volatile("" :: "v"(wf[cur][q])); else asm volatile("" :: "v"(xf[cur][q - 4])); }
                __builtin_amdgcn_sched_barrier(0);
                if (q % 4 == 1) { if (!(ABL & 2)) dma_piece(ds, dbuf, d0 + q / 4); }
                else {
                    const int j = q - (q + 2) / 4;                  // auxiliary slots that are not DMA slots, in order: 0,2,3,4,6,7,8,10 -> 0..7
                    if (j < 8 && !(ABL & 4)) read_frag(nxt, rbase, rks, j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I4 = std::integral_constant<int, 4>;

    for (int s = 0; s < nslab; ++s) {
        const int cb = s & (NSLAB - 1);
        const char* base = smem + cb * SLAB_BYTES;
        const char* nbase = smem + ((s + 1) & (NSLAB - 1)) * SLAB_BYTES;
        kstep(I0{}, base, 1, I4{}, s + NSLAB - 1, (s + NSLAB - 1) & (NSLAB - 1));
        unsigned long long w0 = 0, w1 = 0;
        if (ABL & 16) w0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        if (ABL & 16) w1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ABL & 16) { abl_wait += w1 - w0; abl_bar += __builtin_readcyclecounter() - w1; }
        __builtin_amdgcn_sched_barrier(0);
        kstep(I1{}, nbase, 0, I0{}, s + NSLAB, cb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail DMA must not outlive the workgroup's LDS
    if (ABL && p.part != nullptr && tid == 0 && blockIdx.x < 256) {      // measurement builds: shader cycles and 100-MHz ticks of the main loop
        p.part[2 * blockIdx.x] = (float)(__builtin_readcyclecounter() - abl_c0);
        p.part[2 * blockIdx.x + 1] = (float)(wall_clock64() - abl_r0);
        p.part[512 + 2 * blockIdx.x] = (float)abl_wait;
        p.part[512 + 2 * blockIdx.x + 1] = (float)abl_bar;
    }

    // ---- epilogue (as gemm_kernel.hpp: operands fetched in batches before use) -------------------------------------------------------------
    // acc[mt][nt][r] = C[m][n], m = m0 + (wm*4 + mt)*32 + l32,  n = n0 + (wn*4 + nt)*32 + 8*(r>>2) + 4*hi + (r&3)
    const bool has_bias = p.bias != nullptr;
    const bool has_gate = (EPI == EPI_BIAS_GATE_RES) && p.gate_vid != nullptr;
    const bool has_res = (EPI == EPI_BIAS_GATE_RES) && p.R != nullptr;
    int ncol[NT];
    bool n_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nbase_c = n0 + (wn * NT + nt) * 32;
        n_ok[nt] = nbase_c < p.N;
        ncol[nt] = n_ok[nt] ? nbase_c : 0;
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
        const bool m_ok = m < p.M;
        const int mm = m_ok ? m : p.M - 1;
        f32x4 gv[NT][4];
        u16x4 rv[NT][4];
        {
            const int b = has_gate ? mm / p.rows_per_batch : 0;
            const int t = mm - b * p.rows_per_batch;
            const float* gate = has_gate ? (t < p.n_text ? p.gate_txt : p.gate_vid) + (size_t)b * p.gate_bstride : nullptr;
            auto load_gate = [&]() {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) gv[nt][g] = *(const f32x4*)(gate + ncol[nt] + 8 * g + 4 * hi);
            };
            auto load_res = [&]() {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[nt][g] = *(const u16x4*)(p.R + (size_t)mm * p.ldr + ncol[nt] + 8 * g + 4 * hi);
            };
            if (has_gate && has_res) { load_res(); load_gate(); }
            else if (has_gate) load_gate();
            else if (has_res) load_res();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!n_ok[nt]) continue;                      // wave-uniform
            const int nbase_c = ncol[nt];
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[mt][nt][4 * g + c] + bv[nt][g][c];
                if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = gelu_tanh(v[c]);
                }
                if (EPI == EPI_BIAS_GATE_RES) {
                    if (has_gate) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] *= gv[nt][g][c];
                    }
                    if (has_res) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] += bf16_bits_to_f32(rv[nt][g][c]);
                    }
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
                        *(uint4*)(p.C + (size_t)m * p.ldc + nbase_c + 8 * g + 8 * hi) = o;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (m_ok) {
                        uint2 o = make_uint2(pk[g][0], pk[g][1]);
                        *(uint2*)(p.C + (size_t)m * p.ldc + nbase_c + 8 * g + 4 * hi) = o;
                    }
                }
            }
        }
    }
}

}  // namespace aether
