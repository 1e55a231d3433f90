// Persistent form of the 256x256x64 ping-pong GEMM of gemm_kernel.hpp (DiT linears): C[M,N] = epi(A[M,K] · W[N,K]^T).
//
// Why: with one 128-KiB-LDS workgroup per CU a 256x256x3072 tile is ~80 µs of main loop; around it every tile pays a workgroup
// launch, an exposed first LDS-DMA round trip (nothing to compute under it) and — per launch — a partly filled last round
// (qkv: 2124 tiles = 8.3 rounds of 256) that the one-tile kernel runs as a split-K second launch plus a finalize kernel.  The
// vendor's hand-written kernel for these shapes is built the other way (profiles/r02_vendor_gemm.txt): a grid of 236 = 59 x 4
// workgroups that each own a WHOLE number of tiles (2124 = 236 x 9, 708 = 236 x 3, 2832 = 236 x 12), no tail, the next tile's
// first operands requested while the current one finishes.  This kernel does the same:
//   * grid G workgroups; workgroup i walks tiles i, i + G, i + 2G, ... in the XCD-grouped tile order of gemm_tile_coords, so at
//     any moment the resident workgroups cover one contiguous band of tiles exactly like one round of the one-tile kernel (same
//     L2 footprint).  The launcher picks G by the shape of the last round (measured, profiles/r03_gemm_persistent.txt: a 236-wide
//     grid idles 8 % of the CUs for the whole launch, which costs more than the per-tile overheads it removes when the one-tile
//     scheme's tail is a cheap split-K launch): a short last round (<= 128 tiles: qkv 76, ff-up 16) -> G = 256 over the FULL
//     rounds only, followed by the split-K tail launch of gemm_bf16.hip; otherwise (ff-down / out-proj: 708 = 2 x 256 + 196)
//     -> G = ceil(tiles / rounds) = 236 workgroups owning whole tiles, no tail;
//   * the K loop of consecutive tiles is ONE stream of K tiles through the two LDS buffers: the load slots of a tile's last K
//     tile request the first K tile of the NEXT output tile (only the per-lane row offsets differ), so no tile but the first
//     starts with an exposed DMA round trip;
//   * the two waves of a SIMD keep their ping-pong phase across tiles; both write their epilogue in the same global slot (the
//     early group before its first load slot of the new tile, the late group after its last compute slot of the old one).
// Same arithmetic and K order as the one-tile kernel: results are bit-identical (tests/test_kernels_gpu.py).
#pragma once
#include "gemm_kernel.hpp"

namespace aether {

template <int EPI, bool WIDE_STORE>
__global__ __launch_bounds__(512) void gemm_bf16_persistent_kernel(GemmArgs p) {
    constexpr int WM = 2, WN = 4, MT = 4, NT = 2;
    constexpr int BM = 256, BN = 256;
    constexpr int A_TILE = BM * GEMM_BK * 2, W_TILE = BN * GEMM_BK * 2, BUF_BYTES = A_TILE + W_TILE;
    constexpr int A_ROUNDS = BM / 64, W_ROUNDS = BN / 64, NP = A_ROUNDS + W_ROUNDS;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l32 = lane & 31;

    // Tile walk: XCD x (= blockIdx % 8) owns the contiguous band of tiles the one-tile kernel's xcd_remap gives it (tiles / 8 of the tile
    // order) and its G / 8 workgroups walk through that band G / 8 tiles at a time — the SAME tile-to-XCD-to-time mapping as the one-tile
    // kernel's dispatch order, so consecutive rounds of an XCD reuse the A row tiles (and W column tiles) its L2 already holds.  (A first
    // version that strode through the whole tile range by G lost 3 % to the one-tile kernel: every round started on cold operands.)
    const int G8 = (int)gridDim.x >> 3;                      // workgroups per XCD (the launcher makes the grid a multiple of 8)
    const int tiles = p.ntile_launch;                        // tiles [0, ntile_launch) of the tile order (a split-K tail launch may take the rest)
    const int xcd = (int)blockIdx.x & 7, lidx = (int)blockIdx.x >> 3;
    const int tq = tiles >> 3, trem = tiles & 7;
    const int chunk = tq + (xcd < trem ? 1 : 0);
    const int first = ((xcd < trem) ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq) + lidx;
    const int G = G8;                                        // stride between this workgroup's consecutive tiles
    const int n_my = (chunk - lidx + G8 - 1) / G8;
    if (lidx >= chunk) return;                               // (whole workgroup, before any barrier)
    const int nk = p.K / GEMM_BK;

    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    char* const lds_stage = smem + wave_s * 1024;
    const buf_rsrc_t a_rsrc = make_buf_rsrc(p.A, p.a_bytes), w_rsrc = make_buf_rsrc(p.W, p.w_bytes);
    const int late_wave = __builtin_amdgcn_readfirstlane(wave >> 2);   // waves 4-7 share SIMDs with waves 0-3

    // byte offsets of this lane's staging rows for output tile `tile`
    auto tile_offsets = [&](int tile, unsigned (&a_off)[A_ROUNDS], unsigned (&w_off)[W_ROUNDS], int& m0, int& n0) {
        int tile_m, tile_n;
        gemm_tile_coords(tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
        m0 = tile_m * BM; n0 = tile_n * BN;
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) a_off[r] = 2u * ((unsigned)min(m0 + r * 64 + srow, p.M - 1) * (unsigned)p.lda + schunk * 8);
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) w_off[r] = 2u * ((unsigned)min(n0 + r * 64 + srow, p.N - 1) * (unsigned)p.ldw + schunk * 8);
    };
    // pieces [part*NP/3, (part+1)*NP/3) of K tile kt of the tile whose offsets are given -> buffer `buf`
    auto stage_part = [&](const unsigned (&a_off)[A_ROUNDS], const unsigned (&w_off)[W_ROUNDS], int kt, int buf, int part) {
        const unsigned soff = 2u * (unsigned)(kt * GEMM_BK);
        char* dst = lds_stage + buf * BUF_BYTES;
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r)
            if (r * 3 / NP == part) bglds16(a_rsrc, a_off[r], soff, dst + r * 8192);
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r)
            if ((A_ROUNDS + r) * 3 / NP == part) bglds16(w_rsrc, w_off[r], soff, dst + A_TILE + r * 8192);
    };

    const int swz = (lane >> 1) & 7;
    const int x_row_base = (wm * MT * 32 + l32) * 128;
    const int w_row_base = A_TILE + (wn * NT * 32 + l32) * 128;
    int chunk_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) chunk_off[ks] = (((2 * ks + hi) ^ swz) << 4);

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
    };
    zero_acc();

    bf16x8 wf[NT], xf[MT];
    auto load_frags = [&](const char* base, int slot) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const bf16x8*)(base + w_row_base + nt * 4096 + chunk_off[slot]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const bf16x8*)(base + x_row_base + mt * 4096 + chunk_off[slot]);
    };
    auto mma = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_end = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    unsigned a_cur[A_ROUNDS], w_cur[W_ROUNDS], a_nxt[A_ROUNDS], w_nxt[W_ROUNDS];
    int m0, n0, m0_nxt, n0_nxt;
    tile_offsets(first, a_cur, w_cur, m0, n0);
    // prologue: K tile 0 of the first output tile
#pragma unroll
    for (int part = 0; part < 3; ++part) stage_part(a_cur, w_cur, 0, 0, part);
    drain_and_barrier();

    int par = 0;                                              // LDS buffer of the K tile being consumed
    if (late_wave == 0) {
        for (int ti = 0; ti < n_my; ++ti) {
            for (int kt = 0; kt < nk; ++kt) {
                const char* base = smem + par * BUF_BYTES;
                const bool last_kt = kt == nk - 1;
                if (last_kt) {                                // the K tile staged during this one opens the next output tile
                    if (ti + 1 < n_my) tile_offsets(first + (ti + 1) * G, a_nxt, w_nxt, m0_nxt, n0_nxt);
                    else {                                    // nothing follows: re-fetch this tile's last K tile (uniform vmcnt accounting)
#pragma unroll
                        for (int r = 0; r < A_ROUNDS; ++r) a_nxt[r] = a_cur[r];
#pragma unroll
                        for (int r = 0; r < W_ROUNDS; ++r) w_nxt[r] = w_cur[r];
                        m0_nxt = m0; n0_nxt = n0;
                    }
                }
                const int kt_stage = last_kt ? ((ti + 1 < n_my) ? 0 : nk - 1) : kt + 1;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    load_frags(base, sl);
                    if (sl < 3) {
                        if (last_kt) stage_part(a_nxt, w_nxt, kt_stage, par ^ 1, sl);
                        else stage_part(a_cur, w_cur, kt_stage, par ^ 1, sl);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    slot_end();
                    mma();
                    if (sl == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    slot_end();
                }
                par ^= 1;
            }
            gemm_epilogue<WM, WN, MT, NT, EPI, WIDE_STORE>(p, acc, m0, n0, wm, wn, l32, hi);
            zero_acc();
#pragma unroll
            for (int r = 0; r < A_ROUNDS; ++r) a_cur[r] = a_nxt[r];
#pragma unroll
            for (int r = 0; r < W_ROUNDS; ++r) w_cur[r] = w_nxt[r];
            m0 = m0_nxt; n0 = n0_nxt;
        }
    } else {
        int m0_prev = m0, n0_prev = n0;
        for (int ti = 0; ti < n_my; ++ti) {
            for (int kt = 0; kt < nk; ++kt) {
                const char* base = smem + par * BUF_BYTES;
                const bool last_kt = kt == nk - 1;
                if (ti > 0 || kt > 0) mma();                  // last compute slot of the previous K tile
                if (kt == 0 && ti > 0) {                      // ... which completed the previous output tile
                    gemm_epilogue<WM, WN, MT, NT, EPI, WIDE_STORE>(p, acc, m0_prev, n0_prev, wm, wn, l32, hi);
                    zero_acc();
                }
                slot_end();
                if (last_kt) {
                    if (ti + 1 < n_my) tile_offsets(first + (ti + 1) * G, a_nxt, w_nxt, m0_nxt, n0_nxt);
                    else {
#pragma unroll
                        for (int r = 0; r < A_ROUNDS; ++r) a_nxt[r] = a_cur[r];
#pragma unroll
                        for (int r = 0; r < W_ROUNDS; ++r) w_nxt[r] = w_cur[r];
                        m0_nxt = m0; n0_nxt = n0;
                    }
                }
                const int kt_stage = last_kt ? ((ti + 1 < n_my) ? 0 : nk - 1) : kt + 1;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    load_frags(base, sl);
                    if (sl < 3) {
                        if (last_kt) stage_part(a_nxt, w_nxt, kt_stage, par ^ 1, sl);
                        else stage_part(a_cur, w_cur, kt_stage, par ^ 1, sl);
                    }
                    if (sl == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    slot_end();
                    if (sl < 3) {
                        mma();
                        slot_end();
                    }
                }
                par ^= 1;
            }
            m0_prev = m0; n0_prev = n0;
#pragma unroll
            for (int r = 0; r < A_ROUNDS; ++r) a_cur[r] = a_nxt[r];
#pragma unroll
            for (int r = 0; r < W_ROUNDS; ++r) w_cur[r] = w_nxt[r];
            m0 = m0_nxt; n0 = n0_nxt;
        }
        mma();                                                // last compute slot of the last K tile
        gemm_epilogue<WM, WN, MT, NT, EPI, WIDE_STORE>(p, acc, m0_prev, n0_prev, wm, wn, l32, hi);
    }
}

}  // namespace aether
