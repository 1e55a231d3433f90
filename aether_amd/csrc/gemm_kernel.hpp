// Templated bf16 MFMA GEMM main loop shared by the DiT linears (gemm_bf16.hip) and the VAE's implicit-GEMM
// convolutions (vae_conv.hip):   C[M,N] = epi( A[M,K] · W[N,K]^T ),  fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Geometry: 8 wavefronts as WM (along M) x WN (along N); each wave owns (MT*32) x (NT*32) outputs;
//   workgroup tile BM = WM*MT*32, BN = WN*NT*32, BK = 64.  Instantiations:
//     <2,4,4,2>  256x256  (DiT linears; N a multiple of 256)
//     <4,2,4,2>  512x128  (VAE 128-channel layers; uses all 160 KiB of LDS)
//     <8,1,2,1>  512x32   (VAE conv_out: 32 / 3(+pad) output channels)
// The MFMA is issued "swapped" (W fragment = A operand) so a lane owns 4 consecutive output columns of one row.
// Operand tiles: 16-byte LDS-DMA, double buffered, one barrier per K tile, 128-byte LDS rows XOR-swizzled
// (16-B chunk ^= (row>>1)&7 on the DMA source address and on the ds_read_b128 address).
//
// A-row addressing is a policy:
//   linear:   row m starts at A + m*lda,                       K step kt adds kt*64 elements
//   gathered: row m starts at A + voxel_offset(m) (a zero-padded channels-last activation volume) and K step kt
//             adds tap_off[kt] elements — the (dt,dh,dw) tap and 64-channel block of an implicit-GEMM convolution.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace aether {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_GATE_RES = 2 };

struct GemmArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    bf16_t* C; int ldc;
    int M, N, K;
    const float* bias;          // [N] fp32 or null
    const bf16_t* R; int ldr;   // residual / additive term [M,N] or null
    const float* gate_vid;      // fp32 gate for video rows (index b*gate_bstride + n) or null (=1)
    const float* gate_txt;      // fp32 gate for text rows
    int gate_bstride;
    int rows_per_batch;         // S  (batch index of row m is m / S)
    int n_text;                 // rows [0, n_text) of each batch are text rows
    int tiles_m, tiles_n;
    // gathered-A (implicit-GEMM convolution) only:
    const int* tap_off;         // [K/64] element offsets added per K step
    int oT, oH, oW;             // output volume per batch item: M = NB*oT*oH*oW, row m = ((nb*oT+t)*oH+h)*oW+w
    int iT, iH, iW, iC;         // padded input volume dims (frames, rows, cols, channels per voxel)
    int stride_hw;              // spatial stride (1, or 2 for the down-sampling conv2d)
    // split-K (launches with too few output tiles to fill 256 CUs: the deep, low-resolution VAE layers with K = 27*512):
    int ksplit;                 // 1 = off; otherwise grid = tiles * ksplit and workgroup (tile, slice) accumulates K tiles
                                // [slice*nk/ksplit, (slice+1)*nk/ksplit) and stores its raw fp32 tile to part[slice][M][N]
    float* part;                // fp32 [ksplit][M][N]; summed in slice order (deterministic) by splitk_finalize_kernel
    // a launch may cover only the tiles [tile_base, tile_base + ntile_launch) of the tiles_m x tiles_n grid (DiT tail launch):
    int tile_base, ntile_launch;
    int part_tiled;             // split-K partials stored tile-major [slice][tile - tile_base][BM][BN] instead of [slice][M][N]
    unsigned a_bytes, w_bytes;  // extents of A and W in bytes (< 4 GiB): bounds of the buffer descriptors the LDS-DMA goes through
    unsigned r_bytes;           // extent of R in bytes (< 4 GiB): the staged epilogue fetches the residual by LDS-DMA through a buffer descriptor
};

constexpr int GEMM_BK = 64;
constexpr int GEMM_GROUP_M = 4;

// Compile-time knobs of the main loop, used ONLY by tools/probes/gemm_variants_probe.hip to build the variants of profiles/r06_gemm_energy.txt;
// the library is built with the defaults below (one loop ships).
#ifndef AETHER_GEMM_KSPS
#define AETHER_GEMM_KSPS 1        // k-steps per ping-pong slot (2: half the barriers, 16 MFMAs and 12 fragment reads per slot)
#endif
#ifndef AETHER_GEMM_MFMA_ORDER
#define AETHER_GEMM_MFMA_ORDER 0  // 0: mt outer / nt inner; 1: snake (one operand changes per MFMA); 2: nt outer / mt inner
#endif
#ifndef AETHER_GEMM_SETPRIO
#define AETHER_GEMM_SETPRIO 1     // s_setprio 1 around the MFMA burst
#endif

// tile id -> (tile_m, tile_n): groups of GEMM_GROUP_M row tiles x all column tiles, row tile fastest inside a group
AE_DEV void gemm_tile_coords(int wgid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const int per_group = GEMM_GROUP_M * tiles_n;
    const int group = wgid / per_group;
    const int first_m = group * GEMM_GROUP_M;
    const int gsz = min(GEMM_GROUP_M, tiles_m - first_m);
    const int in_group = wgid - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
}

// ---- fused epilogue of one output tile (shared by the one-tile-per-workgroup kernel below and the persistent kernel) -------------------
// acc[mt][nt][r] = C[m][n], m = m0 + (wm*MT + mt)*32 + l32,  n = n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3)
template <int WM, int WN, int MT, int NT, int EPI, bool WIDE_STORE>
AE_DEV void gemm_epilogue(const GemmArgs& p, const f32x16 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int l32, int hi) {
    // Every operand of the epilogue is fetched BEFORE it is used, in batches: the bias of the wave's column groups once, then per
    // 32-row block its gate vectors and residual rows (16 loads in flight).  (Round 1 issued each of the 12 loads of a 32x32
    // sub-tile behind its own branch, every one followed by s_waitcnt vmcnt(0): 96 dependent round trips ≈ 15 µs per tile — 18 %
    // of the out-projection, 5 % of the other GEMMs.)  Rows / column groups outside the problem load from clamped addresses and
    // are not stored.
    const bool has_bias = p.bias != nullptr;
    const bool has_gate = (EPI == EPI_BIAS_GATE_RES) && p.gate_vid != nullptr;
    const bool has_res = (EPI == EPI_BIAS_GATE_RES) && p.R != nullptr;
    int ncol[NT];
    bool n_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nbase = n0 + (wn * NT + nt) * 32;   // N % 32 == 0: a 32-column group is all in or all out (wave-uniform)
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
            // one basic block per combination: the 16 loads of a row block stay in flight together (a block boundary between the two
            // groups makes the compiler drain the first before issuing the second)
            if (has_gate && has_res) { load_res(); load_gate(); }
            else if (has_gate) load_gate();
            else if (has_res) load_res();
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
                // half-wave exchange: lanes 0-31 end with columns 8g..8g+7, lanes 32-63 with 8(g+1)..8(g+1)+7
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    if (m_ok) {
                        uint4 o = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        *(uint4*)(p.C + (size_t)m * p.ldc + nbase + 8 * g + 8 * hi) = o;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (m_ok) {
                        uint2 o = make_uint2(pk[g][0], pk[g][1]);
                        *(uint2*)(p.C + (size_t)m * p.ldc + nbase + 8 * g + 4 * hi) = o;
                    }
                }
            }
        }
    }
}

// ---- LDS-staged epilogue of one WAVE tile (MT*32 rows x 64 columns), round 6 -------------------------------------------------------------
// The register-path epilogue above moves every operand in the MFMA accumulator layout: a lane owns 4 consecutive columns of one row, so one
// wave-instruction touches 32 rows x 16 B (stores after the half-wave exchange) or 32 rows x 8 B (residual loads) — 32 cache lines for 512 / 256
// useful bytes.  Per-shape traces of the VAE (profiles/r06_vae_by_grid_*_before.md) price that: the SAME convolution launch costs 13 % (128-wide)
// to 26 % (256-wide) more with a residual than without, six times what the residual's bytes cost at HBM speed.
// Here the wave tile goes through a wave-PRIVATE 16-KiB LDS region (the operand buffers are dead once the K loop has passed its last barrier;
// no workgroup barrier inside the epilogue):
//   1. residual rows -> LDS by LDS-DMA, 8 rows x 128 contiguous bytes per wave-instruction (whole cache lines);
//   2. each lane reads its residual values from LDS (ds_read_b64), finishes v = epi(acc) in fp32 exactly as the register path does, rounds ONCE
//      to bf16 and writes the result over the residual it has just consumed (same address, same lane);
//   3. the region is read back row-wise (ds_read_b128) and stored 8 rows x 128 contiguous bytes per wave-instruction.
// LDS position of the 8-byte chunk c (columns 4c..4c+3) of wave-tile row r:  r*128 + ((c ^ 2*((r >> 1) & 7)) * 8): the XOR keeps a 16-byte
// slot = two column-adjacent chunks in order (so steps 1 and 3 move contiguous global bytes) and spreads the accumulator-layout accesses of
// step 2 over the banks (2-way conflicts instead of 32-way).
// RowMap: m (row of the launch's enumeration) -> output row index (int), or -1 if the row is not stored (beyond M / a padding row of the tap-reuse
// enumeration).  Bit-identical to the register path (same fp32 expression, same single rounding).
template <int MT, int EPI, class RowMap>
AE_DEV void staged_epilogue_wave(const GemmArgs& p, const f32x16 (&acc)[MT][2], int m0w, int n0w, char* lds_w, int lane, RowMap rowmap) {
    constexpr int NT = 2;
    const int hi = lane >> 5, l32 = lane & 31;
    const bool has_bias = p.bias != nullptr;
    const bool has_gate = (EPI == EPI_BIAS_GATE_RES) && p.gate_vid != nullptr;
    const bool has_res = (EPI == EPI_BIAS_GATE_RES) && p.R != nullptr;
    // rows this lane moves in steps 1 and 3: r = i*8 + (lane >> 3), i < MT*4; its 16-byte slot holds columns ccol(i) .. ccol(i) + 7
    const int srow = lane >> 3, slot2 = 2 * (lane & 7);
    int orow[MT * 4];                                         // output row index, -1 = not stored
    auto ccol = [&](int i) { return n0w + 4 * (slot2 ^ (2 * (((i * 8 + srow) >> 1) & 7))); };
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) {
        orow[i] = rowmap(m0w + i * 8 + srow);
        if (ccol(i) >= p.N) orow[i] = -1;                     // N % 32 == 0: an 8-column slot is all in or all out
    }
    if (has_res) {
        const buf_rsrc_t r_rsrc = make_buf_rsrc(p.R, p.r_bytes);
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const unsigned voff = orow[i] >= 0 ? (unsigned)(((size_t)orow[i] * p.ldr + ccol(i)) * 2) : 0u;   // dropped rows fetch row 0 and are never stored
            bglds16(r_rsrc, voff, 0u, lds_w + i * 1024);
        }
    }
    int ncol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ncol[nt] = (n0w + nt * 32 < p.N) ? n0w + nt * 32 : 0;
    f32x4 bv[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[nt][g] = has_bias ? *(const f32x4*)(p.bias + ncol[nt] + 8 * g + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the residual tile has landed (the bias with it: needed next anyway)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = mt * 32 + l32;
        const int swz = 2 * ((r >> 1) & 7);
        char* rowp = lds_w + r * 128;
        const float* gate = nullptr;
        if (has_gate) {
            const int m = min(m0w + r, p.M - 1);
            const int b = m / p.rows_per_batch;
            const int t = m - b * p.rows_per_batch;
            gate = (t < p.n_text ? p.gate_txt : p.gate_vid) + (size_t)b * p.gate_bstride;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 gv[4];
            u16x4 rv[4];
            if (has_gate) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gv[g] = *(const f32x4*)(gate + ncol[nt] + 8 * g + 4 * hi);
            }
            if (has_res) {
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[g] = *(const u16x4*)(rowp + (((nt * 8 + 2 * g + hi) ^ swz) << 3));
            }
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
                        for (int c = 0; c < 4; ++c) v[c] *= gv[g][c];
                    }
                    if (has_res) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] += bf16_bits_to_f32(rv[g][c]);
                    }
                }
                *(uint2*)(rowp + (((nt * 8 + 2 * g + hi) ^ swz) << 3)) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's LDS writes are done before it reads the rows back (wave-private region)
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) {
        const uint4 o = *(const uint4*)(lds_w + i * 1024 + lane * 16);
        if (orow[i] >= 0) *(uint4*)(p.C + (size_t)orow[i] * p.ldc + ccol(i)) = o;
    }
}

template <int WM, int WN, int MT, int NT, int EPI, bool WIDE_STORE, bool GATHER>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmArgs p) {
    static_assert(WM * WN == 8, "8 wavefronts per workgroup");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int A_TILE = BM * GEMM_BK * 2, W_TILE = BN * GEMM_BK * 2, BUF_BYTES = A_TILE + W_TILE;
    constexpr int A_ROUNDS = BM / 64;
    constexpr int W_ROUNDS = (BN + 63) / 64;
    static_assert(2 * BUF_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int hi = lane >> 5;
    const int l32 = lane & 31;

    // ---- tile assignment: XCD-aware remap, then groups of GEMM_GROUP_M row tiles x all column tiles ------------
    const int nwg = p.ntile_launch;
    const int kslice = (p.ksplit > 1) ? (int)blockIdx.x / nwg : 0;
    const int wgid = xcd_remap((int)blockIdx.x - kslice * nwg, nwg) + p.tile_base;
    int tile_m, tile_n;
    gemm_tile_coords(wgid, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging addresses ---------------------------------------------------------------------------------------
    // One wave-instruction moves 8 rows x 128 B; round r covers tile rows r*64 + wave*8 + lane/8.
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    unsigned a_off[A_ROUNDS], w_off[W_ROUNDS];
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) {
        const int am = min(m0 + r * 64 + srow, p.M - 1);
        if (GATHER) {
            int w = am % p.oW; int q = am / p.oW;
            int h = q % p.oH; q /= p.oH;            // q = nb*oT + t  (front padding is part of iT)
            const int nb = q / p.oT, t = q - nb * p.oT;
            a_off[r] = (unsigned)((((size_t)nb * p.iT + t) * p.iH + (size_t)h * p.stride_hw) * p.iW + (size_t)w * p.stride_hw) * (unsigned)p.iC + schunk * 8;
        } else {
            a_off[r] = (unsigned)am * (unsigned)p.lda + schunk * 8;
        }
    }
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) {
        const int wr = min(n0 + r * 64 + srow, p.N - 1);
        w_off[r] = (unsigned)wr * (unsigned)p.ldw + schunk * 8;
    }
    // LDS-DMA through buffer descriptors: per-lane byte offsets are loop invariant, the K advance is a scalar offset (a
    // table load for the gathered convolution taps) — no VALU, no v_readfirstlane in the main loop.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    char* const lds_stage = smem + wave_s * 1024;
    const bool w_active = (BN >= 64) || (srow < BN);   // BN = 32: only half of the threads carry weight rows
    const buf_rsrc_t a_rsrc = make_buf_rsrc(p.A, p.a_bytes), w_rsrc = make_buf_rsrc(p.W, p.w_bytes);
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) a_off[r] *= 2;
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) w_off[r] *= 2;

    const int nk_all = p.K / GEMM_BK;
    const int k_first = (p.ksplit > 1) ? (int)((long)kslice * nk_all / p.ksplit) : 0;     // this workgroup's K tiles:
    const int nk = ((p.ksplit > 1) ? (int)((long)(kslice + 1) * nk_all / p.ksplit) : nk_all) - k_first;   // [k_first, k_first + nk)
    auto stage = [&](int kt, int buf) {
        kt += k_first;
        const unsigned a_soff = 2u * (unsigned)(GATHER ? __builtin_amdgcn_readfirstlane(p.tap_off[kt]) : kt * GEMM_BK), w_soff = 2u * (unsigned)(kt * GEMM_BK);
        char* dst = lds_stage + buf * BUF_BYTES;
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) bglds16(a_rsrc, a_off[r], a_soff, dst + r * 8192);
        if (w_active) {
#pragma unroll
            for (int r = 0; r < W_ROUNDS; ++r) bglds16(w_rsrc, w_off[r], w_soff, dst + A_TILE + r * 8192);
        }
    };
    const int late_wave = __builtin_amdgcn_readfirstlane(wave >> 2);   // waves 4-7 share SIMDs with waves 0-3

    // ---- fragment read addresses -----------------------------------------------------------------------------------
    const int swz = (lane >> 1) & 7;
    const int x_row_base = (wm * MT * 32 + l32) * 128;
    const int w_row_base = A_TILE + (wn * NT * 32 + l32) * 128;
    int chunk_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) chunk_off[ks] = (((2 * ks + hi) ^ swz) << 4);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    stage(0, 0);
    drain_and_barrier();

    {
        // ---- "ping-pong" main loop -------------------------------------------------------------------------------
        // The two waves that share a SIMD (w and w+4) alternate roles every slot, phase-locked by s_barrier: while one
        // issues its MFMAs for KSPS k-steps (fragments already in registers) the other reads its next fragments from
        // LDS and issues the LDS-DMA pieces of the next K tile.  With KSPS = 1 (8 slots / tile):
        //   group 0 (waves 0-3):  L0 C0 L1 C1 L2 C2 L3 C3        group 1 (waves 4-7):  C3' L0 C0 L1 C1 L2 C2 L3
        // (C3' = last compute slot of the previous tile); KSPS = 2 halves the number of slots (16 MFMAs per slot).
        // The matrix pipe of every SIMD always has exactly one wave feeding it; the loading wave also issues the LDS-DMA.
        auto pingpong = [&](auto ksps_tag) {
            constexpr int KSPS = decltype(ksps_tag)::value;
            constexpr int NSLOT = 4 / KSPS;            // compute slots per K tile
            bf16x8 wf[KSPS][NT], xf[KSPS][MT];
            auto load_frags = [&](const char* base, int slot) {
#pragma unroll
                for (int k = 0; k < KSPS; ++k) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wf[k][nt] = *(const bf16x8*)(base + w_row_base + nt * 4096 + chunk_off[slot * KSPS + k]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) xf[k][mt] = *(const bf16x8*)(base + x_row_base + mt * 4096 + chunk_off[slot * KSPS + k]);
                }
            };
            // DMA pieces of tile kt split in NPART parts, one per early compute slot
            constexpr int NPART = (KSPS == 1) ? 3 : 1;
            constexpr int NPART0 = NPART;
            // past the last K tile the staging re-fetches that tile into the buffer nobody reads any more: no branch around
            // the DMA pieces, uniform vmcnt accounting
            // gathered A: the tap offset of the tile being staged was fetched (scalar load) one tile earlier — a load issued
            // in the load slot itself would put its latency, and the lgkmcnt wait it shares with the fragment reads, ahead of
            // every DMA piece
            // (readfirstlane: hipcc issues the table read as a vector load and would otherwise wrap every DMA piece that uses
            // it as its scalar offset in a waterfall loop)
            int tap_stage = GATHER ? __builtin_amdgcn_readfirstlane(p.tap_off[min(1, nk - 1) + k_first]) : 0;   // tile staged during tile 0
            int tap_ahead = 0;
            auto stage_part = [&](int kt, int buf, int part, auto nparts_tag) {
                constexpr int NPART = decltype(nparts_tag)::value;
                kt = min(kt, nk - 1) + k_first;
                const unsigned a_soff = 2u * (unsigned)(GATHER ? tap_stage : kt * GEMM_BK), w_soff = 2u * (unsigned)(kt * GEMM_BK);
                char* dst = lds_stage + buf * BUF_BYTES;
                constexpr int NP = A_ROUNDS + W_ROUNDS;
#pragma unroll
                for (int r = 0; r < A_ROUNDS; ++r)
                    if (r * NPART / NP == part) bglds16(a_rsrc, a_off[r], a_soff, dst + r * 8192);
                if (w_active) {
#pragma unroll
                    for (int r = 0; r < W_ROUNDS; ++r)
                        if ((A_ROUNDS + r) * NPART / NP == part) bglds16(w_rsrc, w_off[r], w_soff, dst + A_TILE + r * 8192);
                }
            };
            // The DMA pieces are issued by the wave in its LOAD slot, after its fragment reads: issuing one costs the issuing
            // wave 60-185 cycles (profiles/r01_gemm_ablation.json: the loop runs 20-35 % faster with the DMA removed), which
            // in the MFMA slot came straight out of the matrix pipe's time; VMEM issue of the loading wave overlaps the
            // partner's MFMAs.
            auto mma = [&]() {
                if (AETHER_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int k = 0; k < KSPS; ++k)
#pragma unroll
                    for (int i = 0; i < MT * NT; ++i) {
                        const int mt = (AETHER_GEMM_MFMA_ORDER == 2) ? i % MT : i / NT;
                        int nt = (AETHER_GEMM_MFMA_ORDER == 2) ? i / MT : i % NT;
                        if (AETHER_GEMM_MFMA_ORDER == 1 && (mt & 1)) nt = NT - 1 - nt;
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k][nt], xf[k][mt], acc[mt][nt], 0, 0, 0);
                    }
                if (AETHER_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(0);
            };
            auto slot_end = [&]() {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            // Slot k of tile kt is global slot 8kt+2k (+1 for the compute half) for group 0 and one later for group 1, so the
            // buffer of tile kt+1 (last read in the load slots of tile kt-1, which end at global slot 8kt-1 for group 1) is
            // free for the whole of tile kt; its DMA is waited for (vmcnt(0)) before the barrier that ends global slot 8kt+7.
            if (late_wave == 0) {
                for (int kt = 0; kt < nk; ++kt) {
                    const char* base = smem + (kt & 1) * BUF_BYTES;
                    if (GATHER) tap_ahead = p.tap_off[min(kt + 2, nk - 1) + k_first];
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        load_frags(base, sl);
                        if (sl < NPART0) stage_part(kt + 1, (kt + 1) & 1, sl, std::integral_constant<int, NPART0>{});
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        slot_end();
                        mma();
                        if (sl == NSLOT - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        slot_end();
                    }
                    if (GATHER) tap_stage = __builtin_amdgcn_readfirstlane(tap_ahead);
                }
            } else {
                for (int kt = 0; kt < nk; ++kt) {
                    const char* base = smem + (kt & 1) * BUF_BYTES;
                    if (GATHER) tap_ahead = p.tap_off[min(kt + 2, nk - 1) + k_first];
                    if (kt > 0) mma();                             // last compute slot of the previous tile
                    slot_end();
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        load_frags(base, sl);
                        if (sl < NPART) stage_part(kt + 1, (kt + 1) & 1, sl, std::integral_constant<int, NPART>{});
                        if (sl == NSLOT - 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        slot_end();
                        if (sl < NSLOT - 1) {
                            mma();
                            slot_end();
                        }
                    }
                    if (GATHER) tap_stage = __builtin_amdgcn_readfirstlane(tap_ahead);
                }
                mma();                                             // last compute slot of the last tile
            }
        };
        pingpong(std::integral_constant<int, AETHER_GEMM_KSPS>{});
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    // acc[mt][nt][r] = C[m][n], m = m0 + (wm*MT + mt)*32 + l32,  n = n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3)
    if (p.ksplit > 1) {   // split-K: raw fp32 partial tile; bias / residual / rounding happen in splitk_finalize_kernel
        float* part = p.part_tiled ? p.part + ((size_t)kslice * nwg + (wgid - p.tile_base)) * (BM * BN) : p.part + (size_t)kslice * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mrow = (wm * MT + mt) * 32 + l32;          // row inside the tile
            const int m = m0 + mrow;
            if (m >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ncol = (wn * NT + nt) * 32;              // column group inside the tile
                if (n0 + ncol >= p.N) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    if (p.part_tiled) *(f32x4*)(part + (size_t)mrow * BN + ncol + 8 * g + 4 * hi) = v;
                    else *(f32x4*)(part + (size_t)m * p.N + n0 + ncol + 8 * g + 4 * hi) = v;
                }
            }
        }
        return;
    }
    if constexpr (NT == 2) {
        // LDS-staged epilogue (wave-private 16-KiB regions over the dead operand buffers).  The K loop's last barrier is behind every wave and
        // every LDS read / LDS-DMA of the loop was waited for before it (vmcnt(0) / lgkmcnt(0) in the last slots), so the regions are free.
        static_assert(8 * MT * 32 * 128 <= 2 * BUF_BYTES, "staging regions fit the operand buffers");
        const int M = p.M;
        staged_epilogue_wave<MT, EPI>(p, acc, m0 + wm * MT * 32, n0 + wn * NT * 32, smem + wave_s * (MT * 32 * 128), lane,
                                      [M](int m) -> int { return m < M ? m : -1; });
    } else {
        gemm_epilogue<WM, WN, MT, NT, EPI, WIDE_STORE>(p, acc, m0, n0, wm, wn, l32, hi);
    }
}

}  // namespace aether
