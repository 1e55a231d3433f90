// bf16 GEMM with fused epilogues for the DiT blocks:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Replaces the torch.nn.Linear calls that diffusers' CogVideoXBlock issues under the reference's
// transformer call (aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875): to_q/to_k/to_v, to_out,
// ff.net.0.proj (+GELU-tanh), ff.net.2 (+gate·x+residual), patch_embed.proj/text_proj, proj_out.
//
// gfx950 design: 256x256x64 workgroup tile, 8 wavefronts (2 along M x 4 along N), 128x64 per wave
// on v_mfma_f32_32x32x16_bf16.  The MFMA is issued "swapped" (W fragment as the A operand, activation
// fragment as the B operand) so that each lane ends up owning 4 CONSECUTIVE output columns of one output
// row -> 8-byte (or, with the half-wave exchange, 16-byte) bf16 stores without an LDS transpose.
// Operand tiles are staged HBM->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered
// (2 x 64 KiB), one barrier per K tile; the 128-byte LDS rows are XOR-swizzled (16-B chunk ^= (row>>1)&7,
// applied on the DMA *source* address and again on the ds_read_b128 address) so fragment reads are
// bank-conflict free.  Workgroup ids are remapped XCD-aware and grouped 4(M) x 8(N) for L2 reuse.
#include "common.hpp"
#include "../../include/aether_hip.h"

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
};

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int GROUP_M = 4;

template <int EPI, bool WIDE_STORE>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2;  // 0..1
    const int wn = wave & 3;   // 0..3
    const int hi = lane >> 5;
    const int l32 = lane & 31;

    // ---- tile assignment -------------------------------------------------------------------
    const int nwg = p.tiles_m * p.tiles_n;
    const int wgid = xcd_remap(blockIdx.x, nwg);
    const int per_group = GROUP_M * p.tiles_n;
    const int group = wgid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(GROUP_M, p.tiles_m - first_m);
    const int in_group = wgid - group * per_group;
    const int tile_m = first_m + in_group % gsz;
    const int tile_n = in_group / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging addresses -----------------------------------------------------------------
    // One wave-instruction moves 8 rows x 128 B.  Round r (0..3) covers tile rows r*64 + wave*8 + lane/8.
    // LDS image is linear; the source chunk index is the swizzled one.
    const int srow = wave * 8 + (lane >> 3);                       // row within a 64-row round
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);             // logical 16-B chunk to fetch
    unsigned a_off[4], w_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int am = min(m0 + r * 64 + srow, p.M - 1);
        int wr = min(n0 + r * 64 + srow, p.N - 1);
        a_off[r] = (unsigned)am * (unsigned)p.lda + schunk * 8;
        w_off[r] = (unsigned)wr * (unsigned)p.ldw + schunk * 8;
    }
    char* const lds_stage = smem + wave * 1024;  // + buf*BUF_BYTES + (W? TILE_BYTES) + r*8192

    auto stage = [&](int kt, int buf) {
        const bf16_t* Ak = p.A + kt * BK;
        const bf16_t* Wk = p.W + kt * BK;
        char* dst = lds_stage + buf * BUF_BYTES;
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(Ak + a_off[r], dst + r * 8192);
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(Wk + w_off[r], dst + TILE_BYTES + r * 8192);
    };

    // ---- fragment read addresses -----------------------------------------------------------
    const int swz = (lane >> 1) & 7;  // == ((row>>1)&7) for row = 32*x + l32
    const int x_row_base = (wm * 128 + l32) * 128;                 // + mt*32*128
    const int w_row_base = TILE_BYTES + (wn * 64 + l32) * 128;     // + nt*32*128
    int chunk_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) chunk_off[ks] = (((2 * ks + hi) ^ swz) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    wait_vmcnt<0>();
    block_barrier();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* base = smem + cur * BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 wf[2], xf[4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[nt] = *(const bf16x8*)(base + w_row_base + nt * 4096 + chunk_off[ks]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                xf[mt] = *(const bf16x8*)(base + x_row_base + mt * 4096 + chunk_off[ks]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
        }
        drain_and_barrier();
    }

    // ---- epilogue --------------------------------------------------------------------------
    // acc[mt][nt][r] = C[m][n], m = m0 + wm*128 + mt*32 + l32,
    //                           n = n0 + wn*64 + nt*32 + 8*(r>>2) + 4*hi + (r&3)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 128 + mt * 32 + l32;
        const bool m_ok = m < p.M;
        const float* gate = nullptr;
        if (EPI == EPI_BIAS_GATE_RES && p.gate_vid != nullptr) {
            const int mm = m_ok ? m : 0;
            const int b = mm / p.rows_per_batch;
            const int t = mm - b * p.rows_per_batch;
            gate = (t < p.n_text ? p.gate_txt : p.gate_vid) + (size_t)b * p.gate_bstride;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int nbase = n0 + wn * 64 + nt * 32;   // N % 32 == 0: a 32-column group is all in or all out
            if (nbase >= p.N) continue;                 // wave-uniform
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[mt][nt][4 * g + c];
                if (p.bias != nullptr) {
                    const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] += bv[c];
                }
                if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = gelu_tanh(v[c]);
                }
                if (EPI == EPI_BIAS_GATE_RES) {
                    if (gate != nullptr) {
                        const f32x4 gv = *(const f32x4*)(gate + n);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] *= gv[c];
                    }
                    if (p.R != nullptr) {
                        u16x4 rv = {0, 0, 0, 0};
                        if (m_ok) rv = *(const u16x4*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] += bf16_bits_to_f32(rv[c]);
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

}  // namespace aether

using namespace aether;

extern "C" int aether_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                const float* bias, int epilogue, const void* R, int ldr, const float* gate_vid,
                                const float* gate_txt, int gate_bstride, int rows_per_batch, int n_text, int flags,
                                void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return aether_set_error(AETHER_ERR_SHAPE, "gemm: empty problem");
    if (K % BK != 0) return aether_set_error(AETHER_ERR_SHAPE, "gemm: K must be a multiple of 64");
    if (N % 32 != 0 || (lda % 8) || (ldw % 8) || (ldc % 8) || (R && (ldr % 8)))
        return aether_set_error(AETHER_ERR_SHAPE, "gemm: N must be a multiple of 32 and leading dimensions multiples of 8");
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)R | (uintptr_t)bias | (uintptr_t)gate_vid |
         (uintptr_t)gate_txt) & 15)
        return aether_set_error(AETHER_ERR_ALIGN, "gemm: pointers must be 16-byte aligned");
    if ((size_t)M * (size_t)lda >= (1ull << 32) || (size_t)N * (size_t)ldw >= (1ull << 32))
        return aether_set_error(AETHER_ERR_SHAPE, "gemm: operand exceeds 32-bit element offsets");
    if (epilogue == EPI_BIAS_GATE_RES && (gate_vid != nullptr) != (gate_txt != nullptr))
        return aether_set_error(AETHER_ERR_ARG, "gemm: gate_vid and gate_txt must be given together");
    GemmArgs p;
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias;
    p.R = (const bf16_t*)R; p.ldr = ldr;
    p.gate_vid = gate_vid; p.gate_txt = gate_txt; p.gate_bstride = gate_bstride;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    p.n_text = n_text;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    dim3 grid(p.tiles_m * p.tiles_n), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (flags & AETHER_GEMM_WIDE_STORE) != 0;
#define LAUNCH(E)                                                                              \
    do {                                                                                       \
        if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<E, true>), grid, block, 0, s, p);       \
        else hipLaunchKernelGGL((gemm_bf16_kernel<E, false>), grid, block, 0, s, p);           \
    } while (0)
    switch (epilogue) {
        case EPI_BIAS: LAUNCH(EPI_BIAS); break;
        case EPI_BIAS_GELU: LAUNCH(EPI_BIAS_GELU); break;
        case EPI_BIAS_GATE_RES: LAUNCH(EPI_BIAS_GATE_RES); break;
        default: return aether_set_error(AETHER_ERR_ARG, "gemm: unknown epilogue");
    }
#undef LAUNCH
    return aether_check_launch("gemm_bf16");
}
