// bf16 GEMM with fused epilogues for the DiT blocks:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Replaces the torch.nn.Linear calls that diffusers' CogVideoXBlock issues under the reference's
// transformer call (aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875): to_q/to_k/to_v, to_out,
// ff.net.0.proj (+GELU-tanh), ff.net.2 (+gate·x+residual), patch_embed.proj/text_proj, proj_out.
// Kernel: gemm_kernel.hpp, 256x256x64 tile (2x4 waves, 128x64 per wave).
#include "gemm_kernel.hpp"
#include "../../include/aether_hip.h"

using namespace aether;

namespace aether {

// shared by the linear and the convolution front-ends
int gemm_check_common(const void* A, const void* W, const void* C, const void* R, const float* bias, const float* gate_vid,
                      const float* gate_txt, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue) {
    if (!A || !W || !C) return aether_set_error(AETHER_ERR_ARG, "gemm: null operand");
    if (M <= 0 || N <= 0 || K <= 0) return aether_set_error(AETHER_ERR_SHAPE, "gemm: empty problem");
    if (K % GEMM_BK != 0) return aether_set_error(AETHER_ERR_SHAPE, "gemm: K must be a multiple of 64");
    if (N % 32 != 0 || (lda % 8) || (ldw % 8) || (ldc % 8) || (R && (ldr % 8)))
        return aether_set_error(AETHER_ERR_SHAPE, "gemm: N must be a multiple of 32 and leading dimensions multiples of 8");
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)R | (uintptr_t)bias | (uintptr_t)gate_vid | (uintptr_t)gate_txt) & 15)
        return aether_set_error(AETHER_ERR_ALIGN, "gemm: pointers must be 16-byte aligned");
    if ((size_t)N * (size_t)ldw >= (1ull << 32)) return aether_set_error(AETHER_ERR_SHAPE, "gemm: weight exceeds 32-bit element offsets");
    if (epilogue == EPI_BIAS_GATE_RES && (gate_vid != nullptr) != (gate_txt != nullptr))
        return aether_set_error(AETHER_ERR_ARG, "gemm: gate_vid and gate_txt must be given together");
    if (epilogue < 0 || epilogue > 2) return aether_set_error(AETHER_ERR_ARG, "gemm: unknown epilogue");
    return AETHER_OK;
}

// Tail launch finalize: C tile = epi( sum over K slices of the fp32 partial tile ), partials tile-major
// [slice][tile - tile_base][256][256] (slice order fixed -> deterministic).  One thread = 8 consecutive columns of one row.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_tail_finalize_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256;
    const int tile = blockIdx.y;                                  // tile - tile_base
    int tile_m, tile_n;
    gemm_tile_coords(p.tile_base + tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int item = blockIdx.x * 256 + threadIdx.x;              // BM * BN / 8 items per tile
    const int row = item / (BN / 8), oct = item - row * (BN / 8);
    const int m = tile_m * BM + row, n = tile_n * BN + oct * 8;
    if (m >= p.M || n >= p.N) return;
    const size_t slice = (size_t)p.ntile_launch * BM * BN;
    const float* src = p.part + (size_t)tile * BM * BN + (size_t)row * BN + oct * 8;
    f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
    for (int s = 1; s < p.ksplit; ++s) {
        a0 += *(const f32x4*)(src + s * slice);
        a1 += *(const f32x4*)(src + s * slice + 4);
    }
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    if (p.bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e];
    }
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
    }
    if (EPI == EPI_BIAS_GATE_RES) {
        if (p.gate_vid != nullptr) {
            const int b = m / p.rows_per_batch, t = m - b * p.rows_per_batch;
            const float* gate = (t < p.n_text ? p.gate_txt : p.gate_vid) + (size_t)b * p.gate_bstride;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gate[n + e];
        }
        if (p.R != nullptr) {
            const u16x8 r = *(const u16x8*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf16_bits_to_f32(r[e]);
        }
    }
    *(uint4*)(p.C + (size_t)m * p.ldc + n) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                       pack_bf16x2(v[6], v[7]));
}

}  // namespace aether

extern "C" int aether_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                const float* bias, int epilogue, const void* R, int ldr, const float* gate_vid,
                                const float* gate_txt, int gate_bstride, int rows_per_batch, int n_text, float* splitk_ws,
                                size_t splitk_ws_bytes, int flags, void* stream) {
    int rc = gemm_check_common(A, W, C, R, bias, gate_vid, gate_txt, M, N, K, lda, ldw, ldc, ldr, epilogue);
    if (rc) return rc;
    if ((size_t)M * (size_t)lda * 2 >= (1ull << 32) || (size_t)N * (size_t)ldw * 2 >= (1ull << 32))
        return aether_set_error(AETHER_ERR_SHAPE, "gemm: operand exceeds the 4 GiB a buffer descriptor can address");
    GemmArgs p = {};
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias;
    p.R = (const bf16_t*)R; p.ldr = ldr;
    p.gate_vid = gate_vid; p.gate_txt = gate_txt; p.gate_bstride = gate_bstride;
    p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    p.n_text = n_text;
    p.tiles_m = (M + 255) / 256;
    p.tiles_n = (N + 255) / 256;
    p.a_bytes = (unsigned)(((size_t)(M - 1) * lda + K) * 2);
    p.w_bytes = (unsigned)(((size_t)(N - 1) * ldw + K) * 2);
    if (R) { const size_t rb = ((size_t)(M - 1) * ldr + N) * 2; if (rb >= (1ull << 32)) return aether_set_error(AETHER_ERR_SHAPE, "gemm: residual exceeds the 4 GiB a buffer descriptor can address"); p.r_bytes = (unsigned)rb; }
    dim3 block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (flags & AETHER_GEMM_WIDE_STORE) != 0;
    // Tail balancing: one 128-KiB-LDS workgroup per CU means ceil(tiles/256) rounds; when the last round would hold only a few
    // tiles (qkv: 2124 = 8*256 + 76, ff-up: 2832 = 11*256 + 16) those tiles go to a second launch that splits their K loop over
    // floor(256/rest) workgroups each (fp32 partial tiles, summed in slice order by gemm_tail_finalize_kernel with the epilogue).
    const int NCU = 256, tiles = p.tiles_m * p.tiles_n, nk = K / GEMM_BK;
    const int full = tiles / NCU * NCU, rest = tiles - full;
    int ks = (rest > 0 && rest <= NCU / 2) ? NCU / rest : 1;
    if (ks > nk / 4) ks = nk / 4;
    if (ks < 2 || splitk_ws == nullptr || full == 0 || (((uintptr_t)splitk_ws) & 15) ||
        (size_t)ks * rest * 256 * 256 * sizeof(float) > splitk_ws_bytes)
        ks = 1;
    p.ntile_launch = (ks > 1) ? full : tiles;
#define LAUNCH(E)                                                                                                        \
    do {                                                                                                                 \
        dim3 grid(p.ntile_launch * p.ksplit);                                                                            \
        if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, E, true, false>), grid, block, 0, s, p);               \
        else hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, E, false, false>), grid, block, 0, s, p);                   \
    } while (0)
#define LAUNCH_EPI()                                        \
    switch (epilogue) {                                     \
        case EPI_BIAS: LAUNCH(EPI_BIAS); break;             \
        case EPI_BIAS_GELU: LAUNCH(EPI_BIAS_GELU); break;   \
        default: LAUNCH(EPI_BIAS_GATE_RES); break;          \
    }
    p.ksplit = 1;
    LAUNCH_EPI();
    rc = aether_check_launch("gemm_bf16");
    if (rc || ks == 1) return rc;
    p.tile_base = full; p.ntile_launch = rest; p.ksplit = ks; p.part = splitk_ws; p.part_tiled = 1;
    LAUNCH_EPI();
    rc = aether_check_launch("gemm_bf16 (tail)");
    if (rc) return rc;
    dim3 fgrid(256 * 256 / 8 / 256, rest);
    switch (epilogue) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_tail_finalize_kernel<EPI_BIAS>), fgrid, dim3(256), 0, s, p); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_tail_finalize_kernel<EPI_BIAS_GELU>), fgrid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_tail_finalize_kernel<EPI_BIAS_GATE_RES>), fgrid, dim3(256), 0, s, p); break;
    }
#undef LAUNCH_EPI
#undef LAUNCH
    return aether_check_launch("gemm_tail_finalize");
}
