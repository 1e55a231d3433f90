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

}  // namespace aether

extern "C" int aether_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                const float* bias, int epilogue, const void* R, int ldr, const float* gate_vid,
                                const float* gate_txt, int gate_bstride, int rows_per_batch, int n_text, int flags,
                                void* stream) {
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
    p.stagger = (flags >> 2) & 3;
    p.a_bytes = (unsigned)(((size_t)(M - 1) * lda + K) * 2);
    p.w_bytes = (unsigned)(((size_t)(N - 1) * ldw + K) * 2);
    dim3 grid(p.tiles_m * p.tiles_n), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (flags & AETHER_GEMM_WIDE_STORE) != 0;
#define LAUNCH(E)                                                                                          \
    do {                                                                                                   \
        if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, E, true, false>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, E, false, false>), grid, block, 0, s, p);     \
    } while (0)
    switch (epilogue) {
        case EPI_BIAS: LAUNCH(EPI_BIAS); break;
        case EPI_BIAS_GELU: LAUNCH(EPI_BIAS_GELU); break;
        default: LAUNCH(EPI_BIAS_GATE_RES); break;
    }
#undef LAUNCH
    return aether_check_launch("gemm_bf16");
}
