// Measurement only (profiles/r06_gemm_energy.txt): the shipped 256x256x64 GEMM loop of aether_amd/csrc/gemm_kernel.hpp rebuilt with ONE compile-time knob
// changed per shared object (-DAETHER_GEMM_KSPS=2 | -DAETHER_GEMM_MFMA_ORDER=1|2 | -DAETHER_GEMM_SETPRIO=0), launched on the DiT's qkv / ff-up shapes as
// ONE launch over all tiles (no tail split: the knobs act on the main loop).  Driven by tools/gpu_gemm_variants.py; not part of libaether_hip.so.
#include "../../aether_amd/csrc/gemm_kernel.hpp"

using namespace aether;

extern "C" int run_gemm_variant(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* bias, int epilogue, void* stream) {
    GemmArgs p = {};
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.rows_per_batch = M;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256; p.ntile_launch = p.tiles_m * p.tiles_n; p.ksplit = 1;
    p.a_bytes = (unsigned)(((size_t)(M - 1) * lda + K) * 2); p.w_bytes = (unsigned)(((size_t)(N - 1) * ldw + K) * 2);
    dim3 grid(p.ntile_launch), block(512);
    if (epilogue == 1) hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, EPI_BIAS_GELU, true, false>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<2, 4, 4, 2, EPI_BIAS, true, false>), grid, block, 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
