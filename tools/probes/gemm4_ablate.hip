// Measurement only (not part of libaether_hip.so): the four-wave GEMM main loop of aether_amd/csrc/gemm4_kernel.hpp with parts of
// it switched off through the kernel's ABL template hooks — bit 0 drops the MFMAs, bit 1 the LDS-DMA, bit 2 the fragment reads,
// bit 3 turns the LDS-DMA into plain register loads, bit 4 only adds the timers (shader cycles / 100-MHz ticks of the main loop
// and the cycles spent in its s_waitcnt / s_barrier), bit 5 makes every workgroup stream tile (0,0).  Results are wrong by
// construction; only the times matter.  Driven by tools/gpu_gemm4_ablate.py.
#include "../../aether_amd/csrc/gemm4_kernel.hpp"

using namespace aether;

extern "C" int run_gemm4_ablate(int abl, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const float* bias,
                                float* timers, void* stream) {
    GemmArgs p = {};
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.rows_per_batch = M;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    p.a_bytes = (unsigned)(((size_t)(M - 1) * lda + K) * 2); p.w_bytes = (unsigned)(((size_t)(N - 1) * ldw + K) * 2);
    p.ksplit = 1; p.ntile_launch = p.tiles_m * p.tiles_n; p.part = timers;
    dim3 grid(p.ntile_launch), block(256);
    hipStream_t s = (hipStream_t)stream;
#define CASE(X) case X: hipLaunchKernelGGL((gemm4_bf16_kernel<EPI_BIAS, true, X>), grid, block, 0, s, p); break;
    switch (abl) {
        CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(28) CASE(29) CASE(48) CASE(53)
        default: return -1;
    }
#undef CASE
    return (int)hipGetLastError();
}
