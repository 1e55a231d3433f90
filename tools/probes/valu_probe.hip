// Micro-probe (MI355X): issue cost of the VALU / transcendental / MFMA instructions the attention soft-max is made of,
// alone, mixed inside one wave, and split over the two waves that share a SIMD.  Cycles are shader cycles from
// s_memtime, per loop iteration, for the wave that runs the listed work.
// build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe ; run: ./valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { M_MFMA, M_EXP, M_FMA, M_PKFMA, M_ADD, M_PKADD, M_CVT, M_MAX3, M_EXP_FMA, M_EXP_FMA3, M_MFMA_EXP1, M_MFMA_EXP2, M_MFMA_EXP4,
       M_MFMA_FMA4, M_MFMA_FMA7, M_SPLIT_EXP, M_SPLIT_FMA, M_SPLIT_EXP_HALF, M_MFMA_SOFTMAX, M_SPLIT_SOFTMAX, M_LDEXP, M_NUM };
static const char* kNames[] = {"16 mfma", "16 v_exp", "16 v_fma", "16 v_pk_fma", "16 v_add", "16 v_pk_add", "16 v_cvt_pk_bf16", "16 v_max3",
                               "16 exp + 16 fma (one wave)", "16 exp + 48 fma (one wave)", "16 mfma + 16 exp interleaved (one wave)",
                               "16 mfma + 32 exp interleaved (one wave)", "16 mfma + 64 exp interleaved (one wave)",
                               "16 mfma + 64 fma interleaved (one wave)", "16 mfma + 112 fma interleaved (one wave)",
                               "wave A 16 mfma || wave B 32 exp (same SIMD)", "wave A 16 mfma || wave B 128 fma (same SIMD)",
                               "wave A 16 mfma || wave B 16 exp (same SIMD)", "16 mfma + (32 exp, 32 add, 16 cvt) interleaved (one wave)",
                               "wave A 16 mfma || wave B (32 exp, 32 add, 16 cvt)", "16 v_ldexp"};

#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c2))
#define LDEXP(x) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x) : "v"(ci))
#define MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p1), "v"(p2))
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(p2))
#define CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);   // with 512 threads waves w and w+4 share a SIMD
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[16]; f32x2 pv[16]; unsigned cv[16];
    for (int e = 0; e < 16; ++e) { v[e] = 0.001f * (threadIdx.x + e) - 0.3f; pv[e] = f32x2{v[e], -v[e]}; cv[e] = 0; }
    float c1 = 0.999f, c2 = -0.0001f; f32x2 p1 = {0.999f, 0.998f}, p2 = {-0.0001f, 0.0001f}; int ci = 0;
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(p1), "+v"(p2), "+v"(ci));
#define MF(i) acc[(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(i) & 3], 0, 0, 0)
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_MFMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) MF(i);
        } else if (MODE == M_EXP) {
#pragma unroll
            for (int e = 0; e < 16; ++e) EXP(v[e]);
        } else if (MODE == M_FMA) {
#pragma unroll
            for (int e = 0; e < 16; ++e) FMA(v[e]);
        } else if (MODE == M_PKFMA) {
#pragma unroll
            for (int e = 0; e < 16; ++e) PKFMA(pv[e]);
        } else if (MODE == M_ADD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ADD(v[e]);
        } else if (MODE == M_LDEXP) {
#pragma unroll
            for (int e = 0; e < 16; ++e) LDEXP(v[e]);
        } else if (MODE == M_PKADD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) PKADD(pv[e]);
        } else if (MODE == M_CVT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) CVT(cv[e], v[e], v[(e + 1) & 15]);
        } else if (MODE == M_MAX3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) MAX3(v[e]);
        } else if (MODE == M_EXP_FMA) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { EXP(v[e]); FMA(v[(e + 8) & 15]); }
        } else if (MODE == M_EXP_FMA3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { EXP(v[e]); FMA(v[(e + 8) & 15]); FMA(v[(e + 9) & 15]); FMA(v[(e + 10) & 15]); }
        } else if (MODE == M_MFMA_EXP1 || MODE == M_MFMA_EXP2 || MODE == M_MFMA_EXP4) {
            constexpr int K = MODE == M_MFMA_EXP1 ? 1 : MODE == M_MFMA_EXP2 ? 2 : 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                MF(i);
#pragma unroll
                for (int k = 0; k < K; ++k) EXP(v[(i * K + k) & 15]);
            }
        } else if (MODE == M_MFMA_FMA4 || MODE == M_MFMA_FMA7) {
            constexpr int K = MODE == M_MFMA_FMA4 ? 4 : 7;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                MF(i);
#pragma unroll
                for (int k = 0; k < K; ++k) FMA(v[(i * K + k) & 15]);
            }
        } else if (MODE == M_MFMA_SOFTMAX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                MF(i);
                EXP(v[(2 * i) & 15]); EXP(v[(2 * i + 1) & 15]); ADD(v[(2 * i + 8) & 15]); ADD(v[(2 * i + 9) & 15]);
                CVT(cv[i], v[(2 * i + 4) & 15], v[(2 * i + 5) & 15]);
            }
        } else if (MODE == M_SPLIT_EXP || MODE == M_SPLIT_FMA || MODE == M_SPLIT_EXP_HALF || MODE == M_SPLIT_SOFTMAX) {
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) MF(i);
            } else if (MODE == M_SPLIT_EXP) {
#pragma unroll
                for (int e = 0; e < 32; ++e) EXP(v[e & 15]);
            } else if (MODE == M_SPLIT_EXP_HALF) {
#pragma unroll
                for (int e = 0; e < 16; ++e) EXP(v[e & 15]);
            } else if (MODE == M_SPLIT_FMA) {
#pragma unroll
                for (int e = 0; e < 128; ++e) FMA(v[e & 15]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    EXP(v[(2 * i) & 15]); EXP(v[(2 * i + 1) & 15]); ADD(v[(2 * i + 8) & 15]); ADD(v[(2 * i + 9) & 15]);
                    CVT(cv[i], v[(2 * i + 4) & 15], v[(2 * i + 5) & 15]);
                }
            }
            __builtin_amdgcn_s_barrier();   // both waves advance together: an iteration costs max(A, B) if they overlap
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 16; ++e) s += v[e] + pv[e][0] + pv[e][1] + (float)cv[e];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[2 + wave] = t1 - t0;
}

template <int MODE>
void run(long long* d, int iters, bool two_waves) {
    const int threads = two_waves ? 512 : 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, d, 64);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_iter_wave0\": %.1f, \"cycles_per_iter_wave4\": %.1f, \"wall_us_per_iter\": %.4f, \"implied_GHz\": %.3f}\n",
           kNames[MODE], two_waves ? 2 : 1, (double)h[2] / iters, two_waves ? (double)h[6] / iters : 0.0, ms * 1e3 / iters,
           (double)h[2] / iters / (ms * 1e3 / iters) * 1e-3);
}

template <int M>
void run_all(long long* d, int iters) {
    if constexpr (M < M_NUM) {
        const bool split = (M == M_SPLIT_EXP || M == M_SPLIT_FMA || M == M_SPLIT_EXP_HALF || M == M_SPLIT_SOFTMAX);
        run<M>(d, iters, split);
        if (!split && (M == M_MFMA || M == M_EXP || M == M_FMA || M == M_MFMA_SOFTMAX || M == M_MFMA_EXP2)) run<M>(d, iters, true);
        run_all<M + 1>(d, iters);
    }
}

int main() {
    long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    run_all<0>(d, 20000);
    return 0;
}
