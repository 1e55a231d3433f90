// Micro-probe (MI355X): do MFMA work and VALU work issued by DIFFERENT waves of the same SIMD overlap?
// 512-thread workgroups, one per CU (waves w and w+4 share a SIMD).  Modes:
//   0: waves 0-3 run NM MFMAs per iteration, waves 4-7 idle          (MFMA only)
//   1: waves 0-3 idle, waves 4-7 run NV VALU ops (+NX v_exp) per it. (VALU only)
//   2: both                                                            (overlap?)
//   3: every wave runs MFMA then VALU back to back (one stream)       (serial within a wave)
//   4: every wave runs MFMA and VALU interleaved 1 : NV/NM in program order
// build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe ; run: ./overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = 0.001f * (threadIdx.x + e);
    auto mfma8 = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    };
    auto valu = [&]() {   // 16 x (fma, fma, exp) = 48 VALU ops incl. 16 transcendentals
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float t = __builtin_fmaf(v[e], 1.0001f, -0.25f);
            t = __builtin_fmaf(t, 0.999f, 0.125f);
            v[e] = __builtin_amdgcn_exp2f(t) * 0.5f;
        }
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { if (grp == 0) { mfma8(); mfma8(); } }
        else if (MODE == 1) { if (grp == 1) { valu(); valu(); } }
        else if (MODE == 2) { if (grp == 0) { mfma8(); mfma8(); } else { valu(); valu(); } }
        else if (MODE == 3) { mfma8(); valu(); }
        else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                    for (int e = 2 * (4 * r + i); e < 2 * (4 * r + i) + 2; ++e) {
                        float t = __builtin_fmaf(v[e], 1.0001f, -0.25f);
                        t = __builtin_fmaf(t, 0.999f, 0.125f);
                        v[e] = __builtin_amdgcn_exp2f(t) * 0.5f;
                    }
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 16; ++e) s += v[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    // per iteration: mode 0: 16 MFMA on group 0 ; mode 1: 96 VALU (32 exp) on group 1 ; mode 3/4: 8 MFMA + 48 VALU on every wave
    printf("{\"mfma_only_ms\": %.3f, \"valu_only_ms\": %.3f, \"both_groups_ms\": %.3f, \"serial_in_wave_ms\": %.3f, \"interleaved_in_wave_ms\": %.3f, \"iters\": %d}\n",
           t0, t1, t2, t3, t4, iters);
    // cycles per iteration at 2.4 GHz nominal for reference
    printf("cycles/iter @2.4GHz: mfma %.0f valu %.0f both %.0f serial %.0f interleaved %.0f\n", t0 * 2.4e6 / iters, t1 * 2.4e6 / iters,
           t2 * 2.4e6 / iters, t3 * 2.4e6 / iters, t4 * 2.4e6 / iters);
    return 0;
}
