// One kernel for the element-wise tail of a denoise step (reference: aether/pipelines/aetherv1_pipeline_cogvideox.py:876-916 — the fp32 cast
// of the noise prediction P:877, the classifier-free-guidance combine P:895-899, CogVideoXDPMScheduler.step P:907-915 (diffusers
// schedulers/scheduling_dpm_cogvideox.py, v-prediction, SDE-DPM-Solver++(2M)) and the cast back to the latent dtype P:916).  In PyTorch
// that is ~12 element-wise launches over 3.3 M latent values per step (2.3 ms of the 230 ms step at B = 1); here it is one pass.
//
// BIT-EXACT with the PyTorch sequence: every intermediate is rounded exactly where eager PyTorch rounds it — a bf16 tensor times a
// Python / 0-dim float64 scalar is computed in fp32 and rounded to bf16; fp32 tensors multiply by the scalar cast to fp32; bf16 and
// fp32 operands meet in fp32 — and nothing is contracted into an FMA (`#pragma clang fp contract(off)` below).  The random draws stay in PyTorch (same generator, same order, same shapes and dtype as the reference): the
// kernel receives the drawn noise tensors.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "../../include/aether_hip.h"

// HIP's __f*_rn intrinsics are plain operators (unlike CUDA's they do not stop contraction) and hipcc contracts a*b - c*d into an FMA by
// default: every product here must be rounded on its own, like the separate PyTorch kernels round theirs.
#pragma clang fp contract(off)

namespace aether {

struct DpmArgs {
    const unsigned short* mo;      // model output, bf16 [nb][n] (nb = 2: unconditional, conditional)
    const unsigned short* sample;  // latents, bf16 [n]
    const float* old_x0;           // fp32 [n] or null (first-order update)
    const unsigned short* noise;   // bf16 [n]: the draw the returned sample uses
    float* x0_out;                 // fp32 [n]
    float* prev_f32;               // fp32 [n] or null
    unsigned short* prev_bf16;     // bf16 [n] or null
    long n;
    int nb;
    float guidance, a_sqrt, b_sqrt, m1, m2, m_noise, m3, m4;
};

AE_DEV float bf16_round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
// plain operators written HERE, under the pragma (the __f*_rn helpers of the HIP headers carry their own header's contraction setting)
AE_DEV float mul_rn(float a, float b) { return a * b; }
AE_DEV float add_rn(float a, float b) { return a + b; }
AE_DEV float sub_rn(float a, float b) { return a - b; }

__global__ __launch_bounds__(256) void dpm_step_kernel(DpmArgs p) {
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < p.n; i += stride) {
        const u16x8 s8 = *(const u16x8*)(p.sample + i);
        const u16x8 z8 = *(const u16x8*)(p.noise + i);
        const u16x8 u8 = *(const u16x8*)(p.mo + i);
        u16x8 c8 = u8;
        if (p.nb == 2) c8 = *(const u16x8*)(p.mo + p.n + i);
        float old[8];
        if (p.old_x0 != nullptr) {
            const f32x4 o0 = *(const f32x4*)(p.old_x0 + i), o1 = *(const f32x4*)(p.old_x0 + i + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { old[e] = o0[e]; old[4 + e] = o1[e]; }
        }
        float x0v[8], pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = bf16_bits_to_f32(s8[e]);
            float mo = bf16_bits_to_f32(u8[e]);                                   // noise_pred.float()                       P:877
            if (p.nb == 2) {                                                      // uncond + g * (cond - uncond), fp32       P:895-899
                const float c = bf16_bits_to_f32(c8[e]);
                mo = add_rn(mo, mul_rn(p.guidance, sub_rn(c, mo)));
            }
            // x0 = sqrt(a_t) * sample [bf16 tensor x scalar -> bf16]  -  sqrt(1 - a_t) * model_output [fp32]
            const float x0 = sub_rn(bf16_round(mul_rn(p.a_sqrt, s)), mul_rn(p.b_sqrt, mo));
            float d = x0;
            if (p.old_x0 != nullptr) d = sub_rn(mul_rn(p.m3, x0), mul_rn(p.m4, old[e]));      // (1 + 1/2r) x0 - (1/2r) x0_old
            // prev = m1 * sample [bf16]  -  m2 * d [fp32]  +  m_noise * noise [bf16]
            const float prev = add_rn(sub_rn(bf16_round(mul_rn(p.m1, s)), mul_rn(p.m2, d)),
                                         bf16_round(mul_rn(p.m_noise, bf16_bits_to_f32(z8[e]))));
            x0v[e] = x0; pv[e] = prev;
        }
        *(f32x4*)(p.x0_out + i) = f32x4{x0v[0], x0v[1], x0v[2], x0v[3]};
        *(f32x4*)(p.x0_out + i + 4) = f32x4{x0v[4], x0v[5], x0v[6], x0v[7]};
        if (p.prev_f32 != nullptr) {
            *(f32x4*)(p.prev_f32 + i) = f32x4{pv[0], pv[1], pv[2], pv[3]};
            *(f32x4*)(p.prev_f32 + i + 4) = f32x4{pv[4], pv[5], pv[6], pv[7]};
        }
        if (p.prev_bf16 != nullptr)
            *(uint4*)(p.prev_bf16 + i) = make_uint4(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7]));
    }
}

}  // namespace aether

using namespace aether;

extern "C" int aether_dpm_step(const void* model_out, int nb, float guidance, const void* sample, const float* old_x0, const void* noise,
                               float a_sqrt, float b_sqrt, float m1, float m2, float m_noise, float m3, float m4, float* x0_out,
                               float* prev_f32, void* prev_bf16, long n, void* stream) {
    if (!model_out || !sample || !noise || !x0_out || (!prev_f32 && !prev_bf16)) return aether_set_error(AETHER_ERR_ARG, "dpm_step: null argument");
    if (nb != 1 && nb != 2) return aether_set_error(AETHER_ERR_ARG, "dpm_step: nb must be 1 (no guidance) or 2 (unconditional, conditional)");
    if (n <= 0 || (n & 7)) return aether_set_error(AETHER_ERR_SHAPE, "dpm_step: element count must be a positive multiple of 8");
    if (((uintptr_t)model_out | (uintptr_t)sample | (uintptr_t)old_x0 | (uintptr_t)noise | (uintptr_t)x0_out | (uintptr_t)prev_f32 | (uintptr_t)prev_bf16) & 15)
        return aether_set_error(AETHER_ERR_ALIGN, "dpm_step: pointers must be 16-byte aligned");
    DpmArgs p;
    p.mo = (const unsigned short*)model_out; p.sample = (const unsigned short*)sample; p.old_x0 = old_x0; p.noise = (const unsigned short*)noise;
    p.x0_out = x0_out; p.prev_f32 = prev_f32; p.prev_bf16 = (unsigned short*)prev_bf16; p.n = n; p.nb = nb;
    p.guidance = guidance; p.a_sqrt = a_sqrt; p.b_sqrt = b_sqrt; p.m1 = m1; p.m2 = m2; p.m_noise = m_noise; p.m3 = m3; p.m4 = m4;
    const long items = n / 8;
    const unsigned blocks = (unsigned)std::min<long>(4096, (items + 255) / 256);
    hipLaunchKernelGGL(dpm_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return aether_check_launch("dpm_step");
}
