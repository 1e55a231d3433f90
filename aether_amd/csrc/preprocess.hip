// Input preprocessing on the device: the array branch of the reference's `_preprocess_image`
// (/root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py:451-460) in ONE pass over the raw frames:
//   uint8 -> float32 / 255            (P:455-456, a true fp32 division)
//   imcrop_center                      (aether/utils/preprocess_utils.py:4-39: centred window of the target aspect ratio, zero fill
//                                       where the window leaves the frame)
//   VideoProcessor.preprocess          (P:459: torch.nn.functional.interpolate(size=(H, W)), i.e. NEAREST with PyTorch's index rule
//                                       src = min(floor(dst * float(in) / out), in - 1), then 2x - 1 in fp32)
//   NHWC -> NCHW, one rounding to bf16 (the pipeline's `.to(device, dtype=bfloat16)`, P:476-496)
// The raw clip is uploaded once (uint8: a quarter of the float32 bytes) and never touched by the host again.
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

struct PreArgs {
    const void* src; int is_u8;        // [N, Hs, Ws, C] uint8 or float32
    int N, Hs, Ws, C;
    int top, left, ch, cw;             // crop window in source coordinates (may leave the frame)
    int H, W;                          // output size
    unsigned short* out;               // bf16 [N, C, H, W]
};

__global__ __launch_bounds__(256) void preprocess_frames_kernel(PreArgs p) {
    const long total = (long)p.N * p.C * p.H * p.W;
    const float sy = (float)p.ch / (float)p.H, sx = (float)p.cw / (float)p.W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.W); long r = i / p.W;
        const int y = (int)(r % p.H); r /= p.H;
        const int c = (int)(r % p.C); const int n = (int)(r / p.C);
        // nearest source pixel inside the crop window (identity when the window already has the output size)
        const int cy = (p.ch == p.H) ? y : min((int)floorf((float)y * sy), p.ch - 1);
        const int cx = (p.cw == p.W) ? x : min((int)floorf((float)x * sx), p.cw - 1);
        const int yy = p.top + cy, xx = p.left + cx;
        float v = 0.f;                                                     // zero fill outside the frame
        if (yy >= 0 && yy < p.Hs && xx >= 0 && xx < p.Ws) {
            const size_t o = (((size_t)n * p.Hs + yy) * p.Ws + xx) * p.C + c;
            v = p.is_u8 ? (float)((const unsigned char*)p.src)[o] / 255.0f : ((const float*)p.src)[o];
        }
        p.out[i] = f32_to_bf16_bits(2.0f * v - 1.0f);
    }
}

}  // namespace aether

using namespace aether;

extern "C" int aether_preprocess_frames(const void* src, int is_u8, int N, int Hs, int Ws, int C, int top, int left, int ch, int cw, int H,
                                        int W, void* out, void* stream) {
    if (!src || !out || N <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || ch <= 0 || cw <= 0 || H <= 0 || W <= 0)
        return aether_set_error(AETHER_ERR_ARG, "preprocess_frames: bad arguments");
    PreArgs p{src, is_u8, N, Hs, Ws, C, top, left, ch, cw, H, W, (unsigned short*)out};
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(preprocess_frames_kernel, dim3((unsigned)std::min<long>(16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return aether_check_launch("preprocess_frames");
}
