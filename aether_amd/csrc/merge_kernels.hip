// Sliding-window merge on the device: the per-pixel part of the reference's blend_and_merge_window_results
// (/root/reference/scripts/demo.py:254-422) and of the helpers it calls — compute_scale
// (aether/utils/postprocess_utils.py:847-864: masked least-squares scale of a window's disparity onto the merged one), the
// linear cross-fades of disparity / colour over the overlap (D:303-326) and the back-projection of every frame to a world-space
// point map (project, U:393-403).  HBM-bound streaming kernels, float64 arithmetic like the reference's numpy (fp64 FMA is
// full rate on CDNA4: the passes stay bandwidth bound), the window outputs read straight from the gathered fp32 device buffers.
// The few-dozen-pose camera algebra (raymap -> poses, similarity alignment, slerp) stays on the host.
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

// ---- masked scale fit: num = sum m*p*t, den = sum m*p*p over the overlap, m = p > 0.1; fp32 products like the reference's
//      torch code, float64 accumulation, deterministic two-level reduction -------------------------------------------------
__global__ __launch_bounds__(256) void merge_scale_partial_kernel(const float* __restrict__ pred, const double* __restrict__ target, long n,
                                                                  double* __restrict__ partial) {
    double num = 0.0, den = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float p = pred[i], t = (float)target[i];
        if (p > 0.1f) { num += (double)(p * t); den += (double)(p * p); }
    }
    __shared__ double sn[256], sd[256];
    sn[threadIdx.x] = num; sd[threadIdx.x] = den;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sn[threadIdx.x] += sn[threadIdx.x + s]; sd[threadIdx.x] += sd[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sn[0]; partial[2 * blockIdx.x + 1] = sd[0]; }
}
__global__ void merge_scale_final_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double num = 0.0, den = 0.0;
    for (int i = 0; i < nblk; ++i) { num += partial[2 * i]; den += partial[2 * i + 1]; }
    out[0] = num; out[1] = den;
    out[2] = den != 0.0 ? num / den : 0.0;       // compute_scale: 0 when no pixel passes the mask
}

// ---- one window into the merged arrays ------------------------------------------------------------------------------------
struct MergeArgs {
    const float* w_rgb; const float* w_disp;     // the window's outputs [n_win, HW, 3] / [n_win, HW] fp32
    double* rgb; double* disp;                   // merged arrays, already offset to the window's first frame
    const double* scale;                         // device scalar (out[2] above) or null (first window: scale 1)
    long hw; int n_win, ov;
    double fade[64];                             // np.linspace(1, 0, ov): weight of the already merged frame
};
__global__ __launch_bounds__(256) void merge_window_kernel(MergeArgs p) {
    const long total = (long)p.n_win * p.hw;
    const float sc = p.scale ? (float)p.scale[0] : 1.0f;            // `scale * disparity` is an fp32 product in the reference
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i / p.hw);
        const double wd = (double)(sc * p.w_disp[i]);
        const float r0 = p.w_rgb[3 * i], r1 = p.w_rgb[3 * i + 1], r2 = p.w_rgb[3 * i + 2];
        if (f < p.ov) {
            const double a = p.fade[f], b = 1.0 - a;
            p.disp[i] = p.disp[i] * a + wd * b;
            p.rgb[3 * i] = p.rgb[3 * i] * a + (double)r0 * b;
            p.rgb[3 * i + 1] = p.rgb[3 * i + 1] * a + (double)r1 * b;
            p.rgb[3 * i + 2] = p.rgb[3 * i + 2] * a + (double)r2 * b;
        } else {
            p.disp[i] = wd;
            p.rgb[3 * i] = (double)r0; p.rgb[3 * i + 1] = (double)r1; p.rgb[3 * i + 2] = (double)r2;
        }
    }
}

// ---- back-projection: world = P[:3,:3] · (K^-1 · (u+.5, v+.5, 1) · depth) + P[:3,3],  depth = 1 / clip(disparity, 1e-8, 1e8) ------
__global__ __launch_bounds__(256) void backproject_kernel(const double* __restrict__ disp, const double* __restrict__ Kinv,
                                                          const double* __restrict__ P, double* __restrict__ out, int N, int H, int W) {
    const long hw = (long)H * W, total = (long)N * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / hw);
        const long r = i - (long)n * hw;
        const int v = (int)(r / W), u = (int)(r - (long)v * W);
        const double* K = Kinv + 9 * n;
        const double* M = P + 12 * n;
        const double x = (double)((float)u + 0.5f), y = (double)((float)v + 0.5f);
        const double depth = 1.0 / fmin(fmax(disp[i], 1e-8), 1e8);
        const double cx = (K[0] * x + K[1] * y + K[2]) * depth, cy = (K[3] * x + K[4] * y + K[5]) * depth, cz = (K[6] * x + K[7] * y + K[8]) * depth;
        out[3 * i] = M[0] * cx + M[1] * cy + M[2] * cz + M[3];
        out[3 * i + 1] = M[4] * cx + M[5] * cy + M[6] * cz + M[7];
        out[3 * i + 2] = M[8] * cx + M[9] * cy + M[10] * cz + M[11];
    }
}

}  // namespace aether

using namespace aether;

extern "C" int aether_merge_scale_fit(const float* pred, const double* target, long n, double* scratch, int scratch_doubles, double* out3,
                                      void* stream) {
    if (!pred || !target || !scratch || !out3 || n <= 0) return aether_set_error(AETHER_ERR_ARG, "merge_scale_fit: bad arguments");
    const int nblk = (int)std::min<long>(std::min<long>(2048, scratch_doubles / 2), (n + 255) / 256);
    if (nblk < 1) return aether_set_error(AETHER_ERR_ARG, "merge_scale_fit: scratch too small");
    hipLaunchKernelGGL(merge_scale_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, pred, target, n, scratch);
    hipLaunchKernelGGL(merge_scale_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, nblk, out3);
    return aether_check_launch("merge_scale_fit");
}

extern "C" int aether_merge_window(const float* w_rgb, const float* w_disp, double* rgb, double* disp, int n_win, int ov, long hw,
                                   const double* fade_host, const double* scale_dev, void* stream) {
    if (!w_rgb || !w_disp || !rgb || !disp || n_win <= 0 || ov < 0 || ov > n_win || ov > 64 || hw <= 0 || (ov > 0 && !fade_host))
        return aether_set_error(AETHER_ERR_ARG, "merge_window: bad arguments (overlap <= 64 frames)");
    MergeArgs a;
    a.w_rgb = w_rgb; a.w_disp = w_disp; a.rgb = rgb; a.disp = disp; a.scale = scale_dev; a.hw = hw; a.n_win = n_win; a.ov = ov;
    for (int i = 0; i < ov; ++i) a.fade[i] = fade_host[i];
    const long total = (long)n_win * hw;
    hipLaunchKernelGGL(merge_window_kernel, dim3((unsigned)std::min<long>(16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return aether_check_launch("merge_window");
}

extern "C" int aether_backproject(const double* disp, const double* Kinv, const double* P, double* out, int N, int H, int W, void* stream) {
    if (!disp || !Kinv || !P || !out || N <= 0 || H <= 0 || W <= 0) return aether_set_error(AETHER_ERR_ARG, "backproject: bad arguments");
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(backproject_kernel, dim3((unsigned)std::min<long>(16384, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, disp, Kinv,
                       P, out, N, H, W);
    return aether_check_launch("backproject");
}
