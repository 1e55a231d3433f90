// Kernels of the CogVideoX 3D-causal VAE (diffusers AutoencoderKLCogVideoX), reached from the reference at
// aether/pipelines/aetherv1_pipeline_cogvideox.py:557-618 (encode) and P:931,936 (decode_latents).
//
// Activations are channels-last volumes [NB, T, H, W, C] bf16 (NB = spatial tiles processed as one batch):
// a convolution is then a GEMM whose A rows are voxels and whose K dimension walks (tap, 64-channel block) —
// exactly the MFMA kernel of gemm_kernel.hpp with gathered A rows.  Convolution INPUTS live in zero-bordered
// volumes [NB, T+kt-1, H+ph, W+pw, C]: spatial zero padding and the causal front frames (conv cache / replicated
// first frame) are materialised by the producer (GroupNorm+SiLU apply, resamplers), so the GEMM main loop has no
// bounds checks: row offset = voxel offset, K step = constant tap offset (table) — one scalar add per K tile.
//   aether_conv_gemm_bf16      implicit-GEMM causal conv3d / conv2d (stride 1 or 2), fused bias + residual
//   aether_im2col_first        explicit im2col for the two thin first convs (3 -> 128, 16 -> 512 channels)
//   (GroupNorm / SpatialNorm3D kernels live in vae_norm.hip)
//   aether_time_avgpool_pad / aether_upsample_nearest_pad / aether_pad_copy   resamplers writing padded volumes
#include "conv3_kernel.hpp"
#include "../../include/aether_hip.h"

namespace aether {
int gemm_check_common(const void* A, const void* W, const void* C, const void* R, const float* bias, const float* gate_vid,
                      const float* gate_txt, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue);

// ------------------------------------------------------------------------------------------------
// im2col for thin-channel first convolutions.  x: [C, T_all, H_all, W_all] (strides given), a crop
// (t0.., y0.., x0..) of T x H x W voxels; causal front: frame index < 0 relative to the crop start reads
// frame max(t0 + t - (kt-1) ..., clamp rule below); spatial zero padding at the CROP border.
// Output A [NB? no: single crop] [T*H*W, Kpad], column = ((dt*3+dh)*3+dw)*C + c, zero padded to Kpad.
// ------------------------------------------------------------------------------------------------
struct Im2colArgs {
    const unsigned short* x; long sC, sT, sH, sW;   // element strides of the source
    int C, t0, first_chunk, y0, x0, T, H, W;        // crop; first_chunk: replicate frame t0 instead of reading t0-1,t0-2
    unsigned short* A; int Kpad;
};
// One workgroup per output row (t, h).  The 9 C source rows that row can touch (channel x 3 frames x 3 image rows, W + 2 columns with the zero
// border) are staged in LDS once, the decomposition of the K index (tap-major, channel fastest) into an LDS offset is a table built once per
// workgroup, and a 16-byte piece of A is 8 two-byte LDS reads.  (Round 5: the first version decomposed k with four runtime divisions and fetched
// every element from global memory — 118 us per call, 0.7 TB/s of A written: ALU-bound; 5.3 ms of a 207 ms one-lane encode.)
__global__ __launch_bounds__(256) void im2col_first_kernel(Im2colArgs p) {
    extern __shared__ unsigned short im2col_lds[];
    const int K = 27 * p.C, Wp = p.W + 2, n_rows = 9 * p.C;
    unsigned short* rows = im2col_lds;                               // [n_rows][Wp]
    int* koff = reinterpret_cast<int*>(im2col_lds + (((size_t)n_rows * Wp + 7) & ~(size_t)7));   // [Kpad], 16-byte aligned
    const int t = blockIdx.x / p.H, h = blockIdx.x - t * p.H;
    for (int i = threadIdx.x; i < n_rows * Wp; i += 256) {
        const int r = i / Wp, col = i - r * Wp;
        const int c = r / 9, dt = (r % 9) / 3, dh = r % 3;
        int ts = t + dt - 2;                                         // frame relative to the chunk start
        if (ts < 0 && p.first_chunk) ts = 0;                         // no cache yet: replicate the first frame
        const int hs = h + dh - 1, ws = col - 1;
        unsigned short v = 0;
        if (hs >= 0 && hs < p.H && ws >= 0 && ws < p.W)
            v = p.x[c * p.sC + (long)(p.t0 + ts) * p.sT + (long)(p.y0 + hs) * p.sH + (long)(p.x0 + ws) * p.sW];
        rows[i] = v;
    }
    for (int k = threadIdx.x; k < p.Kpad; k += 256) {
        int o = -1;
        if (k < K) {
            const int c = k % p.C; int tap = k / p.C;
            const int dw = tap % 3; tap /= 3;
            const int dh = tap % 3, dt = tap / 3;
            o = ((c * 3 + dt) * 3 + dh) * Wp + dw;
        }
        koff[k] = o;
    }
    __syncthreads();
    const int groups = p.Kpad / 8;                                   // 16-byte pieces per row of A
    unsigned short* arow = p.A + (size_t)blockIdx.x * p.W * p.Kpad;
    for (int i = threadIdx.x; i < p.W * groups; i += 256) {
        const int w = i / groups, g = i - w * groups;
        u16x8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = koff[g * 8 + e];
            out[e] = o >= 0 ? rows[o + w] : (unsigned short)0;
        }
        *(u16x8*)(arow + (size_t)i * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// Resamplers (all write into a zero-bordered volume [NB, oT, oH, oW, C] at interior offset (pt, ph, pw)).
//  mode 0: copy                                     out(t,h,w) = x(t,h,w)
//  mode 1: temporal average pool k2 s2; if T is odd the first frame is kept (CogVideoXDownsample3D, compress_time)
//  mode 2: nearest x2 in space                      out(t,h,w) = x(t,h/2,w/2)
//  mode 3: nearest x2 in space and time; if T is odd and > 1 the first frame is only spatially up-sampled
//          (CogVideoXUpsample3D, compress_time):  out frames = 1 + 2*(T-1)  [or 2*T when T is even]
// ------------------------------------------------------------------------------------------------
struct ResampleArgs {
    const unsigned short* x; int T, H, W, C;     // source [NB, T, H, W, C]
    unsigned short* y; int nT, nH, nW;           // logical output extent
    int oT, oH, oW, pt, ph, pw; int mode;
};
__global__ __launch_bounds__(256) void resample_pad_kernel(ResampleArgs p) {
    const int nb = blockIdx.y;
    const int oct_per_vox = p.C >> 3;
    const long total = (long)p.nT * p.nH * p.nW * oct_per_vox;
    const unsigned short* xb = p.x + (size_t)nb * p.T * p.H * p.W * p.C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int oct = idx % oct_per_vox;
        long v = idx / oct_per_vox;
        const int w = v % p.nW; long r = v / p.nW;
        const int h = r % p.nH;
        const int t = r / p.nH;
        uint4 out;
        if (p.mode == 1) {
            const bool odd = p.T & 1;
            int ta, tb;
            if (odd) { ta = (t == 0) ? 0 : 2 * t - 1; tb = (t == 0) ? 0 : 2 * t; }
            else { ta = 2 * t; tb = 2 * t + 1; }
            const u16x8 a = *(const u16x8*)(xb + (((size_t)ta * p.H + h) * p.W + w) * p.C + oct * 8);
            const u16x8 b = *(const u16x8*)(xb + (((size_t)tb * p.H + h) * p.W + w) * p.C + oct * 8);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float fa = bf16_bits_to_f32(a[e]), fb = bf16_bits_to_f32(b[e]);
                o[e] = (ta == tb) ? fa : (fa + fb) * 0.5f;
            }
            out = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        } else {
            int ts = t, hs = h, ws = w;
            if (p.mode >= 2) { hs = h >> 1; ws = w >> 1; }
            if (p.mode == 3) {
                if (p.T > 1 && (p.T & 1)) ts = (t == 0) ? 0 : 1 + ((t - 1) >> 1);
                else if (p.T > 1) ts = t >> 1;
                else ts = 0;
            }
            out = *(const uint4*)(xb + (((size_t)ts * p.H + hs) * p.W + ws) * p.C + oct * 8);
        }
        *(uint4*)(p.y + ((((size_t)nb * p.oT + t + p.pt) * p.oH + h + p.ph) * p.oW + w + p.pw) * p.C + oct * 8) = out;
    }
    // everything of the padded volume outside the interior box is zero (round 4: the volume is arena memory of the launch plan, rewritten
    // by every producer, not a persistent zero-bordered pool entry)
    const long vol = (long)p.oT * p.oH * p.oW * oct_per_vox;
    if (vol != total) {
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < vol; idx += (long)gridDim.x * blockDim.x) {
            const int oct = idx % oct_per_vox;
            long v = idx / oct_per_vox;
            const int w = v % p.oW; long r = v / p.oW;
            const int h = r % p.oH;
            const int t = r / p.oH;
            const bool inside = t >= p.pt && t < p.pt + p.nT && h >= p.ph && h < p.ph + p.nH && w >= p.pw && w < p.pw + p.nW;
            if (!inside) *(uint4*)(p.y + ((((size_t)nb * p.oT + t) * p.oH + h) * p.oW + w) * p.C + oct * 8) = zero;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split-K finalize: C[m][n] = bf16( sum_s part[s][m][n] (slice order: deterministic) + bias[n] + R[m][n] )
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ part, int ksplit, int M, int N,
                                                              const float* __restrict__ bias, const unsigned short* __restrict__ R,
                                                              int ldr, unsigned short* __restrict__ C, int ldc) {
    const int oct_per_row = N >> 3;
    const long total = (long)M * oct_per_row;
    const size_t slice = (size_t)M * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int oct = idx % oct_per_row;
        const long m = idx / oct_per_row;
        const float* src = part + (size_t)m * N + oct * 8;
        f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
        for (int s = 1; s < ksplit; ++s) {
            a0 += *(const f32x4*)(src + s * slice);
            a1 += *(const f32x4*)(src + s * slice + 4);
        }
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[oct * 8 + e];
        }
        if (R != nullptr) {
            const u16x8 r = *(const u16x8*)(R + (size_t)m * ldr + oct * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf16_bits_to_f32(r[e]);
        }
        *(uint4*)(C + (size_t)m * ldc + oct * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                              pack_bf16x2(v[6], v[7]));
    }
}

}  // namespace aether

using namespace aether;
#define AE_STREAM ((hipStream_t)stream)

static int grid_for(long total_threads) {
    long b = (total_threads + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

extern "C" int aether_conv_gemm_bf16(const void* X, int NB, int iT, int iH, int iW, int iC, int oT, int oH, int oW, int stride_hw,
                                     const int* tap_off, int n_taps, const void* W, int Cout, void* C, int ldc,
                                     const float* bias, const void* R, int ldr, float* splitk_ws, size_t splitk_ws_bytes, int flags,
                                     void* stream) {
    const int K = n_taps * 64;
    const long Ml = (long)NB * oT * oH * oW;
    if (Ml <= 0 || Ml >= (1l << 31)) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: bad output volume");
    const int M = (int)Ml;
    int rc = gemm_check_common(X, W, C, R, bias, nullptr, nullptr, M, Cout, K, 8, K, ldc, ldr, R ? EPI_BIAS_GATE_RES : EPI_BIAS);
    if (rc) return rc;
    if (!tap_off || n_taps <= 0) return aether_set_error(AETHER_ERR_ARG, "conv_gemm: tap table required");
    if (iC % 64 != 0) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: input channels must be a multiple of 64");
    if (stride_hw != 1 && stride_hw != 2) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: stride must be 1 or 2");
    if ((size_t)NB * iT * iH * iW * iC * 2 >= (1ull << 32) || (size_t)Cout * K * 2 >= (1ull << 32))
        return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: input volume exceeds the 4 GiB a buffer descriptor can address");
    if (splitk_ws != nullptr && (((uintptr_t)splitk_ws) & 15)) return aether_set_error(AETHER_ERR_ALIGN, "conv_gemm: split-K workspace must be 16-byte aligned");
    GemmArgs p = {};
    p.A = (const bf16_t*)X; p.lda = 0;
    p.W = (const bf16_t*)W; p.ldw = K;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = Cout; p.K = K;
    p.bias = bias;
    p.R = (const bf16_t*)R; p.ldr = ldr;
    p.rows_per_batch = M;
    p.tap_off = tap_off;
    p.oT = oT; p.oH = oH; p.oW = oW; p.iT = iT; p.iH = iH; p.iW = iW; p.iC = iC; p.stride_hw = stride_hw;
    const bool wide = (flags & AETHER_GEMM_WIDE_STORE) != 0;
    p.a_bytes = (unsigned)((size_t)NB * iT * iH * iW * iC * 2);
    p.w_bytes = (unsigned)((size_t)Cout * K * 2);
    // the staged epilogue fetches the residual through a buffer descriptor: its extent must fit 32 bits
    if (R) { const size_t rb = ((size_t)(M - 1) * ldr + Cout) * 2; if (rb >= (1ull << 32)) return aether_set_error(AETHER_ERR_SHAPE, "gemm: residual exceeds the 4 GiB a buffer descriptor can address"); p.r_bytes = (unsigned)rb; }
    dim3 block(512);
    // Tap-reuse kernel (conv3_kernel.hpp): 3x3(x3) taps in (dt, dh, channel block, dw) order, unit stride, one-voxel zero border
    // in H and W.  Output rows enumerate the padded plane, so it pays where the border is a small share of the plane.
    if ((flags & AETHER_CONV_TAP_REUSE) && stride_hw == 1 && n_taps % 3 == 0 && iW == oW + 2 && iH == oH + 2 && oH >= 3 &&
        (iT == oT + 2 || iT == oT) && (Cout % 128 == 0 || Cout == 32) && (long)NB * oT * iH * iW < (1l << 31)) {
        p.M = NB * oT * iH * iW;
        p.ksplit = 1;
#define LAUNCH_C3(WM_, WN_, MT_, NT_, BM_, BN_)                                                                                     \
        do {                                                                                                                        \
            p.tiles_m = (p.M + BM_ - 1) / BM_; p.tiles_n = (Cout + BN_ - 1) / BN_; p.ntile_launch = p.tiles_m * p.tiles_n;            \
            dim3 grid(p.ntile_launch);                                                                                              \
            if (R) { if (wide) hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, true>), grid, block, 0, AE_STREAM, p);  \
                     else hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, false>), grid, block, 0, AE_STREAM, p); }   \
            else { if (wide) hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, true>), grid, block, 0, AE_STREAM, p);            \
                   else hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, false>), grid, block, 0, AE_STREAM, p); }               \
        } while (0)
#define LAUNCH_C3_W1(WM_, WN_, MT_, NT_, BM_, BN_)                                                                                  \
        do {                                                                                                                        \
            p.tiles_m = (p.M + BM_ - 1) / BM_; p.tiles_n = (Cout + BN_ - 1) / BN_; p.ntile_launch = p.tiles_m * p.tiles_n;            \
            dim3 grid(p.ntile_launch);                                                                                              \
            if (R) { if (wide) hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, true, true>), grid, block, 0, AE_STREAM, p);  \
                     else hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, false, true>), grid, block, 0, AE_STREAM, p); }   \
            else { if (wide) hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, true, true>), grid, block, 0, AE_STREAM, p);            \
                   else hipLaunchKernelGGL((conv3_gemm_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, false, true>), grid, block, 0, AE_STREAM, p); }               \
        } while (0)
        // 128-wide layers: 512x128 with ONE W buffer (round 6; the 384x128 double-buffered tile of rounds 2-5 measured 2.6-7 % slower per launch
        // in isolation and 0-0.6 % over a whole encode / decode: profiles/r06_conv512_ab.json)
        if (Cout % 256 == 0) LAUNCH_C3(2, 4, 4, 2, 256, 256);
        else if (Cout % 128 == 0) LAUNCH_C3_W1(4, 2, 4, 2, 512, 128);
        else LAUNCH_C3(8, 1, 2, 1, 512, 32);                     // conv_out (3 -> 32 padded output channels)
#undef LAUNCH_C3_W1
#undef LAUNCH_C3
        return aether_check_launch("conv3_gemm_bf16");
    }
    // split-K when the output tiles cannot fill the chip (one 128-KiB-LDS workgroup per CU, 256 CUs): the deep layers have
    // K = 27*512 = 216 K tiles walked serially by a handful of workgroups otherwise.  Slices >= 4 K tiles, total <= 256 WGs.
    auto pick_ksplit = [&](int tiles) {
        if (splitk_ws == nullptr || tiles > 128) return 1;
        int ks = 256 / tiles;
        ks = ks < n_taps / 4 ? ks : n_taps / 4;
        const size_t per_slice = (size_t)M * Cout * sizeof(float);
        if (per_slice == 0) return 1;
        const size_t fit = splitk_ws_bytes / per_slice;
        if ((size_t)ks > fit) ks = (int)fit;
        return ks < 2 ? 1 : ks;
    };
#define LAUNCH_CFG(WM_, WN_, MT_, NT_, BM_, BN_)                                                                                   \
    do {                                                                                                                            \
        p.tiles_m = (M + BM_ - 1) / BM_; p.tiles_n = (Cout + BN_ - 1) / BN_;                                                         \
        p.ksplit = pick_ksplit(p.tiles_m * p.tiles_n); p.part = splitk_ws; p.ntile_launch = p.tiles_m * p.tiles_n;                  \
        dim3 grid(p.tiles_m * p.tiles_n * p.ksplit);                                                                               \
        if (R) { if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, true, true>), grid, block, 0, AE_STREAM, p); \
                 else hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, false, true>), grid, block, 0, AE_STREAM, p); }    \
        else { if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, true, true>), grid, block, 0, AE_STREAM, p);            \
               else hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, false, true>), grid, block, 0, AE_STREAM, p); }               \
    } while (0)
    if (Cout % 256 == 0) LAUNCH_CFG(2, 4, 4, 2, 256, 256);
    else if (Cout % 128 == 0) LAUNCH_CFG(4, 2, 4, 2, 512, 128);
    else LAUNCH_CFG(8, 1, 2, 1, 512, 32);
#undef LAUNCH_CFG
    rc = aether_check_launch("conv_gemm_bf16");
    if (rc || p.ksplit <= 1) return rc;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(grid_for((long)M * (Cout / 8))), dim3(256), 0, AE_STREAM, (const float*)splitk_ws, p.ksplit, M,
                       Cout, bias, (const unsigned short*)R, ldr, (unsigned short*)C, ldc);
    return aether_check_launch("splitk_finalize");
}

extern "C" int aether_im2col_first(const void* x, long sC, long sT, long sH, long sW, int Cin, int t0, int first_chunk, int y0,
                                   int x0, int T, int H, int W, void* A, int Kpad, void* stream) {
    if (!x || !A || Cin <= 0 || T <= 0 || H <= 0 || W <= 0) return aether_set_error(AETHER_ERR_ARG, "im2col_first: bad arguments");
    if (Kpad % 64 != 0 || Kpad < 27 * Cin) return aether_set_error(AETHER_ERR_SHAPE, "im2col_first: Kpad must be a multiple of 64 >= 27*Cin");
    if (!first_chunk && t0 < 2) return aether_set_error(AETHER_ERR_ARG, "im2col_first: later chunks need two preceding frames");
    Im2colArgs p{(const unsigned short*)x, sC, sT, sH, sW, Cin, t0, first_chunk, y0, x0, T, H, W, (unsigned short*)A, Kpad};
    const size_t lds = aether_im2col_first_lds_bytes(Cin, W, Kpad);
    if (lds > AETHER_IM2COL_LDS_LIMIT)
        return aether_set_error(AETHER_ERR_SHAPE, "im2col_first: 9 * Cin * (W + 2) source elements must fit 160 KiB of LDS (the launch plan reports this at "
                                                  "aether_vae_workspace_bytes time)");
    if (lds > 64 * 1024) {                                         // above the default dynamic-LDS limit: raise it for this kernel (a CU has 160 KiB)
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute((const void*)im2col_first_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AETHER_IM2COL_LDS_LIMIT) != hipSuccess)
                return aether_set_error(AETHER_ERR_LAUNCH, "im2col_first: could not raise the dynamic LDS limit");
            raised = true;
        }
    }
    hipLaunchKernelGGL(im2col_first_kernel, dim3(T * H), dim3(256), lds, AE_STREAM, p);
    return aether_check_launch("im2col_first");
}

namespace aether {
// grid (pieces, 4, NB): y = 0,1 -> front frame y of vol; y = 2,3 -> frame y-2 of the cache handed to the next chunk.
__global__ __launch_bounds__(256) void causal_front_kernel(uint4* __restrict__ vol, int Tp, long fvec, const uint4* __restrict__ prev,
                                                           uint4* __restrict__ next) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= fvec) return;
    const int nb = blockIdx.z, y = blockIdx.y;
    uint4* v = vol + (size_t)nb * Tp * fvec;
    int frame = (y < 2) ? y : Tp - 2 + (y - 2);                     // frame of the padded volume this output equals
    uint4 val;
    if (frame >= 2) val = v[(size_t)frame * fvec + i];                // an interior frame (only reached for the cache)
    else val = prev ? prev[((size_t)nb * 2 + frame) * fvec + i] : v[(size_t)2 * fvec + i];
    if (y < 2) v[(size_t)y * fvec + i] = val;
    else next[((size_t)nb * 2 + (y - 2)) * fvec + i] = val;
}
}  // namespace aether

extern "C" int aether_causal_front(void* vol, int NB, int Tp, long frame_elems, const void* prev, void* next, void* stream) {
    if (!vol || !next || NB <= 0 || Tp < 3 || frame_elems <= 0 || (frame_elems % 8) != 0)
        return aether_set_error(AETHER_ERR_ARG, "causal_front: bad arguments (Tp >= 3, frame size a multiple of 8 elements)");
    if ((((uintptr_t)vol | (uintptr_t)prev | (uintptr_t)next) & 15) != 0) return aether_set_error(AETHER_ERR_ALIGN, "causal_front: pointers must be 16-byte aligned");
    const long fvec = frame_elems / 8;
    hipLaunchKernelGGL(causal_front_kernel, dim3((unsigned)((fvec + 255) / 256), 4, NB), dim3(256), 0, AE_STREAM, (uint4*)vol, Tp, fvec,
                       (const uint4*)prev, (uint4*)next);
    return aether_check_launch("causal_front");
}

extern "C" int aether_resample_pad(const void* x, int NB, int T, int H, int W, int C, int mode, void* y, int oT, int oH, int oW,
                                   int pt, int ph, int pw, void* stream) {
    if (!x || !y || C % 8 != 0 || mode < 0 || mode > 3) return aether_set_error(AETHER_ERR_ARG, "resample_pad: bad arguments");
    int nT = T, nH = H, nW = W;
    if (mode == 1) nT = (T & 1) ? (T / 2 + 1) : T / 2;
    if (mode >= 2) { nH = 2 * H; nW = 2 * W; }
    if (mode == 3) nT = (T > 1) ? ((T & 1) ? 2 * T - 1 : 2 * T) : 1;
    if (nT + pt > oT || nH + ph > oH || nW + pw > oW) return aether_set_error(AETHER_ERR_SHAPE, "resample_pad: output volume too small");
    ResampleArgs p{(const unsigned short*)x, T, H, W, C, (unsigned short*)y, nT, nH, nW, oT, oH, oW, pt, ph, pw, mode};
    hipLaunchKernelGGL(resample_pad_kernel, dim3(grid_for((long)nT * nH * nW * (C / 8)), NB), dim3(256), 0, AE_STREAM, p);
    return aether_check_launch("resample_pad");
}
