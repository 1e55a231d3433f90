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
//   aether_groupnorm_stats     deterministic two-level (block partials -> double merge) group statistics
//   aether_groupnorm_apply     normalise + affine (+ SpatialNorm3D: ·conv_y(zq)+conv_b(zq)) + SiLU -> padded volume
//   aether_time_avgpool_pad / aether_upsample_nearest_pad / aether_pad_copy   resamplers writing padded volumes
#include "gemm_kernel.hpp"
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
__global__ __launch_bounds__(256) void im2col_first_kernel(Im2colArgs p) {
    const int K = 27 * p.C;
    const int groups = p.Kpad / 8;                       // 16-byte pieces per row
    const long total = (long)p.T * p.H * p.W * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = idx % groups;
        long m = idx / groups;
        const int w = m % p.W; m /= p.W;
        const int h = m % p.H;
        const int t = m / p.H;
        u16x8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = g * 8 + e;
            unsigned short v = 0;
            if (k < K) {
                const int c = k % p.C;
                int tap = k / p.C;
                const int dw = tap % 3; tap /= 3;
                const int dh = tap % 3;
                const int dt = tap / 3;
                int ts = t + dt - 2;                         // frame relative to the chunk start
                if (ts < 0 && p.first_chunk) ts = 0;         // no cache yet: replicate the first frame
                const int hs = h + dh - 1, ws = w + dw - 1;
                if (hs >= 0 && hs < p.H && ws >= 0 && ws < p.W)
                    v = p.x[c * p.sC + (long)(p.t0 + ts) * p.sT + (long)(p.y0 + hs) * p.sH + (long)(p.x0 + ws) * p.sW];
            }
            out[e] = v;
        }
        *(u16x8*)(p.A + (idx / groups) * p.Kpad + g * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics.  x [NB, V, C]; block b of batch item nb reduces voxels [b*VPB, (b+1)*VPB) into per-channel
// (sum, sum of squares) partials [NB, nblk, 2, C]; the finalize kernel merges blocks and the channels of a group in
// double precision with the parallel-variance formula and emits (mean, rstd) per (nb, group).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const unsigned short* __restrict__ x, int V, int C, int vpb,
                                                                float* __restrict__ part) {
    __shared__ float red[2][256][8];
    const int nb = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int oct_per_vox = C >> 3;                  // 16-byte pieces per voxel
    const int tid = threadIdx.x;
    const int oct = tid % oct_per_vox;               // fixed channel octet of this thread
    const int vlane = tid / oct_per_vox, vstep = 256 / oct_per_vox;
    const int v0 = blk * vpb, v1 = min(V, v0 + vpb);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const unsigned short* base = x + ((size_t)nb * V) * C + oct * 8;
    for (int v = v0 + vlane; v < v1; v += vstep) {
        const u16x8 raw = *(const u16x8*)(base + (size_t)v * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf16_bits_to_f32(raw[e]); s[e] += f; q[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = q[e]; }
    __syncthreads();
    if (tid < oct_per_vox) {                         // fixed-order tree-free reduction: deterministic
        float ts[8], tq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ts[e] = 0.f; tq[e] = 0.f; }
        for (int j = tid; j < 256; j += oct_per_vox)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ts[e] += red[0][j][e]; tq[e] += red[1][j][e]; }
        float* dst = part + (((size_t)nb * nblk + blk) * 2) * C + tid * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dst[e] = ts[e]; dst[C + e] = tq[e]; }
    }
}

__global__ void groupnorm_finalize_kernel(const float* __restrict__ part, int nblk, int C, int G, int V, int vpb, float eps,
                                          float* __restrict__ stats) {
    const int nb = blockIdx.x, g = threadIdx.x;
    if (g >= G) return;
    const int cpg = C / G;
    double n_tot = 0.0, mean = 0.0, m2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const int cnt_v = min(vpb, V - b * vpb);
        const float* ps = part + (((size_t)nb * nblk + b) * 2) * C + g * cpg;
        double s = 0.0, q = 0.0;
        for (int c = 0; c < cpg; ++c) { s += (double)ps[c]; q += (double)ps[C + c]; }
        const double n_b = (double)cnt_v * cpg;
        const double mean_b = s / n_b;
        const double m2_b = fmax(q - s * mean_b, 0.0);
        const double delta = mean_b - mean;
        const double n_new = n_tot + n_b;
        mean += delta * n_b / n_new;
        m2 += m2_b + delta * delta * n_tot * n_b / n_new;
        n_tot = n_new;
    }
    stats[((size_t)nb * G + g) * 2 + 0] = (float)mean;
    stats[((size_t)nb * G + g) * 2 + 1] = (float)(1.0 / sqrt(m2 / n_tot + (double)eps));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm apply (+ SpatialNorm3D) + SiLU -> zero-bordered output volume.
// ------------------------------------------------------------------------------------------------
struct GnApplyArgs {
    const unsigned short* x; int T, H, W, C, G;
    const float* stats; const float* gamma; const float* beta;
    unsigned short* y; int oT, oH, oW, pt, ph, pw;      // output volume dims and interior offset
    int silu;
    // SpatialNorm3D (zq == nullptr: plain GroupNorm)
    const unsigned short* zq; int zT, zH, zW, zC;       // latent volume [NB, zT, zH, zW, zC] channels-last
    const float* wy; const float* by; const float* wb; const float* bb;   // [C, zC], [C]
    int tmap[16];                                        // source latent frame of output frame t (nearest, first-frame rule)
    int rh, rw;                                          // H / zH, W / zW
};
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(GnApplyArgs p) {
    const int nb = blockIdx.y;
    const int oct_per_vox = p.C >> 3;
    const long total = (long)p.T * p.H * p.W * oct_per_vox;
    const int cpg = p.C / p.G;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int oct = idx % oct_per_vox;
        long v = idx / oct_per_vox;
        const int w = v % p.W; long r = v / p.W;
        const int h = r % p.H;
        const int t = r / p.H;
        const int c0 = oct * 8;
        const u16x8 raw = *(const u16x8*)(p.x + (((size_t)nb * p.T * p.H * p.W) + v) * p.C + c0);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / cpg;
            const float mean = p.stats[((size_t)nb * p.G + g) * 2], rstd = p.stats[((size_t)nb * p.G + g) * 2 + 1];
            o[e] = (bf16_bits_to_f32(raw[e]) - mean) * rstd * p.gamma[c0 + e] + p.beta[c0 + e];
        }
        if (p.zq != nullptr) {
            const unsigned short* z = p.zq + ((((size_t)nb * p.zT + p.tmap[t]) * p.zH + h / p.rh) * p.zW + w / p.rw) * p.zC;
            float zv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) zv[j] = j < p.zC ? bf16_bits_to_f32(z[j]) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = p.by[c0 + e], b = p.bb[c0 + e];
                const float* wyr = p.wy + (size_t)(c0 + e) * p.zC;
                const float* wbr = p.wb + (size_t)(c0 + e) * p.zC;
                for (int j = 0; j < p.zC; ++j) { y += wyr[j] * zv[j]; b += wbr[j] * zv[j]; }
                o[e] = o[e] * y + b;
            }
        }
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = silu(o[e]);
        }
        uint4 out = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        unsigned short* dst = p.y + ((((size_t)nb * p.oT + t + p.pt) * p.oH + h + p.ph) * p.oW + w + p.pw) * p.C + c0;
        *(uint4*)dst = out;
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
                                     const float* bias, const void* R, int ldr, int flags, void* stream) {
    const int K = n_taps * 64;
    const long Ml = (long)NB * oT * oH * oW;
    if (Ml <= 0 || Ml >= (1l << 31)) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: bad output volume");
    const int M = (int)Ml;
    int rc = gemm_check_common(X, W, C, R, bias, nullptr, nullptr, M, Cout, K, 8, K, ldc, ldr, R ? EPI_BIAS_GATE_RES : EPI_BIAS);
    if (rc) return rc;
    if (!tap_off || n_taps <= 0) return aether_set_error(AETHER_ERR_ARG, "conv_gemm: tap table required");
    if (iC % 64 != 0) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: input channels must be a multiple of 64");
    if (stride_hw != 1 && stride_hw != 2) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: stride must be 1 or 2");
    if ((size_t)NB * iT * iH * iW * iC >= (1ull << 32)) return aether_set_error(AETHER_ERR_SHAPE, "conv_gemm: input volume exceeds 32-bit element offsets");
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
    dim3 block(512);
#define LAUNCH_CFG(WM_, WN_, MT_, NT_, BM_, BN_)                                                                                   \
    do {                                                                                                                            \
        p.tiles_m = (M + BM_ - 1) / BM_; p.tiles_n = (Cout + BN_ - 1) / BN_;                                                         \
        dim3 grid(p.tiles_m * p.tiles_n);                                                                                          \
        if (R) { if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, true, true>), grid, block, 0, AE_STREAM, p); \
                 else hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS_GATE_RES, false, true>), grid, block, 0, AE_STREAM, p); }    \
        else { if (wide) hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, true, true>), grid, block, 0, AE_STREAM, p);            \
               else hipLaunchKernelGGL((gemm_bf16_kernel<WM_, WN_, MT_, NT_, EPI_BIAS, false, true>), grid, block, 0, AE_STREAM, p); }               \
    } while (0)
    if (Cout % 256 == 0) LAUNCH_CFG(2, 4, 4, 2, 256, 256);
    else if (Cout % 128 == 0) LAUNCH_CFG(4, 2, 4, 2, 512, 128);
    else LAUNCH_CFG(8, 1, 2, 1, 512, 32);
#undef LAUNCH_CFG
    return aether_check_launch("conv_gemm_bf16");
}

extern "C" int aether_im2col_first(const void* x, long sC, long sT, long sH, long sW, int Cin, int t0, int first_chunk, int y0,
                                   int x0, int T, int H, int W, void* A, int Kpad, void* stream) {
    if (!x || !A || Cin <= 0 || T <= 0 || H <= 0 || W <= 0) return aether_set_error(AETHER_ERR_ARG, "im2col_first: bad arguments");
    if (Kpad % 64 != 0 || Kpad < 27 * Cin) return aether_set_error(AETHER_ERR_SHAPE, "im2col_first: Kpad must be a multiple of 64 >= 27*Cin");
    if (!first_chunk && t0 < 2) return aether_set_error(AETHER_ERR_ARG, "im2col_first: later chunks need two preceding frames");
    Im2colArgs p{(const unsigned short*)x, sC, sT, sH, sW, Cin, t0, first_chunk, y0, x0, T, H, W, (unsigned short*)A, Kpad};
    hipLaunchKernelGGL(im2col_first_kernel, dim3(grid_for((long)T * H * W * (Kpad / 8))), dim3(256), 0, AE_STREAM, p);
    return aether_check_launch("im2col_first");
}

extern "C" int aether_groupnorm_stats(const void* x, int NB, int V, int C, int G, float eps, float* partial_ws, int nblk,
                                      float* stats, void* stream) {
    if (!x || !partial_ws || !stats) return aether_set_error(AETHER_ERR_ARG, "groupnorm_stats: null pointer");
    if (C % 8 != 0 || C > 2048 || 256 % (C / 8) != 0 || C % G != 0 || G > 256) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_stats: unsupported C/G");
    if (nblk <= 0) return aether_set_error(AETHER_ERR_ARG, "groupnorm_stats: nblk must be positive");
    const int vpb = (V + nblk - 1) / nblk;
    const int nblk_eff = (V + vpb - 1) / vpb;
    hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(nblk_eff, NB), dim3(256), 0, AE_STREAM, (const unsigned short*)x, V, C, vpb, partial_ws);
    int rc = aether_check_launch("groupnorm_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(NB), dim3(256), 0, AE_STREAM, partial_ws, nblk_eff, C, G, V, vpb, eps, stats);
    return aether_check_launch("groupnorm_finalize");
}

extern "C" int aether_groupnorm_apply(const void* x, int NB, int T, int H, int W, int C, int G, const float* stats,
                                      const float* gamma, const float* beta, int silu_flag, void* y, int oT, int oH, int oW,
                                      int pt, int ph, int pw, const void* zq, int zT, int zH, int zW, int zC, const float* wy,
                                      const float* by, const float* wb, const float* bb, const int* tmap_host, void* stream) {
    if (!x || !y || !stats || !gamma || !beta) return aether_set_error(AETHER_ERR_ARG, "groupnorm_apply: null pointer");
    if (C % 8 != 0 || C % G != 0) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: unsupported C/G");
    if (T + pt > oT || H + ph > oH || W + pw > oW) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: output volume too small");
    GnApplyArgs p = {};
    p.x = (const unsigned short*)x; p.T = T; p.H = H; p.W = W; p.C = C; p.G = G;
    p.stats = stats; p.gamma = gamma; p.beta = beta;
    p.y = (unsigned short*)y; p.oT = oT; p.oH = oH; p.oW = oW; p.pt = pt; p.ph = ph; p.pw = pw; p.silu = silu_flag;
    if (zq != nullptr) {
        if (!wy || !by || !wb || !bb || !tmap_host) return aether_set_error(AETHER_ERR_ARG, "groupnorm_apply: spatial-norm parameters missing");
        if (zC > 16 || T > 16 || zH <= 0 || zW <= 0 || H % zH || W % zW) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: unsupported latent volume");
        p.zq = (const unsigned short*)zq; p.zT = zT; p.zH = zH; p.zW = zW; p.zC = zC;
        p.wy = wy; p.by = by; p.wb = wb; p.bb = bb;
        for (int t = 0; t < T; ++t) p.tmap[t] = tmap_host[t];
        p.rh = H / zH; p.rw = W / zW;
    }
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(grid_for((long)T * H * W * (C / 8)), NB), dim3(256), 0, AE_STREAM, p);
    return aether_check_launch("groupnorm_apply");
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
