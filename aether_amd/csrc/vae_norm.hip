// GroupNorm / SpatialNorm3D of the CogVideoX VAE (diffusers nn.GroupNorm(32, C, eps=1e-6) and CogVideoXSpatialNorm3D,
// reached from aether/pipelines/aetherv1_pipeline_cogvideox.py:557-618 and P:931,936).  HBM-bound by design:
//   1. groupnorm_partial_kernel   one streaming read of x [NB, V, C]: per-block, per-group (sum, sum of squares) in double
//   2. groupnorm_finalize_kernel  adds the block partials in double (fixed order -> deterministic) and folds gamma/beta into a
//                                 per-channel affine table y = x*scale[c] + shift[c]
//   3. spatial_cond_kernel        SpatialNorm3D only: conv_y / conv_b (1x1x1, 16 -> C) evaluated ONCE at latent
//                                 resolution; nearest up-sampling commutes with a 1x1x1 convolution, so the full-
//                                 resolution pass only gathers 2 x 8 floats per 16-byte piece
//   4. groupnorm_apply_kernel     one streaming read of x + one write into the next convolution's zero-bordered
//                                 input volume:  silu( (x*scale+shift) [* Y(zq) + B(zq)] )
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

// Statistics merge: ONE GN_PT-thread workgroup per batch item over the `nblk` per-block per-group partial sums (double, [nblk, G, 2]) of that item.  Thread (sub, g) adds blocks sub, sub + nsub, ... of group g, a fixed-shape LDS
// tree adds the nsub subsets (deterministic: the order depends on nblk and G only), thread g turns the totals into mean / rstd (double
// E[x^2] - E[x]^2 of sums that are exact to fp32 round-off per block) and every channel gets its folded affine pair.
constexpr int GN_PT = 1024;     // threads of a partial-sum workgroup
struct GnFinalArgs { int C, G, V; float eps; const float* gamma; const float* beta; float* stats; float* affine; };
AE_DEV void groupnorm_finalize_block(const double* __restrict__ part_nb, int nblk, int nb, const GnFinalArgs& f, double (*sh)[2], float* sh_mu, float* sh_rstd) {
    const int tid = threadIdx.x, G = f.G, nsub = GN_PT / G;
    const int g = tid % G, sub = tid / G;
    double ssum = 0.0, qsum = 0.0;
    for (int b = sub; b < nblk; b += nsub) {
        const double2 v = *(const double2*)(part_nb + ((size_t)b * G + g) * 2);
        ssum += v.x; qsum += v.y;
    }
    sh[tid][0] = ssum; sh[tid][1] = qsum;
    __syncthreads();
    for (int stride = nsub >> 1; stride > 0; stride >>= 1) {
        if (sub < stride) { sh[tid][0] += sh[tid + stride * G][0]; sh[tid][1] += sh[tid + stride * G][1]; }
        __syncthreads();
    }
    const int cpg = f.C / G;
    if (tid < G) {
        const double N = (double)f.V * cpg;
        const double mu = sh[tid][0] / N;
        const double var = fmax(sh[tid][1] / N - mu * mu, 0.0);
        const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
        sh_mu[tid] = (float)mu; sh_rstd[tid] = rstd;
        f.stats[((size_t)nb * G + tid) * 2 + 0] = (float)mu;
        f.stats[((size_t)nb * G + tid) * 2 + 1] = rstd;
    }
    __syncthreads();
    for (int c = tid; c < f.C; c += GN_PT) {
        const int gg = c / cpg;
        const float sc = sh_rstd[gg] * f.gamma[c];
        f.affine[((size_t)nb * 2 + 0) * f.C + c] = sc;
        f.affine[((size_t)nb * 2 + 1) * f.C + c] = f.beta[c] - sh_mu[gg] * sc;
    }
}

// Streaming read at HBM rate needs many loads in flight: 1024 threads per workgroup, four independent 16-byte loads per thread
// and iteration (256 workgroups x 1024 x 64 B = 16 MiB outstanding on the whole chip).  Per block: per-channel fp32 (sum, sum of squares) by a
// fixed-shape LDS tree, folded to per-GROUP doubles part[nb, blk, g, (sum, sumsq)] in channel order.
// (Round 5 measured the merge inside this kernel — last block to finish, agent-scope fence + ticket counter: the fence pair costs 4-17 us per launch,
// more than the launch it saves; whole encode / decode 2-3 % slower with one or two lanes.  Deleted; profiles/EXPERIMENTS.md.)
__global__ __launch_bounds__(GN_PT) void groupnorm_partial_kernel(const unsigned short* __restrict__ x, int V, int C, int G, int vpb,
                                                                  double* __restrict__ part) {
    __shared__ float red[GN_PT][17];                 // (sum[8], sumsq[8]) per thread, padded against bank conflicts
    const int nb = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int oct_per_vox = C >> 3;                  // 16-byte pieces per voxel (divides 256)
    const int tid = threadIdx.x;
    const int oct = tid % oct_per_vox;               // fixed channel octet of this thread
    const int vlane = tid / oct_per_vox, vstep = GN_PT / oct_per_vox;
    const int v0 = blk * vpb, v1 = min(V, v0 + vpb);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const unsigned short* base = x + ((size_t)nb * V) * C + oct * 8;
    int v = v0 + vlane;
    for (; v + 3 * vstep < v1; v += 4 * vstep) {
        u16x8 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *(const u16x8*)(base + (size_t)(v + u * vstep) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = bf16_bits_to_f32(raw[u][e]); s[e] += f; q[e] += f * f; }
    }
    for (; v < v1; v += vstep) {
        const u16x8 raw = *(const u16x8*)(base + (size_t)v * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf16_bits_to_f32(raw[e]); s[e] += f; q[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = s[e]; red[tid][8 + e] = q[e]; }
    __syncthreads();
    // fixed-shape tree over the threads that share a channel octet (tid, tid + stride, ...): deterministic
    for (int stride = GN_PT / 2; stride >= oct_per_vox; stride >>= 1) {
        if (tid < stride) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[tid][e] += red[tid + stride][e];
        }
        __syncthreads();
    }
    if (tid < G) {                                    // channels of group tid, in channel order, in double
        const int cpg = C / G;
        double ds = 0.0, dq = 0.0;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { ds += (double)red[c >> 3][c & 7]; dq += (double)red[c >> 3][8 + (c & 7)]; }
        *(double2*)(part + (((size_t)nb * nblk + blk) * G + tid) * 2) = make_double2(ds, dq);
    }
}

// Second launch: the merge, one workgroup per batch item.
__global__ __launch_bounds__(GN_PT) void groupnorm_finalize_kernel(const double* __restrict__ part, int nblk, GnFinalArgs fa) {
    __shared__ double sh[GN_PT][2];
    __shared__ float sh_mu[64], sh_rstd[64];
    const int nb = blockIdx.x;
    groupnorm_finalize_block(part + (size_t)nb * nblk * fa.G * 2, nblk, nb, fa, sh, sh_mu, sh_rstd);
}

// cond[nb, zv, 0, c] = by[c] + sum_j wy[c,j] zq[nb,zv,j];   cond[nb, zv, 1, c] = bb[c] + sum_j wb[c,j] zq[nb,zv,j]
// A thread keeps the 2 x zC weights of its channel in registers and walks the block's voxels; the zC latent values of a voxel
// are wave-uniform (staged through LDS once per block), the two outputs of consecutive threads are consecutive floats.
constexpr int SC_ZC = 16;       // latent channels (CogVideoX: 16)
constexpr int SC_VOX = 16;      // voxels per block (64 left the decoder's 2 700-voxel latents on 43 workgroups per tile: latency-bound, 21 us per call)
__global__ __launch_bounds__(256) void spatial_cond_kernel(const unsigned short* __restrict__ zq, int zC, int C,
                                                           const float* __restrict__ wy, const float* __restrict__ by,
                                                           const float* __restrict__ wb, const float* __restrict__ bb,
                                                           float* __restrict__ cond, int total_vox) {
    __shared__ float zs[SC_VOX][SC_ZC];
    const int v0 = blockIdx.x * SC_VOX;
    const int nv = min(SC_VOX, total_vox - v0);
    for (int i = threadIdx.x; i < nv * SC_ZC; i += 256) {
        const int v = i / SC_ZC, j = i - v * SC_ZC;
        zs[v][j] = (j < zC) ? bf16_bits_to_f32(zq[(size_t)(v0 + v) * zC + j]) : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float wyr[SC_ZC], wbr[SC_ZC];
#pragma unroll
        for (int j = 0; j < SC_ZC; ++j) {
            wyr[j] = (j < zC) ? wy[(size_t)c * zC + j] : 0.f;
            wbr[j] = (j < zC) ? wb[(size_t)c * zC + j] : 0.f;
        }
        const float y0 = by[c], b0 = bb[c];
        for (int v = 0; v < nv; ++v) {
            float y = y0, b = b0;
#pragma unroll
            for (int j = 0; j < SC_ZC; ++j) { y += wyr[j] * zs[v][j]; b += wbr[j] * zs[v][j]; }   // same order as the plain loop
            cond[((size_t)(v0 + v) * 2 + 0) * C + c] = y;
            cond[((size_t)(v0 + v) * 2 + 1) * C + c] = b;
        }
    }
}

struct GnApplyArgs {
    const unsigned short* x; int T, H, W, C, log2_opv;
    const float* affine;                                  // [NB, 2, C]
    unsigned short* y; int oT, oH, oW, pt, ph, pw;        // output volume dims and interior offset
    int silu;
    const float* cond; int zT, zH, zW;                    // [NB, zT, zH, zW, 2, C] or null
    int tmap[16];                                         // source latent frame of output frame t
    int rh, log2_cw;                                      // H / zH, log2((W / zW) / RW): runs per latent voxel
    // causal front fused into the store (pt == 2): frames 0, 1 of y = the previous chunk's last two padded frames (front_prev) or,
    // on the first chunk, copies of the first frame; the last two frames of y are saved for the next chunk (front_next).  Both
    // caches are [NB, 2, oH, oW, C] like y's frames; only interiors are read or written (y's border is zero and stays zero).
    int causal;
    const unsigned short* front_prev;
    unsigned short* front_next;
};

// One workgroup per (t, h) row of one batch item.  A thread owns one channel octet (256 % (C/8) == 0, so the octet of item
// tid + 256k does not depend on k: its affine pair lives in registers) and walks runs of RW consecutive voxels that share one
// latent voxel: the SpatialNorm3D pair (4 x 16 B of fp32) is fetched once per run instead of once per voxel, and the RW
// 16-byte loads of a run are all in flight before the first is used.  RW = min(8, W / zW) with conditioning, else the largest
// of 8, 4, 2, 1 that divides W.  Threads per workgroup (round 5): the smallest multiple of the octets per voxel that covers the row's units in
// ceil(units / 256) equal passes — 240 for the 360-voxel, 128-channel rows (three full passes instead of 256 + 256 + 208), 144 for the 144-voxel edge
// tiles (two passes instead of 256 + 32) — instead of always 256.
template <int RW>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(GnApplyArgs p) {
    const int nb = blockIdx.y, bd = blockDim.x;      // bd: a multiple of the octets per voxel, chosen by the host so that the passes over a row are balanced
    const int t = blockIdx.x / p.H, h = blockIdx.x - t * p.H;
    const int opv = 1 << p.log2_opv;
    const int units = (p.W / RW) << p.log2_opv;
    const unsigned short* xrow = p.x + ((((size_t)nb * p.T + t) * p.H + h) * p.W) * p.C;
    unsigned short* yrow = p.y + ((((size_t)nb * p.oT + t + p.pt) * p.oH + h + p.ph) * p.oW + p.pw) * p.C;
    const size_t plane = (size_t)p.oH * p.oW * p.C;                  // elements of one padded frame
    const size_t row_in_plane = ((size_t)(h + p.ph) * p.oW + p.pw) * p.C;
    // row h of saved frame 0 for THIS batch item (saved frame k = frame T - 2 + k of the chunk; k = 1 is `plane` further)
    unsigned short* nrow = p.causal ? p.front_next + (size_t)nb * 2 * plane + row_in_plane : nullptr;
    const int c0 = (threadIdx.x & (opv - 1)) << 3;
    const float* aff = p.affine + (size_t)nb * 2 * p.C + c0;
    const f32x4 s0 = *(const f32x4*)(aff), s1 = *(const f32x4*)(aff + 4);
    const f32x4 h0 = *(const f32x4*)(aff + p.C), h1 = *(const f32x4*)(aff + p.C + 4);
    const float* crow = nullptr;
    if (p.cond != nullptr) crow = p.cond + ((((size_t)nb * p.zT + p.tmap[t]) * p.zH + h / p.rh) * p.zW) * 2 * p.C + c0;
    for (int u = threadIdx.x; u < units; u += bd) {
        const int wb = u >> p.log2_opv;
        const size_t off = (size_t)(wb * RW) * p.C + c0;
        u16x8 raw[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) raw[k] = *(const u16x8*)(xrow + off + (size_t)k * p.C);
        f32x4 y0, y1, b0, b1;
        if (crow != nullptr) {
            const float* cv = crow + (size_t)(wb >> p.log2_cw) * 2 * p.C;
            y0 = *(const f32x4*)(cv); y1 = *(const f32x4*)(cv + 4);
            b0 = *(const f32x4*)(cv + p.C); b1 = *(const f32x4*)(cv + p.C + 4);
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = bf16_bits_to_f32(raw[k][e]) * s0[e] + h0[e];
                o[e + 4] = bf16_bits_to_f32(raw[k][e + 4]) * s1[e] + h1[e];
            }
            if (crow != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = o[e] * y0[e] + b0[e]; o[e + 4] = o[e + 4] * y1[e] + b1[e]; }
            }
            if (p.silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = silu(o[e]);
            }
            const uint4 ov = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            *(uint4*)(yrow + off + (size_t)k * p.C) = ov;
            if (p.causal) {
                const size_t vo = off + (size_t)k * p.C;
                if (t == 0 && p.front_prev == nullptr) {             // first chunk: the two front frames replicate frame 0
                    *(uint4*)(yrow - 2 * plane + vo) = ov;
                    *(uint4*)(yrow - plane + vo) = ov;
                    if (p.T == 1) *(uint4*)(nrow + vo) = ov;          // one-frame chunk: the saved pair is (front frame 1, frame 0)
                }
                if (t >= p.T - 2) *(uint4*)(nrow + (size_t)(t - (p.T - 2)) * plane + vo) = ov;
            }
        }
    }
    // ---- zero border of the padded volume (round 4: the volumes are arena memory of the launch plan, not persistent zero-bordered pool
    // entries): this workgroup owns padded row h + ph of frame t + pt — and of the two causal front frames when t == 0 — i.e. that row's
    // left / right border voxels, plus the rows above (h == 0) and below (h == H - 1) the interior
    {
        const int vec_per_vox = opv;                                  // 16-byte pieces per voxel
        const size_t prow = (size_t)p.oW * p.C;                       // elements of one padded row
        const int right = p.oW - p.pw - p.W, below = p.oH - p.ph - p.H;
        const int nfr = (p.causal && t == 0) ? 3 : 1;                 // frame t (+ the two front frames)
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (int fi = 0; fi < nfr; ++fi) {
            unsigned short* row0 = yrow - (size_t)p.pw * p.C - (size_t)fi * plane;      // start of this padded row in frame t + pt - fi
            for (int u = threadIdx.x; u < (p.pw + right) * vec_per_vox; u += bd) {
                const int vx = u >> p.log2_opv, o8 = (u & (opv - 1)) << 3;
                const int col = vx < p.pw ? vx : p.pw + p.W + (vx - p.pw);
                *(uint4*)(row0 + (size_t)col * p.C + o8) = zero;
            }
            if (h == 0)
                for (int u = threadIdx.x; u < p.ph * p.oW * vec_per_vox; u += bd) *(uint4*)(row0 - (size_t)p.ph * prow + (size_t)u * 8) = zero;
            if (h == p.H - 1)
                for (int u = threadIdx.x; u < below * p.oW * vec_per_vox; u += bd) *(uint4*)(row0 + prow + (size_t)u * 8) = zero;
        }
    }
    if (p.causal && t == 0 && p.front_prev != nullptr) {             // later chunks: front frames = the saved pair, row by row
        const unsigned short* prow = p.front_prev + (size_t)nb * 2 * plane + row_in_plane;
        const int vecs = p.W << p.log2_opv;
        for (int u = threadIdx.x; u < vecs; u += bd) {
            const uint4 a = *(const uint4*)(prow + (size_t)u * 8), b = *(const uint4*)(prow + plane + (size_t)u * 8);
            *(uint4*)(yrow - 2 * plane + (size_t)u * 8) = a;
            *(uint4*)(yrow - plane + (size_t)u * 8) = b;
            if (p.T == 1) *(uint4*)(nrow + (size_t)u * 8) = b;       // one-frame chunk: saved pair = (front frame 1, frame 0)
        }
    }
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace aether

using namespace aether;
#define AE_STREAM ((hipStream_t)stream)

extern "C" int aether_groupnorm_stats(const void* x, int NB, int V, int C, int G, float eps, const float* gamma, const float* beta,
                                      float* partial_ws, int nblk, float* stats, float* affine, void* stream) {
    if (!x || !partial_ws || !stats || !affine || !gamma || !beta) return aether_set_error(AETHER_ERR_ARG, "groupnorm_stats: null pointer");
    if (C % 8 != 0 || C > 2048 || 256 % (C / 8) != 0 || C % G != 0 || G > 64 || 256 % G != 0 || C < 2 * G)
        return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_stats: unsupported C/G");
    if (nblk <= 0 || NB <= 0 || V <= 0) return aether_set_error(AETHER_ERR_ARG, "groupnorm_stats: bad sizes");
    if ((uintptr_t)partial_ws & 15) return aether_set_error(AETHER_ERR_ALIGN, "groupnorm_stats: partial_ws must be 16-byte aligned");
    const int vpb = (V + nblk - 1) / nblk;
    const int nblk_eff = (V + vpb - 1) / vpb;
    GnFinalArgs fa = {C, G, V, eps, gamma, beta, stats, affine};
    double* part = reinterpret_cast<double*>(partial_ws);           // [NB, nblk_eff, G, 2] doubles: 16 G <= 8 C bytes per block
    hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(nblk_eff, NB), dim3(GN_PT), 0, AE_STREAM, (const unsigned short*)x, V, C, G, vpb, part);
    int rc = aether_check_launch("groupnorm_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(NB), dim3(GN_PT), 0, AE_STREAM, part, nblk_eff, fa);
    return aether_check_launch("groupnorm_finalize");
}

extern "C" int aether_spatial_cond(const void* zq, int NB, int zV, int zC, int C, const float* wy, const float* by, const float* wb,
                                   const float* bb, float* cond, void* stream) {
    if (!zq || !wy || !by || !wb || !bb || !cond) return aether_set_error(AETHER_ERR_ARG, "spatial_cond: null pointer");
    if (NB <= 0 || zV <= 0 || zC <= 0 || zC > SC_ZC || C <= 0) return aether_set_error(AETHER_ERR_SHAPE, "spatial_cond: bad sizes (latent channels <= 16)");
    const int total_vox = NB * zV;
    hipLaunchKernelGGL(spatial_cond_kernel, dim3((total_vox + SC_VOX - 1) / SC_VOX), dim3(256), 0, AE_STREAM, (const unsigned short*)zq, zC, C, wy, by,
                       wb, bb, cond, total_vox);
    return aether_check_launch("spatial_cond");
}

static int groupnorm_apply_impl(const void* x, int NB, int T, int H, int W, int C, const float* affine, int silu_flag, void* y, int oT, int oH,
                                int oW, int pt, int ph, int pw, const float* cond, int zT, int zH, int zW, const int* tmap_host, int causal,
                                const void* front_prev, void* front_next, void* stream) {
    if (!x || !y || !affine) return aether_set_error(AETHER_ERR_ARG, "groupnorm_apply: null pointer");
    if (causal && (pt != 2 || oT != T + 2 || !front_next || (((uintptr_t)front_prev | (uintptr_t)front_next) & 15)))
        return aether_set_error(AETHER_ERR_ARG, "groupnorm_apply_causal: needs pt == 2, oT == T + 2 and a 16-byte aligned cache for the next chunk");
    const int l2 = ilog2_exact(C / 8);
    if (C % 8 != 0 || l2 < 0 || l2 > 8) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: C/8 must be a power of two <= 256");
    int rw = 1;
    if (T + pt > oT || H + ph > oH || W + pw > oW) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: output volume too small");
    GnApplyArgs p = {};
    p.x = (const unsigned short*)x; p.T = T; p.H = H; p.W = W; p.C = C; p.log2_opv = l2;
    p.affine = affine;
    p.y = (unsigned short*)y; p.oT = oT; p.oH = oH; p.oW = oW; p.pt = pt; p.ph = ph; p.pw = pw; p.silu = silu_flag;
    p.causal = causal; p.front_prev = (const unsigned short*)front_prev; p.front_next = (unsigned short*)front_next;
    if (cond != nullptr) {
        if (!tmap_host || T > 16 || zT <= 0 || zH <= 0 || zW <= 0 || H % zH || W % zW) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: unsupported latent volume");
        const int lrw = ilog2_exact(W / zW);
        if (lrw < 0) return aether_set_error(AETHER_ERR_SHAPE, "groupnorm_apply: W / zW must be a power of two");
        p.cond = cond; p.zT = zT; p.zH = zH; p.zW = zW; p.rh = H / zH;
        rw = lrw > 3 ? 8 : (1 << lrw);
        p.log2_cw = lrw > 3 ? lrw - 3 : 0;
        for (int t = 0; t < T; ++t) {
            if (tmap_host[t] < 0 || tmap_host[t] >= zT) return aether_set_error(AETHER_ERR_ARG, "groupnorm_apply: time map out of range");
            p.tmap[t] = tmap_host[t];
        }
    } else {
        rw = (W % 8 == 0) ? 8 : (W % 4 == 0) ? 4 : (W % 2 == 0) ? 2 : 1;
    }
    const int units = (W / rw) << l2, opv = 1 << l2;
    const int npass = (units + 255) / 256;
    int bd = ((units + npass - 1) / npass + opv - 1) / opv * opv;
    if (bd > 256 || bd < 64) bd = 256;
    const dim3 grid(T * H, NB), block(bd);
    switch (rw) {
        case 8: hipLaunchKernelGGL(groupnorm_apply_kernel<8>, grid, block, 0, AE_STREAM, p); break;
        case 4: hipLaunchKernelGGL(groupnorm_apply_kernel<4>, grid, block, 0, AE_STREAM, p); break;
        case 2: hipLaunchKernelGGL(groupnorm_apply_kernel<2>, grid, block, 0, AE_STREAM, p); break;
        default: hipLaunchKernelGGL(groupnorm_apply_kernel<1>, grid, block, 0, AE_STREAM, p); break;
    }
    return aether_check_launch("groupnorm_apply");
}

extern "C" int aether_groupnorm_apply(const void* x, int NB, int T, int H, int W, int C, const float* affine, int silu_flag, void* y,
                                      int oT, int oH, int oW, int pt, int ph, int pw, const float* cond, int zT, int zH, int zW,
                                      const int* tmap_host, void* stream) {
    return groupnorm_apply_impl(x, NB, T, H, W, C, affine, silu_flag, y, oT, oH, oW, pt, ph, pw, cond, zT, zH, zW, tmap_host, 0, nullptr, nullptr,
                                stream);
}

extern "C" int aether_groupnorm_apply_causal(const void* x, int NB, int T, int H, int W, int C, const float* affine, int silu_flag, void* y,
                                             int oH, int oW, int ph, int pw, const float* cond, int zT, int zH, int zW, const int* tmap_host,
                                             const void* front_prev, void* front_next, void* stream) {
    return groupnorm_apply_impl(x, NB, T, H, W, C, affine, silu_flag, y, T + 2, oH, oW, 2, ph, pw, cond, zT, zH, zW, tmap_host, 1, front_prev,
                                front_next, stream);
}
