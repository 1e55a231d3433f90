// Whole-VAE entry points: aether_vae_encode / aether_vae_decode = ONE C call per AutoencoderKLCogVideoX.encode /
// .decode as the reference invokes them (aether/pipelines/aetherv1_pipeline_cogvideox.py:557-618 through retrieve_latents,
// P:931,936 through decode_latents; tiling + slicing as scripts/demo.py:229-230 enables them).
//
// Host-side launch plan in C++: spatial tiles of equal shape batched (4/2/2/1 at 480x720), frame chunks with
// the causal-convolution caches threaded from chunk to chunk, every ResNet / resampler / norm as calls of the kernels declared
// in include/aether_hip.h, the tile cross-fade + crop + layout change as one kernel.  Pure enqueue on the caller's stream: no
// allocation (a bump arena over the caller's workspace), no synchronisation, no host<->device copies -> graph capturable, and
// bindable from any language.  Same arithmetic, in the same order, as the Python walk in aether_amd/vae.py (kept there for the
// per-kernel tests and as an A/B check: tests/test_vae_gpu.py::test_c_plan_matches_python_walk, bit-identical).
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <stdint.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace {

struct ConvW {
    const void* w = nullptr; const float* b = nullptr;
    int cout = 0, cout_pad = 0, cin = 0, kt = 1, kh = 1, kw = 1, kcols = 0, blocked = 0;
};
struct NormW {
    const float *gamma = nullptr, *beta = nullptr, *wy = nullptr, *by = nullptr, *wb = nullptr, *bb = nullptr;
    bool spatial() const { return wy != nullptr; }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef std::tuple<int, int, int, int, int, int> Shape5;   // (lane, NB, T, H, W, C)

}  // namespace

struct AetherVae {
    AetherVaeConfig cfg;
    std::map<std::string, ConvW> convs;
    std::map<std::string, NormW> norms;
    // ---- workspace state -------------------------------------------------------------------------------------------
    // [ pool | call arena ]: the pool holds the zero-bordered convolution input volumes, one per distinct shape, whose borders
    // are written once (zeroed when the shape is first seen in THIS workspace) and never again — producers write interiors only.
    char* ws = nullptr; size_t ws_bytes = 0;
    std::map<Shape5, size_t> pool;        // shape -> byte offset inside the pool region
    std::map<std::tuple<int, int, int, int, int, int, int>, size_t> taps;   // (lane,kt,kh,kw,iH,iW,iC) -> offset (int32 table, device generated)
    size_t pool_bytes = 0;                // bytes of the pool region in use
    size_t pool_cap = 0;                  // pool region size of the current workspace
    // ---- second lane (AETHER_VAE_TWO_LANES): the tile groups of one call are independent; half of them are enqueued on a stream of the
    // handle's own, forked from / joined to the caller's stream with events (capturable: the side stream joins the caller's capture)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
};

namespace {

using namespace aether;

// ---- small kernels of the plan ---------------------------------------------------------------------------------------
// tap offsets of an implicit-GEMM convolution in the K order the weights are packed in: (dt, dh, 64-channel block, dw)
__global__ void tap_table_kernel(int* __restrict__ out, int kt, int kh, int kw, int iH, int iW, int iC) {
    const int cb_n = iC / 64;
    const int n = kt * kh * cb_n * kw;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dw = i % kw; int r = i / kw;
    const int cb = r % cb_n; r /= cb_n;
    const int dh = r % kh; const int dt = r / kh;
    out[i] = ((dt * iH + dh) * iW + dw) * iC + cb * 64;
}

// crops of a planar latent z [C, T_all, H_all, W_all] -> channels-last [NB, T, h, w, C]  (SpatialNorm3D's zq per tile batch)
struct CropArgs { const unsigned short* z; unsigned short* out; long sC, sT, sH, sW; int C, t0, T, h, w, NB; int y0[4], x0[4]; };
__global__ void crop_channels_last_kernel(CropArgs p) {
    const long total = (long)p.NB * p.T * p.h * p.w * p.C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % p.C; long r = i / p.C;
        const int x = r % p.w; r /= p.w;
        const int y = r % p.h; r /= p.h;
        const int t = r % p.T; const int nb = r / p.T;
        p.out[i] = p.z[c * p.sC + (long)(p.t0 + t) * p.sT + (long)(p.y0[nb] + y) * p.sH + (long)(p.x0[nb] + x) * p.sW];
    }
}

// Tile assembly = diffusers' blend_v / blend_h / crop / cat of tiled_encode / tiled_decode in ONE pass.  The python loop blends
// IN PLACE and in raster order (tile (i,j) first with the already blended tile above, then with the already blended tile to its
// left), every product rounded to the tensor dtype (bf16) before the sum.  Because blend extents are at most half a tile, the
// value of an output pixel depends on at most four tiles; val() below unrolls exactly that dependency.
struct AsmArgs {
    const unsigned short* tile[16];   // tile (i,j) at index i*ncol + j: channels-last [T, th, tw, ldc]
    int th[16], tw[16];
    int nrow, ncol, T, ldc, C;        // C = channels kept
    int bh, bw, limit_h, limit_w;     // blend extents and the crop of every tile
    int H, W;                         // assembled plane
    unsigned short* out;              // [C, T, H, W]
};
AE_DEV float asm_tile(const AsmArgs& p, int i, int j, int t, int y, int x, int c) {
    const int k = i * p.ncol + j;
    return bf16_bits_to_f32(p.tile[k][(((size_t)t * p.th[k] + y) * p.tw[k] + x) * p.ldc + c]);
}
AE_DEV float asm_mix(float a, float b, int pos, int extent) {     // b*(pos/extent) + a*(1-pos/extent), products rounded to bf16
    const double f = (double)pos / (double)extent;
    const float wa = (float)(1.0 - f), wb = (float)f;
    const float pa = bf16_bits_to_f32(f32_to_bf16_bits(a * wa)), pb = bf16_bits_to_f32(f32_to_bf16_bits(b * wb));
    return bf16_bits_to_f32(f32_to_bf16_bits(pa + pb));
}
__global__ void assemble_kernel(AsmArgs p) {
    const long total = (long)p.C * p.T * p.H * p.W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int X = idx % p.W; long r = idx / p.W;
        const int Y = r % p.H; r /= p.H;
        const int t = r % p.T; const int c = r / p.T;
        const int i = min(Y / p.limit_h, p.nrow - 1), j = min(X / p.limit_w, p.ncol - 1);
        const int y = Y - i * p.limit_h, x = X - j * p.limit_w;
        const int k = i * p.ncol + j;
        float v = asm_tile(p, i, j, t, y, x, c);
        // vertical blend with the (already blended) tile above
        const int ku = (i - 1) * p.ncol + j;
        const int eh = i > 0 ? min(min(p.th[ku], p.th[k]), p.bh) : 0;
        if (y < eh) {
            const int ry = p.th[ku] - eh + y;                        // row of the upper tile; its own blends there: horizontal only
            float a = asm_tile(p, i - 1, j, t, ry, x, c);
            if (j > 0) {
                const int kl = (i - 1) * p.ncol + j - 1;
                const int ew = min(min(p.tw[kl], p.tw[ku]), p.bw);
                if (x < ew) a = asm_mix(asm_tile(p, i - 1, j - 1, t, ry, p.tw[kl] - ew + x, c), a, x, ew);
            }
            v = asm_mix(a, v, y, eh);
        }
        // horizontal blend with the (already blended) tile to the left
        if (j > 0) {
            const int kl = i * p.ncol + j - 1;
            const int ew = min(min(p.tw[kl], p.tw[k]), p.bw);
            if (x < ew) {
                const int cx = p.tw[kl] - ew + x;                    // column of the left tile; its own blends there: vertical only
                float a = asm_tile(p, i, j - 1, t, y, cx, c);
                if (i > 0) {
                    const int kul = (i - 1) * p.ncol + j - 1;
                    const int eh2 = min(min(p.th[kul], p.th[kl]), p.bh);
                    if (y < eh2) a = asm_mix(asm_tile(p, i - 1, j - 1, t, p.th[kul] - eh2 + y, cx, c), a, y, eh2);
                }
                v = asm_mix(a, v, x, ew);
            }
        }
        p.out[idx] = f32_to_bf16_bits(v);
    }
}

// frames produced by one chunk of Tc input frames (the resamplers' odd/even rules, level by level)
int chunk_out_frames(const AetherVaeConfig& c, bool decode, int Tc) {
    const int tlevel = (int)lround(log2((double)c.temporal_compression_ratio));
    int t = Tc;
    for (int i = 0; i < tlevel && i < c.num_levels - 1; ++i)
        t = decode ? (t > 1 ? ((t & 1) ? 2 * t - 1 : 2 * t) : 1) : ((t & 1) ? t / 2 + 1 : t / 2);
    return t;
}
int total_out_frames(const AetherVaeConfig& c, bool decode, int T) {
    const int bs = decode ? 2 : 8, nb = std::max(T / bs, 1), rem = T % bs;
    int tot = 0;
    for (int k = 0; k < nb; ++k) tot += chunk_out_frames(c, decode, std::min(bs * (k + 1) + rem, T) - (bs * k + (k == 0 ? 0 : rem)));
    return tot;
}

// ---- the plan ---------------------------------------------------------------------------------------------------------
struct Act { char* p = nullptr; int NB = 0, T = 0, H = 0, W = 0, C = 0; size_t bytes() const { return (size_t)NB * T * H * W * C * 2; } };

struct Plan {
    AetherVae* h;
    hipStream_t stream;
    bool dry;                 // size the arena only: no launches, no memsets
    char* arena = nullptr;    // call arena base (null when dry)
    size_t top = 0, peak = 0;
    size_t pool_need = 0;     // dry: pool bytes after this call
    std::map<Shape5, size_t> dry_pool;
    std::map<std::tuple<int, int, int, int, int, int, int>, size_t> dry_taps;
    float* splitk = nullptr; size_t splitk_bytes = 0;
    int lane = 0;                     // lane whose groups are being enqueued: selects `stream`, `splitk` and the pool entries
    int n_lanes = 1;
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    float* lane_splitk[4] = {nullptr, nullptr, nullptr, nullptr};
    void set_lane(int l) { lane = l; stream = lane_stream[l]; splitk = lane_splitk[l]; }
    int rc = 0;
    std::string err;

    char* alloc(size_t bytes) {
        const size_t off = top;
        top = align_up(top + bytes, 256);
        if (top > peak) peak = top;
        return dry ? reinterpret_cast<char*>((uintptr_t)256 + off) : arena + off;   // dry: a non-null dummy, never dereferenced
    }
    bool fail(int code, const std::string& m) { if (!rc) { rc = code; err = m; } return false; }
    bool ok(int r, const char* what) { if (r != 0 && !rc) { rc = r; err = what; } return rc == 0; }

    const ConvW* conv(const std::string& n) {
        auto it = h->convs.find(n);
        if (it == h->convs.end()) { fail(AETHER_ERR_ARG, "vae: convolution not registered: " + n); return nullptr; }
        return &it->second;
    }
    const NormW* norm(const std::string& n) {
        auto it = h->norms.find(n);
        if (it == h->norms.end()) { fail(AETHER_ERR_ARG, "vae: norm not registered: " + n); return nullptr; }
        return &it->second;
    }

    // zero-bordered convolution input volume: arena memory with the lifetime of the convolution that reads it.  Its producers write EVERY
    // voxel — interior, causal front frames and the zero border (aether_groupnorm_apply*, aether_resample_pad) — so nothing persists from call
    // to call (rounds 1-3 kept one volume per distinct shape in a pool whose borders were zeroed once: 15.7 GB of the decoder's workspace).
    char* padded(int NB, int T, int H, int W, int C) { return alloc((size_t)NB * T * H * W * C * 2); }
    const int* tap_table(int kt, int kh, int kw, int iH, int iW, int iC, int* n_taps) {
        const auto key = std::make_tuple(lane, kt, kh, kw, iH, iW, iC);
        const int n = kt * kh * kw * (iC / 64);
        *n_taps = n;
        const size_t bytes = align_up((size_t)n * 4, 256);
        if (dry) {
            if (!h->taps.count(key) && !dry_taps.count(key)) { dry_taps[key] = pool_need; pool_need += bytes; }
            return reinterpret_cast<const int*>((uintptr_t)256);
        }
        auto it = h->taps.find(key);
        if (it == h->taps.end()) {
            if (h->pool_bytes + bytes > h->pool_cap) { fail(AETHER_ERR_ARG, "vae: workspace pool region too small"); return nullptr; }
            it = h->taps.emplace(key, h->pool_bytes).first;
            h->pool_bytes += bytes;
            hipLaunchKernelGGL(tap_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (int*)(h->ws + it->second), kt, kh, kw, iH, iW, iC);
        }
        return (const int*)(h->ws + it->second);
    }

    // ---- one kernel entry each ----------------------------------------------------------------------------------------
    Act conv_gemm(const char* vol, int NB, int iT, int iH, int iW, int iC, const ConvW& cw, int oT, int oH, int oW, int stride, const char* residual,
                  char* dst = nullptr) {
        Act out; out.NB = NB; out.T = oT; out.H = oH; out.W = oW; out.C = cw.cout_pad;
        out.p = dst ? dst : alloc(out.bytes());
        int n_taps = 0;
        const int* taps = tap_table(cw.kt, cw.kh, cw.kw, iH, iW, iC, &n_taps);
        if (dry || rc) return out;
        int flags = h->cfg.flags;
        if (cw.blocked && stride == 1 && cw.kh == 3 && cw.kw == 3 && iH == oH + 2 && iW == oW + 2 &&
            (double)iH * iW <= (double)h->cfg.tap_reuse_max_waste * oH * oW)
            flags |= AETHER_CONV_TAP_REUSE;
        ok(aether_conv_gemm_bf16(vol, NB, iT, iH, iW, iC, oT, oH, oW, stride, taps, n_taps, cw.w, cw.cout_pad, out.p, cw.cout_pad, cw.b,
                                 residual, residual ? cw.cout_pad : 0, splitk, splitk_bytes, flags, stream), "aether_conv_gemm_bf16");
        return out;
    }
    Act linear(const char* x, size_t rows, int K, const ConvW& cw, int NB, int T, int H, int W) {
        Act out; out.NB = NB; out.T = T; out.H = H; out.W = W; out.C = cw.cout_pad;
        out.p = alloc(out.bytes());
        if (dry || rc) return out;
        ok(aether_gemm_bf16(x, K, cw.w, cw.kcols, out.p, cw.cout_pad, (int)rows, cw.cout_pad, cw.kcols, cw.b, AETHER_EPI_BIAS, nullptr, 0, nullptr,
                            nullptr, 0, 0, 0, nullptr, 0, h->cfg.flags, stream), "aether_gemm_bf16");
        return out;
    }
    // GroupNorm (+SpatialNorm3D) + SiLU of x into a zero-bordered volume [NB, T+pad_t, H+2p, W+2p, C]
    // front_next != nullptr: the causal front (frames 0, 1 from front_prev or copies of frame 0; last two frames saved to front_next) is
    // written by the same launch (aether_groupnorm_apply_causal)
    char* norm_to_padded(const Act& x, const NormW& nw, int pad_t, int pad_hw, const Act* zq, float eps, const char* front_prev = nullptr,
                         char* front_next = nullptr) {
        const int G = h->cfg.norm_num_groups, V = x.T * x.H * x.W;
        const int nblk = std::max(1, std::min(256, (V + 127) / 128));
        float* part = (float*)alloc((size_t)x.NB * nblk * 2 * x.C * 4);
        float* stats = (float*)alloc((size_t)x.NB * G * 2 * 4);
        float* affine = (float*)alloc((size_t)x.NB * 2 * x.C * 4);
        float* cond = nullptr;
        if (nw.spatial()) cond = (float*)alloc((size_t)zq->NB * zq->T * zq->H * zq->W * 2 * x.C * 4);
        char* vol = padded(x.NB, x.T + pad_t, x.H + 2 * pad_hw, x.W + 2 * pad_hw, x.C);
        if (dry || rc) return vol;
        if (!ok(aether_groupnorm_stats(x.p, x.NB, V, x.C, G, nw.spatial() ? 1e-6f : eps, nw.gamma, nw.beta, part, nblk, stats, affine, stream),
                "aether_groupnorm_stats")) return vol;
        if (nw.spatial()) {
            if (!ok(aether_spatial_cond(zq->p, zq->NB, zq->T * zq->H * zq->W, zq->C, x.C, nw.wy, nw.by, nw.wb, nw.bb, cond, stream), "aether_spatial_cond"))
                return vol;
            int tmap[64];
            if (x.T > 64) { fail(AETHER_ERR_SHAPE, "vae: more than 64 frames in one chunk"); return vol; }
            nearest_time_map(x.T, zq->T, tmap);
            if (front_next)
                ok(aether_groupnorm_apply_causal(x.p, x.NB, x.T, x.H, x.W, x.C, affine, 1, vol, x.H + 2 * pad_hw, x.W + 2 * pad_hw, pad_hw, pad_hw, cond,
                                                 zq->T, zq->H, zq->W, tmap, front_prev, front_next, stream), "aether_groupnorm_apply_causal");
            else
                ok(aether_groupnorm_apply(x.p, x.NB, x.T, x.H, x.W, x.C, affine, 1, vol, x.T + pad_t, x.H + 2 * pad_hw, x.W + 2 * pad_hw, pad_t, pad_hw,
                                          pad_hw, cond, zq->T, zq->H, zq->W, tmap, stream), "aether_groupnorm_apply");
        } else if (front_next) {
            ok(aether_groupnorm_apply_causal(x.p, x.NB, x.T, x.H, x.W, x.C, affine, 1, vol, x.H + 2 * pad_hw, x.W + 2 * pad_hw, pad_hw, pad_hw, nullptr, 0,
                                             0, 0, nullptr, front_prev, front_next, stream), "aether_groupnorm_apply_causal");
        } else {
            ok(aether_groupnorm_apply(x.p, x.NB, x.T, x.H, x.W, x.C, affine, 1, vol, x.T + pad_t, x.H + 2 * pad_hw, x.W + 2 * pad_hw, pad_t, pad_hw,
                                      pad_hw, nullptr, 0, 0, 0, nullptr, stream), "aether_groupnorm_apply");
        }
        return vol;
    }
    // CogVideoXSpatialNorm3D's F.interpolate(mode="nearest") along time: T > 1 and odd -> first frame maps to latent frame 0
    static void nearest_time_map(int T, int zT, int* out) {
        auto nearest = [](int n_out, int n_in, int* o, int base) {
            const double scale = (double)n_in / (double)n_out;
            for (int i = 0; i < n_out; ++i) o[i] = base + std::min((int)floor(i * scale), n_in - 1);
        };
        if (T > 1 && (T & 1)) {
            out[0] = 0;
            if (zT > 1) nearest(T - 1, zT - 1, out + 1, 1);
            else for (int i = 1; i < T; ++i) out[i] = 0;
        } else {
            nearest(T, zT, out, 0);
        }
    }

    // conv caches of one tile group: key -> two buffers used alternately (prev of chunk k = next of chunk k-1)
    struct Cache { char* buf[2] = {nullptr, nullptr}; int filled = 0; };
    std::map<std::string, Cache>* caches = nullptr;
    size_t cache_bytes_hint = 0;

    // the two cache buffers of a conv for this chunk: prev (null on the first chunk) and next; advances the cache
    void causal_buffers(const std::string& key, const char** prev, char** next) {
        Cache& c = (*caches)[key];
        if (c.buf[0] == nullptr) { fail(AETHER_ERR_ARG, "vae: internal: cache not reserved for " + key); *prev = nullptr; *next = nullptr; return; }
        *prev = c.filled ? c.buf[(c.filled - 1) & 1] : nullptr;
        *next = c.buf[c.filled & 1];
        c.filled++;
    }
    // caches are reserved (arena, group lifetime) by a dry walk of the first chunk: reserve_mode records the keys and sizes
    bool reserve_mode = false;
    std::vector<std::pair<std::string, size_t>> reserve_list;

    Act causal_conv(const Act& x, const NormW& nw, const ConvW& cw, const std::string& key, const Act* zq, const char* residual, float eps,
                    char* dst = nullptr) {
        // GroupNorm + SiLU into the padded volume; the same launch writes the causal front (previous chunk's last two frames, or
        // copies of the first frame) and saves this chunk's last two frames for the next one
        const char* prev = nullptr;
        char* next = nullptr;
        if (reserve_mode) reserve_list.emplace_back(key, (size_t)x.NB * 2 * (x.H + 2) * (x.W + 2) * x.C * 2);
        else causal_buffers(key, &prev, &next);
        if (dst == nullptr) dst = alloc((size_t)x.NB * x.T * x.H * x.W * cw.cout_pad * 2);    // the output first: volume + norm scratch sit above it ...
        const size_t mark = top;
        char* vol = norm_to_padded(x, nw, 2, 1, zq, eps, prev, next);
        Act y = conv_gemm(vol, x.NB, x.T + 2, x.H + 2, x.W + 2, x.C, cw, x.T, x.H, x.W, 1, residual, dst);
        top = mark;                                                                          // ... and are released once the convolution is enqueued
        return y;
    }
    Act resnet(const Act& x, const std::string& prefix, const Act* zq) {
        const NormW *n1 = norm(prefix + "norm1"), *n2 = norm(prefix + "norm2");
        const ConvW *c1 = conv(prefix + "conv1"), *c2 = conv(prefix + "conv2");
        if (rc) return x;
        const float eps = h->cfg.norm_eps;
        // the block's output is allocated first; everything else of the block (conv1 output, shortcut, norm scratch) is released
        // when it returns
        char* dst = alloc((size_t)x.NB * x.T * x.H * x.W * c2->cout_pad * 2);
        const size_t mark = top;
        Act hh = causal_conv(x, *n1, *c1, prefix + "conv1", zq, nullptr, eps);
        const char* skip = x.p;
        auto sc = h->convs.find(prefix + "conv_shortcut");
        if (sc != h->convs.end()) skip = linear(x.p, (size_t)x.NB * x.T * x.H * x.W, x.C, sc->second, x.NB, x.T, x.H, x.W).p;
        Act y = causal_conv(hh, *n2, *c2, prefix + "conv2", zq, skip, eps, dst);
        top = mark;
        return y;
    }
    char* resample(const Act& x, int mode, int oT, int oH, int oW, int pt, int ph, int pw) {
        char* vol = padded(x.NB, oT, oH, oW, x.C);
        if (dry || rc) return vol;
        ok(aether_resample_pad(x.p, x.NB, x.T, x.H, x.W, x.C, mode, vol, oT, oH, oW, pt, ph, pw, stream), "aether_resample_pad");
        return vol;
    }
    // explicit im2col of the thin first convolution for NB crops -> [NB, T*H*W, Kpad]
    char* im2col(const void* src, long sC, long sT, long sH, long sW, int Cin, const ConvW& cw, const int (*crops)[2], int NB, int t0, int T, int H, int W,
                 bool first) {
        const size_t per = (size_t)T * H * W * cw.kcols * 2;
        char* A = alloc(per * NB);
        if (aether_im2col_first_lds_bytes(Cin, W, cw.kcols) > AETHER_IM2COL_LDS_LIMIT)      // checked in the dry walk too: aether_vae_workspace_bytes reports it
            fail(AETHER_ERR_SHAPE, "vae: first layer: 9 * Cin * (W + 2) source elements exceed 160 KiB of LDS (enable tiling for inputs this wide)");
        if (dry || rc) return A;
        for (int i = 0; i < NB; ++i)
            if (!ok(aether_im2col_first(src, sC, sT, sH, sW, Cin, t0, first ? 1 : 0, crops[i][0], crops[i][1], T, H, W, A + per * i, cw.kcols, stream),
                    "aether_im2col_first")) break;
        return A;
    }

    // ---- encoder / decoder over one frame chunk of NB equally shaped tiles ------------------------------------------------------
    Act encode_chunk(const void* video, long sC, long sT, long sH, long sW, const int (*crops)[2], int NB, int t0, int T, int H, int W, bool first) {
        const AetherVaeConfig& c = h->cfg;
        const ConvW* cin = conv("encoder.conv_in");
        if (rc) return Act();
        char* A = im2col(video, sC, sT, sH, sW, c.in_channels, *cin, crops, NB, t0, T, H, W, first);
        Act x = linear(A, (size_t)NB * T * H * W, cin->kcols, *cin, NB, T, H, W);
        const int tlevel = (int)lround(log2((double)c.temporal_compression_ratio));
        for (int i = 0; i < c.num_levels && !rc; ++i) {
            for (int j = 0; j < c.layers_per_block && !rc; ++j)
                x = resnet(x, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", nullptr);
            if (i != c.num_levels - 1 && !rc) {
                const ConvW* ds = conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0");
                if (rc) break;
                const bool ct = i < tlevel;
                const int Tn = ct ? ((x.T & 1) ? x.T / 2 + 1 : x.T / 2) : x.T;
                char* out = alloc((size_t)x.NB * Tn * (x.H / 2) * (x.W / 2) * ds->cout_pad * 2);
                const size_t mark = top;
                char* vol = resample(x, ct ? 1 : 0, Tn, x.H + 1, x.W + 1, 0, 0, 0);
                x = conv_gemm(vol, x.NB, Tn, x.H + 1, x.W + 1, x.C, *ds, Tn, x.H / 2, x.W / 2, 2, nullptr, out);
                top = mark;
            }
        }
        for (int j = 0; j < 2 && !rc; ++j) x = resnet(x, "encoder.mid_block.resnets." + std::to_string(j) + ".", nullptr);
        if (rc) return x;
        const NormW* no = norm("encoder.norm_out");
        const ConvW* co = conv("encoder.conv_out");
        if (rc) return x;
        return causal_conv(x, *no, *co, "encoder.conv_out", nullptr, nullptr, 1e-6f);
    }
    Act decode_chunk(const void* z, long sC, long sT, long sH, long sW, const int (*crops)[2], int NB, int t0, int T, int H, int W, bool first) {
        const AetherVaeConfig& c = h->cfg;
        const ConvW* cin = conv("decoder.conv_in");
        if (rc) return Act();
        Act zq; zq.NB = NB; zq.T = T; zq.H = H; zq.W = W; zq.C = c.latent_channels;
        zq.p = alloc(zq.bytes());
        if (!dry) {
            CropArgs a; a.z = (const unsigned short*)z; a.out = (unsigned short*)zq.p; a.sC = sC; a.sT = sT; a.sH = sH; a.sW = sW;
            a.C = c.latent_channels; a.t0 = t0; a.T = T; a.h = H; a.w = W; a.NB = NB;
            for (int i = 0; i < NB; ++i) { a.y0[i] = crops[i][0]; a.x0[i] = crops[i][1]; }
            const long total = (long)NB * T * H * W * c.latent_channels;
            hipLaunchKernelGGL(crop_channels_last_kernel, dim3((unsigned)std::min<long>(4096, (total + 255) / 256)), dim3(256), 0, stream, a);
        }
        char* A = im2col(z, sC, sT, sH, sW, c.latent_channels, *cin, crops, NB, t0, T, H, W, first);
        Act x = linear(A, (size_t)NB * T * H * W, cin->kcols, *cin, NB, T, H, W);
        for (int j = 0; j < 2 && !rc; ++j) x = resnet(x, "decoder.mid_block.resnets." + std::to_string(j) + ".", &zq);
        const int tlevel = (int)lround(log2((double)c.temporal_compression_ratio));
        for (int i = 0; i < c.num_levels && !rc; ++i) {
            for (int j = 0; j < c.layers_per_block + 1 && !rc; ++j)
                x = resnet(x, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", &zq);
            if (i != c.num_levels - 1 && !rc) {
                const ConvW* us = conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0");
                if (rc) break;
                int Tn = x.T, mode = 2;
                if (i < tlevel) { Tn = x.T > 1 ? ((x.T & 1) ? 2 * x.T - 1 : 2 * x.T) : 1; mode = 3; }
                char* out = alloc((size_t)x.NB * Tn * (2 * x.H) * (2 * x.W) * us->cout_pad * 2);
                const size_t mark = top;
                char* vol = resample(x, mode, Tn, 2 * x.H + 2, 2 * x.W + 2, 0, 1, 1);
                x = conv_gemm(vol, x.NB, Tn, 2 * x.H + 2, 2 * x.W + 2, x.C, *us, Tn, 2 * x.H, 2 * x.W, 1, nullptr, out);
                top = mark;
            }
        }
        if (rc) return x;
        const NormW* no = norm("decoder.norm_out");
        const ConvW* co = conv("decoder.conv_out");
        if (rc) return x;
        return causal_conv(x, *no, *co, "decoder.conv_out", &zq, nullptr, 1e-6f);
    }

    // ---- tiles, chunks, assembly --------------------------------------------------------------------------------------------
    static std::vector<std::pair<int, int>> chunks(int n, int bs) {          // diffusers' frame batching: the remainder joins chunk 0
        const int nb = std::max(n / bs, 1), rem = n % bs;
        std::vector<std::pair<int, int>> r;
        for (int k = 0; k < nb; ++k) r.emplace_back(bs * k + (k == 0 ? 0 : rem), std::min(bs * (k + 1) + rem, n));
        return r;
    }

    // whole encode (decode = false) or decode: src planar [C, T, H, W] (contiguous), out planar [Cout, T_out, H_out, W_out]
    bool run(bool decode, const void* src, int T, int H, int W, int tiling, void* out, int* out_T, int* out_H, int* out_W) {
        const AetherVaeConfig& c = h->cfg;
        const int down = 1 << (c.num_levels - 1);
        const int ts_h = c.sample_height / 2, ts_w = c.sample_width / 2;                 // tile_sample_min_*
        const int tl_h = ts_h / down, tl_w = ts_w / down;                               // tile_latent_min_*
        const int tile_h = decode ? tl_h : ts_h, tile_w = decode ? tl_w : ts_w;
        const bool tiled = tiling && (W > tile_w || H > tile_h);
        const int bs = decode ? 2 : 8;                                                   // num_latent_frames_batch_size / num_sample_frames_batch_size
        std::vector<int> rows_y{0}, cols_x{0};
        int th0 = H, tw0 = W, bh = 0, bw = 0, limit_h = 1 << 30, limit_w = 1 << 30;
        if (tiled) {
            const int sh = (int)(tile_h * (1 - 1.0 / 6)), sw = (int)(tile_w * (1 - 1.0 / 5));       // int(tile * (1 - overlap))
            rows_y.clear(); cols_x.clear();
            for (int y = 0; y < H; y += sh) rows_y.push_back(y);
            for (int x = 0; x < W; x += sw) cols_x.push_back(x);
            th0 = tile_h; tw0 = tile_w;
            const int oth = decode ? ts_h : tl_h, otw = decode ? ts_w : tl_w;               // tile size on the OUTPUT side
            bh = (int)(oth * (1.0 / 6)); bw = (int)(otw * (1.0 / 5));
            limit_h = oth - bh; limit_w = otw - bw;
        }
        const int nrow = (int)rows_y.size(), ncol = (int)cols_x.size();
        if (nrow * ncol > 16) return fail(AETHER_ERR_SHAPE, "vae: more than 16 tiles");
        const long sC = (long)T * H * W, sT = (long)H * W, sH = W, sW = 1;
        const auto ch = chunks(T, bs);
        const int To = total_out_frames(c, decode, T);
        const int oC = decode ? c.out_channels : 2 * c.latent_channels;
        // groups of equally shaped tiles, up to 4 per launch batch (every tile's arithmetic is independent of its batch: GroupNorm statistics
        // are per batch item, a convolution row depends on its own voxels only)
        const int NL = (nrow * ncol > 1) ? n_lanes : 1;
        const bool two_lanes = NL > 1;
        const size_t gmax = 4;     // re-measured in round 6 with the staged epilogue: 4 -> 0.1815 / 0.3667 s, 3 (lanes balanced 53 / 47) -> 0.195 / 0.388, 2 -> 0.187 / 0.378
        struct Group { int th, tw; std::vector<int> idx; int lane = 0; };
        std::vector<Group> groups;
        for (int i = 0; i < nrow; ++i)
            for (int j = 0; j < ncol; ++j) {
                const int th = std::min(th0, H - rows_y[i]), tw = std::min(tw0, W - cols_x[j]);
                bool found = false;
                for (auto& g : groups) if (g.th == th && g.tw == tw && g.idx.size() < gmax) { g.idx.push_back(i * ncol + j); found = true; break; }
                if (!found) groups.push_back(Group{th, tw, {i * ncol + j}, 0});
            }
        AsmArgs asmargs{};
        int ldc = 0;
        // full per-tile outputs (call lifetime)
        std::vector<char*> full(nrow * ncol, nullptr);
        std::vector<int> oth(nrow * ncol), otw(nrow * ncol);
        const ConvW* last = conv(decode ? "decoder.conv_out" : "encoder.conv_out");
        if (rc) return false;
        ldc = last->cout_pad;
        for (int k = 0; k < nrow * ncol; ++k) {
            const int i = k / ncol, j = k % ncol;
            const int th = std::min(th0, H - rows_y[i]), tw = std::min(tw0, W - cols_x[j]);
            oth[k] = decode ? th * down : th / down; otw[k] = decode ? tw * down : tw / down;
            full[k] = alloc((size_t)To * oth[k] * otw[k] * ldc * 2);
        }
        // lanes: groups in order of decreasing area, each to the lane with less work so far.  480x720: lane 0 (the caller's stream) takes the
        // batch of the four full tiles (71 % of the pixels), lane 1 the three batches of edge tiles (2 x 240x144, 2 x 80x360, 80x144).  Measured
        // (profiles/r04_vae_lanes_sweep.txt): batches of four on two lanes beat balanced batches of two (encode 0.188 vs 0.197 s, decode 0.377
        // vs 0.387 s: the big batches keep the deep, low-resolution layers out of split-K and halve the launches), and a third or fourth lane
        // loses again (0.203 / 0.398, 0.193 / 0.385)
        if (two_lanes) {
            std::vector<int> order(groups.size());
            for (size_t i = 0; i < groups.size(); ++i) order[i] = (int)i;
            auto area = [&](int i) { return (long)groups[i].th * groups[i].tw * (long)groups[i].idx.size(); };
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return area(a) > area(b); });
            long load[4] = {0, 0, 0, 0};
            for (int i : order) {
                int l = 0;
                for (int k = 1; k < NL; ++k) if (load[k] < load[l]) l = k;
                groups[i].lane = l; load[l] += area(i);
            }
            if (!dry) {
                if (hipEventRecord(h->ev_fork, lane_stream[0]) != hipSuccess) return fail(AETHER_ERR_LAUNCH, "vae: fork of the lanes failed");
                for (int k = 1; k < NL; ++k)
                    if (hipStreamWaitEvent(lane_stream[k], h->ev_fork, 0) != hipSuccess) return fail(AETHER_ERR_LAUNCH, "vae: fork of the lanes failed");
            }
        }
        size_t lane_base = top;
        for (int pass = 0; pass < NL; ++pass) {
          if (pass >= 1) lane_base = align_up(peak, 256);            // lane k's arena starts where lane k-1's ends
          set_lane(pass);
          const size_t mark_call = lane_base;
          for (auto& g : groups) {
            if (g.lane != pass) continue;
            top = mark_call;
            const int NB = (int)g.idx.size();
            int crops[4][2];
            for (int n = 0; n < NB; ++n) { crops[n][0] = rows_y[g.idx[n] / ncol]; crops[n][1] = cols_x[g.idx[n] % ncol]; }
            std::map<std::string, Cache> group_caches;
            caches = &group_caches;
            // reserve the conv caches of this group (sizes from a dry walk of the first chunk's graph)
            {
                const bool was_dry = dry; const size_t t0 = top, p0 = peak;
                dry = true; reserve_mode = true; reserve_list.clear();
                if (decode) decode_chunk(src, sC, sT, sH, sW, crops, NB, ch[0].first, ch[0].second - ch[0].first, g.th, g.tw, true);
                else encode_chunk(src, sC, sT, sH, sW, crops, NB, ch[0].first, ch[0].second - ch[0].first, g.th, g.tw, true);
                dry = was_dry; reserve_mode = false; top = t0; peak = std::max(p0, peak);
                if (rc) return false;
                for (auto& kv : reserve_list) {
                    Cache& cc = group_caches[kv.first];
                    cc.buf[0] = alloc(kv.second);
                    cc.buf[1] = ch.size() > 1 ? alloc(kv.second) : cc.buf[0];
                }
            }
            const size_t mark_group = top;
            int t_out = 0;
            for (size_t k = 0; k < ch.size(); ++k) {
                top = mark_group;
                const int t0 = ch[k].first, Tc = ch[k].second - ch[k].first;
                Act y = decode ? decode_chunk(src, sC, sT, sH, sW, crops, NB, t0, Tc, g.th, g.tw, k == 0)
                               : encode_chunk(src, sC, sT, sH, sW, crops, NB, t0, Tc, g.th, g.tw, k == 0);
                if (rc) return false;
                // chunk output [NB, To_c, oh, ow, ldc] -> per-tile full buffers at frame t_out
                const size_t per_tile = (size_t)y.T * y.H * y.W * y.C * 2;
                if (!dry)
                    for (int n = 0; n < NB; ++n) {
                        const int kk = g.idx[n];
                        if (y.H != oth[kk] || y.W != otw[kk] || y.C != ldc) return fail(AETHER_ERR_SHAPE, "vae: internal: chunk output shape");
                        if (hipMemcpyAsync(full[kk] + (size_t)t_out * oth[kk] * otw[kk] * ldc * 2, y.p + per_tile * n, per_tile, hipMemcpyDeviceToDevice,
                                           stream) != hipSuccess)
                            return fail(AETHER_ERR_LAUNCH, "vae: device copy failed");
                    }
                t_out += y.T;
            }
            if (t_out != To) return fail(AETHER_ERR_SHAPE, "vae: internal: frame count");
          }
        }
        set_lane(0);
        if (two_lanes && !dry) {
            for (int k = 1; k < NL; ++k)
                if (hipEventRecord(h->ev_join[k - 1], lane_stream[k]) != hipSuccess || hipStreamWaitEvent(lane_stream[0], h->ev_join[k - 1], 0) != hipSuccess)
                    return fail(AETHER_ERR_LAUNCH, "vae: join of the lanes failed");
        }
        caches = nullptr;
        // assemble
        int OH = 0, OW = 0;
        for (int i = 0; i < nrow; ++i) OH += std::min(oth[i * ncol], limit_h);
        for (int j = 0; j < ncol; ++j) OW += std::min(otw[j], limit_w);
        *out_T = To; *out_H = OH; *out_W = OW;
        if (dry) return true;
        for (int k = 0; k < nrow * ncol; ++k) { asmargs.tile[k] = (const unsigned short*)full[k]; asmargs.th[k] = oth[k]; asmargs.tw[k] = otw[k]; }
        asmargs.nrow = nrow; asmargs.ncol = ncol; asmargs.T = To; asmargs.ldc = ldc; asmargs.C = oC;
        asmargs.bh = bh; asmargs.bw = bw; asmargs.limit_h = tiled ? limit_h : OH; asmargs.limit_w = tiled ? limit_w : OW;
        asmargs.H = OH; asmargs.W = OW; asmargs.out = (unsigned short*)out;
        const long total = (long)oC * To * OH * OW;
        hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)std::min<long>(16384, (total + 255) / 256)), dim3(256), 0, stream, asmargs);
        return ok(aether_check_launch("vae assemble"), "vae assemble");
    }
};

constexpr size_t kVaeSplitKBytes = (size_t)96 << 20;

int vae_lanes(const AetherVae* h) { return (h->cfg.flags & AETHER_VAE_TWO_LANES) ? 2 : 1; }

int vae_run(AetherVae* h, bool decode, const void* src, int T, int H, int W, int tiling, void* out, void* workspace, size_t workspace_bytes,
            void* stream, size_t* need_out) {
    const int NL = vae_lanes(h);
    Plan dry; dry.h = h; dry.stream = nullptr; dry.dry = true; dry.n_lanes = NL;
    int oT, oH, oW;
    dry.alloc(NL * kVaeSplitKBytes);
    if (!dry.run(decode, src, T, H, W, tiling, nullptr, &oT, &oH, &oW)) return aether_set_error(dry.rc, dry.err.c_str());
    const size_t pool_total = h->pool_bytes + dry.pool_need;
    const size_t need = align_up(pool_total, 256) + dry.peak;
    if (need_out) { *need_out = need; return AETHER_OK; }
    if (!workspace || ((uintptr_t)workspace & 255)) return aether_set_error(AETHER_ERR_ALIGN, "vae: workspace must be non-null and 256-byte aligned");
    if (workspace_bytes < need) return aether_set_error(AETHER_ERR_ARG, "vae: workspace too small (aether_vae_workspace_bytes)");
    if (h->ws != (char*)workspace || h->ws_bytes != workspace_bytes) {        // a new workspace: the tap tables have to be generated in it again
        h->ws = (char*)workspace; h->ws_bytes = workspace_bytes;
        h->pool.clear(); h->taps.clear(); h->pool_bytes = 0;
        Plan d2; d2.h = h; d2.dry = true; d2.n_lanes = NL; d2.alloc(NL * kVaeSplitKBytes);
        if (!d2.run(decode, src, T, H, W, tiling, nullptr, &oT, &oH, &oW)) return aether_set_error(d2.rc, d2.err.c_str());
        if (align_up(d2.pool_need, 256) + d2.peak > workspace_bytes) return aether_set_error(AETHER_ERR_ARG, "vae: workspace too small");
    }
    // the table region may grow up to where this call's arena begins; the arena sits at the END of the workspace
    Plan p; p.h = h; p.stream = (hipStream_t)stream; p.dry = false; p.n_lanes = NL;
    p.arena = (char*)workspace + (workspace_bytes - align_up(dry.peak, 256)) / 256 * 256;
    h->pool_cap = (size_t)(p.arena - (char*)workspace);
    p.lane_stream[0] = (hipStream_t)stream;
    for (int k = 1; k < NL; ++k) {
        if (h->side[k - 1] == nullptr) {
            // HIGH-PRIORITY streams: priority streams get hardware queues of their own, a normal stream may share the caller's queue (4 queues,
            // round robin) and then nothing overlaps (profiles/r03_decode_pair.json)
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&h->side[k - 1], hipStreamNonBlocking, hi) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join[k - 1], hipEventDisableTiming) != hipSuccess ||
                (h->ev_fork == nullptr && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess))
                return aether_set_error(AETHER_ERR_LAUNCH, "vae: could not create a lane's stream / events");
        }
        p.lane_stream[k] = h->side[k - 1];
    }
    for (int k = 0; k < NL; ++k) p.lane_splitk[k] = (float*)p.alloc(kVaeSplitKBytes);
    p.splitk_bytes = kVaeSplitKBytes;
    p.set_lane(0);
    if (!p.run(decode, src, T, H, W, tiling, out, &oT, &oH, &oW)) return aether_set_error(p.rc, p.err.c_str());
    return AETHER_OK;
}

}  // namespace

extern "C" AetherVae* aether_vae_create(const AetherVaeConfig* cfg) {
    if (!cfg) { aether_set_error(AETHER_ERR_ARG, "vae_create: null config"); return nullptr; }
    if (cfg->flags & ~(AETHER_GEMM_WIDE_STORE | AETHER_VAE_TWO_LANES)) {      // AETHER_CONV_TAP_REUSE is chosen per convolution by the plan (tap_reuse_max_waste), not by the caller
        aether_set_error(AETHER_ERR_ARG, "vae_create: undefined flag bits (defined: AETHER_GEMM_WIDE_STORE, AETHER_VAE_TWO_LANES)");
        return nullptr;
    }
    if (cfg->num_levels < 2 || cfg->num_levels > 6 || cfg->latent_channels > 16 || cfg->layers_per_block < 1) {
        aether_set_error(AETHER_ERR_SHAPE, "vae_create: unsupported configuration");
        return nullptr;
    }
    AetherVae* h = new AetherVae();
    h->cfg = *cfg;
    return h;
}

extern "C" void aether_vae_destroy(AetherVae* h) {
    if (!h) return;
    for (int k = 0; k < 3; ++k) {
        if (h->side[k]) hipStreamDestroy(h->side[k]);
        if (h->ev_join[k]) hipEventDestroy(h->ev_join[k]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    delete h;
}

extern "C" int aether_vae_set_conv(AetherVae* h, const char* name, const void* w, const float* b, int cout, int cout_pad, int cin, int kt, int kh,
                                   int kw, int kcols, int blocked) {
    if (!h || !name || !w || !b) return aether_set_error(AETHER_ERR_ARG, "vae_set_conv: null argument");
    if (((uintptr_t)w | (uintptr_t)b) & 15) return aether_set_error(AETHER_ERR_ALIGN, "vae_set_conv: pointers must be 16-byte aligned");
    ConvW c; c.w = w; c.b = b; c.cout = cout; c.cout_pad = cout_pad; c.cin = cin; c.kt = kt; c.kh = kh; c.kw = kw; c.kcols = kcols; c.blocked = blocked;
    h->convs[name] = c;
    return AETHER_OK;
}

extern "C" int aether_vae_set_norm(AetherVae* h, const char* name, const float* gamma, const float* beta, const float* wy, const float* by,
                                   const float* wb, const float* bb) {
    if (!h || !name || !gamma || !beta) return aether_set_error(AETHER_ERR_ARG, "vae_set_norm: null argument");
    if ((wy != nullptr) != (by != nullptr) || (wy != nullptr) != (wb != nullptr) || (wy != nullptr) != (bb != nullptr))
        return aether_set_error(AETHER_ERR_ARG, "vae_set_norm: give all four SpatialNorm3D tensors or none");
    NormW n; n.gamma = gamma; n.beta = beta; n.wy = wy; n.by = by; n.wb = wb; n.bb = bb;
    h->norms[name] = n;
    return AETHER_OK;
}

extern "C" size_t aether_vae_workspace_bytes(AetherVae* h, int decode, int T, int H, int W, int tiling) {
    if (!h || T <= 0 || H <= 0 || W <= 0) return 0;
    size_t need = 0;
    if (vae_run(h, decode != 0, nullptr, T, H, W, tiling, nullptr, nullptr, 0, nullptr, &need) != AETHER_OK) return 0;
    return need;
}

extern "C" int aether_vae_output_shape(AetherVae* h, int decode, int T, int H, int W, int* oC, int* oT, int* oH, int* oW) {
    if (!h || !oC || !oT || !oH || !oW) return aether_set_error(AETHER_ERR_ARG, "vae_output_shape: null argument");
    const AetherVaeConfig& c = h->cfg;
    const int down = 1 << (c.num_levels - 1);
    *oC = decode ? c.out_channels : 2 * c.latent_channels;
    *oT = total_out_frames(c, decode != 0, T);
    *oH = decode ? H * down : H / down;
    *oW = decode ? W * down : W / down;
    return AETHER_OK;
}

extern "C" int aether_vae_encode(AetherVae* h, const void* x, int T, int H, int W, int tiling, void* moments, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!h || !x || !moments) return aether_set_error(AETHER_ERR_ARG, "vae_encode: null argument");
    return vae_run(h, false, x, T, H, W, tiling, moments, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int aether_vae_decode(AetherVae* h, const void* z, int T, int H, int W, int tiling, void* sample, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (!h || !z || !sample) return aether_set_error(AETHER_ERR_ARG, "vae_decode: null argument");
    return vae_run(h, true, z, T, H, W, tiling, sample, workspace, workspace_bytes, stream, nullptr);
}
