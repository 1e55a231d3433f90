// HBM-bound kernels of the DiT forward (everything that is not a GEMM or attention):
//   LayerNorm(+affine)+AdaLN modulation, the small fp32 GEMVs of the timestep / AdaLN path,
//   sinusoidal timestep features, patchify / un-patchify, q/k LayerNorm + 3-D RoPE + head-major re-layout
//   and the V transpose that feeds the attention kernel.
// All of them stream 16 bytes per lane (bf16 x 8), keep fp32 statistics, and round to bf16 once.
// Reference call site for all of them: the transformer call at
// aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875 (diffusers CogVideoXTransformer3DModel.forward).
#include "common.hpp"
#include "../../include/aether_hip.h"

namespace aether {

// ------------------------------------------------------------------------------------------------
// LayerNorm + modulate: one wavefront per row, the row (<= 4096 elements) lives in registers.
// ------------------------------------------------------------------------------------------------
struct LnArgs {
    const bf16_t* x; int ldx; bf16_t* y; int ldy; int rows, D; float eps;
    const float* w; const float* b;
    const float* shift_vid; const float* scale_vid; const float* shift_txt; const float* scale_txt;
    int mod_bstride, rows_per_batch, n_text;
};

// Rows are handed out in BLOCKS of 4 (one row per wave) that never straddle a (batch item, row type) segment — text rows [0, n_text) or video
// rows [n_text, S) of a batch item share their modulation vectors.  The grid is PERSISTENT (at most three workgroups per CU: 48 KiB of LDS each
// at D = 3072): workgroup g walks blocks g, g + G, g + 2G, ... and stages the four fp32 vectors a row needs (LayerNorm weight / bias, AdaLN
// scale / shift: 48 KiB, EIGHT times the 6 KiB of the row itself) in LDS only when the segment changes — once or twice per launch — instead
// of re-reading them from L2 for every row (round 1: 1.5 GB of L2 traffic per launch, the kernel sat on the L2 roof at 3.9 TB/s) or once per
// 16 rows (rounds 2-3; 943 such workgroups on 768 resident slots also left a 23 % second round).  A wave owns one row at a time (the row
// lives in registers), software-pipelined by one block; the arithmetic and its order are unchanged.
constexpr int LN_RPB = 4;
template <int NCH>  // D = NCH * 512
__global__ __launch_bounds__(256) void layernorm_modulate_kernel(LnArgs p, int blk_txt, int blk_vid, int nblocks) {
    __shared__ __attribute__((aligned(16))) float vec[4][NCH * 512];     // w, b, scale, shift
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_batch = blk_txt + blk_vid;
    // block id -> (segment key, row of this wave or -1)
    auto locate = [&](int blk, int& key) -> int {
        const int bidx = blk / per_batch, r = blk - bidx * per_batch;
        const bool txt = r < blk_txt;
        key = 2 * bidx + (txt ? 1 : 0);
        const int seg0 = bidx * p.rows_per_batch + (txt ? 0 : p.n_text);
        const int seg1 = min(p.rows, bidx * p.rows_per_batch + (txt ? p.n_text : p.rows_per_batch));
        const int row = seg0 + (txt ? r : r - blk_txt) * LN_RPB + wave;
        return row < seg1 ? row : -1;
    };
    auto load_row = [&](int row, u16x8 (&dst)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) dst[c] = *(const u16x8*)(p.x + (size_t)row * p.ldx + c * 512 + lane * 8);
    };
    int staged = -1;
    bool mod = false;
    u16x8 cur[NCH], nxt[NCH];
    int key = 0, blk = blockIdx.x;
    int row = blk < nblocks ? locate(blk, key) : -1;
    if (row >= 0) load_row(row, cur);
    for (; blk < nblocks; blk += gridDim.x) {
        // the next block's row is requested BEFORE this one is reduced and stored (x and y are not declared disjoint: the compiler may not
        // hoist the loads above the stores by itself); it stays packed (bf16) until it becomes the current one
        int key_next = 0;
        const int row_next = blk + (int)gridDim.x < nblocks ? locate(blk + gridDim.x, key_next) : -1;
        if (row_next >= 0) load_row(row_next, nxt);
        if (key != staged) {                                  // workgroup-uniform: every wave of the workgroup is at the same block
            __syncthreads();                                  // (all waves are done with the previous segment's vectors)
            const int bidx = key >> 1;
            const bool txt = key & 1;
            const float* shift = nullptr; const float* scale = nullptr;
            if (p.shift_vid != nullptr) {
                shift = (txt ? p.shift_txt : p.shift_vid) + (size_t)bidx * p.mod_bstride;
                scale = (txt ? p.scale_txt : p.scale_vid) + (size_t)bidx * p.mod_bstride;
            }
            mod = shift != nullptr;
            for (int i = threadIdx.x * 4; i < NCH * 512; i += 1024) {
                *(f32x4*)&vec[0][i] = p.w ? *(const f32x4*)(p.w + i) : f32x4{1.f, 1.f, 1.f, 1.f};
                *(f32x4*)&vec[1][i] = p.w ? *(const f32x4*)(p.b + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)&vec[2][i] = scale ? *(const f32x4*)(scale + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4*)&vec[3][i] = shift ? *(const f32x4*)(shift + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            staged = key;
        }
        if (row >= 0) {
            float v[NCH][8];
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[c][e] = bf16_bits_to_f32(cur[c][e]); sum += v[c][e]; }
            }
            const float mean = wave_sum(sum) / (float)p.D;
            float sq = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
            const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
            bf16_t* yr = p.y + (size_t)row * p.ldy;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int d0 = c * 512 + lane * 8;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd;
                if (p.w != nullptr) {
                    const f32x4 w0 = *(const f32x4*)&vec[0][d0], w1 = *(const f32x4*)&vec[0][d0 + 4];
                    const f32x4 b0 = *(const f32x4*)&vec[1][d0], b1 = *(const f32x4*)&vec[1][d0 + 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = o[e] * w0[e] + b0[e]; o[e + 4] = o[e + 4] * w1[e] + b1[e]; }
                }
                if (mod) {
                    const f32x4 s0 = *(const f32x4*)&vec[2][d0], s1 = *(const f32x4*)&vec[2][d0 + 4];
                    const f32x4 h0 = *(const f32x4*)&vec[3][d0], h1 = *(const f32x4*)&vec[3][d0 + 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = o[e] * (1.0f + s0[e]) + h0[e];
                        o[e + 4] = o[e + 4] * (1.0f + s1[e]) + h1[e];
                    }
                }
                uint4 out = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                                       pack_bf16x2(o[6], o[7]));
                *(uint4*)(yr + d0) = out;
            }
        }
        row = row_next; key = key_next;
        if (row_next >= 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out[b,n] = act_out(bias[n] + sum_k act_in(x[b,k]) W[n,k]) — one wavefront per output feature n.
// ------------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ x, int K, const bf16_t* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, int N,
                                                        int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* wr = W + (size_t)n * K;
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        const u16x8 raw = *(const u16x8*)(wr + k0);
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = bf16_bits_to_f32(raw[e]);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const f32x4 x0 = *(const f32x4*)(x + (size_t)b * K + k0), x1 = *(const f32x4*)(x + (size_t)b * K + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = x0[e], a1 = x1[e];
                if (act_in == 1) { a0 = silu(a0); a1 = silu(a1); }
                acc[b] += a0 * wv[e] + a1 * wv[e + 4];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        float r = wave_sum(acc[b]);
        if (lane == 0) {
            if (bias != nullptr) r += bias[n];
            if (act_out == 1) r = silu(r);
            out[(size_t)b * N + n] = r;
        }
    }
}

__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, int B, int dim, float* __restrict__ out) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const float freq = expf(-9.210340371976184f * (float)i / (float)half);  // ln(10000)
    const float arg = t[b] * freq;
    out[(size_t)b * dim + i] = cosf(arg);          // flip_sin_to_cos=True: [cos | sin]
    out[(size_t)b * dim + half + i] = sinf(arg);
}

// ------------------------------------------------------------------------------------------------
// patchify: x[B,F,C,H,W] -> A[(b,f,ph,pw), (c,dy,dx)]      (p = 2: each thread makes 4 consecutive columns)
// ------------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ A, int BF, int C,
                                int H, int W, int p) {
    const int PH = H / p, PW = W / p, pp = p * p;
    const size_t total = (size_t)BF * PH * PW * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = idx % C;
        size_t r = idx / C;
        const int pw = r % PW; r /= PW;
        const int ph = r % PH; r /= PH;
        const size_t bf = r;
        const unsigned short* src = x + ((bf * C + c) * H + (size_t)ph * p) * W + (size_t)pw * p;
        unsigned short* dst = A + (((bf * PH + ph) * PW + pw) * C + c) * pp;
        for (int dy = 0; dy < p; ++dy)
            for (int dx = 0; dx < p; ++dx) dst[dy * p + dx] = src[(size_t)dy * W + dx];
    }
}

// un-patchify: Y[(b,f,ph,pw), (dy,dx,c)] -> out[B,F,Cout,H,W]
// (diffusers: output.reshape(B,F,H/p,W/p,-1,p,p).permute(0,1,4,2,5,3,6): proj_out column = c*p*p + dy*p + dx)
__global__ void unpatchify_kernel(const unsigned short* __restrict__ Y, int ldy, unsigned short* __restrict__ out, int BF,
                                  int Cout, int H, int W, int p) {
    const int PH = H / p, PW = W / p;
    const size_t total = (size_t)BF * Cout * H * W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int w = idx % W;
        size_t r = idx / W;
        const int h = r % H; r /= H;
        const int c = r % Cout; r /= Cout;
        const size_t bf = r;
        const int ph = h / p, dy = h - ph * p, pw = w / p, dx = w - pw * p;
        out[idx] = Y[((bf * PH + ph) * PW + pw) * (size_t)ldy + (size_t)c * p * p + dy * p + dx];
    }
}

// ------------------------------------------------------------------------------------------------
// ONE pass over the fused qkv projection: q/k LayerNorm(64) + 3-D RoPE (+ softmax scale on q) -> head-major Qh / Kh,
// V -> V^T (through LDS).  One workgroup per (64-token tile, batch·head): it reads the
// three 64 x 64 blocks of its head out of the [B, S, 3·H·64] projection (128-byte row segments) and writes three 8 KiB
// tiles.  Round 1 did this in two kernels (q/k rows, then V transpose + a re-read of Kh for the norms: 648 MB of traffic per
// layer at S = 15 076); fused it moves each element once in, once out (554 MB).
// ------------------------------------------------------------------------------------------------
struct QkArgs {
    const bf16_t* qkv; int S, H, n_text, Spad;
    const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b; float eps;
    const float* cos_t; const float* sin_t; float q_scale;
    bf16_t* Qh; bf16_t* Kh; unsigned short* Vt;
};

// LayerNorm(64) of one head row spread over 8 lanes (8 values each) + affine + RoPE (adjacent pairs) + scale, rounded to bf16;
// every operand is already in registers (all loads of the workgroup are issued up front: one memory round trip, not eight)
AE_DEV uint4 qk_row(const u16x8 raw, const f32x4 (&nw)[2], const f32x4 (&nb)[2], float eps, const f32x4 (&cs)[2], const f32x4 (&sn)[2], bool rope,
                    float sc) {
    float v[8];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = bf16_bits_to_f32(raw[e]); sum += v[e]; }
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
    const float mean = sum * (1.0f / 64.0f);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * nw[e >> 2][e & 3] + nb[e >> 2][e & 3];
    if (rope) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float x0 = v[e], x1 = v[e + 1];
            v[e] = x0 * cs[e >> 2][e & 3] - x1 * sn[e >> 2][e & 3];
            v[e + 1] = x1 * cs[(e + 1) >> 2][(e + 1) & 3] + x0 * sn[(e + 1) >> 2][(e + 1) & 3];
        }
    }
    return make_uint4(pack_bf16x2(v[0] * sc, v[1] * sc), pack_bf16x2(v[2] * sc, v[3] * sc), pack_bf16x2(v[4] * sc, v[5] * sc),
                      pack_bf16x2(v[6] * sc, v[7] * sc));
}

__global__ __launch_bounds__(256) void qkv_prepare_kernel(QkArgs p) {
    __shared__ unsigned short tile[64][66];
    // head fastest in the grid: the workgroups running together cover the same 64 token rows across all heads, i.e. whole
    // contiguous 18 KiB rows of the projection between them
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int tile_idx = blockIdx.y;
    const int s0 = tile_idx * 64;
    const int HD = p.H * 64;
    const int tid = threadIdx.x;
    const int ch = tid & 7;                       // 16-byte chunk of the 128-byte head row (the same in both passes)
    // ---- every load of this thread, issued before anything is consumed ------------------------------------------------------------
    u16x8 rq[2], rk[2], rv[2];
    f32x4 cs[2][2], sn[2][2];
    bool ok[2], rope[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = s0 + (tid >> 3) + 32 * i;   // token of piece tid + 256 i
        ok[i] = s < p.S;
        const bf16_t* row = p.qkv + ((size_t)b * p.S + (ok[i] ? s : p.S - 1)) * 3 * HD + h * 64 + ch * 8;
        rq[i] = *(const u16x8*)row;
        rk[i] = *(const u16x8*)(row + HD);
        rv[i] = *(const u16x8*)(row + 2 * HD);
        const int vtok = s - p.n_text;
        rope[i] = ok[i] && vtok >= 0;
        const size_t to = (size_t)(rope[i] ? vtok : 0) * 64 + ch * 8;     // (n_text == S: no table; never dereferenced then)
        if (p.cos_t != nullptr) {
            cs[i][0] = *(const f32x4*)(p.cos_t + to); cs[i][1] = *(const f32x4*)(p.cos_t + to + 4);
            sn[i][0] = *(const f32x4*)(p.sin_t + to); sn[i][1] = *(const f32x4*)(p.sin_t + to + 4);
        }
    }
    f32x4 qw[2], qb[2], kw[2], kb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qw[j] = *(const f32x4*)(p.qn_w + ch * 8 + 4 * j); qb[j] = *(const f32x4*)(p.qn_b + ch * 8 + 4 * j);
        kw[j] = *(const f32x4*)(p.kn_w + ch * 8 + 4 * j); kb[j] = *(const f32x4*)(p.kn_b + ch * 8 + 4 * j);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = (tid >> 3) + 32 * i;
        const int s = s0 + sl;
        const uint4 q = qk_row(rq[i], qw, qb, p.eps, cs[i], sn[i], rope[i], p.q_scale);
        const uint4 k = qk_row(rk[i], kw, kb, p.eps, cs[i], sn[i], rope[i], 1.0f);
        if (ok[i]) {
            const size_t o = (((size_t)b * p.H + h) * p.S + s) * 64 + ch * 8;
            *(uint4*)(p.Qh + o) = q;
            *(uint4*)(p.Kh + o) = k;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[ch * 8 + e][sl] = ok[i] ? rv[i][e] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = tid + i * 256;          // d = piece/8, token chunk = piece%8; pad columns (s >= S) are zero
        const int d = piece >> 3, ch = piece & 7;
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[d][ch * 8 + e];
        *(u16x8*)(p.Vt + ((size_t)bh * 64 + d) * p.Spad + s0 + ch * 8) = o;
    }
}

}  // namespace aether

using namespace aether;

#define AE_STREAM ((hipStream_t)stream)

extern "C" int aether_layernorm_modulate(const void* x, int ldx, void* y, int ldy, int rows, int D, float eps, const float* w,
                                         const float* b, const float* shift_vid, const float* scale_vid,
                                         const float* shift_txt, const float* scale_txt, int mod_bstride,
                                         int rows_per_batch, int n_text, void* stream) {
    if (!x || !y || rows <= 0) return aether_set_error(AETHER_ERR_ARG, "layernorm: bad arguments");
    if (D % 512 != 0 || D > 4096 || D <= 0) return aether_set_error(AETHER_ERR_SHAPE, "layernorm: D must be a multiple of 512, <= 4096");
    if ((ldx % 8) || (ldy % 8)) return aether_set_error(AETHER_ERR_ALIGN, "layernorm: leading dimensions must be multiples of 8");
    if ((w == nullptr) != (b == nullptr)) return aether_set_error(AETHER_ERR_ARG, "layernorm: w and b must be given together");
    const int nmod = (shift_vid != nullptr) + (scale_vid != nullptr) + (shift_txt != nullptr) + (scale_txt != nullptr);
    if (nmod != 0 && nmod != 4) return aether_set_error(AETHER_ERR_ARG, "layernorm: give all four modulation vectors or none");
    LnArgs p{(const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, D, eps, w, b, shift_vid, scale_vid, shift_txt, scale_txt,
             mod_bstride, rows_per_batch > 0 ? rows_per_batch : rows, n_text};
    // 4-row blocks never straddle a (batch item, row type) segment: per batch item ceil(n_text / 4) + ceil((S - n_text) / 4)
    const int S = p.rows_per_batch, nb = (rows + S - 1) / S;
    const int nt = (nmod == 4) ? std::min(std::max(n_text, 0), S) : 0;     // without modulation the whole batch item is one segment
    p.n_text = nt;
    const int blk_txt = (nt + LN_RPB - 1) / LN_RPB, blk_vid = (S - nt + LN_RPB - 1) / LN_RPB;
    const int nblocks = nb * (blk_txt + blk_vid);
    const int resident = 256 * std::max(1, std::min(8, (160 * 1024) / (4 * D * 4)));          // workgroups the chip holds at once (LDS: 16 D bytes each)
    dim3 grid(std::min(nblocks, resident)), block(256);
    switch (D / 512) {
#define LN_CASE(n) case n: hipLaunchKernelGGL((layernorm_modulate_kernel<n>), grid, block, 0, AE_STREAM, p, blk_txt, blk_vid, nblocks); break;
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
#undef LN_CASE
    }
    return aether_check_launch("layernorm_modulate");
}

extern "C" int aether_gemv_rows(const float* x, int B, int K, const void* W, const float* bias, float* out, int N, int act_in,
                                int act_out, void* stream) {
    if (!x || !W || !out) return aether_set_error(AETHER_ERR_ARG, "gemv: null pointer");
    if (B < 1 || B > 8) return aether_set_error(AETHER_ERR_SHAPE, "gemv: 1 <= B <= 8");
    if (K % 8 != 0 || K <= 0 || N <= 0) return aether_set_error(AETHER_ERR_SHAPE, "gemv: K must be a positive multiple of 8");
    dim3 grid((N + 3) / 4), block(256);
    const bf16_t* Wp = (const bf16_t*)W;
    switch (B) {
#define GV_CASE(n) case n: hipLaunchKernelGGL((gemv_rows_kernel<n>), grid, block, 0, AE_STREAM, x, K, Wp, bias, out, N, act_in, act_out); break;
        GV_CASE(1) GV_CASE(2) GV_CASE(3) GV_CASE(4) GV_CASE(5) GV_CASE(6) GV_CASE(7) GV_CASE(8)
#undef GV_CASE
    }
    return aether_check_launch("gemv_rows");
}

extern "C" int aether_timestep_sinusoid(const float* t_dev, int B, int dim, float* out, void* stream) {
    if (!t_dev || !out || B <= 0 || dim <= 0 || (dim & 1)) return aether_set_error(AETHER_ERR_ARG, "timestep_sinusoid: bad arguments");
    const int total = B * (dim / 2);
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, AE_STREAM, t_dev, B, dim, out);
    return aether_check_launch("timestep_sinusoid");
}

extern "C" int aether_patchify(const void* x, void* A, int B, int F, int C, int H, int W, int p, void* stream) {
    if (!x || !A || p <= 0 || H % p || W % p) return aether_set_error(AETHER_ERR_SHAPE, "patchify: H, W must be multiples of p");
    const size_t total = (size_t)B * F * (H / p) * (W / p) * C;
    const int blocks = (int)min((size_t)8192, (total + 255) / 256);
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, AE_STREAM, (const unsigned short*)x, (unsigned short*)A, B * F, C, H, W, p);
    return aether_check_launch("patchify");
}

extern "C" int aether_unpatchify(const void* Y, int ldy, void* out, int B, int F, int Cout, int H, int W, int p, void* stream) {
    if (!Y || !out || p <= 0 || H % p || W % p || ldy < Cout * p * p) return aether_set_error(AETHER_ERR_SHAPE, "unpatchify: bad shape");
    const size_t total = (size_t)B * F * Cout * H * W;
    const int blocks = (int)min((size_t)8192, (total + 255) / 256);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks), dim3(256), 0, AE_STREAM, (const unsigned short*)Y, ldy, (unsigned short*)out, B * F, Cout, H, W, p);
    return aether_check_launch("unpatchify");
}

extern "C" int aether_qk_norm_rope(const void* qkv, int B, int S, int H, int n_text, const float* qn_w, const float* qn_b,
                                   const float* kn_w, const float* kn_b, float eps, const float* cos_t, const float* sin_t,
                                   float q_scale, void* Qh, void* Kh, void* Vt, int Spad, void* stream) {
    if (!qkv || !Qh || !Kh || !Vt || !qn_w || !qn_b || !kn_w || !kn_b) return aether_set_error(AETHER_ERR_ARG, "qk_norm_rope: null pointer");
    if (B <= 0 || S <= 0 || H <= 0 || n_text < 0 || n_text > S) return aether_set_error(AETHER_ERR_SHAPE, "qk_norm_rope: bad shape");
    if (n_text < S && (!cos_t || !sin_t)) return aether_set_error(AETHER_ERR_ARG, "qk_norm_rope: rope tables required");
    if (Spad % 64 != 0 || Spad < S) return aether_set_error(AETHER_ERR_SHAPE, "qk_norm_rope: Spad must be roundup(S,64)");
    QkArgs p{(const bf16_t*)qkv, S, H, n_text, Spad, qn_w, qn_b, kn_w, kn_b, eps, cos_t, sin_t, q_scale, (bf16_t*)Qh, (bf16_t*)Kh,
             (unsigned short*)Vt};
    hipLaunchKernelGGL(qkv_prepare_kernel, dim3(B * H, Spad / 64), dim3(256), 0, AE_STREAM, p);
    return aether_check_launch("qkv_prepare");
}

