// Host-side launch plan of one CogVideoXTransformer3DModel.forward as the reference invokes it at
// aether/pipelines/aetherv1_pipeline_cogvideox.py:865-875 (hidden [B,F,96,H,W] + text [B,226,4096] + timestep
// + 3-D rope -> noise prediction [B,F,56,H,W]).  Pure enqueue: no allocation, no synchronisation, so the
// whole forward can be captured into a hipGraph by the caller.
//
// Token layout in HBM: one residual stream X[B, S, D] bf16 with the 226 text rows first, then the
// F·(H/2)·(W/2) video rows of each batch (the order diffusers' attention processor concatenates them in).
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/aether_hip.h"

struct AetherDit {
    AetherDitConfig cfg;
    std::map<std::string, const void*> w;
    bool weights_checked = false;     // every name of kRequired registered (re-evaluated after each set_weight, not per forward)
    const void* pos_emb = nullptr;    // positional table added after the patch embedding: [pos_rows, D] bf16 (text rows first)
    int pos_rows = 0;
    // optional per-kernel-class timing (hipEvents recorded on the launch stream around each enqueue)
    bool profile = false;
    std::vector<hipEvent_t> events;   // pairs (start, stop)
    std::vector<int> event_cat;       // class of pair i
    size_t pairs_used = 0;
};

namespace {

const char* const kRequired[] = {
    "patch_w", "patch_b", "text_w", "text_b", "time_w1", "time_b1", "time_w2", "time_b2", "adaln_w", "adaln_b",
    "ln1_w", "ln1_b", "qkv_w", "qkv_b", "qn_w", "qn_b", "kn_w", "kn_b", "o_w", "o_b", "ln2_w", "ln2_b",
    "ff1_w", "ff1_b", "ff2_w", "ff2_b", "normf_w", "normf_b", "normo_w", "normo_b", "proj_w", "proj_b"};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr size_t kSplitKBytes = (size_t)256 * 256 * 256 * sizeof(float);   // fp32 partial tiles of a GEMM tail launch (<= 256 workgroups)

struct Plan {
    int D, FF, Nt, Nv, S, M, Spad, Kp, Np, Nmod, PH, PW;
    size_t off_x, off_xn, off_qkv, off_qh, off_kh, off_vt, off_attn, off_ff, off_patch, off_tsin, off_t1, off_temb,
        off_mod, off_proj, off_splitk, total;
};

Plan make_plan(const AetherDitConfig& c, int B, int F, int H, int W) {
    Plan p;
    p.D = c.num_heads * c.head_dim;
    p.FF = p.D * c.ff_mult;
    p.Nt = c.max_text_len;
    p.PH = H / c.patch_size;
    p.PW = W / c.patch_size;
    p.Nv = F * p.PH * p.PW;
    p.S = p.Nt + p.Nv;
    p.M = B * p.S;
    p.Spad = (p.S + 63) / 64 * 64;
    p.Kp = c.in_channels * c.patch_size * c.patch_size;
    p.Np = c.out_channels * c.patch_size * c.patch_size;
    p.Nmod = c.num_layers * 12 * p.D + 2 * p.D;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    p.off_x = take((size_t)p.M * p.D * 2);
    p.off_xn = take((size_t)p.M * p.D * 2);
    p.off_qkv = take((size_t)p.M * 3 * p.D * 2);
    p.off_qh = take((size_t)p.M * p.D * 2);
    p.off_kh = take((size_t)p.M * p.D * 2);
    p.off_vt = take((size_t)B * c.num_heads * 64 * p.Spad * 2);
    p.off_attn = take((size_t)p.M * p.D * 2);
    p.off_ff = take((size_t)p.M * p.FF * 2);
    p.off_patch = take((size_t)B * p.Nv * p.Kp * 2);
    p.off_tsin = take((size_t)B * p.D * 4);
    p.off_t1 = take((size_t)B * c.time_embed_dim * 4);
    p.off_temb = take((size_t)B * c.time_embed_dim * 4);
    p.off_mod = take((size_t)B * p.Nmod * 4);
    p.off_proj = take((size_t)B * p.Nv * p.Np * 2);
    p.off_splitk = take(kSplitKBytes);
    p.total = o;
    return p;
}

}  // namespace

// flag bits a transformer handle understands; anything else is a caller error (flags of deleted kernel variants must not be accepted silently)
static constexpr int kDitFlags = AETHER_GEMM_WIDE_STORE | AETHER_ATTN_EXACT_MAX;

extern "C" AetherDit* aether_dit_create(const AetherDitConfig* cfg) {
    if (!cfg) { aether_set_error(AETHER_ERR_ARG, "dit_create: null config"); return nullptr; }
    if (cfg->flags & ~kDitFlags) { aether_set_error(AETHER_ERR_ARG, "dit_create: undefined flag bits (defined: AETHER_GEMM_WIDE_STORE, AETHER_ATTN_EXACT_MAX)"); return nullptr; }
    if (cfg->head_dim != 64) { aether_set_error(AETHER_ERR_SHAPE, "dit_create: head_dim must be 64"); return nullptr; }
    const int D = cfg->num_heads * cfg->head_dim;
    if (D % 512 != 0 || D > 4096) { aether_set_error(AETHER_ERR_SHAPE, "dit_create: hidden size must be a multiple of 512, <= 4096"); return nullptr; }
    const int Kp = cfg->in_channels * cfg->patch_size * cfg->patch_size;
    const int Np = cfg->out_channels * cfg->patch_size * cfg->patch_size;
    if (Kp % 64 != 0 || cfg->text_dim % 64 != 0 || Np % 32 != 0 || cfg->time_embed_dim % 8 != 0) {
        aether_set_error(AETHER_ERR_SHAPE, "dit_create: in_channels*p*p and text_dim must be multiples of 64, out_channels*p*p of 32");
        return nullptr;
    }
    AetherDit* h = new AetherDit();
    h->cfg = *cfg;
    return h;
}

extern "C" void aether_dit_destroy(AetherDit* h) {
    if (!h) return;
    for (hipEvent_t e : h->events) hipEventDestroy(e);
    delete h;
}

extern "C" int aether_dit_set_weight(AetherDit* h, const char* name, const void* dev_ptr) {
    if (!h || !name) return aether_set_error(AETHER_ERR_ARG, "dit_set_weight: null argument");
    if (((uintptr_t)dev_ptr) & 15) return aether_set_error(AETHER_ERR_ALIGN, "dit_set_weight: pointer must be 16-byte aligned");
    if (std::string(name) == "pos_emb")
        return aether_set_error(AETHER_ERR_ARG, "dit_set_weight: the positional table carries a row count: use aether_dit_set_pos_embedding");
    h->w[name] = dev_ptr;
    h->weights_checked = false;
    return AETHER_OK;
}

extern "C" int aether_dit_set_pos_embedding(AetherDit* h, const void* table, int rows) {
    if (!h) return aether_set_error(AETHER_ERR_ARG, "dit_set_pos_embedding: null handle");
    if (((uintptr_t)table) & 15) return aether_set_error(AETHER_ERR_ALIGN, "dit_set_pos_embedding: pointer must be 16-byte aligned");
    if ((table == nullptr) != (rows <= 0)) return aether_set_error(AETHER_ERR_ARG, "dit_set_pos_embedding: give a table and its row count, or (null, 0)");
    h->pos_emb = table;
    h->pos_rows = table ? rows : 0;
    return AETHER_OK;
}

extern "C" int aether_dit_set_flags(AetherDit* h, int flags) {
    if (!h) return aether_set_error(AETHER_ERR_ARG, "dit_set_flags: null handle");
    if (flags & ~kDitFlags) return aether_set_error(AETHER_ERR_ARG, "dit_set_flags: undefined flag bits (defined: AETHER_GEMM_WIDE_STORE, AETHER_ATTN_EXACT_MAX)");
    h->cfg.flags = flags;
    return AETHER_OK;
}

extern "C" size_t aether_dit_workspace_bytes(const AetherDit* h, int B, int F, int H, int W) {
    if (!h || B <= 0 || F <= 0 || H <= 0 || W <= 0) return 0;
    return make_plan(h->cfg, B, F, H, W).total;
}

static int prof_begin(AetherDit* h, int cat, hipStream_t s) {
    if (!h->profile) return -1;
    if (h->pairs_used * 2 + 2 > h->events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        h->events.push_back(a);
        h->events.push_back(b);
        h->event_cat.push_back(cat);
    }
    const size_t i = h->pairs_used++;
    h->event_cat[i] = cat;
    hipEventRecord(h->events[2 * i], s);
    return (int)i;
}
static void prof_end(AetherDit* h, int idx, hipStream_t s) {
    if (idx >= 0) hipEventRecord(h->events[2 * idx + 1], s);
}

// AE_RUN(class, call): enqueue one kernel (group), optionally bracketed by timing events
#define AE_RUN(cat, expr)                                        \
    do {                                                         \
        const int pi_ = prof_begin(h, (cat), (hipStream_t)stream); \
        int rc_ = (expr);                                        \
        prof_end(h, pi_, (hipStream_t)stream);                   \
        if (rc_ != 0) return rc_;                                \
    } while (0)
#define AE_TRY(expr) AE_RUN(AETHER_PROF_OTHER, expr)

extern "C" int aether_dit_set_profile(AetherDit* h, int enable) {
    if (!h) return aether_set_error(AETHER_ERR_ARG, "dit_set_profile: null handle");
    h->profile = enable != 0;
    h->pairs_used = 0;
    return AETHER_OK;
}

extern "C" int aether_dit_get_profile(AetherDit* h, float* ms_per_class, int* launches_per_class) {
    if (!h || !ms_per_class || !launches_per_class) return aether_set_error(AETHER_ERR_ARG, "dit_get_profile: null argument");
    for (int c = 0; c < AETHER_PROF_NUM; ++c) { ms_per_class[c] = 0.f; launches_per_class[c] = 0; }
    for (size_t i = 0; i < h->pairs_used; ++i) {
        if (hipEventSynchronize(h->events[2 * i + 1]) != hipSuccess) return aether_set_error(AETHER_ERR_LAUNCH, "dit_get_profile: event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->events[2 * i], h->events[2 * i + 1]) != hipSuccess) return aether_set_error(AETHER_ERR_LAUNCH, "dit_get_profile: elapsed failed");
        ms_per_class[h->event_cat[i]] += ms;
        launches_per_class[h->event_cat[i]] += 1;
    }
    h->pairs_used = 0;
    return AETHER_OK;
}

extern "C" int aether_dit_forward(AetherDit* h, const void* hidden, const void* text, const float* timesteps,
                                  const float* rope_cos, const float* rope_sin, void* out, int B, int F, int H, int W,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !hidden || !text || !timesteps || !out || !workspace) return aether_set_error(AETHER_ERR_ARG, "dit_forward: null argument");
    const AetherDitConfig& c = h->cfg;
    if (B < 1 || B > 8) return aether_set_error(AETHER_ERR_SHAPE, "dit_forward: 1 <= B <= 8");
    if (H % c.patch_size || W % c.patch_size) return aether_set_error(AETHER_ERR_SHAPE, "dit_forward: H, W must be multiples of patch_size");
    if (!rope_cos || !rope_sin) return aether_set_error(AETHER_ERR_ARG, "dit_forward: rotary tables are required");
    if (!h->weights_checked) {
        for (const char* name : kRequired)
            if (h->w.find(name) == h->w.end() || h->w[name] == nullptr) {
                std::string msg = std::string("dit_forward: weight not registered: ") + name;
                return aether_set_error(AETHER_ERR_ARG, msg.c_str());
            }
        h->weights_checked = true;
    }
    const Plan p = make_plan(c, B, F, H, W);
    if (c.use_pos_embedding) {
        if (!h->pos_emb) return aether_set_error(AETHER_ERR_ARG, "dit_forward: use_pos_embedding set but no table registered (aether_dit_set_pos_embedding)");
        if (h->pos_rows < p.S) return aether_set_error(AETHER_ERR_SHAPE, "dit_forward: positional table has fewer rows than text + video tokens");
    }
    if (workspace_bytes < p.total) return aether_set_error(AETHER_ERR_ARG, "dit_forward: workspace too small");
    if (((uintptr_t)workspace) & 255) return aether_set_error(AETHER_ERR_ALIGN, "dit_forward: workspace must be 256-byte aligned");

    char* ws = (char*)workspace;
    char* x = ws + p.off_x; char* xn = ws + p.off_xn; char* qkv = ws + p.off_qkv;
    char* qh = ws + p.off_qh; char* kh = ws + p.off_kh; char* vt = ws + p.off_vt;
    char* attn = ws + p.off_attn; char* ff = ws + p.off_ff; char* patch = ws + p.off_patch;
    float* tsin = (float*)(ws + p.off_tsin); float* t1 = (float*)(ws + p.off_t1); float* temb = (float*)(ws + p.off_temb);
    float* mod = (float*)(ws + p.off_mod); char* proj = ws + p.off_proj; float* skws = (float*)(ws + p.off_splitk);
    const int D = p.D, FF = p.FF, S = p.S, M = p.M, Nt = p.Nt, Nv = p.Nv, L = c.num_layers, fl = c.flags;
    auto W_ = [&](const char* n) { return (const char*)h->w[n]; };
    auto Wf = [&](const char* n) { return (const float*)h->w[n]; };

    // ---- embeddings ------------------------------------------------------------------------------
    AE_TRY(aether_patchify(hidden, patch, B, F, c.in_channels, H, W, c.patch_size, stream));
    const char* pos = c.use_pos_embedding ? (const char*)h->pos_emb : nullptr;
    for (int b = 0; b < B; ++b) {
        char* xb = x + (size_t)b * S * D * 2;
        AE_TRY(aether_gemm_bf16((const char*)text + (size_t)b * Nt * c.text_dim * 2, c.text_dim, W_("text_w"), c.text_dim, xb, D,
                                Nt, D, c.text_dim, Wf("text_b"), pos ? AETHER_EPI_BIAS_GATE_RES : AETHER_EPI_BIAS, pos, D,
                                nullptr, nullptr, 0, 0, 0, skws, kSplitKBytes, fl, stream));
        AE_TRY(aether_gemm_bf16(patch + (size_t)b * Nv * p.Kp * 2, p.Kp, W_("patch_w"), p.Kp, xb + (size_t)Nt * D * 2, D, Nv, D,
                                p.Kp, Wf("patch_b"), pos ? AETHER_EPI_BIAS_GATE_RES : AETHER_EPI_BIAS,
                                pos ? pos + (size_t)Nt * D * 2 : nullptr, D, nullptr, nullptr, 0, 0, 0, skws, kSplitKBytes, fl, stream));
    }
    // ---- timestep embedding and every AdaLN modulation vector of the forward in one GEMV ---------
    AE_TRY(aether_timestep_sinusoid(timesteps, B, D, tsin, stream));
    AE_TRY(aether_gemv_rows(tsin, B, D, W_("time_w1"), Wf("time_b1"), t1, c.time_embed_dim, 0, 1, stream));
    AE_TRY(aether_gemv_rows(t1, B, c.time_embed_dim, W_("time_w2"), Wf("time_b2"), temb, c.time_embed_dim, 0, 0, stream));
    AE_TRY(aether_gemv_rows(temb, B, c.time_embed_dim, W_("adaln_w"), Wf("adaln_b"), mod, p.Nmod, 1, 0, stream));

    // ---- transformer blocks ----------------------------------------------------------------------
    // softmax scale 1/sqrt(64) and log2(e) folded into Q in fp32 before its single rounding to bf16: scores arrive in
    // the log2 domain.
    const float q_scale = 0.125f * 1.4426950408889634f;
    for (int i = 0; i < L; ++i) {
        const float* m1 = mod + (size_t)i * 12 * D;  // shift, scale, gate, enc_shift, enc_scale, enc_gate
        const float* m2 = m1 + 6 * D;
        AE_RUN(AETHER_PROF_LN, aether_layernorm_modulate(x, D, xn, D, M, D, c.norm_eps, Wf("ln1_w") + (size_t)i * D, Wf("ln1_b") + (size_t)i * D,
                                         m1, m1 + D, m1 + 3 * D, m1 + 4 * D, p.Nmod, S, Nt, stream));
        AE_RUN(AETHER_PROF_GEMM_QKV, aether_gemm_bf16(xn, D, W_("qkv_w") + (size_t)i * 3 * D * D * 2, D, qkv, 3 * D, M, 3 * D, D,
                                Wf("qkv_b") + (size_t)i * 3 * D, AETHER_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 0, 0, skws, kSplitKBytes, fl, stream));
        // (q/k norm + RoPE + V^T as the qkv GEMM's EPILOGUE was built and measured in round 3: parity green, net 0 +- 0.5 ms per step — 8 waves
        // x 256 registers leave the epilogue nothing to hide its table fetches behind — and retired in round 4: profiles/r03_fused_qkv_epilogue.txt)
        AE_RUN(AETHER_PROF_QKROPE, aether_qk_norm_rope(qkv, B, S, c.num_heads, Nt, Wf("qn_w") + i * 64, Wf("qn_b") + i * 64, Wf("kn_w") + i * 64,
                                   Wf("kn_b") + i * 64, c.qk_norm_eps, rope_cos, rope_sin, q_scale, qh, kh, vt, p.Spad, stream));
        AE_RUN(AETHER_PROF_ATTN, aether_flash_attn_fwd(qh, kh, vt, attn, B, c.num_heads, S, p.Spad, fl, stream));
        AE_RUN(AETHER_PROF_GEMM_O, aether_gemm_bf16(attn, D, W_("o_w") + (size_t)i * D * D * 2, D, x, D, M, D, D, Wf("o_b") + (size_t)i * D,
                                AETHER_EPI_BIAS_GATE_RES, x, D, m1 + 2 * D, m1 + 5 * D, p.Nmod, S, Nt, skws, kSplitKBytes, fl, stream));
        AE_RUN(AETHER_PROF_LN, aether_layernorm_modulate(x, D, xn, D, M, D, c.norm_eps, Wf("ln2_w") + (size_t)i * D, Wf("ln2_b") + (size_t)i * D,
                                         m2, m2 + D, m2 + 3 * D, m2 + 4 * D, p.Nmod, S, Nt, stream));
        AE_RUN(AETHER_PROF_GEMM_FF1, aether_gemm_bf16(xn, D, W_("ff1_w") + (size_t)i * FF * D * 2, D, ff, FF, M, FF, D, Wf("ff1_b") + (size_t)i * FF,
                                AETHER_EPI_BIAS_GELU, nullptr, 0, nullptr, nullptr, 0, 0, 0, skws, kSplitKBytes, fl, stream));
        AE_RUN(AETHER_PROF_GEMM_FF2, aether_gemm_bf16(ff, FF, W_("ff2_w") + (size_t)i * D * FF * 2, FF, x, D, M, D, FF, Wf("ff2_b") + (size_t)i * D,
                                AETHER_EPI_BIAS_GATE_RES, x, D, m2 + 2 * D, m2 + 5 * D, p.Nmod, S, Nt, skws, kSplitKBytes, fl, stream));
    }

    // ---- norm_final -> norm_out (AdaLayerNorm, shift first) -> proj_out -> un-patchify -----------
    const float* mf = mod + (size_t)L * 12 * D;  // shift | scale
    AE_TRY(aether_layernorm_modulate(x, D, xn, D, M, D, c.norm_eps, Wf("normf_w"), Wf("normf_b"), nullptr, nullptr, nullptr,
                                     nullptr, 0, S, Nt, stream));
    AE_TRY(aether_layernorm_modulate(xn, D, x, D, M, D, c.norm_eps, Wf("normo_w"), Wf("normo_b"), mf, mf + D, mf, mf + D, p.Nmod,
                                     S, Nt, stream));
    for (int b = 0; b < B; ++b)
        AE_TRY(aether_gemm_bf16(x + ((size_t)b * S + Nt) * D * 2, D, W_("proj_w"), D, proj + (size_t)b * Nv * p.Np * 2, p.Np, Nv,
                                p.Np, D, Wf("proj_b"), AETHER_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 0, 0, skws, kSplitKBytes, fl, stream));
    AE_TRY(aether_unpatchify(proj, p.Np, out, B, F, c.out_channels, H, W, c.patch_size, stream));
    return AETHER_OK;
}
