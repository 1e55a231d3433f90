/*
 * aether_hip.h — C ABI of libaether_hip.so, the MI355X (gfx950) implementation of the AetherV1
 * latent-video denoising hot path.
 *
 * The reference (InternRobotics/Aether) has no native interface: its hot path is Python that calls three
 * diffusers objects at a handful of call sites.  Each entry point below states the reference call site
 * (file:line in /root/reference) whose arithmetic it replaces.  P: = aether/pipelines/aetherv1_pipeline_cogvideox.py.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked host; the caller (PyTorch) owns every buffer;
 *     the library never allocates or frees device memory and keeps no pointer past a call except the
 *     weight table registered on an explicit handle (aether_dit_*, aether_vae_*).
 *   - bf16 tensors are passed as void* (2 bytes / element, row-major, 16-byte aligned base + rows).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are only enqueued.
 *   - return value: 0 = ok, negative = AETHER_ERR_*; text via aether_last_error() (thread local).
 *   - no exception crosses this boundary.
 */
#ifndef AETHER_HIP_H
#define AETHER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AETHER_OK 0
#define AETHER_ERR_ARG (-1)    /* null / inconsistent argument                 */
#define AETHER_ERR_SHAPE (-2)  /* unsupported shape (see the entry's contract) */
#define AETHER_ERR_ALIGN (-3)  /* pointer or leading dimension not 16-B aligned */
#define AETHER_ERR_ARCH (-4)   /* device is not gfx950                          */
#define AETHER_ERR_LAUNCH (-5) /* HIP launch error                              */

const char* aether_last_error(void);
int aether_version(void);
/* 0 if the current HIP device is a gfx950 part, AETHER_ERR_ARCH otherwise (host-side query). */
int aether_check_device(void);

/* ------------------------------------------------------------------------------------------------
 * Per-kernel entry points (used by the native transformer / VAE and by the parity tests)
 * ------------------------------------------------------------------------------------------------ */

/* GEMM epilogues */
#define AETHER_EPI_BIAS 0          /* C = A·Wᵀ + bias                                           */
#define AETHER_EPI_BIAS_GELU 1     /* C = gelu_tanh(A·Wᵀ + bias)                                 */
#define AETHER_EPI_BIAS_GATE_RES 2 /* C = R + gate[b(m),type(m),:] ⊙ (A·Wᵀ + bias)               */
#define AETHER_GEMM_WIDE_STORE 1   /* flags bit: 16-byte stores through a half-wave exchange (register-path epilogue: the 32-column conv_out tile;
                                    * the 128 x 64 wave tiles store whole 128-byte rows through LDS since round 6 whatever this bit says) */
/* Main loop (one; the lock-step, two-k-steps-per-slot, fragment-reads-in-the-compute-slot, four-wave and persistent-grid loops measured in
 * rounds 1-3 are recorded in profiles/r0*_gemm_*): "ping-pong" — the two waves that share a SIMD alternate "read fragments from LDS + issue
 * the next tile's LDS-DMA" and "issue MFMAs" slots, phase-locked by s_barrier. */

/* C[M,N] = epi(A[M,K] · W[N,K]ᵀ), bf16 in / bf16 out / fp32 accumulate on MFMA.
 * Replaces nn.Linear in CogVideoXBlock / CogVideoXPatchEmbed / proj_out under P:865-875
 * (to_q,to_k,to_v fused as one [3D,D] weight; to_out.0; ff.net.0.proj; ff.net.2; patch_embed.proj as a
 * GEMM over 2x2 patches; text_proj; proj_out).  K % 64 == 0, N % 32 == 0, ld* % 8 == 0.
 * bias fp32 [N] or NULL.  R (bf16 [M,N], ldr) and the gates apply to AETHER_EPI_BIAS_GATE_RES only:
 * row m belongs to batch b = m / rows_per_batch and is a text row iff (m % rows_per_batch) < n_text;
 * gate = (text ? gate_txt : gate_vid)[b*gate_bstride + n]; NULL gates mean 1.  R may alias C.
 * splitk_ws (fp32 scratch of splitk_ws_bytes >= 64 MiB, may be NULL): tail balancing — when the last round of 256 tiles would
 * hold <= 128 tiles, those tiles run as a second launch with their K loop split over floor(256/rest) workgroups each and a
 * finalize kernel (fixed summation order) applies the epilogue.  Without scratch the GEMM is a single launch. */
int aether_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                     const float* bias, int epilogue, const void* R, int ldr, const float* gate_vid,
                     const float* gate_txt, int gate_bstride, int rows_per_batch, int n_text, float* splitk_ws,
                     size_t splitk_ws_bytes, int flags, void* stream);

/* y = LayerNorm(x; eps)·w + b, then optionally y·(1+scale)+shift with per-batch, per-row-type
 * modulation vectors (fp32).  Replaces CogVideoXLayerNormZero.norm + modulation, norm_final, and
 * AdaLayerNorm (norm_out) of the transformer called at P:865-875.  x,y bf16 [rows, D]; D % 512 == 0,
 * D <= 4096; w,b fp32 [D] or NULL; the shift_x / scale_x vectors are fp32, indexed [b*mod_bstride + d] or all NULL. */
int aether_layernorm_modulate(const void* x, int ldx, void* y, int ldy, int rows, int D, float eps,
                              const float* w, const float* b, const float* shift_vid, const float* scale_vid,
                              const float* shift_txt, const float* scale_txt, int mod_bstride,
                              int rows_per_batch, int n_text, void* stream);

/* out[b,n] = act_out(bias[n] + sum_k act_in(x[b,k]) · W[n,k]);  x fp32 [B,K], W bf16 [N,K], out fp32.
 * act codes: 0 none, 1 SiLU.  One wavefront per output feature; B <= 8; K % 8 == 0.
 * Replaces TimestepEmbedding.linear_1/linear_2 and every CogVideoXLayerNormZero.linear /
 * AdaLayerNorm.linear (all layers in ONE launch: their weights are concatenated along N). */
int aether_gemv_rows(const float* x, int B, int K, const void* W, const float* bias, float* out, int N,
                     int act_in, int act_out, void* stream);

/* Sinusoidal timestep features, diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0):
 * out[b, 0:dim/2] = cos(t·f), out[b, dim/2:] = sin(t·f), f_i = exp(-ln(1e4)·i/(dim/2)).  t_dev: fp32 [B] on device. */
int aether_timestep_sinusoid(const float* t_dev, int B, int dim, float* out, void* stream);

/* Patchify for CogVideoXPatchEmbed (patch_size_t = None): x bf16 [B,F,C,H,W] ->
 * A bf16 [B*F*(H/p)*(W/p), C*p*p] with column order (c, dy, dx) = Conv2d weight.flatten(1) order. */
int aether_patchify(const void* x, void* A, int B, int F, int C, int H, int W, int p, void* stream);

/* Inverse of the un-patchify reshape at the end of CogVideoXTransformer3DModel.forward:
 * Y bf16 [B*F*(H/p)*(W/p), ldy] (first p*p*Cout columns used, column order (c, dy, dx)) -> out bf16 [B,F,Cout,H,W]. */
int aether_unpatchify(const void* Y, int ldy, void* out, int B, int F, int Cout, int H, int W, int p, void* stream);

/* q/k LayerNorm(head_dim) + 3-D RoPE + head-major re-layout.  Replaces CogVideoXAttnProcessor2_0's
 * norm_q/norm_k + apply_rotary_emb (adjacent-pair convention, fp32) under P:865-875.
 * qkv bf16 [B,S,3*H*64] (q | k | v thirds) -> Qh,Kh bf16 [B,H,S,64] and Vt bf16 [B,H,64,Spad]
 * (V transposed, Spad = roundup(S,64), pad columns zeroed).  Rows [0,n_text) of each batch are text rows
 * (no RoPE); cos,sin fp32 [S-n_text, 64].  Q is additionally multiplied by q_scale in fp32 before its single rounding
 * to bf16; the attention kernel expects q_scale = log2(e)/sqrt(64) (scores in the log2 domain). */
int aether_qk_norm_rope(const void* qkv, int B, int S, int H, int n_text, const float* qn_w, const float* qn_b,
                        const float* kn_w, const float* kn_b, float eps, const float* cos_t, const float* sin_t,
                        float q_scale, void* Qh, void* Kh, void* Vt, int Spad, void* stream);

#define AETHER_ATTN_EXACT_MAX 32  /* flags bit 5: conservative path only: no shift-0 sweep — every row's shift is a true score maximum from
                                     its first tile on (generic tiles with the a-posteriori check); data-independent cost           */

/* Non-causal flash attention, head_dim 64: O[b,s,h*64+d] = softmax_2(Qh·Khᵀ)·V where softmax_2 uses base 2, i.e.
 * Qh must carry softmax_scale·log2(e) (see aether_qk_norm_rope).  Replaces F.scaled_dot_product_attention in
 * CogVideoXAttnProcessor2_0.  Qh,Kh [B,H,S,64], Vt [B,H,64,Spad], O bf16 [B,S,H*64].
 * Soft-max = exact on every path (soft-max is invariant under any per-row shift; the running maximum of the online algorithm only
 * keeps exp2 in range).  Conservative path: each row keeps a shift m that is a true score maximum (of its first tile, then of any
 * tile that forced a refresh); a tile is exponentiated against m directly — no tile maximum, no subtraction (m rides in the C
 * operand of the QK^T MFMA), no rescale — and checked afterwards: a partial tile sum above 2^100 (or NaN) makes the wave take the
 * classic online step on the scores it still holds (tile maximum, shift update, rescale) and exponentiate again.  Default path:
 * the same with shift 0 for the whole sweep in a two-tile software pipeline; finished rows whose sum is not in [2^-100, inf) or
 * whose accumulators are not finite make the WORKGROUP redo its sweep on the conservative path.  Either way the results are those
 * of an exact fp32 soft-max; only speed depends on the data (|log2-domain score| > 100 is needed to leave the fast path).
 * flags: AETHER_GEMM_WIDE_STORE (16-byte epilogue stores), AETHER_ATTN_EXACT_MAX. */
int aether_flash_attn_fwd(const void* Qh, const void* Kh, const void* Vt, void* O, int B, int H, int S, int Spad, int flags, void* stream);

/* Element-wise tail of one denoise step in ONE pass (aetherv1_pipeline_cogvideox.py:876-916): fp32 cast of the noise prediction (P:877),
 * classifier-free-guidance combine uncond + g·(cond − uncond) when nb = 2 (P:895-899), CogVideoXDPMScheduler.step for v-prediction
 * (P:907-915; x0 = sqrt(a_t)·sample − sqrt(1−a_t)·model_output, prev = m1·sample − m2·d + m_noise·noise with d = x0 on a first-order
 * step (old_x0 NULL) and m3·x0 − m4·old_x0 otherwise) and the cast back to bf16 (P:916).  model_out bf16 [nb][n], sample / noise bf16 [n]
 * (the noise is the draw the RETURNED sample uses; the caller makes the reference's draws with its generator, in the reference's
 * order), scalars = the float64 host values of the schedule cast to fp32.  Writes x0_out fp32 [n] (`pred_original_sample`) and the new
 * sample as fp32 (prev_f32) and / or bf16 (prev_bf16).  Bit-identical to the eager PyTorch sequence (every intermediate rounded where
 * PyTorch rounds it, no FMA contraction). */
int aether_dpm_step(const void* model_out, int nb, float guidance, const void* sample, const float* old_x0, const void* noise,
                    float a_sqrt, float b_sqrt, float m1, float m2, float m_noise, float m3, float m4, float* x0_out, float* prev_f32,
                    void* prev_bf16, long n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3D-causal VAE kernels (diffusers AutoencoderKLCogVideoX: encode at P:557-618, decode_latents at P:931,936).
 * Activations are channels-last volumes [NB, T, H, W, C] bf16, NB = spatial tiles batched together.
 * Convolution inputs are zero-bordered volumes that already contain the causal front frames.
 * ------------------------------------------------------------------------------------------------ */

/* Implicit-GEMM convolution: out[M = NB*oT*oH*oW, Cout] = bias + R + sum over taps of X(voxel + tap) · Wᵀ.
 * Replaces CogVideoXCausalConv3d.conv (3x3x3, input volume [NB, oT+2, oH+2, oW+2, iC]), the resamplers' Conv2d
 * (3x3 stride 1 on [NB, oT, oH+2, oW+2, iC]; 3x3 stride 2 on [NB, oT, 2*oH+1, 2*oW+1, iC]) and conv_shortcut (1x1x1).
 * tap_off: device int32 [n_taps], element offset of K step k inside the padded volume (tap base + 64*channel block);
 * W bf16 [Cout, n_taps*64] with the matching K order (tap-major, then channel); iC % 64 == 0; Cout % 32 == 0.
 * R (bf16 [M, Cout], ldr) is added when non-NULL (ResNet skip).
 * splitk_ws (fp32 scratch of splitk_ws_bytes, may be NULL): launches with <= 128 output tiles (the deep low-resolution
 * layers, K = 27*512) split their K loop over up to 256/tiles workgroups per tile; the fp32 partial tiles are summed in
 * slice order (deterministic) by a finalize kernel that also applies bias / R and rounds to bf16.
 * flags bit 7 (AETHER_CONV_TAP_REUSE): the caller states that the K order is (dt, dh, 64-channel block, dw) with dw fastest,
 * i.e. tap_off[3j+1] = tap_off[3j] + iC and tap_off[3j+2] = tap_off[3j] + 2*iC.  For a 3x3(x3) stride-1 convolution with
 * Cout % 128 == 0 the library may then run the tap-reuse kernel (one staged input tile serves the three dw taps); results
 * equal the plain kernel's up to fp32 summation order (the K order inside an output is the same: bit-identical). */
#define AETHER_CONV_TAP_REUSE 128
int aether_conv_gemm_bf16(const void* X, int NB, int iT, int iH, int iW, int iC, int oT, int oH, int oW, int stride_hw,
                          const int* tap_off, int n_taps, const void* W, int Cout, void* C, int ldc, const float* bias,
                          const void* R, int ldr, float* splitk_ws, size_t splitk_ws_bytes, int flags, void* stream);

/* Explicit im2col for the thin first convolutions (encoder conv_in 3->128, decoder conv_in 16->512):
 * x is a strided [Cin, T_all, H_all, W_all] tensor (element strides sC,sT,sH,sW); the crop starts at frame t0,
 * row y0, column x0 and spans T x H x W; zero padding at the crop border; causal front = the two frames before t0
 * (first_chunk != 0: the crop's first frame replicated).  A bf16 [T*H*W, Kpad], column ((dt*3+dh)*3+dw)*Cin + c.
 * One workgroup per output row stages the 9*Cin source rows in LDS: 9*Cin*(W+2) elements (+ the K-offset table) must fit the CU's 160 KiB
 * (AETHER_ERR_SHAPE otherwise; the VAE launch plan reports such a geometry already from aether_vae_workspace_bytes). */
int aether_im2col_first(const void* x, long sC, long sT, long sH, long sW, int Cin, int t0, int first_chunk, int y0, int x0,
                        int T, int H, int W, void* A, int Kpad, void* stream);

/* GroupNorm statistics of x [NB, V, C] -> stats fp32 [NB, G, 2] = (mean, rstd) and the folded per-channel affine
 * table affine fp32 [NB, 2, C] (scale = rstd*gamma, shift = beta - mean*rstd*gamma).  Deterministic: per-block
 * per-group partial sums in double (partial_ws: NB*nblk*2*C floats of 16-byte aligned scratch, used as [NB, nblk, G, 2]
 * doubles; needs C >= 2 G) merged in a fixed order in double by a second launch. */
int aether_groupnorm_stats(const void* x, int NB, int V, int C, int G, float eps, const float* gamma, const float* beta,
                           float* partial_ws, int nblk, float* stats, float* affine, void* stream);

/* CogVideoXSpatialNorm3D conditioning at LATENT resolution (nearest up-sampling commutes with a 1x1x1 conv):
 * cond fp32 [NB, zV, 2, C]: [.,.,0,:] = conv_y(zq), [.,.,1,:] = conv_b(zq); zq bf16 [NB, zV, zC] channels-last;
 * wy,wb fp32 [C, zC]; by,bb fp32 [C]. */
int aether_spatial_cond(const void* zq, int NB, int zV, int zC, int C, const float* wy, const float* by, const float* wb,
                        const float* bb, float* cond, void* stream);

/* y = [silu]( (x*scale+shift) [* cond_y + cond_b] ) written at interior offset (pt,ph,pw) of the zero-bordered
 * volume y [NB, oT, oH, oW, C].  cond != NULL selects SpatialNorm3D: cond fp32 [NB, zT, zH, zW, 2, C] from
 * aether_spatial_cond, gathered at the nearest latent voxel (tmap_host[t] = source latent frame of frame t, host
 * array of T <= 16 ints; H % zH == 0; W / zW a power of two).  C/8 must be a power of two. */
int aether_groupnorm_apply(const void* x, int NB, int T, int H, int W, int C, const float* affine, int silu_flag, void* y,
                           int oT, int oH, int oW, int pt, int ph, int pw, const float* cond, int zT, int zH, int zW,
                           const int* tmap_host, void* stream);

/* The same with the causal front of CogVideoXCausalConv3d's input written by the same launch (the conv that follows pads two
 * frames in front: autoencoder_kl_cogvideox.py CogVideoXCausalConv3d.fake_context_parallel_forward / conv_cache, reached from
 * P:557-618 and P:931,936): y is [NB, T+2, oH, oW, C]; frames 0, 1 <- front_prev (the last two padded frames of the previous
 * frame chunk, [NB, 2, oH, oW, C]) or, when front_prev is NULL (first chunk), copies of normalised frame 0; the last two frames
 * of y are also stored to front_next (same shape) for the next chunk.  Only interiors of the caches are read or written.
 * Replaces aether_groupnorm_apply + aether_causal_front (which stays for callers that pad a volume they did not normalise). */
int aether_groupnorm_apply_causal(const void* x, int NB, int T, int H, int W, int C, const float* affine, int silu_flag, void* y,
                                  int oH, int oW, int ph, int pw, const float* cond, int zT, int zH, int zW, const int* tmap_host,
                                  const void* front_prev, void* front_next, void* stream);

/* Causal front of a convolution input vol [NB, Tp, Hp, Wp, C] whose frames 2..Tp-1 are already written (CogVideoXCausalConv3d:
 * two frames of temporal context in front, from the previous chunk's `conv_cache` or by repeating the first frame):
 * frames 0,1 := prev [NB, 2, Hp, Wp, C], or copies of frame 2 when prev is NULL;  next [NB, 2, Hp, Wp, C] := the last two
 * frames of vol after that fill (the cache for the following chunk).  frame_elems = Hp*Wp*C.  One launch. */
int aether_causal_front(void* vol, int NB, int Tp, long frame_elems, const void* prev, void* next, void* stream);

/* Resample x [NB,T,H,W,C] into a zero-bordered volume y [NB,oT,oH,oW,C] at interior offset (pt,ph,pw).
 * mode 0 copy; 1 temporal avg-pool k2 s2 (first frame kept when T is odd; CogVideoXDownsample3D);
 * 2 nearest x2 in space; 3 nearest x2 in space and time with the first-frame rule of CogVideoXUpsample3D. */
int aether_resample_pad(const void* x, int NB, int T, int H, int W, int C, int mode, void* y, int oT, int oH, int oW, int pt,
                        int ph, int pw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input preprocessing on the device: the array branch of `_preprocess_image` (P:451-460) fused into one pass — uint8 / 255,
 * imcrop_center's centred window (preprocess_utils.py:4-39, zero fill outside the frame), nearest resize with PyTorch's index
 * rule (VideoProcessor.preprocess -> F.interpolate(size=...)), 2x - 1, NHWC -> NCHW, one rounding to bf16.
 * src [N, Hs, Ws, C] uint8 (is_u8 != 0) or float32 on the device; (top, left, ch, cw) = the crop window in source pixels
 * (aether_amd/preprocess.py::crop_window); out bf16 [N, C, H, W].
 * ------------------------------------------------------------------------------------------------ */
int aether_preprocess_frames(const void* src, int is_u8, int N, int Hs, int Ws, int C, int top, int left, int ch, int cw, int H, int W,
                             void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sliding-window merge on the device: the per-pixel part of blend_and_merge_window_results (scripts/demo.py:254-422) and of
 * compute_scale / project (aether/utils/postprocess_utils.py:847-864, 393-403).  float64 arithmetic like the reference's numpy.
 * ------------------------------------------------------------------------------------------------ */
/* out3[0] = sum m*p*t, out3[1] = sum m*p*p (m = p > 0.1; fp32 products, float64 deterministic accumulation) over n pixels of the
 * overlap, out3[2] = their ratio (0 when no pixel passes the mask).  pred fp32 = the new window's disparity, target float64 =
 * the merged disparity so far.  scratch: >= 2 doubles per partial block (4096 doubles are plenty).  D:295-301. */
int aether_merge_scale_fit(const float* pred, const double* target, long n, double* scratch, int scratch_doubles, double* out3, void* stream);
/* One window [n_win, hw(,3)] fp32 into the merged float64 arrays (pointers at the window's first frame): the first `ov` frames are
 * cross-faded with what is there (weights fade_host[f] for the merged frame, 1 - fade_host[f] for the window; host array of ov
 * doubles = np.linspace(1, 0, ov)), the others are written; disparity is first multiplied by the device scalar *scale_dev
 * (NULL: 1) as an fp32 product.  D:303-326. */
int aether_merge_window(const float* w_rgb, const float* w_disp, double* rgb, double* disp, int n_win, int ov, long hw,
                        const double* fade_host, const double* scale_dev, void* stream);
/* World-space point map of N frames: out[n,v,u,:] = P[n][:3,:3] · (Kinv[n] · (u+.5, v+.5, 1) / clip(disp, 1e-8, 1e8)) + P[n][:3,3];
 * Kinv float64 [N,3,3], P float64 [N,3,4] (camera-to-world), disp float64 [N,H,W], out float64 [N,H,W,3].  U:393-403. */
int aether_backproject(const double* disp, const double* Kinv, const double* P, double* out, int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-VAE entries (one call = AutoencoderKLCogVideoX.encode at P:557-618 / .decode through decode_latents at P:931,936,
 * with tiling and frame batching exactly as scripts/demo.py:229-230 enables them).  The launch plan (tile batches of equal
 * shape, frame chunks with threaded causal-conv caches, ResNets, resamplers, tile cross-fade) is C++ inside the library: pure
 * enqueue on `stream`, no allocation, no synchronisation, no host<->device copy -> graph capturable.
 * ------------------------------------------------------------------------------------------------ */
typedef struct AetherVaeConfig {
    int in_channels;                 /* 3 */
    int out_channels;                /* 3 */
    int latent_channels;             /* 16 (<= 16) */
    int layers_per_block;            /* 3 */
    int num_levels;                  /* 4 = len(block_out_channels); channel widths come with the registered weights */
    int norm_num_groups;             /* 32 */
    int temporal_compression_ratio;  /* 4 */
    int sample_height, sample_width; /* 480, 720: tiles are half of it, overlaps 1/6 and 1/5 (diffusers) */
    float norm_eps;                  /* 1e-6 */
    float tap_reuse_max_waste;       /* padded-plane / output-plane ratio up to which the tap-reuse convolution runs (1.06) */
    int flags;                       /* AETHER_GEMM_* flags forwarded to the GEMMs | AETHER_VAE_TWO_LANES */
} AetherVaeConfig;
#define AETHER_VAE_TWO_LANES 256 /* flags bit 8: the batches of equally shaped spatial tiles of one encode / decode (independent of each other until the
                                    cross-fade) are enqueued on TWO streams — the caller's and one owned by the handle (high priority, forked from / joined to
                                    the caller's stream by events; capturable) — assigned by tile area (480x720: the four full tiles | the five edge tiles), so
                                    that the small launches of one lane (512-channel levels at latent resolution, GroupNorm statistics, split-K finalizes)
                                    fill the gaps of the other.  Same kernels, same batches, same arithmetic: bit-identical to the one-lane plan.           */

typedef struct AetherVae AetherVae; /* opaque host-side handle: weight table + workspace bookkeeping */

AetherVae* aether_vae_create(const AetherVaeConfig* cfg);
void aether_vae_destroy(AetherVae* h);
/* Register a packed convolution under its diffusers module path ("encoder.conv_in", "decoder.up_blocks.0.resnets.1.conv2",
 * "encoder.down_blocks.0.downsamplers.0", "...conv_shortcut"): w bf16 [cout_pad, kcols] in the K order of
 * aether_conv_gemm_bf16 (blocked != 0: (dt, dh, 64-channel block, dw)) or, for the thin first convolutions, (dt, dh, dw, cin)
 * zero-padded to kcols; b fp32 [cout_pad].  Packing: aether_amd/vae.py `_Conv`. */
int aether_vae_set_conv(AetherVae* h, const char* name, const void* w, const float* b, int cout, int cout_pad, int cin, int kt, int kh,
                        int kw, int kcols, int blocked);
/* Register a GroupNorm ("encoder...norm1": gamma/beta fp32 [C], the other four NULL) or a CogVideoXSpatialNorm3D
 * ("decoder...norm1": norm_layer gamma/beta + conv_y / conv_b as fp32 [C, zC] matrices and [C] biases). */
int aether_vae_set_norm(AetherVae* h, const char* name, const float* gamma, const float* beta, const float* wy, const float* by,
                        const float* wb, const float* bb);
/* Workspace the NEXT call of this kind needs, given what the handle already keeps in the caller's current workspace (the
 * zero-bordered convolution input volumes persist there from call to call; a different workspace pointer resets them). */
size_t aether_vae_workspace_bytes(AetherVae* h, int decode, int T, int H, int W, int tiling);
int aether_vae_output_shape(AetherVae* h, int decode, int T, int H, int W, int* oC, int* oT, int* oH, int* oW);
/* x bf16 [in_channels, T, H, W] contiguous in [-1, 1] -> moments bf16 [2*latent_channels, T', H/8, W/8] (mean | logvar: the
 * caller samples the posterior, P:233-245).  workspace: 256-byte aligned device memory of >= aether_vae_workspace_bytes. */
int aether_vae_encode(AetherVae* h, const void* x, int T, int H, int W, int tiling, void* moments, void* workspace, size_t workspace_bytes,
                      void* stream);
/* z bf16 [latent_channels, T, h, w] contiguous (already divided by the scaling factor, P:931) -> sample bf16 [out_channels, T_out, 8h, 8w]. */
int aether_vae_decode(AetherVae* h, const void* z, int T, int H, int W, int tiling, void* sample, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-transformer entry (one call = CogVideoXTransformer3DModel.forward as invoked at P:865-875)
 * ------------------------------------------------------------------------------------------------ */
typedef struct AetherDitConfig {
    int num_layers;         /* 42 */
    int num_heads;          /* 48 */
    int head_dim;           /* 64 (only 64 is implemented) */
    int in_channels;        /* 96 */
    int out_channels;       /* 56 */
    int patch_size;         /* 2 */
    int text_dim;           /* 4096 */
    int time_embed_dim;     /* 512 */
    int ff_mult;            /* 4 */
    int max_text_len;       /* 226 */
    float norm_eps;         /* 1e-5 */
    float qk_norm_eps;      /* 1e-6 */
    int use_pos_embedding;  /* add pos_embedding [text+video, D] after patch embed */
    int flags;              /* AETHER_GEMM_* flags forwarded to the GEMMs */
} AetherDitConfig;

typedef struct AetherDit AetherDit; /* opaque host-side handle: weight table + launch plan */

AetherDit* aether_dit_create(const AetherDitConfig* cfg);
void aether_dit_destroy(AetherDit* h);
/* Register a device weight by its diffusers state-dict-derived name (see aether_amd/transformer.py for the
 * packing: fused qkv, concatenated AdaLN linears, fp32 biases/norm params).  Pointer must stay valid. */
int aether_dit_set_weight(AetherDit* h, const char* name, const void* dev_ptr);
/* Positional table added to the patch/text embedding when cfg.use_pos_embedding: bf16 [rows, D], text rows first
 * (diffusers CogVideoXPatchEmbed: the learned `pos_embedding` when the clip has `sample_frames` frames, otherwise the
 * 3-D sin-cos table of the actual size — the caller picks, aether_amd/transformer.py).  aether_dit_forward refuses a
 * table with fewer rows than text + video tokens (no out-of-bounds read).  (NULL, 0) unregisters. */
int aether_dit_set_pos_embedding(AetherDit* h, const void* table, int rows);
/* Replace the flags given at creation (e.g. AETHER_ATTN_EXACT_MAX for a measurement).  aether_dit_create / aether_dit_set_flags accept
 * AETHER_GEMM_WIDE_STORE | AETHER_ATTN_EXACT_MAX, aether_vae_create AETHER_GEMM_WIDE_STORE | AETHER_VAE_TWO_LANES; any other bit is
 * AETHER_ERR_ARG (create: NULL + aether_last_error), so flags of kernel variants that no longer exist are refused, not ignored. */
int aether_dit_set_flags(AetherDit* h, int flags);
/* Bytes of scratch the forward needs for batch B and a latent grid F x H x W (latent pixels). */
size_t aether_dit_workspace_bytes(const AetherDit* h, int B, int F, int H, int W);
/* hidden bf16 [B,F,in_channels,H,W]; text bf16 [B,max_text_len,text_dim]; timesteps fp32 [B] (device);
 * rope cos/sin fp32 [F*(H/p)*(W/p), 64]; out bf16 [B,F,out_channels,H,W]; workspace >= workspace_bytes. */
int aether_dit_forward(AetherDit* h, const void* hidden, const void* text, const float* timesteps,
                       const float* rope_cos, const float* rope_sin, void* out, int B, int F, int H, int W,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Optional per-kernel-class timing of aether_dit_forward: hipEvents are recorded on the launch stream around
 * every enqueue while enabled (do not enable during hipGraph capture).  aether_dit_get_profile synchronises on
 * the recorded events, ADDS UP milliseconds and launch counts per class since the last call, and resets. */
#define AETHER_PROF_OTHER 0     /* embeddings, timestep/AdaLN GEMVs, final norms, proj_out, (un)patchify */
#define AETHER_PROF_LN 1        /* LayerNorm + modulation                                                */
#define AETHER_PROF_GEMM_QKV 2
#define AETHER_PROF_QKROPE 3    /* q/k LayerNorm + RoPE + V transpose                                    */
#define AETHER_PROF_ATTN 4      /* flash attention                                                       */
#define AETHER_PROF_GEMM_O 5
#define AETHER_PROF_GEMM_FF1 6
#define AETHER_PROF_GEMM_FF2 7
#define AETHER_PROF_NUM 8
int aether_dit_set_profile(AetherDit* h, int enable);
int aether_dit_get_profile(AetherDit* h, float* ms_per_class /*[AETHER_PROF_NUM]*/, int* launches_per_class);

#ifdef __cplusplus
}
#endif

/* internal helpers shared by the translation units of the library (not part of the ABI) */
#ifdef __cplusplus
extern "C" int aether_set_error(int code, const char* msg);
extern "C" int aether_check_launch(const char* what);
#endif

#endif /* AETHER_HIP_H */
