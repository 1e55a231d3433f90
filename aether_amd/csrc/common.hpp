// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of aether_amd.
// Wavefront = 64 lanes everywhere in this tree; nothing here is portable to 32-wide hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aether {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define AE_DEV __device__ __forceinline__

AE_DEV float bf16_bits_to_f32(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950); matches torch's .to(bfloat16)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
AE_DEV unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
AE_DEV unsigned pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// 16-byte async copy HBM -> LDS. The LDS destination is wave-uniform `lds_wave_base` + lane*16;
// the global source is per lane (swizzles go on the SOURCE address, never on the destination).
AE_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same through a buffer descriptor: address = base(rsrc, wave-uniform SGPRs) + voff (per lane, bytes) + soff (wave-uniform,
// bytes).  Only `soff` changes from tile to tile, so the per-tile address arithmetic is scalar; bytes at or beyond the
// descriptor's size are not fetched (no clamping, no fault): the destination then holds zeros or stale data — finite either
// way once the slot has been written once — and callers mask such elements.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
AE_DEV buf_rsrc_t make_buf_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);   // raw buffer, stride 0, 32-bit data format
}
AE_DEV void bglds16(buf_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int N>
AE_DEV void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
AE_DEV void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// retire this wave's LDS-DMA (vmcnt) and LDS reads (lgkmcnt) before meeting the other waves: after the barrier
// the buffer just read may be overwritten by another wave's DMA, and the buffer just filled may be read.
AE_DEV void drain_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
AE_DEV void block_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

AE_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
AE_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// tanh-approximated GELU exactly as torch.nn.functional.gelu(x, approximate="tanh") evaluates it in fp32
// (0.5·(1+tanh(u)) == sigmoid(2u); evaluated through v_exp_f32 / v_rcp_f32 instead of a libm tanh)
AE_DEV float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f;  // sqrt(2/pi)
    const float k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float e = __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

AE_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }

// XCD-aware, bijective remap of the flat workgroup id (block b runs on XCD b % 8 on MI355X; speed only).
AE_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

}  // namespace aether

// dynamic LDS of aether_im2col_first (three source rows of every (dt, dh) tap + the K-offset table): shared by the kernel's entry point and the
// VAE launch plan, which refuses an over-wide untiled first layer when the workspace is sized, not in the middle of a run
constexpr size_t AETHER_IM2COL_LDS_LIMIT = 160 * 1024;
inline size_t aether_im2col_first_lds_bytes(int Cin, int W, int Kpad) {
    return ((((size_t)9 * Cin * (W + 2)) + 7) & ~(size_t)7) * 2 + (size_t)Kpad * 4;
}
