// Shared device helpers for libupsparts_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/upsparts_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// IEEE half: the forward tensors of precision-critical scopes (the mask decoder) are stored and multiplied as fp16 -- 10
// mantissa bits instead of bf16's 7 at the same MFMA rate; gradients stay bf16 (range)
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

void ups_set_error(const char* fmt, ...);

// hipFuncSetAttribute is per DEVICE: the "already raised the LDS limit of this kernel" flag of a launcher must be too.  Drop-in
// for the `static bool done = false; if (!done) { ...; done = true; }` idiom: one bit per device ordinal of the calling thread's
// current device (a second GPU driven from the same process would otherwise launch with > 64 KB of dynamic LDS un-permitted).
// set by the kernel launchers of ups_conv_igemm that wrote ups_conv_desc.sign_out (conv_igemm.hip; ups_conv_sign_out_written())
extern thread_local int g_ups_sign_written;

struct UpsPerDevice {
    unsigned long long bits = 0;
    static int cur() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool operator!() const { return !((bits >> cur()) & 1ull); }
    UpsPerDevice& operator=(bool v) { if (v) bits |= 1ull << cur(); else bits &= ~(1ull << cur()); return *this; }
};

#define UPS_CHECK_ARG(cond)                                                        \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ups_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
            return UPS_E_ARG;                                                      \
        }                                                                          \
    } while (0)

#define UPS_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ups_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return UPS_E_LAUNCH;                                                   \
        }                                                                          \
    } while (0)

// 16-byte chunk of T
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int N = 4;
    __device__ static inline void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Chunk<bf16> {
    static constexpr int N = 8;
    __device__ static inline void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    __device__ static inline unsigned pk(float a, float b) {
        union { bf16 h[2]; unsigned u; } x;
        x.h[0] = (bf16)a; x.h[1] = (bf16)b;   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
        return x.u;
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
};

template <> struct Chunk<f16> {
    static constexpr int N = 8;
    __device__ static inline void unpack(const uint4& u, float* f) {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            union { unsigned u; f16x2 h; } x; x.u = w[i];
            f[2 * i] = (float)x.h[0]; f[2 * i + 1] = (float)x.h[1];
        }
    }
    __device__ static inline unsigned pk(float a, float b) {
        // RNE; SATURATING at +-65504 (v_med3_f32; NaN stays NaN): a residual stream or a restored checkpoint that leaves the
        // fp16 range clips instead of turning into inf / NaN that nothing notices before the next log step
        union { f16x2 h; unsigned u; } x;
        x.h[0] = (f16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); x.h[1] = (f16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
        return x.u;
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
};
// two 16-bit elements of a dword as floats, by storage type
template <typename T> __device__ __forceinline__ void ups_unpack2(unsigned w, float& lo, float& hi);
template <> __device__ __forceinline__ void ups_unpack2<bf16>(unsigned w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void ups_unpack2<f16>(unsigned w, float& lo, float& hi) {
    union { unsigned u; f16x2 h; } x; x.u = w;
    lo = (float)x.h[0]; hi = (float)x.h[1];
}

__device__ inline float ups_act(float x, int act, float slope) {
    if (act == UPS_ACT_LRELU) return x > 0.f ? x : slope * x;
    if (act == UPS_ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == UPS_ACT_ELU) return x > 0.f ? x : expm1f(x);
    return x;
}
// branch-free activation for the staging paths: leaky-relu(slope) and relu (slope_eff = 0) share one formula,
// max(x, slope * x) for 0 <= slope <= 1 (the launchers check the range): the same product and hence the same bits as
// x > 0 ? x : slope * x.  v_max_f32 is written out so that no canonicalising max(x, x) is put in front of it.
__device__ __forceinline__ float ups_vmax(float a, float b) {
#ifdef UPS_VMAX_BUILTIN         // (hazard hunt, tools/probes/rows_hazard.sh: the compiler-visible form)
    return __builtin_fmaxf(a, b);
#else
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ inline float ups_act_ns(float x, float slope_eff) { return ups_vmax(x, slope_eff * x); }
__device__ inline float ups_slope_eff(int act, float slope) { return act == UPS_ACT_LRELU ? slope : 0.f; }
typedef float ups_f32x2 __attribute__((ext_vector_type(2)));
// activation-on-load of a 16-byte chunk: 6 VALU per bf16 pair (unpack 2, v_pk_mul_f32, 2 max, v_cvt_pk_bf16_f32)
// TIGHT: leave the pack order to the compiler (its interleaved form costs 8 more VALU per chunk but peaks at fewer live
// registers: the 128-wide patch kernel at two blocks per CU has none to spare)
template <bool TIGHT = false>
__device__ __forceinline__ uint4 ups_act_chunk(uint4 u, float s, bf16*) {
    unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ups_f32x2 x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
        const ups_f32x2 sx = x * s;
        const float lo = ups_vmax(x[0], sx[0]), hi = ups_vmax(x[1], sx[1]);
        // written out: the vectoriser otherwise converts (lo, lo) / (hi, hi) pairs of two words and re-interleaves them
        if constexpr (TIGHT) w[i] = Chunk<bf16>::pk(lo, hi);
        else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(lo), "v"(hi));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// fp16: packed half math (v_pk_mul_f16 + v_pk_max_f16 per pair); slope * x is rounded to fp16 once, as a stored
// act(x) tensor would be
template <bool TIGHT = false>
__device__ __forceinline__ uint4 ups_act_chunk(uint4 u, float s, f16*) {
    unsigned w[4] = {u.x, u.y, u.z, u.w};
    const f16x2 sv = {(f16)s, (f16)s};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        union { unsigned u; f16x2 h; } x; x.u = w[i];
        const f16x2 sx = x.h * sv;
        x.h = __builtin_elementwise_max(x.h, sx);
        w[i] = x.u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// fp16 tensor -> activation -> bf16 chunk (weight-gradient staging of a layer whose forward input is stored as fp16: the
// gradient operand is bf16, and an MFMA takes one type)
__device__ __forceinline__ uint4 ups_act_chunk_f16_to_bf16(uint4 u, float s, bool act) {
    unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo, hi;
        ups_unpack2<f16>(w[i], lo, hi);
        if (act) { lo = ups_vmax(lo, s * lo); hi = ups_vmax(hi, s * hi); }
        w[i] = Chunk<bf16>::pk(lo, hi);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <bool TIGHT = false>
__device__ __forceinline__ uint4 ups_act_chunk(uint4 u, float s, float*) {
    float f[4];
    Chunk<float>::unpack(u, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = ups_act_ns(f[e], s);
    return Chunk<float>::pack(f);
}

__device__ inline float ups_dact(float x, int act, float slope) {
    if (act == UPS_ACT_LRELU) return x > 0.f ? 1.f : slope;
    if (act == UPS_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (act == UPS_ACT_ELU) return x > 0.f ? 1.f : expf(x);
    return 1.f;
}

template <typename T> __device__ inline float ld_as_float(const T* p);
template <> __device__ inline float ld_as_float<float>(const float* p) { return *p; }
template <> __device__ inline float ld_as_float<bf16>(const bf16* p) { return (float)*p; }
template <> __device__ inline float ld_as_float<f16>(const f16* p) { return (float)*p; }
template <typename T> __device__ inline void st_from_float(T* p, float v);
template <> __device__ inline void st_from_float<f16>(f16* p, float v) { *p = (f16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
template <> __device__ inline void st_from_float<float>(float* p, float v) { *p = v; }
template <> __device__ inline void st_from_float<bf16>(bf16* p, float v) { *p = (bf16)v; }

// ln(x) for x well above the denormal range (> 1e-37): the bare v_log_f32 (1 ulp on log2) times ln 2, two instructions.  hipcc expands
// __logf into the denormal-safe, extended-precision sequence (15 instructions) -- in the pixel-per-lane kernels, whose arguments are
// P * m + 1e-20 or a sum of exponentials >= 1, that was a third of the instruction stream.
__device__ __forceinline__ float ups_log_fast(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }

// Sign byte of eight stored 16-bit values (ups_conv_desc.sign_out): bit e = element e > 0, i.e. positive and non-zero as a 16-bit
// integer (bf16 and fp16 alike).
// (round 6, late: two packed 16-bit min / max per word -- clamp every half to {0, 1} -- and five bit operations, instead of eight
// compares, eight selects and their wait states: 14 instructions for 30)
typedef short ups_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ups_pos16(unsigned w) {       // per 16-bit half: 1 if the half is > 0 as a signed integer, else 0
    ups_s16x2 v;
    __builtin_memcpy(&v, &w, 4);
    v = __builtin_elementwise_max(__builtin_elementwise_min(v, (ups_s16x2){1, 1}), (ups_s16x2){0, 0});
    unsigned r;
    __builtin_memcpy(&r, &v, 4);
    return r;
}
__device__ __forceinline__ unsigned ups_sign_byte(const uint4& u) {
    // flags of the even elements (low halves) at bits 0, 2, 4, 6; of the odd elements (high halves) at bits 16, 18, 20, 22
    const unsigned b = ups_pos16(u.x) | (ups_pos16(u.y) << 2) | (ups_pos16(u.z) << 4) | (ups_pos16(u.w) << 6);
    return (b & 0x55u) | ((b >> 15) & 0xaau);
}

// Wave-wide reductions on the DPP data path: four row-local steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane of
// a 16-lane row then holds the row's result) and four v_readlane for the rows -- 12 VALU / SALU instructions, no LDS, result
// uniform.  (Until round 5 these were six __shfl_xor steps, which hipcc lowers to ds_bpermute_b32 + s_waitcnt each: the 70
// reductions at the end of a pixel-per-lane block -- 420 LDS round trips in a dependent chain -- cost more than the block's tiles.)
__device__ __forceinline__ float ups_dpp(float v, const int ctrl_sel) {
    const int i = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xf, 0xf, false); break;      // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xf, 0xf, false); break;      // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xf, 0xf, false); break;     // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xf, 0xf, false); break;    // row_mirror
    }
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float ups_row(float v, const int row) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row));
}
// (`_full`: EVERY lane of the wave must be active -- a DPP read of an inactive lane returns the reader's own value and v_readlane
// whatever the register holds; the shuffle forms below tolerate partial waves: ds_bpermute returns 0 for inactive lanes.)
__device__ inline float wave_sum_full(float v) {
    v += ups_dpp(v, 0); v += ups_dpp(v, 1); v += ups_dpp(v, 2); v += ups_dpp(v, 3);
    return (ups_row(v, 0) + ups_row(v, 1)) + (ups_row(v, 2) + ups_row(v, 3));
}
__device__ inline float wave_max_full(float v) {
    v = fmaxf(v, ups_dpp(v, 0)); v = fmaxf(v, ups_dpp(v, 1)); v = fmaxf(v, ups_dpp(v, 2)); v = fmaxf(v, ups_dpp(v, 3));
    return fmaxf(fmaxf(ups_row(v, 0), ups_row(v, 1)), fmaxf(ups_row(v, 2), ups_row(v, 3)));
}
__device__ inline int wave_sum_full_i(int v) {          // (all 64 lanes active, as wave_sum_full)
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for 256-thread blocks; result valid in thread 0 (and broadcast via smem[0])
__device__ inline float block_sum_256(float v, float* smem4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem4[wid] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// running maximum of non-negative floats in a global slot: the slot only grows, so a wave whose value does not exceed what it
// reads needs no atomic (a stale read only costs a redundant atomic) -- thousands of blocks otherwise serialise on 64 addresses
__device__ __forceinline__ void ups_amax_slot(float* slot, float m) {
    if (m > __builtin_nontemporal_load(slot)) atomicMax((unsigned*)slot, __float_as_uint(m));
}

static inline int ups_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
