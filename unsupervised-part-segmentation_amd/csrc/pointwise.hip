// HBM-bound NHWC streaming kernels: legacy-TF bilinear x2, activation + global mean, 2x2 max pool,
// channel copies, the perceptual-loss pre-processing and L1 reductions, dtype conversion.
// All of them move 16 bytes per lane (8 bf16 / 4 f32) with lanes along the channel axis -> fully coalesced.
#include "common.h"

namespace {

template <typename T> struct V16 {
    static constexpr int N = Chunk<T>::N;
    __device__ static inline void ld(const T* p, float* f) { uint4 u = *(const uint4*)p; Chunk<T>::unpack(u, f); }
    __device__ static inline void st(T* p, const float* f) { *(uint4*)p = Chunk<T>::pack(f); }
};

// ------------------------------------------------------------------ bilinear x2 (cub/code/nn.py:834-847; Appendix A.2)
// out[2i] = in[i]; out[2i+1] = 0.5*(in[i] + in[min(i+1, n-1)])  (rows first, then columns, like the oracle)
// One thread per INPUT pixel chunk: the four neighbours it needs are loaded once and its 2x2 block of outputs is written
// (one load per output chunk instead of 2.25; same operation order as before: rows first, then columns).
// fp8 copy of a stored bf16 chunk for the consuming convolution (ups_conv_desc.in_f8): act -> running max -> * scale -> 8 bytes
struct F8Emit {
    unsigned char* out;      // NULL: record the maximum only
    const float* scale; float* amax;
    int act, e5m2; float slope;
};
// u = the 8 bf16 values as they are stored (the copy is the quantisation of the STORED tensor): unpacked with a shift / a mask per
// pair, activation and scale on packed pairs (v_pk_mul_f32), the running maximum two values at a time (v_max3_f32)
__device__ __forceinline__ void f8_emit_words(const F8Emit& q, float sc, float ns, long long elem, uint4 u, float& amax) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    ups_f32x2 f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (ups_f32x2){__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
        if (q.act != UPS_ACT_NONE) {
            const ups_f32x2 sx = f[k] * ns;
            f[k] = (ups_f32x2){ups_vmax(f[k][0], sx[0]), ups_vmax(f[k][1], sx[1])};
        }
        amax = fmaxf(fmaxf(amax, fabsf(f[k][0])), fabsf(f[k][1]));
    }
    if (q.out) {
        int d0 = 0, d1 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = f[k] * sc;
        if (q.e5m2) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                f[k] = (ups_f32x2){__builtin_amdgcn_fmed3f(f[k][0], -57344.f, 57344.f), __builtin_amdgcn_fmed3f(f[k][1], -57344.f, 57344.f)};
            d0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0][0], f[0][1], d0, false); d0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[1][0], f[1][1], d0, true);
            d1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2][0], f[2][1], d1, false); d1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[3][0], f[3][1], d1, true);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                f[k] = (ups_f32x2){__builtin_amdgcn_fmed3f(f[k][0], -448.f, 448.f), __builtin_amdgcn_fmed3f(f[k][1], -448.f, 448.f)};
            d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0][0], f[0][1], d0, false); d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[1][0], f[1][1], d0, true);
            d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2][0], f[2][1], d1, false); d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[3][0], f[3][1], d1, true);
        }
        *(uint2*)(q.out + elem) = make_uint2((unsigned)d0, (unsigned)d1);
    }
}
// store the chunk as T and (EMIT instances: T = bf16) hand the stored words to the fp8 copy
template <typename T, bool EMIT>
__device__ __forceinline__ void st_emit(T* ptr, const float* v, const F8Emit& q, float sc, float ns, long long elem, float& amax) {
    if constexpr (EMIT) {
        static_assert(sizeof(T) == 2, "fp8 copies accompany bf16 tensors");
        const uint4 u = Chunk<T>::pack(v);
        *(uint4*)ptr = u;
        f8_emit_words(q, sc, ns, elem, u, amax);
    } else {
        V16<T>::st(ptr, v);
    }
}
__device__ __forceinline__ void f8_emit_finish(const F8Emit& q, float amax) {
    const float m = wave_max(amax);
    if ((threadIdx.x & 63) == 0) ups_amax_slot(q.amax + (blockIdx.x & 63), m);
}

// (chunk k, column, row, image) of a flat index over [n][h][w][cc]: shifts when all three extents are powers of two (the shipped
// shapes), one 32-bit division each otherwise -- the 64-bit % and / by run-time values that stood here were ~360 of the ~450
// instructions a thread spent per 64 output bytes: the up-sampling kernels ran on the vector ALU, not on HBM (0.59 of the roof).
struct Idx4 { int k, x, y, b; };
struct IdxDec {
    unsigned cc, w, h;
    int lcc, lw, lh;            // log2 when a power of two, else -1
};
__host__ __device__ inline int ups_log2_exact(unsigned v) {
    if (v == 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1u << l) < v) ++l;
    return l;
}
__device__ __forceinline__ Idx4 idx_decode(const IdxDec& d, long long idx) {
    Idx4 r;
    if (d.lcc >= 0 && d.lw >= 0 && d.lh >= 0) {
        const unsigned long long u = (unsigned long long)idx;
        r.k = (int)((unsigned)u & (d.cc - 1));
        const unsigned long long t = u >> d.lcc;
        r.x = (int)((unsigned)t & (d.w - 1));
        const unsigned long long t2 = t >> d.lw;
        r.y = (int)((unsigned)t2 & (d.h - 1));
        r.b = (int)(t2 >> d.lh);
    } else if (idx < (1ll << 31)) {
        const unsigned u = (unsigned)idx;
        const unsigned t = u / d.cc; r.k = (int)(u - t * d.cc);
        const unsigned t2 = t / d.w; r.x = (int)(t - t2 * d.w);
        const unsigned t3 = t2 / d.h; r.y = (int)(t2 - t3 * d.h);
        r.b = (int)t3;
    } else {
        r.k = (int)(idx % d.cc);
        long long t = idx / d.cc;
        r.x = (int)(t % d.w); t /= d.w;
        r.y = (int)(t % d.h);
        r.b = (int)(t / d.h);
    }
    return r;
}

template <typename T, bool EMIT = false>
// ons >= 0: the stored value is max(v, ons * v) (post-activation storage of the up-sampled tensor, ups_bilinear2x_fwd_act); the
// interpolation itself always runs on the un-activated values
__global__ void bilinear2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, F8Emit q = F8Emit(),
                                      float ons = -1.f, unsigned char* __restrict__ sbits = nullptr) {
    // sbits (16-bit T): the sign byte of every stored 8-element chunk (ups_conv_desc.sign_out layout), for the input gradient of the
    // convolution that consumes the up-sampled tensor
    auto stv = [&](T* ptr, const float* v) __attribute__((always_inline)) {
        float t[V16<T>::N];
#pragma unroll
        for (int e = 0; e < V16<T>::N; ++e) t[e] = ons >= 0.f ? ups_vmax(v[e], ons * v[e]) : v[e];
        if constexpr (sizeof(T) == 2) {
            const uint4 u = Chunk<T>::pack(t);
            *(uint4*)ptr = u;
            if (sbits) {
                // (the sign of the STORED 16-bit value: a tiny positive fp32 that rounds to zero must not count as positive)
                sbits[(ptr - y) >> 3] = (unsigned char)ups_sign_byte(u);
            }
        } else {
            V16<T>::st(ptr, t);
        }
    };
    constexpr int E = V16<T>::N;
    float amax = 0.f;
    const float sc = (EMIT && q.out) ? *q.scale : 1.f, ns = EMIT ? ups_slope_eff(q.act, q.slope) : 0.f;
    const int cc = c / E;
    const long long total = (long long)n * h * w * cc;
    const IdxDec dec = {(unsigned)cc, (unsigned)w, (unsigned)h, ups_log2_exact((unsigned)cc), ups_log2_exact((unsigned)w), ups_log2_exact((unsigned)h)};
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const Idx4 id = idx_decode(dec, idx);
        const int k = id.k, x0 = id.x, y0 = id.y, b = id.b;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const T* base = x + (long long)b * h * w * c + k * E;
        float a00[E], a01[E], a10[E], a11[E], o[E];
        V16<T>::ld(base + ((long long)y0 * w + x0) * c, a00);
        V16<T>::ld(base + ((long long)y0 * w + x1) * c, a01);
        V16<T>::ld(base + ((long long)y1 * w + x0) * c, a10);
        V16<T>::ld(base + ((long long)y1 * w + x1) * c, a11);
        T* ob = y + (((long long)b * 2 * h + 2 * y0) * (2 * w) + 2 * x0) * c + k * E;
        const long long orow = (long long)2 * w * c;
        if constexpr (EMIT) st_emit<T, true>(ob, a00, q, sc, ns, ob - y, amax); else stv(ob, a00);               // (2y, 2x)
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = 0.5f * (a00[e] + a01[e]);
        if constexpr (EMIT) st_emit<T, true>(ob + c, o, q, sc, ns, ob + c - y, amax); else stv(ob + c, o);       // (2y, 2x+1)
#pragma unroll
        for (int e = 0; e < E; ++e) { a00[e] = 0.5f * (a00[e] + a10[e]); a01[e] = 0.5f * (a01[e] + a11[e]); }
        if constexpr (EMIT) st_emit<T, true>(ob + orow, a00, q, sc, ns, ob + orow - y, amax); else stv(ob + orow, a00);   // (2y+1, 2x)
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = 0.5f * (a00[e] + a01[e]);
        if constexpr (EMIT) st_emit<T, true>(ob + orow + c, o, q, sc, ns, ob + orow + c - y, amax); else stv(ob + orow + c, o);   // (2y+1, 2x+1)
    }
    if constexpr (EMIT) f8_emit_finish(q, amax);
}

// gather form of the transpose: 1-D weights of output o on input i: o=2i ->1, o=2i+1 ->.5 (+.5 if i==n-1), o=2i-1 ->.5
__device__ inline float up_w(int o, int i, int n) {
    if (o == 2 * i) return 1.f;
    if (o == 2 * i + 1) return (i == n - 1) ? 1.f : 0.5f;
    return 0.5f;  // o == 2i-1
}

template <typename T, bool EMIT = false>
__global__ void bilinear2x_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int n, int h, int w, int c, F8Emit q = F8Emit()) {
    constexpr int E = V16<T>::N;
    float amax = 0.f;
    const float sc = (EMIT && q.out) ? *q.scale : 1.f;
    const int cc = c / E;
    const long long total = (long long)n * h * w * cc;
    const IdxDec dec = {(unsigned)cc, (unsigned)w, (unsigned)h, ups_log2_exact((unsigned)cc), ups_log2_exact((unsigned)w), ups_log2_exact((unsigned)h)};
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const Idx4 id = idx_decode(dec, idx);
        const int k = id.k, ix = id.x, iy = id.y, b = id.b;
        const T* base = gy + (long long)b * 4 * h * w * c + k * E;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        for (int oy = max(2 * iy - 1, 0); oy <= 2 * iy + 1; ++oy) {
            const float wy = up_w(oy, iy, h);
            for (int ox = max(2 * ix - 1, 0); ox <= 2 * ix + 1; ++ox) {
                const float ww = wy * up_w(ox, ix, w);
                float g[E];
                V16<T>::ld(base + ((long long)oy * 2 * w + ox) * c, g);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += ww * g[e];
            }
        }
        st_emit<T, EMIT>(gx + idx * E, acc, q, sc, 0.f, idx * E, amax);
    }
    if constexpr (EMIT) f8_emit_finish(q, amax);
}

// ------------------------------------------------------------------ activate + global mean (model.py:50-51)
template <typename T>
__global__ void act_mean_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int hw, int c, int act, float slope) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c) return;
    const int b = idx / c, ch = idx - b * c;
    float s = 0.f;
    for (int p = 0; p < hw; ++p) s += ups_act(ld_as_float<T>(x + ((long long)b * hw + p) * c + ch), act, slope);
    st_from_float<T>(y + idx, s / (float)hw);
}
template <typename T>
__global__ void act_mean_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, int n, int hw,
                                    int c, int act, float slope) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * hw * c) return;
    const int ch = (int)(idx % c);
    const int b = (int)(idx / ((long long)hw * c));
    const float g = ld_as_float<T>(gy + (long long)b * c + ch) / (float)hw;
    st_from_float<T>(gx + idx, g * ups_dact(ld_as_float<T>(x + idx), act, slope));
}

// ------------------------------------------------------------------ sign bytes of a stored 16-bit tensor (ups_conv_desc.sign_out)
__global__ void sign_pack_kernel(const uint4* __restrict__ x, unsigned char* __restrict__ bits, long long chunks) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
        const uint4 u = x[i];
        bits[i] = (unsigned char)ups_sign_byte(u);
    }
}

// ------------------------------------------------------------------ elu (nn.py:747-758): the one activation that is materialised
template <typename T>
__global__ void elu_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ out, long long chunks) {
    constexpr int E = V16<T>::N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
        float a[E], g[E];
        V16<T>::ld(x + i * E, a);
        if (gy) {
            V16<T>::ld(gy + i * E, g);
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = g[e] * ups_dact(a[e], UPS_ACT_ELU, 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = ups_act(a[e], UPS_ACT_ELU, 0.f);
        }
        V16<T>::st(out + i * E, a);
    }
}

// ------------------------------------------------------------------ 2x2 max pool
// EMIT (bf16): the pooled tensor also leaves as the e4m3 copy its consuming convolution stages (ups_conv_desc.in_f8; round 5: the
// first convolution of every block of the perceptual trunk reads a pooled map, and the chain of copies through a block starts here)
template <typename T, bool EMIT = false>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, F8Emit q8 = F8Emit()) {
    constexpr int E = V16<T>::N;
    float amax = 0.f;
    const float sc = (EMIT && q8.out) ? *q8.scale : 1.f, ns = EMIT ? ups_slope_eff(q8.act, q8.slope) : 0.f;
    const int cc = c / E, ho = h / 2, wo = w / 2;
    const long long total = (long long)n * ho * wo * cc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        long long t = idx / cc;
        const int ox = (int)(t % wo); t /= wo;
        const int oy = (int)(t % ho);
        const int b = (int)(t / ho);
        const T* base = x + (((long long)b * h + 2 * oy) * w + 2 * ox) * c + k * E;
        float a[E], q[E];
        V16<T>::ld(base, a);
        V16<T>::ld(base + c, q);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], q[e]);
        V16<T>::ld(base + (long long)w * c, q);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], q[e]);
        V16<T>::ld(base + (long long)w * c + c, q);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], q[e]);
        st_emit<T, EMIT>(y + idx * E, a, q8, sc, ns, idx * E, amax);
    }
    if constexpr (EMIT) f8_emit_finish(q8, amax);
}
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, int n, int h,
                                    int w, int c) {
    constexpr int E = V16<T>::N;
    const int cc = c / E, ho = h / 2, wo = w / 2;
    const long long total = (long long)n * ho * wo * cc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        long long t = idx / cc;
        const int ox = (int)(t % wo); t /= wo;
        const int oy = (int)(t % ho);
        const int b = (int)(t / ho);
        const long long o00 = (((long long)b * h + 2 * oy) * w + 2 * ox) * c + k * E;
        const long long offs[4] = {o00, o00 + c, o00 + (long long)w * c, o00 + (long long)w * c + c};
        float v[4][E], g[E], m[E];
#pragma unroll
        for (int q = 0; q < 4; ++q) V16<T>::ld(x + offs[q], v[q]);
        V16<T>::ld(gy + idx * E, g);
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = fmaxf(fmaxf(v[0][e], v[1][e]), fmaxf(v[2][e], v[3][e]));
        bool taken[E];
#pragma unroll
        for (int e = 0; e < E; ++e) taken[e] = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool hit = !taken[e] && v[q][e] == m[e];
                o[e] = hit ? g[e] : 0.f;
                taken[e] = taken[e] || hit;
            }
            V16<T>::st(gx + offs[q], o);
        }
    }
}

// ------------------------------------------------------------------ channel copy / add between tensors of different widths
template <typename T, bool ADD>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, long long rows, int c) {
    constexpr int E = V16<T>::N;
    const int cc = c / E;
    const long long total = rows * cc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        const long long r = idx / cc;
        float a[E];
        V16<T>::ld(src + r * lds + k * E, a);
        if (ADD) {
            float d[E];
            V16<T>::ld(dst + r * ldd + k * E, d);
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] += d[e];
        }
        V16<T>::st(dst + r * ldd + k * E, a);
    }
}

// ------------------------------------------------------------------ the other up-sampling methods of nn.upsample (N:820-849)
// "subpixel": tf.depth_to_space(x, 2) of a [n,h,w,4C] convolution output: y[b, 2i+di, 2j+dj, c] = x[b, i, j, (2 di + dj) C + c].
// C need not be a multiple of 8 (physical widths ldx / ldy are): element-wise, one thread per element of the RESULT; pad channels
// are written as zero.  bwd = the inverse gather (space_to_depth of the gradient).
template <typename T>
__global__ void depth_to_space_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int C, int ldx, int ldy, int bwd) {
    const long long total = bwd ? (long long)n * h * w * ldx : (long long)n * 4 * h * w * ldy;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (!bwd) {                       // dst = y [n, 2h, 2w, ldy]
            const int c = (int)(idx % ldy);
            long long r = idx / ldy;
            const int x2 = (int)(r % (2 * w)); r /= 2 * w;
            const int y2 = (int)(r % (2 * h));
            const int b = (int)(r / (2 * h));
            if (c < C) v = ld_as_float<T>(src + (((long long)b * h + (y2 >> 1)) * w + (x2 >> 1)) * ldx + ((y2 & 1) * 2 + (x2 & 1)) * C + c);
        } else {                          // dst = gx [n, h, w, ldx], src = gy [n, 2h, 2w, ldy]
            const int k = (int)(idx % ldx);
            long long r = idx / ldx;
            const int j = (int)(r % w); r /= w;
            const int i = (int)(r % h);
            const int b = (int)(r / h);
            if (k < 4 * C) {
                const int q = k / C, c = k - q * C;
                v = ld_as_float<T>(src + (((long long)b * 2 * h + 2 * i + (q >> 1)) * (2 * w) + 2 * j + (q & 1)) * ldy + c);
            }
        }
        st_from_float<T>(dst + idx, v);
    }
}
// "nearest_neighbor": y[b, 2i+di, 2j+dj, :] = x[b, i, j, :]; bwd: gx = the sum of the four.  One thread per 16-byte chunk of x / gx.
template <typename T>
__global__ void nearest2x_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int c, int bwd) {
    constexpr int E = V16<T>::N;
    const int cc = c / E;
    const long long total = (long long)n * h * w * cc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        long long r = idx / cc;
        const int j = (int)(r % w); r /= w;
        const int i = (int)(r % h);
        const int b = (int)(r / h);
        const long long big = (((long long)b * 2 * h + 2 * i) * (2 * w) + 2 * j) * c + k * E;        // element (2i, 2j) of the 2x tensor
        if (!bwd) {
            const uint4 u = *(const uint4*)(src + idx * E);
            *(uint4*)(dst + big) = u; *(uint4*)(dst + big + c) = u;
            *(uint4*)(dst + big + (long long)2 * w * c) = u; *(uint4*)(dst + big + (long long)2 * w * c + c) = u;
        } else {
            float a[E], t[E];
            V16<T>::ld(src + big, a);
            V16<T>::ld(src + big + c, t);
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] += t[e];
            V16<T>::ld(src + big + (long long)2 * w * c, t);
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] += t[e];
            V16<T>::ld(src + big + (long long)2 * w * c + c, t);
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] += t[e];
            V16<T>::st(dst + idx * E, a);
        }
    }
}

// ------------------------------------------------------------------ crop window (perceptual_input: resize256_crop224)
// fwd: y[n, i, j, :] = x[n, oy + i, ox + j, :] for the ho x wo window whose corner (oy, ox) is READ ON THE DEVICE (one random
// window per step for the whole batch, tf.random_crop; a device scalar keeps the launch valid inside a captured HIP graph);
// bwd: gx = gy inside the window, zero elsewhere.  One thread per 16-byte chunk of the result.
template <typename T>
__global__ void crop_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int c, int ho, int wo,
                            const int* __restrict__ yx, int bwd) {
    constexpr int E = V16<T>::N;
    const int cc = c / E;
    const int oy = min(max(yx[0], 0), h - ho), ox = min(max(yx[1], 0), w - wo);
    const int rh = bwd ? h : ho, rw = bwd ? w : wo;            // shape of the result
    const long long total = (long long)n * rh * rw * cc;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        long long r = idx / cc;
        const int j = (int)(r % rw); r /= rw;
        const int i = (int)(r % rh);
        const int img = (int)(r / rh);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (!bwd) {
            v = *(const uint4*)(src + (((long long)img * h + oy + i) * w + ox + j) * c + k * E);
        } else if ((unsigned)(i - oy) < (unsigned)ho && (unsigned)(j - ox) < (unsigned)wo) {
            v = *(const uint4*)(src + (((long long)img * ho + i - oy) * wo + j - ox) * c + k * E);
        }
        *(uint4*)(dst + idx * E) = v;
    }
}

// ------------------------------------------------------------------ VGG pre-processing (edflow VGG19Features, UNVERIFIED)
// y[pix] = {b*127.5+127.5-103.939, g*..-116.779, r*..-123.68, 0,0,0,0,0}
template <typename T, typename TX>
__global__ void vgg_pre_fwd_kernel(const TX* __restrict__ x, int ldx, T* __restrict__ y, long long pixels) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pixels) return;
    const float r = ld_as_float<TX>(x + p * ldx), g = ld_as_float<TX>(x + p * ldx + 1), b = ld_as_float<TX>(x + p * ldx + 2);
    T* o = y + p * 8;
    st_from_float<T>(o + 0, (b + 1.f) * 127.5f - 103.939f);
    st_from_float<T>(o + 1, (g + 1.f) * 127.5f - 116.779f);
    st_from_float<T>(o + 2, (r + 1.f) * 127.5f - 123.68f);
#pragma unroll
    for (int e = 3; e < 8; ++e) st_from_float<T>(o + e, 0.f);
}
template <typename T>
__global__ void vgg_pre_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int ldgx, long long pixels) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pixels) return;
    const T* g = gy + p * 8;
    T* o = gx + p * ldgx;
    st_from_float<T>(o + 0, 127.5f * ld_as_float<T>(g + 2));
    st_from_float<T>(o + 1, 127.5f * ld_as_float<T>(g + 1));
    st_from_float<T>(o + 2, 127.5f * ld_as_float<T>(g + 0));
    for (int e = 3; e < ldgx; ++e) st_from_float<T>(o + e, 0.f);
}

// ------------------------------------------------------------------ L1 feature distance (sum |act(a)-act(b)|)
template <typename T>
__global__ __launch_bounds__(256) void l1_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, long long rows,
                                                     int c, int ld, int act, float* __restrict__ partial) {
    __shared__ float red[4];
    constexpr int E = V16<T>::N;
    const int cc = ld / E;
    const long long total = rows * cc;
    float s = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        float x[E], y[E];
        V16<T>::ld(a + idx * E, x);
        V16<T>::ld(b + idx * E, y);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (k * E + e < c) s += fabsf(ups_act(x[e], act, 0.f) - ups_act(y[e], act, 0.f));
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
template <typename T>
__global__ void l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ gb, long long rows, int c,
                              int ld, int act, const float* __restrict__ scale_dev, float scale) {
    constexpr int E = V16<T>::N;
    const int cc = ld / E;
    const long long total = rows * cc;
    const float sc = scale * (scale_dev ? scale_dev[0] : 1.f);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cc);
        float x[E], y[E], g[E];
        V16<T>::ld(a + idx * E, x);
        V16<T>::ld(b + idx * E, y);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float d = ups_act(x[e], act, 0.f) - ups_act(y[e], act, 0.f);
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            g[e] = (k * E + e < c) ? -sc * sg * ups_dact(y[e], act, 0.f) : 0.f;
        }
        V16<T>::st(gb + idx * E, g);
    }
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ partial, int n, float scale,
                                                        float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * s;
}

template <typename TS, typename TD>
__global__ void convert_kernel(const TS* __restrict__ s, TD* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        st_from_float<TD>(d + i, ld_as_float<TS>(s + i));
}

template <typename T>
__global__ void pad_convert_kernel(const float* __restrict__ s, int c, T* __restrict__ d, int ldd, long long rows) {
    const long long total = rows * ldd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldd);
        const long long r = i / ldd;
        st_from_float<T>(d + i, k < c ? s[r * c + k] : 0.f);
    }
}

inline int grid_for(long long work, int cap = 16384) {
    long long g = (work + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define UPS_DISPATCH(dtype, KERNEL, grid, s, ...)                                                         \
    do {                                                                                                  \
        if ((dtype) == UPS_F32) hipLaunchKernelGGL(KERNEL<float>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else if ((dtype) == UPS_BF16) hipLaunchKernelGGL(KERNEL<bf16>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else if ((dtype) == UPS_F16) hipLaunchKernelGGL(KERNEL<f16>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else { ups_set_error("bad dtype %d", (int)(dtype)); return UPS_E_ARG; }                           \
        UPS_LAUNCH_CHECK();                                                                               \
    } while (0)

extern "C" int ups_bilinear2x_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    UPS_CHECK_ARG(x && y && c % 8 == 0);
    const long long work = (long long)n * h * w * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(bilinear2x_fwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w, c);
    else if (dtype == UPS_F16) hipLaunchKernelGGL(bilinear2x_fwd_kernel<f16>, dim3(grid_for(work)), dim3(256), 0, s, (const f16*)x, (f16*)y, n, h, w, c);
    else hipLaunchKernelGGL(bilinear2x_fwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_bilinear2x_fwd_act(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act,
                                      float slope, void* stream) {
    UPS_CHECK_ARG(x && y && c % 8 == 0 && slope >= 0.f && slope <= 1.f && act >= UPS_ACT_NONE && act <= UPS_ACT_RELU);
    const long long work = (long long)n * h * w * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    const float ons = act == UPS_ACT_NONE ? -1.f : (act == UPS_ACT_LRELU ? slope : 0.f);
    if (dtype == UPS_F32) hipLaunchKernelGGL(bilinear2x_fwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w, c, F8Emit(), ons);
    else if (dtype == UPS_F16) hipLaunchKernelGGL(bilinear2x_fwd_kernel<f16>, dim3(grid_for(work)), dim3(256), 0, s, (const f16*)x, (f16*)y, n, h, w, c, F8Emit(), ons);
    else hipLaunchKernelGGL(bilinear2x_fwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w, c, F8Emit(), ons);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_bilinear2x_bwd(const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    UPS_CHECK_ARG(gy && gx && c % 8 == 0);
    const long long work = (long long)n * h * w * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(bilinear2x_bwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)gy, (float*)gx, n, h, w, c);
    else hipLaunchKernelGGL(bilinear2x_bwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)gy, (bf16*)gx, n, h, w, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_bilinear2x_fwd_bits(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act,
                                       float slope, void* sign_bits, void* stream) {
    UPS_CHECK_ARG(x && y && sign_bits && c % 8 == 0 && slope >= 0.f && slope <= 1.f && act >= UPS_ACT_NONE && act <= UPS_ACT_RELU);
    UPS_CHECK_ARG(dtype == UPS_BF16 || dtype == UPS_F16);
    const long long work = (long long)n * h * w * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    const float ons = act == UPS_ACT_NONE ? -1.f : (act == UPS_ACT_LRELU ? slope : 0.f);
    if (dtype == UPS_F16) hipLaunchKernelGGL(bilinear2x_fwd_kernel<f16>, dim3(grid_for(work)), dim3(256), 0, s, (const f16*)x, (f16*)y, n, h, w, c, F8Emit(), ons, (unsigned char*)sign_bits);
    else hipLaunchKernelGGL(bilinear2x_fwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w, c, F8Emit(), ons, (unsigned char*)sign_bits);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_sign_pack(const void* x, int32_t dtype, int64_t chunks, void* sign_bits, void* stream) {
    UPS_CHECK_ARG(x && sign_bits && chunks > 0 && (dtype == UPS_BF16 || dtype == UPS_F16) && ((uintptr_t)x & 15) == 0);
    hipLaunchKernelGGL(sign_pack_kernel, dim3(grid_for(chunks)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (unsigned char*)sign_bits,
                       (long long)chunks);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
// bf16 bilinear x2 that also hands its output to an fp8 convolution: max |act(y)| into amax[64]; with y_f8 != NULL the e4m3
// (e5m2 != 0: e5m2) bytes of act(y) * *scale next to y (ups_conv_desc.in_f8 of the consumer)
extern "C" int ups_bilinear2x_fwd_f8(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* y_f8,
                                     const float* scale, float* amax, int32_t act, float slope, int32_t e5m2, void* stream) {
    UPS_CHECK_ARG(x && y && amax && c % 8 == 0 && (!y_f8 || scale) && slope >= 0.f && slope <= 1.f);
    const long long work = (long long)n * h * w * (c / 8);
    F8Emit q; q.out = (unsigned char*)y_f8; q.scale = scale; q.amax = amax; q.act = act; q.e5m2 = e5m2; q.slope = slope;
    hipLaunchKernelGGL((bilinear2x_fwd_kernel<bf16, true>), dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (bf16*)y, n, h, w, c, q);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_bilinear2x_bwd_f8(const void* gy, void* gx, int32_t n, int32_t h, int32_t w, int32_t c, void* gx_f8,
                                     const float* scale, float* amax, int32_t e5m2, void* stream) {
    UPS_CHECK_ARG(gy && gx && amax && c % 8 == 0 && (!gx_f8 || scale));
    const long long work = (long long)n * h * w * (c / 8);
    F8Emit q; q.out = (unsigned char*)gx_f8; q.scale = scale; q.amax = amax; q.act = UPS_ACT_NONE; q.e5m2 = e5m2; q.slope = 0.f;
    hipLaunchKernelGGL((bilinear2x_bwd_kernel<bf16, true>), dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const bf16*)gy,
                       (bf16*)gx, n, h, w, c, q);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_depth_to_space(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t C, int32_t ldx,
                                  int32_t ldy, int32_t bwd, void* stream) {
    UPS_CHECK_ARG(src && dst && n > 0 && h > 0 && w > 0 && C > 0 && 4 * C <= ldx && C <= ldy);
    const long long work = bwd ? (long long)n * h * w * ldx : (long long)n * 4 * h * w * ldy;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(depth_to_space_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)src, (float*)dst, n, h, w, C, ldx, ldy, bwd);
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(depth_to_space_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n, h, w, C, ldx, ldy, bwd);
    else if (dtype == UPS_F16) hipLaunchKernelGGL(depth_to_space_kernel<f16>, dim3(grid_for(work)), dim3(256), 0, s, (const f16*)src, (f16*)dst, n, h, w, C, ldx, ldy, bwd);
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_nearest2x(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bwd, void* stream) {
    UPS_CHECK_ARG(src && dst && n > 0 && h > 0 && w > 0 && c % 8 == 0);
    const long long work = (long long)n * h * w * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(nearest2x_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)src, (float*)dst, n, h, w, c, bwd);
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(nearest2x_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n, h, w, c, bwd);
    else if (dtype == UPS_F16 && !bwd) hipLaunchKernelGGL(nearest2x_kernel<f16>, dim3(grid_for(work)), dim3(256), 0, s, (const f16*)src, (f16*)dst, n, h, w, c, bwd);
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
static int crop_launch(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo,
                       const int32_t* yx_dev, int bwd, void* stream) {
    UPS_CHECK_ARG(src && dst && yx_dev && n > 0 && ho > 0 && wo > 0 && ho <= h && wo <= w && c % 8 == 0);
    const long long work = (long long)n * (bwd ? h : ho) * (bwd ? w : wo) * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(crop_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)src, (float*)dst, n, h, w, c, ho, wo, yx_dev, bwd);
    else if (dtype == UPS_BF16 || dtype == UPS_F16)
        hipLaunchKernelGGL(crop_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n, h, w, c, ho, wo, yx_dev, bwd);
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_crop_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo,
                            const int32_t* yx_dev, void* stream) {
    return crop_launch(x, y, dtype, n, h, w, c, ho, wo, yx_dev, 0, stream);
}
extern "C" int ups_crop_bwd(const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ho, int32_t wo,
                            const int32_t* yx_dev, void* stream) {
    return crop_launch(gy, gx, dtype, n, h, w, c, ho, wo, yx_dev, 1, stream);
}
extern "C" int ups_act_mean_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t hw, int32_t c, int32_t act, float slope, void* stream) {
    UPS_CHECK_ARG(x && y);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ups_cdiv((long long)n * c, 256);
    if (dtype == UPS_F32) hipLaunchKernelGGL(act_mean_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n, hw, c, act, slope);
    else hipLaunchKernelGGL(act_mean_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, hw, c, act, slope);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_act_mean_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int32_t n, int32_t hw, int32_t c, int32_t act, float slope, void* stream) {
    UPS_CHECK_ARG(x && gy && gx);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ups_cdiv((long long)n * hw * c, 256);
    if (dtype == UPS_F32) hipLaunchKernelGGL(act_mean_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)gy, (float*)gx, n, hw, c, act, slope);
    else hipLaunchKernelGGL(act_mean_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (const bf16*)gy, (bf16*)gx, n, hw, c, act, slope);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
static int elu_launch(const void* x, const void* gy, void* out, int32_t dtype, int64_t n, void* stream) {
    UPS_CHECK_ARG(x && out && n > 0 && n % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(elu_kernel<float>, dim3(grid_for(n / 4)), dim3(256), 0, s, (const float*)x, (const float*)gy, (float*)out, (long long)(n / 4));
    else if (dtype == UPS_F16) hipLaunchKernelGGL(elu_kernel<f16>, dim3(grid_for(n / 8)), dim3(256), 0, s, (const f16*)x, (const f16*)gy, (f16*)out, (long long)(n / 8));
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(elu_kernel<bf16>, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16*)x, (const bf16*)gy, (bf16*)out, (long long)(n / 8));
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_elu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream) { return elu_launch(x, nullptr, y, dtype, n, stream); }
extern "C" int ups_elu_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int64_t n, void* stream) {
    UPS_CHECK_ARG(gy != nullptr);
    return elu_launch(x, gy, gx, dtype, n, stream);
}
extern "C" int ups_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    UPS_CHECK_ARG(x && y && c % 8 == 0 && h % 2 == 0 && w % 2 == 0);
    const long long work = (long long)n * (h / 2) * (w / 2) * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w, c);
    else hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
// bf16 2x2 max pool that also hands its output to an fp8 convolution: max |act(y)| into amax[64]; with y_f8 != NULL the e4m3 bytes
// of act(y) * *scale next to y (as ups_bilinear2x_fwd_f8)
extern "C" int ups_maxpool2_fwd_f8(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* y_f8, const float* scale,
                                   float* amax, int32_t act, float slope, void* stream) {
    UPS_CHECK_ARG(x && y && amax && c % 8 == 0 && h % 2 == 0 && w % 2 == 0 && (!y_f8 || scale) && slope >= 0.f && slope <= 1.f);
    const long long work = (long long)n * (h / 2) * (w / 2) * (c / 8);
    F8Emit q; q.out = (unsigned char*)y_f8; q.scale = scale; q.amax = amax; q.act = act; q.e5m2 = 0; q.slope = slope;
    hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16, true>), dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y,
                       n, h, w, c, q);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_maxpool2_bwd(const void* x, const void* gy, void* gx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    UPS_CHECK_ARG(x && gy && gx && c % 8 == 0 && h % 2 == 0 && w % 2 == 0);
    const long long work = (long long)n * (h / 2) * (w / 2) * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)x, (const float*)gy, (float*)gx, n, h, w, c);
    else hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)x, (const bf16*)gy, (bf16*)gx, n, h, w, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_copy_channels(const void* src, int32_t lds, void* dst, int32_t ldd, int32_t dtype, int64_t rows, int32_t c, void* stream) {
    UPS_CHECK_ARG(src && dst && c % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0);
    const long long work = (long long)rows * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL((copy_channels_kernel<float, false>), dim3(grid_for(work)), dim3(256), 0, s, (const float*)src, lds, (float*)dst, ldd, (long long)rows, c);
    else hipLaunchKernelGGL((copy_channels_kernel<bf16, false>), dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)src, lds, (bf16*)dst, ldd, (long long)rows, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_add_channels(const void* src, int32_t lds, void* dst, int32_t ldd, int32_t dtype, int64_t rows, int32_t c, void* stream) {
    UPS_CHECK_ARG(src && dst && c % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0);
    const long long work = (long long)rows * (c / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL((copy_channels_kernel<float, true>), dim3(grid_for(work)), dim3(256), 0, s, (const float*)src, lds, (float*)dst, ldd, (long long)rows, c);
    else hipLaunchKernelGGL((copy_channels_kernel<bf16, true>), dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)src, lds, (bf16*)dst, ldd, (long long)rows, c);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_vgg_preprocess_fwd(const void* x, int32_t x_is_f32, int32_t ldx, void* y, int32_t dtype, int64_t pixels, void* stream) {
    UPS_CHECK_ARG(x && y && ldx >= 3);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ups_cdiv(pixels, 256);
    if (dtype == UPS_F32) hipLaunchKernelGGL((vgg_pre_fwd_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)y, (long long)pixels);
    else if (x_is_f32) hipLaunchKernelGGL((vgg_pre_fwd_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (bf16*)y, (long long)pixels);
    else hipLaunchKernelGGL((vgg_pre_fwd_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)x, ldx, (bf16*)y, (long long)pixels);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_vgg_preprocess_bwd(const void* gy, void* gx, int32_t dtype, int32_t ldgx, int64_t pixels, void* stream) {
    UPS_CHECK_ARG(gy && gx && ldgx >= 3);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ups_cdiv(pixels, 256);
    if (dtype == UPS_F32) hipLaunchKernelGGL(vgg_pre_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)gy, (float*)gx, ldgx, (long long)pixels);
    else hipLaunchKernelGGL(vgg_pre_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)gy, (bf16*)gx, ldgx, (long long)pixels);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_l1_fwd(const void* a, const void* b, int32_t dtype, int64_t rows, int32_t c, int32_t ld, int32_t act, float* partial, int32_t nblocks, void* stream) {
    UPS_CHECK_ARG(a && b && partial && ld % 8 == 0 && c <= ld && nblocks >= 1);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(l1_fwd_kernel<float>, dim3(nblocks), dim3(256), 0, s, (const float*)a, (const float*)b, (long long)rows, c, ld, act, partial);
    else hipLaunchKernelGGL(l1_fwd_kernel<bf16>, dim3(nblocks), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (long long)rows, c, ld, act, partial);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_l1_bwd(const void* a, const void* b, void* gb, int32_t dtype, int64_t rows, int32_t c, int32_t ld, int32_t act, const float* scale_dev, float scale, void* stream) {
    UPS_CHECK_ARG(a && b && gb && ld % 8 == 0 && c <= ld);
    const long long work = (long long)rows * (ld / (dtype == UPS_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(l1_bwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)gb, (long long)rows, c, ld, act, scale_dev, scale);
    else hipLaunchKernelGGL(l1_bwd_kernel<bf16>, dim3(grid_for(work)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)gb, (long long)rows, c, ld, act, scale_dev, scale);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_sum_scale(const float* partial, int32_t n, float scale, float* out, int32_t accumulate, void* stream) {
    UPS_CHECK_ARG(partial && out && n >= 1);
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, scale, out, accumulate);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_convert(const void* src, int32_t sd, void* dst, int32_t dd, int64_t count, void* stream) {
    UPS_CHECK_ARG(src && dst && count >= 0);
    if (count == 0) return UPS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(count);
    if (sd == UPS_F32 && dd == UPS_BF16) hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16*)dst, (long long)count);
    else if (sd == UPS_BF16 && dd == UPS_F32) hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (float*)dst, (long long)count);
    else if (sd == UPS_F32 && dd == UPS_F32) hipLaunchKernelGGL((convert_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, (long long)count);
    else if (sd == UPS_F32 && dd == UPS_F16) hipLaunchKernelGGL((convert_kernel<float, f16>), dim3(grid), dim3(256), 0, s, (const float*)src, (f16*)dst, (long long)count);
    else if (sd == UPS_F16 && dd == UPS_F32) hipLaunchKernelGGL((convert_kernel<f16, float>), dim3(grid), dim3(256), 0, s, (const f16*)src, (float*)dst, (long long)count);
    else if (sd == UPS_BF16 && dd == UPS_BF16) hipLaunchKernelGGL((convert_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, (long long)count);
    else { ups_set_error("ups_convert: bad dtypes"); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_pad_convert(const float* src, int32_t c, void* dst, int32_t dtype, int32_t ldd, int64_t rows, void* stream) {
    UPS_CHECK_ARG(src && dst && ldd >= c && rows >= 0);
    if (rows == 0) return UPS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((long long)rows * ldd);
    if (dtype == UPS_F32) hipLaunchKernelGGL(pad_convert_kernel<float>, dim3(grid), dim3(256), 0, s, src, c, (float*)dst, ldd, (long long)rows);
    else if (dtype == UPS_F16) hipLaunchKernelGGL(pad_convert_kernel<f16>, dim3(grid), dim3(256), 0, s, src, c, (f16*)dst, ldd, (long long)rows);
    else hipLaunchKernelGGL(pad_convert_kernel<bf16>, dim3(grid), dim3(256), 0, s, src, c, (bf16*)dst, ldd, (long long)rows);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

// ------------------------------------------------------------------ thin-plate-spline warp (cub/code/SB_model48i/model.py:282-311)
// The reference calls the un-vendored eddata.utils.tps ("adapted from CompVis/unsupervised-disentangling",
// train_cub_subset_tps.yaml:188): TPS spatial transformer, source position of every output pixel
//   (x_s, y_s) = T @ [1, x, y, phi_1 .. phi_K],  phi_i = d2 * log(d2 + 1e-6), d2 = |(x,y) - coord_i|^2   on linspace(-1,1),
// then the classic `_interpolate`: pixel = (coord + 1) * size / 2, corner indices clamped, weights from the clamped corners.
// One thread per output pixel; T and the control points of the image sit in LDS.  Semantics UNVERIFIED (parity unpinned).
namespace {
constexpr int TPS_MAXK = 32;
__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ img, const float* __restrict__ T,
                                                        const float* __restrict__ coord, float* __restrict__ out,
                                                        int h, int w, int c, int K) {
    __shared__ float sT[2 * (TPS_MAXK + 3)], sC[2 * TPS_MAXK];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * (K + 3); i += 256) sT[i] = T[(long long)n * 2 * (K + 3) + i];
    for (int i = threadIdx.x; i < 2 * K; i += 256) sC[i] = coord[(long long)n * 2 * K + i];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= h * w) return;
    const int yy = q / w, xx = q - yy * w;
    const float x = w > 1 ? -1.f + 2.f * (float)xx / (float)(w - 1) : -1.f;
    const float y = h > 1 ? -1.f + 2.f * (float)yy / (float)(h - 1) : -1.f;
    float xs = sT[0] + sT[1] * x + sT[2] * y;
    float ys = sT[K + 3] + sT[K + 4] * x + sT[K + 5] * y;
    for (int i = 0; i < K; ++i) {
        const float dx = x - sC[2 * i], dy = y - sC[2 * i + 1];
        const float d2 = dx * dx + dy * dy;
        const float r = d2 * logf(d2 + 1e-6f);
        xs += sT[3 + i] * r;
        ys += sT[K + 6 + i] * r;
    }
    const float px = (xs + 1.f) * (float)w * 0.5f, py = (ys + 1.f) * (float)h * 0.5f;
    const float x0 = floorf(px), y0 = floorf(py);
    const float x0c = fminf(fmaxf(x0, 0.f), (float)(w - 1)), x1c = fminf(fmaxf(x0 + 1.f, 0.f), (float)(w - 1));
    const float y0c = fminf(fmaxf(y0, 0.f), (float)(h - 1)), y1c = fminf(fmaxf(y0 + 1.f, 0.f), (float)(h - 1));
    const float wa = (x1c - px) * (y1c - py), wb = (x1c - px) * (py - y0c), wc = (px - x0c) * (y1c - py), wd = (px - x0c) * (py - y0c);
    const long long base = (long long)n * h * w;
    const float* pa = img + (base + (long long)y0c * w + (long long)x0c) * c;
    const float* pb = img + (base + (long long)y1c * w + (long long)x0c) * c;
    const float* pc = img + (base + (long long)y0c * w + (long long)x1c) * c;
    const float* pd = img + (base + (long long)y1c * w + (long long)x1c) * c;
    float* o = out + (base + q) * c;
    for (int k = 0; k < c; ++k) o[k] = wa * pa[k] + wb * pb[k] + wc * pc[k] + wd * pd[k];
}
}  // namespace

extern "C" int ups_tps_warp(const float* img, const float* T, const float* coord, float* out, int32_t n, int32_t h, int32_t w,
                            int32_t c, int32_t K, void* stream) {
    UPS_CHECK_ARG(img && T && coord && out && n > 0 && h > 0 && w > 0 && c > 0 && K >= 1 && K <= TPS_MAXK);
    hipLaunchKernelGGL(tps_warp_kernel, dim3(ups_cdiv((long long)h * w, 256), n), dim3(256), 0, (hipStream_t)stream, img, T, coord,
                       out, h, w, c, K);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
