// Cooperative staging of [pixels][P] fp32 maps between HBM and LDS for the part-path kernels.
//
// The maps of the part path are NHWC with P (3 ... 64, typically 10 / 25) floats per pixel.  A lane-per-part layout makes
// every wave instruction move only (64 / GP) * P * 4 bytes (160 B at P = 10): far too few bytes in flight to cover the HBM
// latency (the round-1 kernels sat at 14-32 % of the HBM roof).  Here a block moves a whole tile of consecutive pixels with
// 16-byte loads / stores, 4 per thread in flight, into an LDS image with an ODD pixel pitch PP = P | 1 (conflict-free for
// the lane-per-part reads that follow); compute then runs out of LDS in whatever lane layout suits the reductions.
#pragma once
#include "common.h"

__device__ __forceinline__ int tile_pitch(int P) { return P | 1; }

// HBM [count][P] (contiguous, src = first pixel) -> lds[px * PP + c].  Optional `add` (same shape) is summed in while the
// tile is staged and optional `echo` receives the staged values (l = mean + eps goes back out with the same 16-byte index).
__device__ inline void tile_load_f32(const float* __restrict__ src, int count, int P, int PP, float* __restrict__ lds,
                                     const float* __restrict__ add = nullptr, float* __restrict__ echo = nullptr) {
    const int nfl = count * P;
    const int tid = threadIdx.x, nt = blockDim.x;
    int done = 0;
    if (((((unsigned long long)src) | ((unsigned long long)add) | ((unsigned long long)echo)) & 15ull) == 0) {
        const int nv = nfl >> 2;
        const float4* __restrict__ s4 = (const float4*)src;
        const float4* __restrict__ a4 = (const float4*)add;
        float4* __restrict__ e4 = (float4*)echo;
        for (int i0 = 0; i0 < nv; i0 += 4 * nt) {
            float4 u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * nt + tid;
                u[k] = i < nv ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (a4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * nt + tid;
                    if (i < nv) { const float4 a = a4[i]; u[k].x += a.x; u[k].y += a.y; u[k].z += a.z; u[k].w += a.w; }
                }
            }
            if (e4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * nt + tid;
                    if (i < nv) e4[i] = u[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * nt + tid;
                if (i < nv) {
                    const int e = 4 * i;
                    int px = e / P, c = e - px * P;
                    const float w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        lds[px * PP + c] = w[j];
                        if (++c == P) { c = 0; ++px; }
                    }
                }
            }
        }
        done = nv << 2;
    }
    for (int e = done + tid; e < nfl; e += nt) {
        const int px = e / P, c = e - px * P;
        float v = src[e];
        if (add) v += add[e];
        if (echo) echo[e] = v;
        lds[px * PP + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-phase staging (round 4).  tile_load_f32 requests a map's pieces and waits for them before it scatters them into LDS: a
// block that stages five maps pays five dependent HBM round trips (~2.5 us each) before its first pixel is computed -- that, not
// bytes or arithmetic, is what held the prior / moment / un-pool kernels at 12-34 % of the HBM roof.  Here a block first REQUESTS
// the 16-byte pieces of every map it needs (K per thread and map, all in flight at once), then commits them to LDS; inside a tile
// loop the next tile is requested before the current one is computed.
template <int K>
struct TileReq {
    float4 u[K];
    int nv;                  // 16-byte pieces of the tile (0: unaligned source, everything goes through the scalar tail)
};

template <int K>
__device__ __forceinline__ void tile_request(TileReq<K>& r, const float* __restrict__ src, int count, int P) {
    const int nfl = count * P;
    r.nv = ((((unsigned long long)src) & 15ull) == 0 && src != nullptr) ? (nfl >> 2) : 0;
    const float4* __restrict__ s4 = (const float4*)src;
    const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = k * nt + tid;
        r.u[k] = i < r.nv ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ void tile_scatter4(float* __restrict__ lds, int i, int P, int PP, const float4& v) {
    const int e = 4 * i;
    int px = e / P, c = e - px * P;
    const float w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lds[px * PP + c] = w[j];
        if (++c == P) { c = 0; ++px; }
    }
}

// pieces [0, K * blockDim) come out of the request registers; whatever the tile has beyond them (large P), its last < 4 floats
// and an unaligned source are fetched here
template <int K>
__device__ __forceinline__ void tile_commit(const TileReq<K>& r, const float* __restrict__ src, int count, int P, int PP,
                                            float* __restrict__ lds) {
    const int nfl = count * P;
    const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = k * nt + tid;
        if (i < r.nv) tile_scatter4(lds, i, P, PP, r.u[k]);
    }
    const float4* __restrict__ s4 = (const float4*)src;
    for (int i = K * nt + tid; i < r.nv; i += nt) tile_scatter4(lds, i, P, PP, s4[i]);
    for (int e = (r.nv << 2) + tid; e < nfl; e += nt) {
        const int px = e / P, c = e - px * P;
        lds[px * PP + c] = src[e];
    }
}

// lds[px * PP + c] -> HBM [count][P]; `f` maps the staged value to the stored one
struct TileIdent { __device__ float operator()(float v) const { return v; } };
template <typename F = TileIdent>
__device__ inline void tile_store_f32(float* __restrict__ dst, int count, int P, int PP, const float* __restrict__ lds, F f = F()) {
    const int nfl = count * P;
    const int tid = threadIdx.x, nt = blockDim.x;
    int done = 0;
    if ((((unsigned long long)dst) & 15ull) == 0) {
        const int nv = nfl >> 2;
        float4* __restrict__ d4 = (float4*)dst;
        for (int i = tid; i < nv; i += nt) {
            const int e = 4 * i;
            int px = e / P, c = e - px * P;
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = f(lds[px * PP + c]);
                if (++c == P) { c = 0; ++px; }
            }
            d4[i] = make_float4(w[0], w[1], w[2], w[3]);
        }
        done = nv << 2;
    }
    for (int e = done + tid; e < nfl; e += nt) {
        const int px = e / P, c = e - px * P;
        dst[e] = f(lds[px * PP + c]);
    }
}

// pixels per tile so that `maps` LDS images of pitch PP stay within `budget` bytes (multiple of 64, at most 256)
static inline int tile_pixels(int P, int maps, int budget = 40 * 1024) {
    int t = 256;
    while (t > 64 && (size_t)t * (P | 1) * 4 * maps > (size_t)budget) t >>= 1;
    return t;
}
