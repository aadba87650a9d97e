// The part path between the mask decoder and the image decoder (cub/code/SB_model48i/model.py:414-484):
// per-pixel soft-max over parts + hard max, spatial soft-max moments -> rectangle centres, part-masked
// appearance images, part-feature un-pooling.  All HBM-bound.  Data moves between HBM and LDS in whole pixel tiles with
// 16-byte accesses (tile.h: enough bytes in flight to cover the HBM latency); compute runs out of LDS with lanes along the
// part / feature axis so the part reductions are wavefront shuffles.
#include <stdlib.h>
#include "common.h"
#include "tile.h"

namespace {

// ------------------------------------------------------------------ softmax_P + hard_max (nn.py:58-62, 134-136)
// One block = one tile of `tpx` consecutive pixels: l = mean (+ eps) is staged (and echoed to `l`) with 16-byte accesses,
// GP lanes per pixel reduce over the parts, the tile then holds +m / -m (sign = hard-max flag: m of a maximum is >= 1/P > 0)
// and goes out as the two maps m and hard.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
struct TileAbs { __device__ float operator()(float v) const { return fabsf(v); } };
struct TileSign { __device__ float operator()(float v) const { return (__float_as_uint(v) >> 31) ? 1.f : 0.f; } };

// thread = pixel: the P values of the pixel are walked in LDS (odd pixel pitch: conflict-free), no cross-lane traffic at all
// (a lane-per-part layout spends 16 ds_bpermute per 4 pixels on the max / sum / arg-max reductions: LDS-issue bound)
// mom (optional, [blocks][P][5] int32): per part the pixel count and the sums of iy, ix, iy^2, ix^2 over the tile's hard-mask
// pixels (integer: exact and order-independent, so the block's LDS atomics keep the result reproducible) -- the spatial soft-max
// moments of the ONE-HOT hard map have a closed form in these (hard_moments_finalize_kernel), which spares the separate
// pass over the 40-byte-per-pixel map that ups_spatial_moments(hard) was.
__global__ __launch_bounds__(256) void part_softmax_kernel(const float* __restrict__ mean, const float* __restrict__ eps,
                                                           float* __restrict__ l, float* __restrict__ m, float* __restrict__ hard,
                                                           long long* __restrict__ amax, unsigned* __restrict__ bits,
                                                           long long pixels, int P, int tpx, int* __restrict__ mom, int img_w,
                                                           int img_hw) {
    extern __shared__ __attribute__((aligned(16))) float ts[];          // [tpx][PP] (+ [P][5] ints when mom)
    const int PP = tile_pitch(P);
    const long long pix0 = (long long)blockIdx.x * tpx;
    const int cnt = (int)min((long long)tpx, pixels - pix0);
    int* red = (int*)(ts + (size_t)tpx * PP);
    if (mom) { for (int i = threadIdx.x; i < P * 5; i += 256) red[i] = 0; }
    tile_load_f32(mean + pix0 * P, cnt, P, PP, ts, eps ? eps + pix0 * P : nullptr, (eps && l) ? l + pix0 * P : nullptr);
    __syncthreads();
    for (int px = threadIdx.x; px < cnt; px += 256) {
        float* row = ts + px * PP;
        float mx = -INFINITY;
        for (int c = 0; c < P; ++c) mx = fmaxf(mx, row[c]);
        float sm = 0.f;
        for (int c = 0; c < P; ++c) { const float e = expf(row[c] - mx); row[c] = e; sm += e; }
        float mm = -1.f;
        for (int c = 0; c < P; ++c) { const float pm = row[c] / sm; row[c] = pm; mm = fmaxf(mm, pm); }
        int first = -1;
        unsigned bm = 0u;
        for (int c = 0; c < P; ++c) {
            const float pm = row[c];
            if (pm == mm) {
                if (first < 0) first = c;
                if (c < 32) bm |= 1u << c;
                row[c] = -pm;                        // sign = hard-max flag
            }
        }
        if (amax) amax[pix0 + px] = first;
        if (bits) bits[pix0 + px] = bm;
        if (mom) {
            const int q = (int)((pix0 + px) % img_hw), iy = q / img_w, ix = q - iy * img_w;
            for (unsigned b = bm; b; b &= b - 1) {
                int* r = red + 5 * (__ffs(b) - 1);
                atomicAdd(r, 1); atomicAdd(r + 1, iy); atomicAdd(r + 2, ix); atomicAdd(r + 3, iy * iy); atomicAdd(r + 4, ix * ix);
            }
        }
    }
    __syncthreads();
    if (mom) { for (int i = threadIdx.x; i < P * 5; i += 256) mom[(long long)blockIdx.x * P * 5 + i] = red[i]; }
    tile_store_f32(m + pix0 * P, cnt, P, PP, ts, TileAbs());
    if (hard) tile_store_f32(hard + pix0 * P, cnt, P, PP, ts, TileSign());
}

// Round 5: the same pass pixel-per-lane for P = 10 at 128- / 256-wide maps (the benchmark's [2B, H, W, 10] logits: 419 MB of traffic,
// 0.51 of the HBM roof in the LDS-walking form above).  mean and eps tiles arrive linearly by LDS-DMA (two slots), a thread owns one
// pixel with its ten parts in registers (exp / rcp instead of expf and ten divisions), l / m / hard leave through a linear LDS tile
// with 16-byte stores, arg-max and bit set straight from registers.  Hard-mask moments: a wave's 64 pixels lie in one image row, so
// per part present in the wave the pixel count is a popcount, the iy sums are scalar products and only ix, ix^2 take a DPP sum;
// lane 0 adds the five integers into the block's LDS record (exact, order-independent), one record per block.
template <int P>
__global__ __launch_bounds__(256, 2) void part_softmax_px_kernel(const float* __restrict__ mean, const float* __restrict__ eps,
                                                                  float* __restrict__ l, float* __restrict__ m, float* __restrict__ hard,
                                                                  long long* __restrict__ amax, unsigned* __restrict__ bits,
                                                                  const int tiles_img, const int tiles_per_block, int* __restrict__ mom,
                                                                  const int img_w) {
    static_assert(P % 2 == 0 && ((P / 2) & 1) == 1, "conflict-free 8-byte LDS accesses need P / 2 odd");
    constexpr int TB = 256 * P * 4, PCS = TB / 1024, PW = 2 * PCS / 4;
    static_assert(TB % 1024 == 0 && (2 * PCS) % 4 == 0, "piece counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* OUT = smem + 2 * 2 * TB;          // (l, m, hard) of the tile
    int* red = (int*)(smem + 2 * 2 * TB + 3 * TB);   // [P][5]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpi = tiles_img / tiles_per_block;
    const int n = blockIdx.x / bpi, bi = blockIdx.x - n * bpi;
    const int t_begin = bi * tiles_per_block, t_end = t_begin + tiles_per_block;
    const long long img = (long long)n * tiles_img * 256;                 // first pixel of the image
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* g0 = (const unsigned char*)(mean + img * P);
    const unsigned char* g1 = (const unsigned char*)((eps ? eps : mean) + img * P);
    const unsigned voff = (unsigned)lane * 16u;
    const bool has_eps = eps != nullptr;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const unsigned base = smem_lds + (unsigned)(((t - t_begin) & 1) * 2 * TB);
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int q = wid + 4 * k;
            const unsigned char* src = (q < PCS ? g0 : g1) + (long long)t * TB + (q < PCS ? q : q - PCS) * 1024;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)q * 1024u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(src) : "memory", "m0");
        }
    };
    issue(t_begin);
    if (mom) { for (int i = tid; i < P * 5; i += 256) red[i] = 0; }
    const bool col_fixed = mom && (256 % img_w) == 0;
    unsigned mcnt[P], msy[P], msyy[P];
#pragma unroll
    for (int c = 0; c < P; ++c) mcnt[c] = msy[c] = msyy[c] = 0u;
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) { issue(t + 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* tm = (const float*)(smem + ((t - t_begin) & 1) * 2 * TB);
        const float* te = tm + 256 * P;
        float v[P];
#pragma unroll
        for (int c = 0; c < P; c += 2) {
            const float2 a = *(const float2*)(tm + tid * P + c);
            v[c] = a.x; v[c + 1] = a.y;
        }
        if (has_eps) {
#pragma unroll
            for (int c = 0; c < P; c += 2) {
                const float2 b = *(const float2*)(te + tid * P + c);
                v[c] += b.x; v[c + 1] += b.y;
            }
        }
        float* to = (float*)OUT;
        if (l) {
#pragma unroll
            for (int c = 0; c < P; c += 2) *(float2*)(to + tid * P + c) = make_float2(v[c], v[c + 1]);
        }
        float mx = v[0];
#pragma unroll
        for (int c = 1; c < P; ++c) mx = fmaxf(mx, v[c]);
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < P; ++c) { v[c] = __expf(v[c] - mx); sm += v[c]; }
        float r = __builtin_amdgcn_rcpf(sm);
        r = r * (2.f - sm * r);                       // one Newton step: the quotient to ~1 ulp
        float mm = 0.f;
#pragma unroll
        for (int c = 0; c < P; ++c) { v[c] *= r; mm = fmaxf(mm, v[c]); }
        unsigned bm = 0u;
#pragma unroll
        for (int c = 0; c < P; ++c) bm |= (v[c] == mm) ? (1u << c) : 0u;
#pragma unroll
        for (int c = 0; c < P; c += 2) {
            *(float2*)(to + 256 * P + tid * P + c) = make_float2(v[c], v[c + 1]);
            if (hard) *(float2*)(to + 2 * 256 * P + tid * P + c) = make_float2((bm >> c) & 1u ? 1.f : 0.f, (bm >> (c + 1)) & 1u ? 1.f : 0.f);
        }
        const long long px = img + ((long long)t << 8) + tid;
        if (amax) amax[px] = (long long)(__ffs(bm) - 1);
        if (bits) bits[px] = bm;
        if (mom) {
            const int q = (t << 8) + tid;
            const int iy = q / img_w, ix = q - iy * img_w;
            if (col_fixed) {
                // the thread's column is the same in every tile of the block (img_w | 256): per part only the pixel count and the
                // row sums are carried per thread (4 instructions per part and pixel, whatever the masks look like); the column sums
                // follow as ix * count at the end of the block
                const unsigned iy2 = (unsigned)(iy * iy);
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    const unsigned bit = (bm >> c) & 1u;
                    mcnt[c] += bit; msy[c] += bit * (unsigned)iy; msyy[c] += bit * iy2;
                }
            } else {
                const bool one_row = (img_w & 63) == 0;                    // a wave's 64 pixels lie in one image row: scalar iy sums
                const int iy_u = __builtin_amdgcn_readfirstlane(iy);
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    const bool sel = (bm >> c) & 1u;
                    const unsigned long long mask = __ballot(sel);
                    if (mask) {                                                         // (uniform)
                        const int cnt = __popcll(mask);
                        const int sx = wave_sum_full_i(sel ? ix : 0), sxx = wave_sum_full_i(sel ? ix * ix : 0);
                        int sy, syy;
                        if (one_row) { sy = iy_u * cnt; syy = iy_u * iy_u * cnt; }
                        else { sy = wave_sum_full_i(sel ? iy : 0); syy = wave_sum_full_i(sel ? iy * iy : 0); }
                        if (lane == 0) {
                            int* rr = red + 5 * c;
                            atomicAdd(rr, cnt); atomicAdd(rr + 1, sy); atomicAdd(rr + 2, sx); atomicAdd(rr + 3, syy); atomicAdd(rr + 4, sxx);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float4* o4 = (const float4*)OUT;
        const long long o16 = (img + ((long long)t << 8)) * P / 4;                     // first 16-byte piece of the tile in a map
#pragma unroll
        for (int k = 0; k < 3; ++k) {                // l, m, hard: 640 pieces each
            float* dstp = k == 0 ? l : (k == 1 ? m : hard);
            if (!dstp) continue;                     // (uniform)
            float4* dst = (float4*)dstp + o16;
#pragma unroll
            for (int j = tid; j < TB / 16; j += 256) dst[j] = o4[k * (TB / 16) + j];
        }
    }
    if (mom) {
        if (col_fixed) {
            const int ix = tid % img_w;
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const int cnt = wave_sum_full_i((int)mcnt[c]);
                if (cnt) {                                                              // (uniform)
                    const int sy = wave_sum_full_i((int)msy[c]), syy = wave_sum_full_i((int)msyy[c]);
                    const int sx = wave_sum_full_i(ix * (int)mcnt[c]), sxx = wave_sum_full_i(ix * ix * (int)mcnt[c]);
                    if (lane == 0) {
                        int* rr = red + 5 * c;
                        atomicAdd(rr, cnt); atomicAdd(rr + 1, sy); atomicAdd(rr + 2, sx); atomicAdd(rr + 3, syy); atomicAdd(rr + 4, sxx);
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < P * 5; i += 256) mom[((long long)n * bpi + bi) * P * 5 + i] = red[i];
    }
}

// ------------------------------------------------------------------ spatial soft-max moments (nn.py:65-71, 1541-1587)
// partial[n][slab][p][8] = {max, Z, S0, Sy, Sx, Q, Qy, -} relative to the slab max (Q = sum e*k*(gy^2+gx^2), Qy = sum e*k*gy^2)
// threads = (part c, sub-lane s) with NS = 256 / P sub-lanes per part (250 of 256 lanes busy at P = 10, against 160 with a
// power-of-two lane group); each walks the column c of the staged tile with its pixel coordinates advanced incrementally.
// kl_partial (optional, [n][nslab]): sum over the slab of x * log(P * x + 1e-20) -- the categorical KL of the map itself
// (model.py:21-25): view 1's only other prior term, so its separate pass over the map (prior_fwd, view 1) is not needed.
__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ x, int h, int w, int P, float gamma,
                                                              const int* __restrict__ rc, int hh, int hw_half,
                                                              int rows_per_slab, int tpx, float* __restrict__ partial,
                                                              float* __restrict__ kl_partial) {
    extern __shared__ __attribute__((aligned(16))) float ts[];          // [tpx][PP], then red[NS][P][7], then 4 floats
    const int PP = tile_pitch(P);
    const int NS = 256 / P;
    float* red = ts + (size_t)tpx * PP;
    float* red4 = red + (size_t)NS * P * 7;
    float kl = 0.f;
    const int n = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
    const int c = threadIdx.x % P, sl = threadIdx.x / P;
    const bool act = sl < NS;
    const int y0 = slab * rows_per_slab, y1 = min(h, y0 + rows_per_slab);
    float mx = -INFINITY, Z = 0.f, S0 = 0.f, Sy = 0.f, Sx = 0.f, Q = 0.f, Qy = 0.f;
    int cy = 0, cx = 0;
    if (rc) { cy = rc[((long long)n * P + c) * 2]; cx = rc[((long long)n * P + c) * 2 + 1]; }
    const float sy = h > 1 ? 2.f / (float)(h - 1) : 0.f, sx = w > 1 ? 2.f / (float)(w - 1) : 0.f;
    const int q0 = y0 * w, q1 = max(y1, y0) * w;
    const float* img = x + (long long)n * h * w * P;
    const int dyy = NS / w, dxx = NS - dyy * w;                         // pixel step NS in (row, column) form
    TileReq<3> rq;                            // the next tile's pieces are in flight while this one is summed (tile.h)
    if (q0 < q1) tile_request(rq, img + (long long)q0 * P, min(tpx, q1 - q0), P);
    for (int t0 = q0; t0 < q1; t0 += tpx) {
        const int cnt = min(tpx, q1 - t0);
        __syncthreads();                      // the previous tile has been consumed
        tile_commit(rq, img + (long long)t0 * P, cnt, P, PP, ts);
        if (t0 + tpx < q1) tile_request(rq, img + (long long)(t0 + tpx) * P, min(tpx, q1 - t0 - tpx), P);
        __syncthreads();
        if (act) {
            int yy = (t0 + sl) / w, xx = (t0 + sl) - yy * w;
            for (int px = sl; px < cnt; px += NS) {
                const float xv = ts[px * PP + c];
                if (kl_partial) kl += xv * __logf((float)P * xv + 1e-20f);
                const float v = gamma * xv;
                if (v > mx) {
                    const float sc = __expf(mx - v);  // exp(-inf) = 0 on the first element
                    Z *= sc; S0 *= sc; Sy *= sc; Sx *= sc; Q *= sc; Qy *= sc;
                    mx = v;
                }
                const float e = __expf(v - mx);
                Z += e;
                float k = 1.f;
                if (rc && abs(yy - cy) <= hh && abs(xx - cx) <= hw_half) k = 0.f;
                const float gy = -1.f + sy * (float)yy, gx = -1.f + sx * (float)xx;
                const float ek = e * k;
                S0 += ek; Sy += ek * gy; Sx += ek * gx; Q += ek * (gy * gy + gx * gx); Qy += ek * gy * gy;
                yy += dyy; xx += dxx;
                if (xx >= w) { xx -= w; ++yy; }
            }
        }
    }
    __syncthreads();
    if (act) {
        float* d = red + ((size_t)sl * P + c) * 7;
        d[0] = mx; d[1] = Z; d[2] = S0; d[3] = Sy; d[4] = Sx; d[5] = Q; d[6] = Qy;
    }
    __syncthreads();
    if (threadIdx.x < P) {
        float M = -INFINITY;
        for (int q = 0; q < NS; ++q) M = fmaxf(M, red[((size_t)q * P + c) * 7]);
        float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < NS; ++q) {
            const float* r = red + ((size_t)q * P + c) * 7;
            const float sc = (r[0] == -INFINITY) ? 0.f : __expf(r[0] - M);
            for (int k = 0; k < 6; ++k) o[k] += sc * r[1 + k];
        }
        float* dst = partial + (((long long)n * nslab + slab) * P + c) * 8;
        dst[0] = M;
        for (int k = 0; k < 6; ++k) dst[1 + k] = o[k];
        dst[7] = 0.f;
    }
    if (kl_partial) {
        const float v = block_sum_256(kl, red4);
        if (threadIdx.x == 0) kl_partial[(long long)n * nslab + slab] = v;
    }
}

// Round 5: the same partial sums pixel-per-lane for P = 10 at 128- / 256-wide maps (the soft map of view 1: 42 MB read once, the
// launch was VALU-bound at 1.3 TB/s -- 85 instructions per (pixel, part) element in the (part, sub-lane) form above).  Tiles of 512
// pixels arrive linearly by LDS-DMA (three slots, two tiles in flight); a thread owns two pixels of a tile with its ten parts'
// running (max, Z, S0, Sy, Sx, Q, Qy) in registers -- one branch-free rescale per part and tile -- and the block leaves ONE record
// per part (shuffle trees, then the four waves through LDS): the launcher hands moments_combine_kernel `blocks per image` as its
// slab count.
template <int P, int LW>
__global__ __launch_bounds__(256) void moments_px_kernel(const float* __restrict__ x, const int h, const float gamma,
                                                            const int* __restrict__ rc, const int hh, const int hw_half,
                                                            const int tiles_per_block, float* __restrict__ partial,
                                                            float* __restrict__ kl_partial) {
    static_assert(P % 2 == 0 && ((P / 2) & 1) == 1, "conflict-free 8-byte LDS reads need P / 2 odd");
    constexpr int W = 1 << LW;
    constexpr int TPX = 512, TB = TPX * P * 4, PW = TB / 1024 / 4;          // 20 pieces per tile: five per wave
    static_assert(TB % 4096 == 0, "piece counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];    // 3 tiles (2 when the block has only 2), then [4 waves][P][7] + 4 floats
    float* red = (float*)(smem + min(3, tiles_per_block) * TB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hwp = h * W, tiles_img = hwp / TPX;
    const int bpi = tiles_img / tiles_per_block;
    const int n = blockIdx.x / bpi, bi = blockIdx.x - n * bpi;
    const int t_begin = bi * tiles_per_block, t_end = t_begin + tiles_per_block;
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* src0 = (const unsigned char*)(x + (long long)n * hwp * P);
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const unsigned base = smem_lds + (unsigned)(((t - t_begin) % 3) * TB);
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int q = wid + 4 * k;
            const unsigned char* src = src0 + (long long)t * TB + q * 1024;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)q * 1024u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(src) : "memory", "m0");
        }
    };
    issue(t_begin);
    if (t_begin + 1 < t_end) issue(t_begin + 1);
    int cy[P], cx[P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        cy[c] = rc ? rc[((long long)n * P + c) * 2] : 0;
        cx[c] = rc ? rc[((long long)n * P + c) * 2 + 1] : 0;
    }
    const float sy = h > 1 ? 2.f / (float)(h - 1) : 0.f, sx = W > 1 ? 2.f / (float)(W - 1) : 0.f;
    float mx[P], Z[P], S0[P], Sy[P], Sx[P], Q[P], Qy[P];
#pragma unroll
    for (int c = 0; c < P; ++c) { mx[c] = -INFINITY; Z[c] = S0[c] = Sy[c] = Sx[c] = Q[c] = Qy[c] = 0.f; }
    float kl = 0.f;
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < t_end) issue(t + 2);
        const float* tile = (const float*)(smem + ((t - t_begin) % 3) * TB);
        float v0[P], v1[P];
#pragma unroll
        for (int c = 0; c < P; c += 2) {
            const float2 a = *(const float2*)(tile + tid * P + c), b = *(const float2*)(tile + (256 + tid) * P + c);
            v0[c] = a.x; v0[c + 1] = a.y; v1[c] = b.x; v1[c + 1] = b.y;
        }
        // a wave's 64 pixels lie in one row (W >= 64): the row index is wave-uniform (scalar rectangle tests, scalar gy); the thread's
        // second pixel sits 256 / W rows below the first, in the SAME column
        const int q0 = t * TPX + tid;
        const int y0 = __builtin_amdgcn_readfirstlane(q0 >> LW), x0 = q0 & (W - 1), y1 = y0 + 256 / W, x1 = x0;
        const float gy0 = -1.f + sy * (float)y0, gx0 = -1.f + sx * (float)x0;
        const float gy1 = -1.f + sy * (float)y1, gx1 = -1.f + sx * (float)x1;
        const float r0 = gy0 * gy0 + gx0 * gx0, r1 = gy1 * gy1 + gx1 * gx1;
        const float yy0 = gy0 * gy0, yy1 = gy1 * gy1;
#pragma unroll
        for (int c = 0; c < P; ++c) {
            const float a = v0[c], b = v1[c];
            if (kl_partial) kl += a * ups_log_fast((float)P * a + 1e-20f) + b * ups_log_fast((float)P * b + 1e-20f);
            const float ga = gamma * a, gb = gamma * b;
            const float nm = fmaxf(mx[c], fmaxf(ga, gb));
            const float sc = __expf(mx[c] - nm);          // exp(-inf) = 0 on the first tile
            const float ea = __expf(ga - nm), eb = __expf(gb - nm);
            float ka = ea, kb = eb;
            if (rc) {
                if (abs(y0 - cy[c]) <= hh && abs(x0 - cx[c]) <= hw_half) ka = 0.f;
                if (abs(y1 - cy[c]) <= hh && abs(x1 - cx[c]) <= hw_half) kb = 0.f;
            }
            mx[c] = nm;
            Z[c] = Z[c] * sc + (ea + eb);
            S0[c] = S0[c] * sc + (ka + kb);
            Sy[c] = Sy[c] * sc + (ka * gy0 + kb * gy1);
            Sx[c] = Sx[c] * sc + (ka * gx0 + kb * gx1);
            Q[c] = Q[c] * sc + (ka * r0 + kb * r1);
            Qy[c] = Qy[c] * sc + (ka * yy0 + kb * yy1);
        }
    }
    // ---- one record per part: wave trees (max, then the rescaled sums), the four waves through LDS
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const float M = wave_max_full(mx[c]);
        const float sc = __expf(mx[c] - M);
        const float z = wave_sum_full(Z[c] * sc), s0 = wave_sum_full(S0[c] * sc), s1 = wave_sum_full(Sy[c] * sc), s2 = wave_sum_full(Sx[c] * sc);
        const float s3 = wave_sum_full(Q[c] * sc), s4 = wave_sum_full(Qy[c] * sc);
        if (lane == 0) {
            float* d = red + (wid * P + c) * 7;
            d[0] = M; d[1] = z; d[2] = s0; d[3] = s1; d[4] = s2; d[5] = s3; d[6] = s4;
        }
    }
    kl = wave_sum_full(kl);
    if (lane == 0) red[4 * P * 7 + wid] = kl;
    __syncthreads();
    if (tid < P) {
        const int c = tid;
        float M = red[c * 7];
        for (int w = 1; w < 4; ++w) M = fmaxf(M, red[(w * P + c) * 7]);
        float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < 4; ++w) {
            const float* r = red + (w * P + c) * 7;
            const float sc = __expf(r[0] - M);
            for (int k = 0; k < 6; ++k) o[k] += sc * r[1 + k];
        }
        float* dst = partial + (((long long)n * bpi + bi) * P + c) * 8;
        dst[0] = M;
        for (int k = 0; k < 6; ++k) dst[1 + k] = o[k];
        dst[7] = 0.f;
    }
    if (kl_partial && tid == 0)
        kl_partial[(long long)n * bpi + bi] = (red[4 * P * 7] + red[4 * P * 7 + 1]) + (red[4 * P * 7 + 2] + red[4 * P * 7 + 3]);
}

// stats[n][p] = {max, Z, S0, Sy, Sx, Q, Qy, 0} of spatial_softmax(gamma * hard) (no rectangle) from the integer sums of the
// hard pixels: e = 1 on them, exp(-gamma) elsewhere (all e = 1 when the part owns no pixel); grid sums of the linspace(-1, 1)
// coordinates in closed form (sum g = 0, sum g^2 = n (n + 1) / (3 (n - 1)))
// (one WAVE per (image, part): lane b sums the tiles b, b + 64, ... and a shuffle tree adds the lanes -- a thread per (image, part)
// walking all tiles serially made this tail 28 us behind a 95 us soft-max pass)
__global__ __launch_bounds__(256) void hard_moments_finalize_kernel(const int* __restrict__ mom, int count_n, int blocks_per_img, int P,
                                                                    int h, int w, float gamma, float* __restrict__ stats) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= count_n * P) return;
    const int n = idx / P, c = idx - n * P;
    long long a[5] = {0, 0, 0, 0, 0};
    for (int b = lane; b < blocks_per_img; b += 64) {
        const int* r = mom + ((long long)(n * blocks_per_img + b) * P + c) * 5;
        for (int k = 0; k < 5; ++k) a[k] += r[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
    if (lane != 0) return;
    const double N = (double)h * w, N1 = (double)a[0];
    const double sy = h > 1 ? 2.0 / (h - 1) : 0.0, sx = w > 1 ? 2.0 / (w - 1) : 0.0;
    const double hy = -N1 + sy * a[1], hx = -N1 + sx * a[2];                                   // sum over the hard pixels of gy, gx
    const double hyy = N1 - 2.0 * sy * a[1] + sy * sy * a[3], hxx = N1 - 2.0 * sx * a[2] + sx * sx * a[4];
    const double ayy = h > 1 ? (double)w * h * (h + 1.0) / (3.0 * (h - 1.0)) : N;                // sum over ALL pixels of gy^2
    const double axx = w > 1 ? (double)h * w * (w + 1.0) / (3.0 * (w - 1.0)) : N;
    float* d = stats + (long long)idx * 8;
    if (a[0] > 0) {
        const double e = exp(-(double)gamma), Z = N1 + e * (N - N1);
        const double Qy = (1.0 - e) * hyy + e * ayy, Qx = (1.0 - e) * hxx + e * axx;
        d[0] = gamma; d[1] = (float)Z; d[2] = (float)Z; d[3] = (float)((1.0 - e) * hy); d[4] = (float)((1.0 - e) * hx);
        d[5] = (float)(Qy + Qx); d[6] = (float)Qy; d[7] = 0.f;
    } else {
        d[0] = 0.f; d[1] = (float)N; d[2] = (float)N; d[3] = 0.f; d[4] = 0.f; d[5] = (float)(ayy + axx); d[6] = (float)ayy; d[7] = 0.f;
    }
}

// (grid: ceil(n * P / 4) blocks -- four waves, one (image, part) each -- + one more that sums the KL partials in a fixed order)
__global__ __launch_bounds__(256) void moments_combine_kernel(const float* __restrict__ partial, int count_n, int nslab, int P,
                                                              float* __restrict__ stats, const float* __restrict__ kl_partial,
                                                              float* __restrict__ kl_sum) {
    if (blockIdx.x == gridDim.x - 1 && kl_sum) {
        __shared__ float red4[4];
        float a = 0.f;
        for (int i = threadIdx.x; i < count_n * nslab; i += 256) a += kl_partial[i];
        const float v = block_sum_256(a, red4);
        if (threadIdx.x == 0) { kl_sum[0] = v; for (int k = 1; k < 16; ++k) kl_sum[k] = 0.f; }
        return;
    }
    // one wave per (image, part): lane s holds slab s (, s + 64, ...), shuffle trees for the max and the rescaled sums
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= count_n * P) return;
    const int n = idx / P, c = idx - n * P;
    float M = -INFINITY;
    for (int s = lane; s < nslab; s += 64) M = fmaxf(M, partial[(((long long)n * nslab + s) * P + c) * 8]);
    M = wave_max(M);
    float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = lane; s < nslab; s += 64) {
        const float* src = partial + (((long long)n * nslab + s) * P + c) * 8;
        const float sc = (src[0] == -INFINITY) ? 0.f : expf(src[0] - M);
        for (int k = 0; k < 6; ++k) o[k] += sc * src[1 + k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = wave_sum(o[k]);
    if (lane != 0) return;
    float* d = stats + (long long)idx * 8;
    d[0] = M; d[1] = o[0]; d[2] = o[1]; d[3] = o[2]; d[4] = o[3]; d[5] = o[4]; d[6] = o[5]; d[7] = 0.f;
}

// px = (row centre, column centre) of the rectangle draw_rect paints.  xy_order: the external tfutils.draw_rect reads the
// (y, x) pair it is handed (M:441-442) as (x, y), so the row centre comes from mu_x and the column centre from mu_y.
__global__ void moments_to_px_kernel(const float* __restrict__ stats, int count, int h, int xy_order, int* __restrict__ px) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const float* s = stats + (long long)idx * 8;
    const float muy = s[3] / s[1], mux = s[4] / s[1];
    const int cy = (int)(muy * (float)h / 2.0f + (float)h / 2.0f);   // tf.cast(float->int32): truncation
    const int cx = (int)(mux * (float)h / 2.0f + (float)h / 2.0f);
    px[idx * 2 + 0] = xy_order ? cx : cy;
    px[idx * 2 + 1] = xy_order ? cy : cx;
}

__global__ void draw_rect_kernel(const int* __restrict__ px, int n, int h, int w, int P, int hh, int hwid, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * h * w * P) return;
    const int c = (int)(idx % P);
    long long t = idx / P;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    const int cy = px[((long long)b * P + c) * 2], cx = px[((long long)b * P + c) * 2 + 1];
    out[idx] = (abs(y - cy) <= hh && abs(x - cx) <= hwid) ? 1.f : 0.f;
}

// ------------------------------------------------------------------ mask_parts (model.py:176-187) + part-major transpose (nn.py:97-103)
// block = 256 consecutive pixels (of the batch): the hard tile comes in with 16-byte coalesced loads (tile.h; a thread-per-pixel read of P
// floats at a 4 P byte stride touched 20-50 cache lines per wave instruction, P times over: 0.38-0.44 of the HBM roof at P = 16 / 20 / 25
// with rotating operands), then thread = pixel writes P 16-byte pieces, each coalesced along the pixels of one part image
template <typename T>
__global__ __launch_bounds__(256) void mask_parts_fwd_kernel(const float* __restrict__ view, const float* __restrict__ hard, T* __restrict__ out,
                                                             int B, long long hw, int P) {
    extern __shared__ __attribute__((aligned(16))) float ts[];          // [256][PP]
    const int PP = tile_pitch(P);
    const long long total = (long long)B * hw;
    const long long pix0 = (long long)blockIdx.x * 256;
    const int cnt = (int)min(256ll, total - pix0);
    const long long idx = pix0 + threadIdx.x;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (threadIdx.x < cnt) { v0 = view[idx * 3]; v1 = view[idx * 3 + 1]; v2 = view[idx * 3 + 2]; }
    tile_load_f32(hard + pix0 * P, cnt, P, PP, ts);
    __syncthreads();
    if (threadIdx.x >= cnt) return;
    const int b = (int)(idx / hw);
    const long long px = idx - (long long)b * hw;
    const float* hrow = ts + threadIdx.x * PP;
    for (int p = 0; p < P; ++p) {
        const float hm = hrow[p];
        float f[8] = {v0 * hm, v1 * hm, v2 * hm, 0.f, 0.f, 0.f, 0.f, 0.f};
        T* o = out + (((long long)p * B + b) * hw + px) * 8;
        if (sizeof(T) == 2) *(uint4*)o = Chunk<bf16>::pack(f);
        else { *(uint4*)o = Chunk<float>::pack(f); *(uint4*)((float*)o + 4) = Chunk<float>::pack(f + 4); }
    }
}
// thread = pixel: P 16-byte loads (coalesced along the pixels of one part image), results staged as a [256][P] tile and
// written with 16-byte stores (a thread-per-pixel store of P floats at a 4 P byte stride ran at 7 % of the HBM roof)
template <typename T>
__global__ __launch_bounds__(256) void mask_parts_bwd_kernel(const float* __restrict__ view, const T* __restrict__ g,
                                                             float* __restrict__ gh, int B, long long hw, int P) {
    extern __shared__ __attribute__((aligned(16))) float ts[];          // [256][PP]
    const int PP = tile_pitch(P);
    const long long total = (long long)B * hw;
    const long long pix0 = (long long)blockIdx.x * 256;
    const int cnt = (int)min(256ll, total - pix0);
    const long long idx = pix0 + threadIdx.x;
    if (threadIdx.x < cnt) {
        const int b = (int)(idx / hw);
        const long long px = idx - (long long)b * hw;
        const float v0 = view[idx * 3], v1 = view[idx * 3 + 1], v2 = view[idx * 3 + 2];
        for (int p = 0; p < P; ++p) {
            const T* gp = g + (((long long)p * B + b) * hw + px) * 8;
            float f[4];
            if (sizeof(T) == 2) {
                const uint2 u = *(const uint2*)gp;
                f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u); f[2] = __uint_as_float(u.y << 16);
            } else {
                const float4 u = *(const float4*)gp;
                f[0] = u.x; f[1] = u.y; f[2] = u.z;
            }
            ts[threadIdx.x * PP + p] = f[0] * v0 + f[1] * v1 + f[2] * v2;
        }
    }
    __syncthreads();
    tile_store_f32(gh + pix0 * P, cnt, P, PP, ts);
}

// ------------------------------------------------------------------ unpool_features + concat (model.py:225-249, 482-484)
// block = 256 consecutive pixels of one image: the hard tile and feat[b] sit in LDS; one item per (pixel, 8-channel chunk)
// -> consecutive items write consecutive 16-byte chunks.  Chunks >= F/8 carry the hard mask itself, then zero padding.
template <typename T>
__global__ __launch_bounds__(256) void unpool_fwd_kernel(const float* __restrict__ hard, const float* __restrict__ feat,
                                                         T* __restrict__ out, long long hw, int P, int F, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float ts[];          // hard [256][PP], feat [P][F]
    const int PP = tile_pitch(P);
    float* fs = ts + 256 * PP;
    const int b = blockIdx.y;
    const long long q0 = (long long)blockIdx.x * 256;
    const int cnt = (int)min(256ll, hw - q0);
    const long long pix0 = (long long)b * hw + q0;
    tile_load_f32(hard + pix0 * P, cnt, P, PP, ts);
    for (int i = threadIdx.x; i < P * F; i += 256) fs[i] = feat[(long long)b * P * F + i];
    int* act = (int*)(fs + P * F);                        // per pixel: its single active part, -1 (none) or -2 (several: ties)
    __syncthreads();
    // the hard mask is one-hot except at ties: find each pixel's active part ONCE (every 16-byte output item walked all P parts
    // before: 100 LDS reads per pixel at P = 10 for ~10 useful ones)
    for (int px = threadIdx.x; px < cnt; px += 256) {
        const float* hrow = ts + px * PP;
        int a = -1;
        for (int p = 0; p < P; ++p)
            if (hrow[p] != 0.f) a = a == -1 ? p : -2;
        act[px] = a;
    }
    __syncthreads();
    const int cpp = ldo / 8;
    for (int i = threadIdx.x; i < cnt * cpp; i += 256) {
        const int px = i / cpp, k = i - px * cpp;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* hrow = ts + px * PP;
        if (k * 8 < F) {
            const int a = act[px];
            if (a >= 0) {
                const float hm = hrow[a];
                const float* fr = fs + a * F + k * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = hm * fr[e];
            } else if (a == -2) {
                for (int p = 0; p < P; ++p) {
                    const float hm = hrow[p];
                    if (hm != 0.f) {
                        const float* fr = fs + p * F + k * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += hm * fr[e];
                    }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = k * 8 + e - F;
                if (ch < P) f[e] = hrow[ch];
            }
        }
        T* o = out + (pix0 + px) * ldo + k * 8;
        if (sizeof(T) == 2) *(uint4*)o = Chunk<bf16>::pack(f);
        else { *(uint4*)o = Chunk<float>::pack(f); *(uint4*)((float*)o + 4) = Chunk<float>::pack(f + 4); }
    }
}

// Backward of unpool_features + concat in ONE pass over the gradient (round 4; it was two kernels, each reading g):
//   g_hard[b][px][p]  = sum_f g[b][px][f] * feat[b][p][f] + g[b][px][F+p]
//   g_feat[b][p][f]   = sum_px hard[b][px][p] * g[b][px][f]          (per-slab partials, reduced by unpool_feat_reduce_kernel)
// block = (image b, slab of the image's pixels), walked in tiles of tpx pixels: the hard tile (odd pitch) and the gradient rows
// (raw 16-byte pieces) of a tile are requested together and the NEXT tile's pieces are in flight while this one is computed
// (tile.h).  Phase 1: thread = (part, pixel lane) -> g_hard tile, stored with 16-byte accesses.  Phase 2:
// lane = feature, each wave walks the tile's pixels and adds the gradient row into the LDS column of the (usually single)
// active part.
template <typename T>
__global__ __launch_bounds__(256) void unpool_bwd_kernel(const float* __restrict__ hard, const float* __restrict__ feat,
                                                         const T* __restrict__ g, float* __restrict__ gh,
                                                         float* __restrict__ gfeat_partial, long long hw, int P, int F, int ldo,
                                                         int slab_px, int tpx) {
    extern __shared__ __attribute__((aligned(16))) float ts[];  // acc [4][P][64], out [tpx][PP], hard [tpx][PP], feat [P][F+1], g [tpx][ldo] (T)
    const int PP = tile_pitch(P);
    float* acc = ts;
    float* os = acc + 4 * P * 64;
    float* hs = os + (((size_t)tpx * PP + 3) & ~(size_t)3);
    float* fs = hs + (((size_t)tpx * PP + 3) & ~(size_t)3);
    T* gs = (T*)(fs + (((size_t)P * (F + 1) + 3) & ~(size_t)3));
    const int b = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
    const int f = threadIdx.x & 63, pl = threadIdx.x >> 6;
    // phase-1 role: thread = (part c, pixel lane pg), 256 / P pixel lanes (250 of 256 threads busy at P = 10)
    const int PL = 256 / P;
    const int pg = threadIdx.x / P;
    const int c = pg < PL ? threadIdx.x - pg * P : P;           // c == P: idle
    constexpr int KG = 5;                                       // 16-byte pieces of a gradient tile per thread (128 px x 80 ch bf16)
    for (int i = threadIdx.x; i < 4 * P * 64; i += 256) acc[i] = 0.f;
    for (int i = threadIdx.x; i < P * F; i += 256) fs[(i / F) * (F + 1) + (i % F)] = feat[(long long)b * P * F + i];
    const long long p0 = (long long)slab * slab_px, p1 = min(hw, p0 + (long long)slab_px);
    float* my = acc + (long long)pl * P * 64;
    TileReq<2> rh;
    uint4 rg[KG];
    auto request = [&](long long t0) {
        const int cnt = (int)min((long long)tpx, p1 - t0);
        const long long pix0 = (long long)b * hw + t0;
        tile_request(rh, hard + pix0 * P, cnt, P);
        const uint4* src = (const uint4*)(g + pix0 * ldo);
        const int nv = (int)((size_t)cnt * ldo * sizeof(T) / 16);
#pragma unroll
        for (int k = 0; k < KG; ++k) { const int i = k * 256 + threadIdx.x; rg[k] = i < nv ? src[i] : make_uint4(0u, 0u, 0u, 0u); }
    };
    if (p0 < p1) request(p0);
    // feat[b][c][:] of this thread's part in registers: the inner product of phase 1 then costs 8 LDS reads (the gradient row) per
    // (pixel, part) instead of 72 -- the launch was LDS-issue bound on those scalar reads
    // 16-bit gradients: the row as PACKED bf16 pairs, contracted with v_dot2c_f32_bf16 (two MACs per instruction, no unpacking of
    // the gradient; the launch was VALU-bound on 64 conversions + 64 FMAs per (pixel, part)).  In bf16 mode feat is the float view of
    // a bf16 tensor, so the packing is exact; an fp32 feat given with bf16 gradients is rounded to bf16 here.
    float frr[sizeof(T) == 2 ? 1 : 64];
    unsigned fpk[sizeof(T) == 2 ? 32 : 1];
    __syncthreads();                                              // (fs is complete)
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int k = 0; k < 32; ++k)
            fpk[k] = (c < P && 2 * k + 1 < F) ? Chunk<bf16>::pk(fs[c * (F + 1) + 2 * k], fs[c * (F + 1) + 2 * k + 1]) : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < 64; ++k) frr[k] = (c < P && k < F) ? fs[c * (F + 1) + k] : 0.f;
    }
    for (long long t0 = p0; t0 < p1; t0 += tpx) {
        const int cnt = (int)min((long long)tpx, p1 - t0);
        const long long pix0 = (long long)b * hw + t0;
        __syncthreads();                                          // the previous tile has been consumed (and stored)
        tile_commit(rh, hard + pix0 * P, cnt, P, PP, hs);
        {
            const uint4* src = (const uint4*)(g + pix0 * ldo);
            const int nv = (int)((size_t)cnt * ldo * sizeof(T) / 16);
#pragma unroll
            for (int k = 0; k < KG; ++k) { const int i = k * 256 + threadIdx.x; if (i < nv) ((uint4*)gs)[i] = rg[k]; }
            for (int i = KG * 256 + threadIdx.x; i < nv; i += 256) ((uint4*)gs)[i] = src[i];
        }
        if (t0 + tpx < p1) request(t0 + tpx);
        __syncthreads();
        if (c < P) {                                              // phase 1: d / d hard (the part's feature row sits in registers)
            for (int px = pg; px < cnt; px += PL) {
                const T* row = gs + (size_t)px * ldo;
                float a = ld_as_float<T>(row + F + c);
#pragma unroll
                for (int ff = 0; ff < 64; ff += 8) {
                    if (ff < F) {
                        if constexpr (sizeof(T) == 2) {
                            const uint4 u = *(const uint4*)(row + ff);
                            const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                bf16x2_t gv, fv;
                                __builtin_memcpy(&gv, &w4[e], 4); __builtin_memcpy(&fv, &fpk[ff / 2 + e], 4);
                                a = __builtin_amdgcn_fdot2_f32_bf16(gv, fv, a, false);
                            }
                        } else {
                            float v[8];
                            uint4 u0 = *(const uint4*)(row + ff), u1 = *(const uint4*)((const float*)(row + ff) + 4);
                            Chunk<float>::unpack(u0, v); Chunk<float>::unpack(u1, v + 4);
#pragma unroll
                            for (int e = 0; e < 8; ++e) a += v[e] * frr[ff + e];
                        }
                    }
                }
                os[px * PP + c] = a;
            }
        }
        for (int px = pl; px < cnt; px += 4) {                    // phase 2: d / d feat
            const float gf = (f < F) ? ld_as_float<T>(gs + (size_t)px * ldo + f) : 0.f;
            const float hv = (f < P) ? hs[px * PP + f] : 0.f;
            unsigned long long m = __ballot(hv != 0.f);
            while (m) {
                const int pp = __ffsll((long long)m) - 1;
                m &= m - 1;
                my[pp * 64 + f] += __shfl(hv, pp, 64) * gf;
            }
        }
        __syncthreads();
        tile_store_f32(gh + pix0 * P, cnt, P, PP, os);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * 64; i += 256) {
        const int pp = i / 64, ff = i % 64;
        if (ff < F)
            gfeat_partial[(((long long)b * nslab + slab) * P + pp) * F + ff] =
                acc[i] + acc[P * 64 + i] + acc[2 * P * 64 + i] + acc[3 * P * 64 + i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the same backward on the matrix cores for the benchmark shape (bf16 gradient, F = 64 features, P = 10 parts, 80-channel
// rows).  unpool_bwd_kernel issued 60 M wave instructions per launch -- a pixel's feature row added into its part's accumulator row
// by one wave at a time (27 instructions per pixel), 32 dot2 per (pixel, part) -- and the SIMDs' issue slots, not HBM, set its
// 137 us (SQ counters, profiles/round5_pmc_part.txt).  Both products are small GEMMs:
//   g_hard[px][p] = sum_f G[px][f] feat[p][f] (+ G[px][64 + p])   : A = G rows as they lie in LDS, B = feat, K = 64 features
//   g_feat[p][f]  = sum_px hard[px][p] G[px][f]                   : A = hard^T (hi + lo bf16 pair: exact to 2^-17 for any
//                                                                   fp32 mask, exact for the 0 / 1 masks the model passes),
//                                                                   B = G read column-wise with ds_read_b64_tr_b16, K = 32 pixels
// ~100 wave instructions per 128-pixel tile instead of ~460.  Tiles (gradient rows + hard rows) arrive LINEARLY by LDS-DMA, three
// slots (two tiles in flight), g_hard leaves through an LDS tile with 16-byte stores; waits are counted (loads, stores and LDS-DMA
// retire in issue order).  80 KB of LDS: two blocks per CU.
// Round 6: templated on the part count (P = 10 / 16 / 20 / 25 in rows of LD = 80 / 80 / 88 / 96 channels -- the BASELINE configs and
// every shipped yaml); more than 16 parts take two 16-part blocks in both products.
constexpr int UB_TP = 128, UB_F = 64;
template <int P, int LD> struct UB {
    static constexpr int GT = UB_TP * LD * 2;               // gradient tile (bf16): 20 / 22 / 24 KiB
    static constexpr int HT = UB_TP * P * 4;                // hard tile (and g_hard tile), fp32
    static constexpr int SLOT = GT + HT;
    static constexpr int GP = GT / 1024, HP = (HT + 1023) / 1024, PIECES = GP + HP;       // LDS-DMA pieces of 1 KiB (the last hard piece may be partial)
    static constexpr int NPB = (P + 15) / 16;               // 16-part blocks
    static constexpr int NOUT = HT / 16;                    // 16-byte pieces of the g_hard tile
    static constexpr size_t SHMEM = 3 * (size_t)SLOT + HT;
    static_assert(GT % 1024 == 0 && HT % 16 == 0 && P <= 32 && LD >= UB_F + P && LD % 8 == 0, "tile geometry");
};

__device__ __forceinline__ void ub_wait_vm(int young) {
#define UPS_UB_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    switch (young) {
        UPS_UB_W(0) UPS_UB_W(1) UPS_UB_W(2) UPS_UB_W(3) UPS_UB_W(4) UPS_UB_W(5) UPS_UB_W(6) UPS_UB_W(7) UPS_UB_W(8) UPS_UB_W(9)
        UPS_UB_W(10) UPS_UB_W(11) UPS_UB_W(12) UPS_UB_W(13) UPS_UB_W(14) UPS_UB_W(15) UPS_UB_W(16) UPS_UB_W(17) UPS_UB_W(18) UPS_UB_W(19)
        UPS_UB_W(20) UPS_UB_W(21) UPS_UB_W(22) UPS_UB_W(23) UPS_UB_W(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef UPS_UB_W
}

template <int P, int LD>
__global__ __launch_bounds__(256, P <= 10 ? 2 : 1) void unpool_bwd_mfma_kernel(const float* __restrict__ hard, const float* __restrict__ feat,
                                                                  const bf16* __restrict__ g, float* __restrict__ gh,
                                                                  float* __restrict__ gfeat_partial, const long long hw,
                                                                  const int tiles_per_block, const int slabs_per_block, const int nslab) {
    typedef UB<P, LD> U;
    constexpr int NPB = U::NPB;
    typedef __attribute__((ext_vector_type(4))) short v4s;
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* OS = smem + 3 * U::SLOT;          // g_hard tile [128][P] fp32, linear
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x, sb = blockIdx.y;       // image, block of the image
    const long long px0 = (long long)sb * tiles_per_block * UB_TP;       // first pixel of this block
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* gsrc = (const unsigned char*)(g + ((long long)b * hw + px0) * LD);
    const unsigned char* hsrc = (const unsigned char*)(hard + ((long long)b * hw + px0) * P);
    const unsigned voff = (unsigned)lane * 16u;
    // LDS-DMA pieces / 16-byte stores this wave issues per tile (wave-instructions: what vmcnt counts)
    constexpr int KQ = (U::PIECES + 3) / 4, KS = (U::NOUT + 255) / 256;
    const int L = (U::PIECES - wid + 3) / 4, S = (U::NOUT - 64 * wid + 255) / 256;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const unsigned base = smem_lds + (unsigned)((t % 3) * U::SLOT);
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int q = wid + 4 * k;
            if (q < U::PIECES) {
                const unsigned char* src = q < U::GP ? gsrc + (long long)t * U::GT + q * 1024
                                                     : hsrc + (long long)t * U::HT + (q - U::GP) * 1024;
                const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)q * 1024u);
                // (the last piece of a hard tile whose size is not a multiple of 1 KiB -- P = 25: 12.5 KiB -- moves its first lanes only)
                const bool part = (U::HT & 1023) != 0 && q == U::PIECES - 1;
                if (!part || voff < (unsigned)(U::HT & 1023))
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(src) : "memory", "m0");
            }
        }
    };
    issue(0);
    if (1 < tiles_per_block) issue(1);
    // B operand of the first product: feat[b][part 16 nb + li][features 8 lg .. + 7] and [32 + 8 lg .. + 7] as bf16 (exact: in bf16 mode
    // feat is the float view of a bf16 tensor)
    bf16x8 fb[NPB][2];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const int pi = 16 * pb + li;
        const float* fr = feat + ((long long)b * P + min(pi, P - 1)) * UB_F;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) fb[pb][h][e] = (__bf16)(pi < P ? fr[32 * h + 8 * lg + e] : 0.f);
    }
    f32x4 acc[NPB][4];                                // g_feat: parts 16 pb + 4 lg + i x features 16 nb + li, this wave's pixels
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[pb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < tiles_per_block; ++t) {
        // younger than tile t's pieces: the stores of up to two tiles and tile t + 1's pieces
        ub_wait_vm(min(t, 2) * S + (t + 1 < tiles_per_block ? L : 0));
        __builtin_amdgcn_s_barrier();                 // tile t is in; everyone is done with tile t - 1 (its slot, the g_hard tile)
        if (t + 2 < tiles_per_block) issue(t + 2);
        const unsigned char* G = smem + (t % 3) * U::SLOT;
        const unsigned char* H = G + U::GT;
        const int R0 = wid * 32;                      // this wave's 32 pixels of the tile
        // ---- g_hard: two 16-pixel blocks, K = 64 features in two steps
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int row = R0 + 16 * mb + li;
            const bf16x8 a0 = *(const bf16x8*)(G + row * (LD * 2) + 16 * lg);
            const bf16x8 a1 = *(const bf16x8*)(G + row * (LD * 2) + 64 + 16 * lg);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, fb[pb][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, fb[pb][1], d, 0, 0, 0);
                // lane: pixels R0 + 16 mb + 4 lg + i, part 16 pb + li; + the gradient's own part channel
                const int pi = 16 * pb + li;
                if (pi < P) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = R0 + 16 * mb + 4 * lg + i;
                        const unsigned short u = *(const unsigned short*)(G + r * (LD * 2) + 2 * (UB_F + pi));
                        *(float*)(OS + r * (P * 4) + 4 * pi) = d[i] + __uint_as_float((unsigned)u << 16);
                    }
                }
            }
        }
        // ---- g_feat: K = this wave's 32 pixels; k-index j of lane group lg <-> rows R0 + 4 lg + j (j < 4), R0 + 16 + 4 lg + j - 4
        // (two 4-row blocks per 16-lane group that a 32-lane half reads from eight DIFFERENT rows: conflict-free at 160-byte rows)
        bf16x8 ah[NPB], al[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int pi = 16 * pb + li;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = R0 + (j < 4 ? 4 * lg + j : 16 + 4 * lg + j - 4);
                const float v = *(const float*)(H + r * (P * 4) + 4 * min(pi, P - 1));
                const float hv = pi < P ? v : 0.f;
                const __bf16 hi = (__bf16)hv;
                ah[pb][j] = hi;
                al[pb][j] = (__bf16)(hv - (float)hi);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            // lane 4 q + p of a group supplies row q, columns 4 p .. 4 p + 3 of the 4 x 16 block (guide T10); EXEC is all ones here
            const int q = li >> 2, pp = li & 3;
            const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(G + (R0 + 4 * lg + q) * (LD * 2) + 2 * (16 * nb + 4 * pp)));
            const v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(G + (R0 + 16 + 4 * lg + q) * (LD * 2) + 2 * (16 * nb + 4 * pp)));
            bf16x8 bb;
            __builtin_memcpy(&bb, &b0, 8);
            __builtin_memcpy((char*)&bb + 8, &b1, 8);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                acc[pb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[pb], bb, acc[pb][nb], 0, 0, 0);
                acc[pb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[pb], bb, acc[pb][nb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // the g_hard tile is complete
        {
            float4* dst = (float4*)(gh + ((long long)b * hw + px0 + (long long)t * UB_TP) * P);
            const float4* o4 = (const float4*)OS;
#pragma unroll
            for (int k = 0; k < KS; ++k)
                if (256 * k + tid < U::NOUT) dst[256 * k + tid] = o4[256 * k + tid];
        }
    }
    // ---- the four waves' g_feat blocks -> one record of the block's slab group (the other records of the group: zero)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = (float*)smem;                        // [4 waves][16 NPB parts][64 features]
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wid * 16 * NPB + 16 * pb + 4 * lg + i) * 64 + 16 * nb + li] = acc[pb][nb][i];
    __syncthreads();
    float* out = gfeat_partial + (((long long)b * nslab + (long long)sb * slabs_per_block) * P) * UB_F;
    for (int i = tid; i < slabs_per_block * P * UB_F; i += 256) {
        float v = 0.f;
        if (i < P * UB_F) {
            const int pp = i >> 6, ff = i & 63;
            v = (red[pp * 64 + ff] + red[(16 * NPB + pp) * 64 + ff]) + (red[(32 * NPB + pp) * 64 + ff] + red[(48 * NPB + pp) * 64 + ff]);
        }
        out[i] = v;
    }
}

__global__ void unpool_feat_reduce_kernel(const float* __restrict__ partial, int B, int nslab, int PF, float* __restrict__ gfeat) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * PF) return;
    const int b = idx / PF, r = idx - b * PF;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < nslab; ++k) s += partial[((long long)b * nslab + k) * PF + r];      // (independent loads, one chain of adds)
    gfeat[idx] = s;
}

constexpr int UNPOOL_SLABS = 32;

// kernels whose LDS images can exceed the 64 KB default at P = 64: raise the limit once per kernel
template <typename K>
bool allow_big_lds(K kernel, UpsPerDevice& done) {
    if (!done) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        done = true;
    }
    return true;
}

}  // namespace

// Launch of part_softmax_px_kernel when the shape allows it (P = 10, whole 256-pixel tiles, 16-byte aligned maps); images of
// img_hw pixels in rows of img_w (no moments: no image structure needed, the pixels are cut into pseudo-images).
// Returns blocks per image (> 0) when launched, 0 when the caller has to take the generic kernel.
static int part_softmax_px_launch(const float* mean, const float* eps, float* l, float* m, float* hard, int64_t* argmax,
                                  uint32_t* hard_bits, long long pixels, int P, int img_w, long long img_hw, int* mom, hipStream_t s) {
    const char* e = getenv("UPS_SOFTMAX_PX");
    if ((e && e[0] == '0') || P != 10 || pixels % 256 != 0) return 0;
    if (((((uintptr_t)mean) | ((uintptr_t)eps) | ((uintptr_t)l) | ((uintptr_t)m) | ((uintptr_t)hard)) & 15) != 0) return 0;
    if (eps && !l) return 0;
    if (!mom) { img_w = 128; img_hw = pixels % 16384 == 0 ? 16384 : 256; }
    if (img_hw % 256 != 0 || pixels % img_hw != 0 || img_w < 1) return 0;
    const int tiles_img = (int)(img_hw / 256);
    const long long n = pixels / img_hw;
    // blocks per image: two blocks per CU over the batch; at most 32 tiles per block (int32 sums of iy^2 over <= 8192 pixels of rows
    // < 2^11); the block records must fit the caller's scratch (one record per 512 pixels: ups_part_softmax_moments_ints)
    int bpi = 1;
    while ((n * bpi < 512 || tiles_img / bpi > 32) && tiles_img % (2 * bpi) == 0 && 2 * bpi * 512 <= img_hw) bpi *= 2;
    if (tiles_img / bpi > 32 || (mom && img_hw / img_w > 256)) return 0;      // (int32 sums of iy^2: rows < 256, <= 8192 pixels per block)
    constexpr size_t shm = (2 * 2 + 3) * (size_t)(256 * 10 * 4) + 10 * 5 * sizeof(int);
    static UpsPerDevice at;
    if (!at) {
        if (hipFuncSetAttribute((const void*)part_softmax_px_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 0;
        at = true;
    }
    hipLaunchKernelGGL((part_softmax_px_kernel<10>), dim3((unsigned)(n * bpi)), dim3(256), shm, s, mean, eps, l, m, hard,
                       (long long*)argmax, (unsigned*)hard_bits, tiles_img, tiles_img / bpi, mom, img_w);
    return bpi;
}

extern "C" int ups_part_softmax_fwd(const float* mean, const float* eps, float* l, float* m, float* hard, int64_t* argmax,
                                    uint32_t* hard_bits, int64_t pixels, int32_t P, void* stream) {
    UPS_CHECK_ARG(mean && m && pixels > 0 && P >= 1 && P <= 64);
    UPS_CHECK_ARG(!hard_bits || P <= 32);
    hipStream_t s = (hipStream_t)stream;
    if (part_softmax_px_launch(mean, eps, l, m, hard, argmax, hard_bits, pixels, P, 0, 0, nullptr, s) > 0) {
        UPS_LAUNCH_CHECK();
        return UPS_OK;
    }
    const int tpx = tile_pixels(P, 1, 24 * 1024);
    const int grid = ups_cdiv(pixels, tpx);
    const size_t shm = (size_t)tpx * (P | 1) * sizeof(float);
    hipLaunchKernelGGL(part_softmax_kernel, dim3(grid), dim3(256), shm, s, mean, eps, l, m, hard, (long long*)argmax,
                       (unsigned*)hard_bits, (long long)pixels, P, tpx, (int*)nullptr, 1, 1);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int32_t ups_part_softmax_moments_tile(int32_t P) { return tile_pixels(P, 1, 24 * 1024); }

extern "C" size_t ups_part_softmax_moments_ints(int64_t pixels, int32_t P) {
    return (size_t)ups_cdiv(pixels, tile_pixels(P, 1, 24 * 1024)) * P * 5;
}

extern "C" int ups_part_softmax_moments_fwd(const float* mean, const float* eps, float* l, float* m, float* hard, int64_t* argmax,
                                            uint32_t* hard_bits, int32_t n, int32_t h, int32_t w, int32_t P, float gamma,
                                            float* stats, int32_t* scratch, void* stream) {
    UPS_CHECK_ARG(mean && m && stats && scratch && n > 0 && h > 0 && w > 0 && P >= 1 && P <= 32 && gamma > 0.f);
    hipStream_t s = (hipStream_t)stream;
    const long long pixels = (long long)n * h * w;
    const int tpx = tile_pixels(P, 1, 24 * 1024);
    UPS_CHECK_ARG((h * w) % tpx == 0);           // a tile never straddles two images
    {
        const int bpi = part_softmax_px_launch(mean, eps, l, m, hard, argmax, hard_bits, pixels, P, w, (long long)h * w, (int*)scratch, s);
        if (bpi > 0) {
            UPS_LAUNCH_CHECK();
            hipLaunchKernelGGL(hard_moments_finalize_kernel, dim3(ups_cdiv((long long)n * P, 4)), dim3(256), 0, s, (const int*)scratch, n,
                               bpi, P, h, w, gamma, stats);
            UPS_LAUNCH_CHECK();
            return UPS_OK;
        }
    }
    const int grid = ups_cdiv(pixels, tpx);
    const size_t shm = (size_t)tpx * (P | 1) * sizeof(float) + (size_t)P * 5 * sizeof(int);
    hipLaunchKernelGGL(part_softmax_kernel, dim3(grid), dim3(256), shm, s, mean, eps, l, m, hard, (long long*)argmax,
                       (unsigned*)hard_bits, pixels, P, tpx, (int*)scratch, w, h * w);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(hard_moments_finalize_kernel, dim3(ups_cdiv((long long)n * P, 4)), dim3(256), 0, s, (const int*)scratch, n,
                       (h * w) / tpx, P, h, w, gamma, stats);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

constexpr int MOMENT_SLABS = 32;      // row slabs per image (n x 32 blocks: 2 048 at B = 64; two 256-pixel tiles each at 128x128)

static int spatial_moments_launch(const float* x, int32_t n, int32_t h, int32_t w, int32_t P, float gamma, const int32_t* rect_c,
                                  int32_t half_h, int32_t half_w, float* stats, float* kl_sum, void* stream) {
    UPS_CHECK_ARG(x && stats && n > 0 && P >= 1 && P <= 64);
    hipStream_t s = (hipStream_t)stream;
    // the stats buffer doubles as the workspace: n*P*8 floats of result, then n*MOMENT_SLABS*P*8 floats of per-slab partials and
    // n*MOMENT_SLABS floats of KL partials (ups_spatial_moments_floats)
    {   // pixel-per-lane form (round 5): P = 10, 128- / 256-wide maps in whole 512-pixel tiles
        const char* e = getenv("UPS_MOMENTS_PX");
        const long long hwp = (long long)h * w;
        if (!(e && e[0] == '0') && P == 10 && (w == 128 || w == 256) && hwp % 512 == 0 && (((uintptr_t)x) & 15) == 0) {
            const int tiles_img = (int)(hwp / 512);
            int bpi = 1;          // two blocks per CU over the batch; at most MOMENT_SLABS records per image
            const char* eb = getenv("UPS_MOMENTS_PX_BLOCKS");
            const long long want_blocks = eb ? atoll(eb) : 512;
            while (2 * bpi <= MOMENT_SLABS && (long long)n * bpi < want_blocks && tiles_img % (2 * bpi) == 0) bpi *= 2;
            float* partial_px = stats + (long long)n * P * 8;
            float* klp_px = kl_sum ? partial_px + (long long)n * MOMENT_SLABS * P * 8 : nullptr;
            constexpr size_t shm_max = 3 * (size_t)(512 * 10 * 4) + (4 * 10 * 7 + 4) * sizeof(float);
            const size_t shm_px = (size_t)(tiles_img / bpi < 3 ? tiles_img / bpi : 3) * (512 * 10 * 4) + (4 * 10 * 7 + 4) * sizeof(float);
            static UpsPerDevice a7;
            if (!a7) {
                if (hipFuncSetAttribute((const void*)moments_px_kernel<10, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_max) != hipSuccess ||
                    hipFuncSetAttribute((const void*)moments_px_kernel<10, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_max) != hipSuccess)
                    return UPS_E_LAUNCH;
                a7 = true;
            }
            if (w == 128)
                hipLaunchKernelGGL((moments_px_kernel<10, 7>), dim3(n * bpi), dim3(256), shm_px, s, x, h, gamma, rect_c, half_h, half_w,
                                   tiles_img / bpi, partial_px, klp_px);
            else
                hipLaunchKernelGGL((moments_px_kernel<10, 8>), dim3(n * bpi), dim3(256), shm_px, s, x, h, gamma, rect_c, half_h, half_w,
                                   tiles_img / bpi, partial_px, klp_px);
            UPS_LAUNCH_CHECK();
            hipLaunchKernelGGL(moments_combine_kernel, dim3(ups_cdiv(n * P, 4) + 1), dim3(256), 0, s, partial_px, n, bpi, P, stats, klp_px, kl_sum);
            UPS_LAUNCH_CHECK();
            return UPS_OK;
        }
    }
    const int nslab = MOMENT_SLABS;
    float* partial = stats + (long long)n * P * 8;
    float* klp = kl_sum ? partial + (long long)n * nslab * P * 8 : nullptr;
    const int rows = ups_cdiv(h, nslab);
    const int tpx = tile_pixels(P, 1, 24 * 1024);
    const size_t shm = ((size_t)tpx * (P | 1) + (size_t)(256 / P) * P * 7 + 4) * sizeof(float);
    hipLaunchKernelGGL(moments_partial_kernel, dim3(n, nslab), dim3(256), shm, s, x, h, w, P, gamma, rect_c, half_h, half_w, rows, tpx,
                       partial, klp);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(moments_combine_kernel, dim3(ups_cdiv(n * P, 4) + 1), dim3(256), 0, s, partial, n, nslab, P, stats, klp, kl_sum);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_spatial_moments(const float* x, int32_t n, int32_t h, int32_t w, int32_t P, float gamma,
                                   const int32_t* rect_c, int32_t half_h, int32_t half_w, float* stats, void* stream) {
    return spatial_moments_launch(x, n, h, w, P, gamma, rect_c, half_h, half_w, stats, nullptr, stream);
}

extern "C" int ups_spatial_moments_kl(const float* x, int32_t n, int32_t h, int32_t w, int32_t P, float gamma,
                                      const int32_t* rect_c, int32_t half_h, int32_t half_w, float* stats, float* kl_sums16,
                                      void* stream) {
    UPS_CHECK_ARG(kl_sums16);
    return spatial_moments_launch(x, n, h, w, P, gamma, rect_c, half_h, half_w, stats, kl_sums16, stream);
}

extern "C" size_t ups_spatial_moments_floats(int32_t n, int32_t P) {
    return (size_t)n * P * 8 + (size_t)n * MOMENT_SLABS * P * 8 + (size_t)n * MOMENT_SLABS;
}
extern "C" size_t ups_unpool_bwd_floats(int32_t B, int32_t P, int32_t F) { return (size_t)B * P * F * (1 + UNPOOL_SLABS); }

extern "C" int ups_moments_to_px(const float* stats, int32_t count, int32_t h, int32_t xy_order, int32_t* px, void* stream) {
    UPS_CHECK_ARG(stats && px && count > 0);
    hipLaunchKernelGGL(moments_to_px_kernel, dim3(ups_cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, stats, count, h, xy_order, px);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_draw_rect(const int32_t* px, int32_t n, int32_t h, int32_t w, int32_t P, int32_t half_h, int32_t half_w,
                             float* out, void* stream) {
    UPS_CHECK_ARG(px && out);
    hipLaunchKernelGGL(draw_rect_kernel, dim3(ups_cdiv((long long)n * h * w * P, 256)), dim3(256), 0, (hipStream_t)stream,
                       px, n, h, w, P, half_h, half_w, out);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_mask_parts_fwd(const float* view, const float* hard, void* out, int32_t dtype, int32_t B, int64_t hw,
                                  int32_t P, void* stream) {
    UPS_CHECK_ARG(view && hard && out && P >= 1 && P <= 64);
    const int grid = ups_cdiv((long long)B * hw, 256);
    const size_t shm = (size_t)256 * (P | 1) * sizeof(float);
    static UpsPerDevice a0, a1;
    if (!allow_big_lds(mask_parts_fwd_kernel<float>, a0) || !allow_big_lds(mask_parts_fwd_kernel<bf16>, a1)) return UPS_E_LAUNCH;
    if (dtype == UPS_F32) hipLaunchKernelGGL(mask_parts_fwd_kernel<float>, dim3(grid), dim3(256), shm, (hipStream_t)stream, view, hard, (float*)out, B, (long long)hw, P);
    else hipLaunchKernelGGL(mask_parts_fwd_kernel<bf16>, dim3(grid), dim3(256), shm, (hipStream_t)stream, view, hard, (bf16*)out, B, (long long)hw, P);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_mask_parts_bwd(const float* view, const void* g_out, float* g_hard, int32_t dtype, int32_t B, int64_t hw,
                                  int32_t P, void* stream) {
    UPS_CHECK_ARG(view && g_out && g_hard);
    UPS_CHECK_ARG(P >= 1 && P <= 64);
    const int grid = ups_cdiv((long long)B * hw, 256);
    const size_t shm = (size_t)256 * (P | 1) * sizeof(float);
    static UpsPerDevice a0, a1;
    if (!allow_big_lds(mask_parts_bwd_kernel<float>, a0) || !allow_big_lds(mask_parts_bwd_kernel<bf16>, a1)) return UPS_E_LAUNCH;
    if (dtype == UPS_F32) hipLaunchKernelGGL(mask_parts_bwd_kernel<float>, dim3(grid), dim3(256), shm, (hipStream_t)stream, view, (const float*)g_out, g_hard, B, (long long)hw, P);
    else hipLaunchKernelGGL(mask_parts_bwd_kernel<bf16>, dim3(grid), dim3(256), shm, (hipStream_t)stream, view, (const bf16*)g_out, g_hard, B, (long long)hw, P);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_unpool_fwd(const float* hard, const float* feat, void* out, int32_t dtype, int32_t B, int64_t hw, int32_t P,
                              int32_t F, int32_t ldo, void* stream) {
    UPS_CHECK_ARG(hard && feat && out && F % 8 == 0 && ldo % 8 == 0 && ldo >= F + P && P >= 1 && P <= 64);
    const dim3 grid(ups_cdiv(hw, 256), B);
    const size_t shm = ((size_t)256 * (P | 1) + (size_t)P * F + 256) * sizeof(float);
    static UpsPerDevice a0, a1;
    if (!allow_big_lds(unpool_fwd_kernel<float>, a0) || !allow_big_lds(unpool_fwd_kernel<bf16>, a1)) return UPS_E_LAUNCH;
    if (dtype == UPS_F32) hipLaunchKernelGGL(unpool_fwd_kernel<float>, grid, dim3(256), shm, (hipStream_t)stream, hard, feat, (float*)out, (long long)hw, P, F, ldo);
    else hipLaunchKernelGGL(unpool_fwd_kernel<bf16>, grid, dim3(256), shm, (hipStream_t)stream, hard, feat, (bf16*)out, (long long)hw, P, F, ldo);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

// g_feat: caller provides B*P*F floats followed by B*UNPOOL_SLABS*P*F floats of workspace.
extern "C" int ups_unpool_bwd(const float* hard, const float* feat, const void* g, float* g_hard, float* g_feat, int32_t dtype,
                              int32_t B, int64_t hw, int32_t P, int32_t F, int32_t ldo, void* stream) {
    UPS_CHECK_ARG(hard && feat && g && g_hard && g_feat && F <= 64 && F % 8 == 0 && P <= 64 && ldo >= F + P && ldo % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    float* partial = g_feat + (long long)B * P * F;
    const size_t esz = dtype == UPS_F32 ? 4 : 2;
    {   // matrix-core form (round 5; round 6: P = 16 / 20 / 25 too): bf16 gradient, 64 features + P parts in round8(64 + P)-channel rows,
        // whole 128-pixel tiles
        const char* e = getenv("UPS_UNPOOL_MFMA");          // (read at every call: the unit test compares both forms)
        const bool mf_on = !(e && e[0] == '0');
        const bool al16 = ((((uintptr_t)hard) | ((uintptr_t)g) | ((uintptr_t)g_hard)) & 15) == 0;
        const bool shape = (P == 10 && ldo == 80) || (P == 16 && ldo == 80) || (P == 20 && ldo == 88) || (P == 25 && ldo == 96);
        if (mf_on && dtype == UPS_BF16 && shape && F == UB_F && hw % UB_TP == 0 && al16) {
            const int tiles_img = (int)(hw / UB_TP);
            // blocks per image: two per CU over the batch, a power of two dividing the tiles and the slab records
            int bpi = 1;
            while (2 * bpi <= UNPOOL_SLABS && (long long)B * bpi < 512 && tiles_img % (2 * bpi) == 0) bpi *= 2;
            // (one record per block, `bpi` records per image: the reduction walks 8 records instead of UNPOOL_SLABS = 32)
#define UPS_UB_LAUNCH(PV, LDV) do { \
                static UpsPerDevice am; \
                if (!am) { \
                    if (hipFuncSetAttribute((const void*)unpool_bwd_mfma_kernel<PV, LDV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)UB<PV, LDV>::SHMEM) != hipSuccess) return UPS_E_LAUNCH; \
                    am = true; \
                } \
                hipLaunchKernelGGL((unpool_bwd_mfma_kernel<PV, LDV>), dim3(B, bpi), dim3(256), (UB<PV, LDV>::SHMEM), s, hard, feat, (const bf16*)g, \
                                   g_hard, partial, (long long)hw, tiles_img / bpi, 1, bpi); } while (0)
            if (P == 10) UPS_UB_LAUNCH(10, 80);
            else if (P == 16) UPS_UB_LAUNCH(16, 80);
            else if (P == 20) UPS_UB_LAUNCH(20, 88);
            else UPS_UB_LAUNCH(25, 96);
#undef UPS_UB_LAUNCH
            UPS_LAUNCH_CHECK();
            const int total = B * P * F;
            hipLaunchKernelGGL(unpool_feat_reduce_kernel, dim3(ups_cdiv(total, 256)), dim3(256), 0, s, partial, B, bpi, P * F, g_feat);
            UPS_LAUNCH_CHECK();
            return UPS_OK;
        }
    }
    int tpx = 128;
    auto lds_bytes = [&](int t) {
        return ((size_t)4 * P * 64 + 2 * (((size_t)t * (P | 1) + 3) & ~(size_t)3) + (((size_t)P * (F + 1) + 3) & ~(size_t)3)) * sizeof(float)
               + (size_t)t * ldo * esz;
    };
    while (tpx > 32 && lds_bytes(tpx) > 52 * 1024) tpx >>= 1;
    // whole tiles per slab, so that every tile's first pixel (and its 16-byte pieces) is aligned
    const int slab_px = ups_cdiv(ups_cdiv(hw, UNPOOL_SLABS), tpx) * tpx;
    const size_t shmem = lds_bytes(tpx);
    UPS_CHECK_ARG(shmem <= 160 * 1024);
    const dim3 grid(B, UNPOOL_SLABS);
    static UpsPerDevice ah0, ah1;
    if (!allow_big_lds(unpool_bwd_kernel<float>, ah0) || !allow_big_lds(unpool_bwd_kernel<bf16>, ah1)) return UPS_E_LAUNCH;
    if (dtype == UPS_F32)
        hipLaunchKernelGGL(unpool_bwd_kernel<float>, grid, dim3(256), shmem, s, hard, feat, (const float*)g, g_hard, partial, (long long)hw,
                           P, F, ldo, slab_px, tpx);
    else
        hipLaunchKernelGGL(unpool_bwd_kernel<bf16>, grid, dim3(256), shmem, s, hard, feat, (const bf16*)g, g_hard, partial, (long long)hw,
                           P, F, ldo, slab_px, tpx);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(unpool_feat_reduce_kernel, dim3(ups_cdiv(B * P * F, 256)), dim3(256), 0, s, partial, B, UNPOOL_SLABS, P * F, g_feat);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
